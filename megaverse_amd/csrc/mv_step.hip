// megaverse_amd/csrc/mv_step.hip -- one simulation tick for every env: actions -> kinematic
// The tick itself (physics, scenario logic, episode swap-in) lives in mv_tick_tower.h; this file holds the kernels and their launchers.
// character physics -> TowerBuilding scenario logic -> timers/done.
//
// Replaces, per env (reference paths relative to src/libs):
//   Env::step                                   env/src/env.cpp:83-152
//   DefaultKinematicAgent look/accelerate/jump  env/src/agent.cpp:100-161
//   KinematicCharacterController::setAcceleration / preStep / playerStep / stepUp /
//     stepForwardAndStrafe / stepDown / recoverFromPenetration / updateTargetPositionBasedOnCollision
//                                               env/src/kinematic_character_controller.cpp:156-442,519-602,753-792
//   Bullet 2.89 ghost convexSweepTest + contact manifolds [third party, not vendored]: restated as
//     conservative advancement on exact closest points (DESIGN.md "physics model")
//   ObjectStackingComponent::step/onInteractAction  scenarios/include/scenarios/component_object_stacking.hpp:45-168
//   FallDetectionComponent::step                scenarios/include/scenarios/component_fall_detection.hpp:33-55
//   TowerBuildingScenario::step + callbacks     scenarios/src/scenario_tower_building.cpp:179-261
//   Scenario::rewardAgent/rewardTeam            env/include/env/scenario.hpp:259-298
//   done bookkeeping of VectorEnv::step         env/src/vector_env.cpp:93-105 (the reset itself: mv_reset.hip)
//
// Mapping: ONE WAVEFRONT PER ENV.  The env's colliders (<=16 layout slabs, <=80 movable boxes,
// <=8 agent capsules) live in VGPRs, two per lane.  A sweep is "every lane casts against its two
// colliders, then a 64-bit (fraction, slot) wave-min picks the winner"; depenetration and object
// lookups are ballot + find-first-lane.  Agents inside an env are order dependent (they collide
// with each other and share voxels) so they run one after another with wave-uniform state.
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>

#include <algorithm>

#include "mv_tick_tower.h"

namespace mv {

using namespace tick_tower;

// One workgroup per env: wave 0 runs the tick (one wave per env: physics, scenario logic, auto-reset), the others wait at the barrier (several
// agents: they take their share of the character controllers first);
// then the workgroup builds the lists of the env's frames (mv_frame.h).  `render` = 0: mv_step_no_render.
//   one agent:  STEP_THREADS (128) threads work on the env's one frame together.  The tick needs ~150 VGPRs, i.e. 3 waves per SIMD: with
//               2 waves per env 1024 envs are resident at once (with 4 they take two rounds, and a launch lasts as long as its slowest
//               tick PER ROUND: measured 41 us vs 25 us);
//   A agents:   64 min(A, 4) threads, every wave sets up its own frame(s): a frame setup is a chain of dependent loads (~6 us), A of them
//               one after the other would cost more than the launch the fusion saves.
template <int A_MAX>
__global__ __launch_bounds__(256) void step_kernel(GymView gv, int W, int H, int render)
{
    __shared__ FrameScratch s_fs[A_MAX == 1 ? 1 : 4];
    const int env = blockIdx.x;
    MV_T_BEGIN
#ifdef MV_TICK_TIMING
    const unsigned long long rt0 = __builtin_amdgcn_s_memrealtime();   // 100 MHz
#endif
    if (A_MAX > 1) tower_tick<A_MAX, (A_MAX > 1)>(gv, env);   // (several agents: every wave takes part, the controllers are shared out, mv_tick_tower.h)
    else if (threadIdx.x < 64) tower_tick<A_MAX>(gv, env);
#ifdef MV_TICK_TIMING
    if (!render && gv.dbg && threadIdx.x == 0) {
        const unsigned long long rt1 = __builtin_amdgcn_s_memrealtime();
        unsigned long long *d = gv.dbg + (size_t)env * 64;
        d[48] += rt1 - rt0; d[49] += 1; d[50] = rt0; d[51] = rt1;
        if (gv.hdr[env].num_frames == 0) { d[52] += rt1 - rt0; d[53] += 1; }      // (tick-only launches: 52..55 = ticks that regenerated the env, longest other tick)
        else if (rt1 - rt0 > d[54]) { d[54] = rt1 - rt0; for (int k = 0; k < 16; ++k) d[32 + k] = d[16 + k]; }
    }
#endif
    if (!render) return;
    __syncthreads();   // the tick's stores (same CU: same L1) before the frame setup's loads
    MV_T(6);           // the whole tick as wave 0 saw it (incl. the generator of a finished env), up to the barrier
    if (A_MAX == 1) {
        frame_setup_body<STEP_THREADS, false>(gv, env, W, H, s_fs[0]);
        MV_T(7);   // frame setup
#ifdef MV_TICK_TIMING
        if (gv.dbg && threadIdx.x == 0) {
            const unsigned long long rt1 = __builtin_amdgcn_s_memrealtime();
            unsigned long long *d = gv.dbg + (size_t)env * 64;
            if (!d[49]) { d[52] += rt1 - rt0; d[53] += 1; d[54] = rt0; d[55] = rt1; }
        }
#endif
    } else {
        const int A = gv.num_agents, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), nw = blockDim.x >> 6;
        for (int a = wave; a < A; a += nw) frame_setup_body<64, true>(gv, env * A + a, W, H, s_fs[wave]);
    }
}

// done: an event that completes with the launch, carried by its dispatch packet (cf. mv_raster.h)
void launch_step(const GymView &gv, hipStream_t stream, int W, int H, int render, hipEvent_t done)
{
    const dim3 grid(gv.num_envs), block(gv.num_agents == 1 ? STEP_THREADS : 64 * std::min(gv.num_agents, 4));
    if (gv.num_agents == 1) hipExtLaunchKernelGGL(step_kernel<1>, grid, block, 0, stream, nullptr, done, 0, gv, W, H, render);
    else hipExtLaunchKernelGGL(step_kernel<MAX_AGENTS>, grid, block, 0, stream, nullptr, done, 0, gv, W, H, render);   // (agent loops are real loops: one multi-agent build)
}

}  // namespace mv
