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
#include <cstdio>
#include <cstdlib>

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
#ifdef MV_STEP_PRIO
    __builtin_amdgcn_s_setprio(MV_STEP_PRIO);
#endif
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
        // (tick-only launches: 52..55 = ticks that regenerated the env, longest other tick)
        if (gv.hdr[env].num_frames == 0) { d[52] += rt1 - rt0; d[53] += 1; }
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

// k consecutive ticks of every env with ONE launch (a batched open-loop call, mv_step_n with a device-side random policy): the envs are
// independent and every tick draws its own actions, so nothing orders env A's tick j + 1 behind env B's tick j -- only k launches did, each as
// long as its slowest env, and each having to find room for 1024 two-wave workgroups of ~150 VGPRs beside the observation pass of the previous
// call (kernel traces, r04l: 70-190 us per step kernel while a pass runs, 20 us alone; the chain of step kernels, not the pass, set the pipelined
// rate).  Here an env's workgroup becomes resident once and runs tick, frame setup (into slot j's lists), tick, ...: gv[j] is tick j's view
// (its hand-over slot, its staging outputs, its action index, its cost histogram).  The frame setups of such a launch do not clear the next pass's
// histogram (lpt_no_clear: inside one launch env 0's clearing would race with the envs that are a tick ahead): every pass that draws from one clears it
// when its last workgroup has looked its frame up (mv_raster.hip: hist_done; mv_api.hip: take_hist).
// One agent: ONE wave per env (the single-tick kernel's second wave only helps with the frame setup, and idles through the tick): the
// workgroups stay resident for the whole call beside the observation passes of the previous one, and every wave of ~150 VGPRs they hold is
// two or three waves the pass cannot have (measured: 21.1 M obs/s with two waves per env, 22.3 M with one).
// MV_STEP_TICKS_WAVES_PER_SIMD: the register budget of the one-wave-per-env multi-tick kernels (512 / n VGPRs).  Their waves stay resident for a whole
// batched call beside the observation passes, and what they hold the passes cannot have: left to itself hipcc takes 236 VGPRs for the TowerBuilding
// tick (launch bound 64: nothing asks it to be frugal), the one-launch-per-tick kernel does the same work in 97.
#ifndef MV_STEP_TICKS_WAVES_PER_SIMD
#define MV_STEP_TICKS_WAVES_PER_SIMD 4
#endif
template <int A_MAX, class Args>
__device__ __forceinline__ void step_ticks_body(const Args &a, int W, int H)
{
    __shared__ FrameScratch s_fs[A_MAX == 1 ? 1 : 4];
    const int env = blockIdx.x;
#ifdef MV_STEP_PRIO
    __builtin_amdgcn_s_setprio(MV_STEP_PRIO);
#endif
    for (int j = 0; j < a.n; ++j) {
        const GymView &gv = a.view(j);
        if (A_MAX == 1) {
            tower_tick<A_MAX>(gv, env);
            wave_sync();   // the tick's stores before the frame setup's loads (one wave: no barrier needed)
            frame_setup_body<64, true>(gv, env, W, H, s_fs[0]);
        } else {
            tower_tick<A_MAX, (A_MAX > 1)>(gv, env);
            __syncthreads();   // the tick's stores (same CU: same L1) before the frame setup's loads
            const int A = gv.num_agents, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), nw = blockDim.x >> 6;
            for (int q = wave; q < A; q += nw) frame_setup_body<64, true>(gv, env * A + q, W, H, s_fs[wave]);
            __syncthreads();   // every frame of the env is set up (the state they read) before the next tick changes it
        }
    }
}
// (two kernels, not one template: a launch bound that depends on a template parameter is not applied)
template <class Args> __global__ __launch_bounds__(64, MV_STEP_TICKS_WAVES_PER_SIMD) void step_ticks_kernel(Args a,
                                                   int W, int H) { step_ticks_body<1>(a, W, H); }
template <class Args> __global__ __launch_bounds__(256, 3) void step_ticks_agents_kernel(Args a, int W, int H) { step_ticks_body<MAX_AGENTS>(a, W, H); }

// Software-pipelined (one agent per env): TWO waves per env.  Wave 0 runs tick j + 1 while wave 1 sets tick j's frame up (mv_frame.h) -- the two halves of a
// tick's work that step_ticks_body runs back to back in one wave, each a chain of dependent loads and a few thousand vector instructions of ONE wave on its
// SIMD (48 % of the resident wave's cycles were spent in s_waitcnt, r08z_pmc_SQ2.csv).  The frame setup reads the simulator state in place, so the two waves
// meet at two workgroup barriers per tick:
//   A(j): tick j's state is written (wave 0: behind its write-back and the episode swap-in of a finished env; wave 1: before it reads anything)
//   B(j): tick j's state is read    (wave 1: behind the record loads of its last round of slots; wave 0: before tick j + 1's write-back, tower_tick's
// pipe_wait) An iteration lasts max(tick, frame setup) instead of their sum; the last frame setup runs alone.
template <class Args>
__global__ __launch_bounds__(128, MV_STEP_TICKS_WAVES_PER_SIMD) void step_ticks_pipe_kernel(Args a, int W, int H)
{
    __shared__ FrameScratch s_fs;
    const int env = blockIdx.x;
#ifdef MV_STEP_PRIO
    __builtin_amdgcn_s_setprio(MV_STEP_PRIO);
#endif
    if (__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)) == 0) {
        for (int j = 0; j < a.n; ++j) {
            tower_tick<1>(a.view(j), env, j > 0);   // (j > 0: B(j - 1) inside, before the write-back)
            __syncthreads();                        // A(j)
        }
        __syncthreads();                            // B(n - 1): the last frame setup's
    } else {
        for (int j = 0; j < a.n; ++j) {
            __syncthreads();                        // A(j)
            frame_setup_body<64, true, true>(a.view(j), env, W, H, s_fs);   // B(j) inside
        }
    }
}

// When the two-wave kernels run.  ALONE on the chip their launch is a quarter shorter (TowerBuilding 16.6 -> 12.3 us per tick: 98 us per 8 ticks, Empty 16.9 ->
// 14.5, r09a); beside the observation passes of a FULL chip -- 1024 envs: a resident wave on every SIMD -- its length follows the passes' vector load, not its
// own dependent chains, and the second wave per env is registers the passes lose: TowerBuilding 32.8 / 32.1 M obs/s (one wave / two), ObstaclesHard 28.6 / 26.1
// (r09z, r10r).  With fewer envs than SIMDs the second wave is free and the step launch is what a call waits for (r10s, one wave / two, M obs/s):
//   envs           256            512            768
//   TowerBuilding  11.6 / 16.9    24.0 / 25.2    28.5 / 28.2
//   ObstaclesHard  11.6 / 14.0    20.7 / 23.1    26.0 / 26.3      (512: one GPU's share of BASELINE configs[2])
//   ObstaclesEasy  12.3 / 14.0    21.3 / 23.8    26.5 / 27.3
//   Empty          15.4 / 17.9    26.1 / 33.1    35.4 / 44.6      (1024: 45.8 / 47.0)
// So: up to 512 envs always, the Obstacles family up to 768, Empty at any size.  MV_STEP_PIPE=0 / 1 forces one or the other (read at every launch: tests switch).
bool step_pipe_enabled(const GymView &gv)
{
    const char *e = getenv("MV_STEP_PIPE");
    if (e && *e) return atoi(e) != 0;
    const bool obstFamily = gv.scenario == SCN_OBSTACLES || gv.scenario == SCN_EMPTY;
    return gv.num_envs <= 512 || (obstFamily && gv.num_envs <= 768) || gv.scenario == SCN_EMPTY;
}

void launch_step_ticks(const GymView *views, int k, hipStream_t stream, int W, int H, hipEvent_t done)
{
    const GymView &gv = views[0];
    // several agents: TWO waves per env (measured at 512 envs x 4 agents: one wave 20.3 M obs/s, two 21.7, four 16.1 -- four waves of ~180 VGPRs per env,
    // resident for the whole call, are what the observation passes beside them cannot have; one-launch-per-tick: 19.4)
    const dim3 grid(gv.num_envs), block(gv.num_agents == 1 ? 64 : 64 * std::min(gv.num_agents, 2));
    StepTicksArgs8 a;   // (k <= 8: the views are the launch's arguments)
    a.n = k; a.pad = 0;
    for (int j = 0; j < 8; ++j) a.gv[j] = views[std::min(j, k - 1)];
    if (gv.num_agents == 1 && step_pipe_enabled(gv)) hipExtLaunchKernelGGL(step_ticks_pipe_kernel<StepTicksArgs8>,
        grid, dim3(128), 0, stream, nullptr, done, 0, a, W, H);
    else if (gv.num_agents == 1) hipExtLaunchKernelGGL(step_ticks_kernel<StepTicksArgs8>, grid, block, 0, stream, nullptr, done, 0, a, W, H);
    else hipExtLaunchKernelGGL(step_ticks_agents_kernel<StepTicksArgs8>, grid, block, 0, stream, nullptr, done, 0, a, W, H);
}

// done: an event that completes with the launch, carried by its dispatch packet (cf. mv_raster.h)
void launch_step(const GymView &gv, hipStream_t stream, int W, int H, int render, hipEvent_t done)
{
    const dim3 grid(gv.num_envs), block(gv.num_agents == 1 ? STEP_THREADS : 64 * std::min(gv.num_agents, 4));
    if (gv.num_agents == 1) hipExtLaunchKernelGGL(step_kernel<1>, grid, block, 0, stream, nullptr, done, 0, gv, W, H, render);
    // (agent loops are real loops: one multi-agent build)
    else hipExtLaunchKernelGGL(step_kernel<MAX_AGENTS>, grid, block, 0, stream, nullptr, done, 0, gv, W, H, render);
}

}  // namespace mv
