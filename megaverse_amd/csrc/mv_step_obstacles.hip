// megaverse_amd/csrc/mv_step_obstacles.hip -- one simulation tick + episode swap-in for the Obstacles family
// The tick itself (physics, scenario logic, episode swap-in) lives in mv_tick_obstacles.h; this file holds the kernels and their launchers.
// (ObstaclesEasy / Medium / Hard / Walls / Steps / Lava; BASELINE.json configs[2]).
//
// Replaces, per env (reference paths relative to src/libs):
//   Env::step                                   env/src/env.cpp:83-152            (shared pieces: mv_physics.h)
//   ObstaclesScenario::step / agentTouchedLava  scenarios/src/scenario_obstacles.cpp:197-239,268-278
//   ObjectStackingComponent (default callbacks) scenarios/include/scenarios/component_object_stacking.hpp:45-168
//   FallDetectionComponent                      scenarios/include/scenarios/component_fall_detection.hpp:33-55
//   Scenario::rewardTeam/rewardAll/doneWithTimer env/include/env/scenario.hpp:114-117,259-307
//   VectorEnv::step done bookkeeping + Env::reset of finished envs (env/src/vector_env.cpp:93-105): the reset
//   kernel below swaps in the episode the host generator (mv_gen_obstacles.cpp) left resident in HBM.
//
// Same mapping as the TowerBuilding kernel (one wavefront per env, colliders in VGPRs) with four colliders per
// lane: 128 merged layout slabs, 80 movable boxes, 8 agent capsules.  The level is a long chain of platforms,
// so voxel questions ("is this cell solid / lava / exit / holding a diamond?") are answered from the box
// lists with ballots instead of a dense chunk; column occupancy for drops and teleports is a 128-bit wave OR.
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>

#include "mv_tick_obstacles.h"

namespace mv {

using namespace tick_obstacles;

// One workgroup per env: wave 0 runs the tick (one wave per env: physics, scenario logic, auto-reset), the others wait at the barrier;
// then the workgroup builds the lists of the env's frames (mv_frame.h).  `render` = 0: mv_step_no_render.
//   one agent:  STEP_THREADS (128) threads work on the env's one frame together.  The tick needs ~150 VGPRs, i.e. 3 waves per SIMD: with
//               2 waves per env 1024 envs are resident at once (with 4 they take two rounds, and a launch lasts as long as its slowest
//               tick PER ROUND: measured 41 us vs 25 us);
//   A agents:   64 min(A, 4) threads, every wave sets up its own frame(s): a frame setup is a chain of dependent loads (~6 us), A of them
//               one after the other would cost more than the launch the fusion saves.
template <int A_MAX>
__global__ __launch_bounds__(256) void step_obstacles_kernel(GymView gv, int W, int H, int render)
{
    __shared__ FrameScratch s_fs[A_MAX == 1 ? 1 : 4];
    const int env = blockIdx.x;
    if (threadIdx.x < 64) obstacles_tick<A_MAX>(gv, env);
    if (!render) return;
    __syncthreads();   // the tick's stores (same CU: same L1) before the frame setup's loads
    if (A_MAX == 1) frame_setup_body<STEP_THREADS, false>(gv, env, W, H, s_fs[0]);
    else {
        const int A = gv.num_agents, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), nw = blockDim.x >> 6;
        for (int a = wave; a < A; a += nw) frame_setup_body<64, true>(gv, env * A + a, W, H, s_fs[wave]);
    }
}

// k consecutive ticks of every env with one launch (one agent per env; see step_ticks_kernel, mv_step.hip, for why): one wave per env, resident for the
// whole batched call; gv[j] is tick j's view.  An env that finishes swaps its next resident episode in at the tail of its tick as always -- a
// batched call only ever spans ticks of gyms whose episodes are long (mv_step_n steps the others tick by tick), so the two resident episodes
// outlast it.
#ifndef MV_STEP_TICKS_WAVES_PER_SIMD
#define MV_STEP_TICKS_WAVES_PER_SIMD 4   // (the register budget of the resident multi-tick waves: mv_step.hip)
#endif
template <class Args>
__global__ __launch_bounds__(64, MV_STEP_TICKS_WAVES_PER_SIMD) void step_obstacles_ticks_kernel(Args a, int W, int H)
{
    __shared__ FrameScratch s_fs;
    const int env = blockIdx.x;
#ifdef MV_STEP_PRIO
    __builtin_amdgcn_s_setprio(MV_STEP_PRIO);
#endif
    for (int j = 0; j < a.n; ++j) {
        const GymView &gv = a.view(j);
        obstacles_tick<1>(gv, env);
        wave_sync();   // the tick's stores before the frame setup's loads (one wave: no barrier needed)
        frame_setup_body<64, true>(gv, env, W, H, s_fs);
    }
}

// software-pipelined: two waves per env, wave 0 ticks while wave 1 sets the previous tick's frame up (mv_step.hip: step_ticks_pipe_kernel)
template <class Args>
__global__ __launch_bounds__(128, MV_STEP_TICKS_WAVES_PER_SIMD) void step_obstacles_ticks_pipe_kernel(Args a, int W, int H)
{
    __shared__ FrameScratch s_fs;
    const int env = blockIdx.x;
#ifdef MV_STEP_PRIO
    __builtin_amdgcn_s_setprio(MV_STEP_PRIO);
#endif
    if (__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)) == 0) {
        for (int j = 0; j < a.n; ++j) {
            obstacles_tick<1>(a.view(j), env, j > 0);
            __syncthreads();
        }
        __syncthreads();
    } else {
        for (int j = 0; j < a.n; ++j) {
            __syncthreads();
            frame_setup_body<64, true, true>(a.view(j), env, W, H, s_fs);
        }
    }
}

bool step_pipe_enabled(const GymView &gv);   // mv_step.hip

void launch_step_obstacles_ticks(const GymView *views, int k, hipStream_t stream, int W, int H, hipEvent_t done)
{
    StepTicksArgs8 a8;   // (k <= 8: the views are the launch's arguments, mv_types.h)
    a8.n = k; a8.pad = 0;
    for (int j = 0; j < 8; ++j) a8.gv[j] = views[std::min(j, k - 1)];
    if (step_pipe_enabled(views[0])) hipExtLaunchKernelGGL(step_obstacles_ticks_pipe_kernel<StepTicksArgs8>,
        dim3(views[0].num_envs), dim3(128), 0, stream, nullptr, done, 0, a8, W, H);
    else hipExtLaunchKernelGGL(step_obstacles_ticks_kernel<StepTicksArgs8>, dim3(views[0].num_envs), dim3(64), 0, stream, nullptr, done, 0, a8, W, H);
}

__global__ __launch_bounds__(64) void reset_obstacles_kernel(GymView gv, const EpisodeBlob *blobs, int *status, int force_all)
{
    const int env = blockIdx.x;
    if (env >= gv.num_envs) return;
    if (!force_all && !gv.hdr[env].done) return;
    swap_in_episode(gv, blobs, status, env, force_all);
}

void launch_step_obstacles(const GymView &gv, hipStream_t stream, int W, int H, int render)
{
    const dim3 grid(gv.num_envs), block(gv.num_agents == 1 ? STEP_THREADS : 64 * std::min(gv.num_agents, 4));
    if (gv.num_agents == 1) hipLaunchKernelGGL(step_obstacles_kernel<1>, grid, block, 0, stream, gv, W, H, render);
    // (agent loops are real loops: one multi-agent build)
    else hipLaunchKernelGGL(step_obstacles_kernel<MAX_AGENTS>, grid, block, 0, stream, gv, W, H, render);
}

void launch_reset_obstacles(const GymView &gv, const EpisodeBlob *blobs, int *status, int force_all, hipStream_t stream)
{
    hipLaunchKernelGGL(reset_obstacles_kernel, dim3(gv.num_envs), dim3(64), 0, stream, gv, blobs, status, force_all);
}

}  // namespace mv
