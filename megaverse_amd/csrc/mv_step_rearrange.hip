// megaverse_amd/csrc/mv_step_rearrange.hip -- one simulation tick + episode swap-in for the Rearrange scenario.
// The tick itself (physics, scenario logic, episode swap-in) lives in mv_tick_rearrange.h; this file holds the kernels and their launchers.
//
// Replaces, per env (reference paths relative to src/libs):
//   Env::step                                   env/src/env.cpp:83-152            (shared pieces: mv_physics.h)
//   RearrangeScenario::step / canPlaceObject / placedObject / pickedObject / checkDone / countMatchingObjects
//                                               scenarios/src/scenario_rearrange.cpp:125-180
//   ObjectStackingComponent                     scenarios/include/scenarios/component_object_stacking.hpp:45-168
//   VectorEnv::step done bookkeeping + Env::reset of finished envs (env/src/vector_env.cpp:93-105)
//
// The scene is tiny and almost entirely static: 5 room slabs, 9 static boxes (raised floor + two stepped pedestals,
// addStaticCollidingBox), the target arrangement (<= 7 static items) and its movable copy.  One wavefront per env,
// ONE collider per lane, in the order collision tie-breaks need: lanes 0-4 room, 5-13 static boxes, 14-21 target items,
// 22-29 movable items, 30-37 other agents.  The room is fixed (19 x H x 14), so "is this cell solid?" is arithmetic.
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>

#include "mv_tick_rearrange.h"

namespace mv {

using namespace tick_rearrange;

// One workgroup per env: wave 0 runs the tick (one wave per env: physics, scenario logic, auto-reset), the others wait at the barrier;
// then the workgroup builds the lists of the env's frames (mv_frame.h).  `render` = 0: mv_step_no_render.
//   one agent:  STEP_THREADS (128) threads work on the env's one frame together.  The tick needs ~150 VGPRs, i.e. 3 waves per SIMD: with
//               2 waves per env 1024 envs are resident at once (with 4 they take two rounds, and a launch lasts as long as its slowest
//               tick PER ROUND: measured 41 us vs 25 us);
//   A agents:   64 min(A, 4) threads, every wave sets up its own frame(s): a frame setup is a chain of dependent loads (~6 us), A of them
//               one after the other would cost more than the launch the fusion saves.
template <int A_MAX>
__global__ __launch_bounds__(256) void step_rearrange_kernel(GymView gv, int W, int H, int render)
{
    __shared__ FrameScratch s_fs[A_MAX == 1 ? 1 : 4];
    const int env = blockIdx.x;
    if (threadIdx.x < 64) rearrange_tick<A_MAX>(gv, env);
    if (!render) return;
    __syncthreads();   // the tick's stores (same CU: same L1) before the frame setup's loads
    if (A_MAX == 1) frame_setup_body<STEP_THREADS, false>(gv, env, W, H, s_fs[0]);
    else {
        const int A = gv.num_agents, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), nw = blockDim.x >> 6;
        for (int a = wave; a < A; a += nw) frame_setup_body<64, true>(gv, env * A + a, W, H, s_fs[wave]);
    }
}

__global__ __launch_bounds__(64) void reset_rearrange_kernel(GymView gv, const RearrangeBlob *blobs, int *status, int force_all)
{
    const int env = blockIdx.x;
    if (env >= gv.num_envs) return;
    if (!force_all && !gv.hdr[env].done) return;
    swap_in_episode(gv, blobs, status, env, force_all);
}

// k consecutive ticks of every env with one launch (one agent per env; see step_ticks_kernel, mv_step.hip, for why): one wave per env, resident for the
// whole batched call; gv[j] is tick j's view.  (Episodes come from the host: a batched call only ever spans ticks of gyms whose episodes are long,
// mv_step_n steps the others tick by tick, so the two resident episodes outlast it.)
#ifndef MV_STEP_TICKS_WAVES_PER_SIMD
#define MV_STEP_TICKS_WAVES_PER_SIMD 4   // (the register budget of the resident multi-tick waves: mv_step.hip)
#endif
template <class Args>
__global__ __launch_bounds__(64, MV_STEP_TICKS_WAVES_PER_SIMD) void step_rearrange_ticks_kernel(Args a, int W, int H)
{
    __shared__ FrameScratch s_fs;
    const int env = blockIdx.x;
    for (int j = 0; j < a.n; ++j) {
        const GymView &gv = a.view(j);
        rearrange_tick<1>(gv, env);
        wave_sync();   // the tick's stores before the frame setup's loads (one wave: no barrier needed)
        frame_setup_body<64, true>(gv, env, W, H, s_fs);
    }
}

void launch_step_rearrange_ticks(const GymView *views, int k, hipStream_t stream, int W, int H, hipEvent_t done)
{
    StepTicksArgs8 a8;   // (k <= 8: the views are the launch's arguments, mv_types.h)
    a8.n = k; a8.pad = 0;
    for (int j = 0; j < 8; ++j) a8.gv[j] = views[std::min(j, k - 1)];
    hipExtLaunchKernelGGL(step_rearrange_ticks_kernel<StepTicksArgs8>, dim3(views[0].num_envs), dim3(64), 0, stream, nullptr, done, 0, a8, W, H);
}

void launch_step_rearrange(const GymView &gv, hipStream_t stream, int W, int H, int render)
{
    const dim3 grid(gv.num_envs), block(gv.num_agents == 1 ? STEP_THREADS : 64 * std::min(gv.num_agents, 4));
    if (gv.num_agents == 1) hipLaunchKernelGGL(step_rearrange_kernel<1>, grid, block, 0, stream, gv, W, H, render);
    // (agent loops are real loops: one multi-agent build)
    else hipLaunchKernelGGL(step_rearrange_kernel<MAX_AGENTS>, grid, block, 0, stream, gv, W, H, render);
}

void launch_reset_rearrange(const GymView &gv, const RearrangeBlob *blobs, int *status, int force_all, hipStream_t stream)
{
    hipLaunchKernelGGL(reset_rearrange_kernel, dim3(gv.num_envs), dim3(64), 0, stream, gv, blobs, status, force_all);
}

}  // namespace mv
