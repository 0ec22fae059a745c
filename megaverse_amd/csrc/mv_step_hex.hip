// megaverse_amd/csrc/mv_step_hex.hip -- one simulation tick + episode swap-in for the HexMemory and HexExplore scenarios
// The tick itself (physics, scenario logic, episode swap-in) lives in mv_tick_hex.h; this file holds the kernels and their launchers.
// (members of the reference's multi-task set, scenarios/init.hpp:47-48).
//
// Replaces, per env (reference paths relative to src/libs):
//   Env::step                                   env/src/env.cpp:83-152            (shared pieces: mv_physics.h)
//   HexExploreScenario::step                    scenarios/src/scenario_hex_explore.cpp:43-58
//   HexMemoryScenario::step                     scenarios/src/scenario_hex_memory.cpp:84-127
//   VectorEnv::step done bookkeeping + Env::reset of finished envs (env/src/vector_env.cpp:93-105)
// (no ObjectStackingComponent, no FallDetectionComponent in these scenarios: "interact" does nothing, nobody is put back)
//
// The maze is a list of boxes, each axis-aligned in the world or in one of three frames rotated about Y (mv_gen_hex.cpp): the
// colliding ones -- the floor and up to 294 walls -- are the list's prefix.  As in Collect, the wave streams that prefix once per
// agent and tick (64 per pass), keeps the boxes that meet a conservative envelope of everything the capsule can do this tick, and
// compacts them in list order into an LDS candidate list; the shared controller then works on one candidate per lane.  The
// collectables (<= 128) live two per lane.
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>

#include "mv_tick_hex.h"

namespace mv {

using namespace tick_hex;

// One workgroup per env: wave 0 runs the tick, then the workgroup builds the lists of the env's frames (see mv_step_collect.hip)
template <int A_MAX>
__global__ __launch_bounds__(256) void step_hex_kernel(GymView gv, int W, int H, int render)
{
    __shared__ FrameScratch s_fs[A_MAX == 1 ? 1 : 4];
    __shared__ DepthSortScratch s_ds[A_MAX == 1 ? 1 : 4];   // (long lists: mv_frame.h)
    const int env = blockIdx.x;
    if (threadIdx.x < 64) hex_tick<A_MAX>(gv, env);
    if (!render) return;
    __syncthreads();
    if (A_MAX == 1) frame_setup_body<STEP_THREADS, false>(gv, env, W, H, s_fs[0], &s_ds[0]);
    else {
        const int A = gv.num_agents, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), nw = blockDim.x >> 6;
        for (int a = wave; a < A; a += nw) frame_setup_body<64, true>(gv, env * A + a, W, H, s_fs[wave], &s_ds[wave]);
    }
}

__global__ __launch_bounds__(64) void reset_hex_kernel(GymView gv, const HexBlob *blobs, int *status, int force_all)
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
__global__ __launch_bounds__(64, MV_STEP_TICKS_WAVES_PER_SIMD) void step_hex_ticks_kernel(Args a, int W, int H)
{
    __shared__ FrameScratch s_fs;
    __shared__ DepthSortScratch s_ds;
    const int env = blockIdx.x;
    for (int j = 0; j < a.n; ++j) {
        const GymView &gv = a.view(j);
        hex_tick<1>(gv, env);
        wave_sync();   // the tick's stores before the frame setup's loads (one wave: no barrier needed)
        frame_setup_body<64, true>(gv, env, W, H, s_fs, &s_ds);
    }
}

void launch_step_hex_ticks(const GymView *views, int k, hipStream_t stream, int W, int H, hipEvent_t done)
{
    StepTicksArgs8 a8;   // (k <= 8: the views are the launch's arguments, mv_types.h)
    a8.n = k; a8.pad = 0;
    for (int j = 0; j < 8; ++j) a8.gv[j] = views[std::min(j, k - 1)];
    hipExtLaunchKernelGGL(step_hex_ticks_kernel<StepTicksArgs8>, dim3(views[0].num_envs), dim3(64), 0, stream, nullptr, done, 0, a8, W, H);
}

void launch_step_hex(const GymView &gv, hipStream_t stream, int W, int H, int render)
{
    const dim3 grid(gv.num_envs), block(gv.num_agents == 1 ? STEP_THREADS : 64 * std::min(gv.num_agents, 4));
    if (gv.num_agents == 1) hipLaunchKernelGGL(step_hex_kernel<1>, grid, block, 0, stream, gv, W, H, render);
    else hipLaunchKernelGGL(step_hex_kernel<MAX_AGENTS>, grid, block, 0, stream, gv, W, H, render);
}

void launch_reset_hex(const GymView &gv, const HexBlob *blobs, int *status, int force_all, hipStream_t stream)
{
    hipLaunchKernelGGL(reset_hex_kernel, dim3(gv.num_envs), dim3(64), 0, stream, gv, blobs, status, force_all);
}

}  // namespace mv
