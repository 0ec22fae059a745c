// megaverse_amd/csrc/mv_step_sokoban.hip -- one simulation tick + episode swap-in for Sokoban.
// The tick itself (physics, scenario logic, episode swap-in) lives in mv_tick_sokoban.h; this file holds the kernels and their launchers.
//
// Replaces, per env (reference paths relative to src/libs):
//   Env::step                                   env/src/env.cpp:83-152            (shared pieces: mv_physics.h)
//   SokobanScenario::step (push logic)          scenarios/src/scenario_sokoban.cpp:172-236
//   SokobanScenario::addEpisodeDrawables        scenarios/src/scenario_sokoban.cpp:243-294  (collision shape of the boxes :275-293; drawing: mv_frame.h)
//   Scenario::rewardTeam / doneWithTimer        env/include/env/scenario.hpp:114-117,259-307
//   VectorEnv::step done bookkeeping + Env::reset of finished envs (env/src/vector_env.cpp:93-105): the reset kernel below swaps in
//   the episode the host generator (mv_gen_sokoban.cpp: Boxoban level files, per-env shuffled level list) left resident in HBM.
//
// Same mapping as the Obstacles kernel (one wavefront per env, four colliders per lane): <= 128 merged layout slabs (the floor and the
// undrawn two-voxel walls; one voxel is 2 units wide), <= 80 pushable boxes, <= 8 agent capsules.  There is no ObjectStackingComponent
// and no FallDetectionComponent in this scenario: the Interact action pushes the box in front of the agent one cell further when the
// agent stands in the adjacent cell and the cell behind the box is free (no wall, no box, no agent).  Box lookups are ballots over the
// lanes that hold the boxes; wall / goal lookups read the level's 32 x 32 cell map.
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>

#include "mv_tick_sokoban.h"

namespace mv {

using namespace tick_sokoban;

// One workgroup per env: wave 0 runs the tick, then the workgroup builds the lists of the env's frames (see mv_step.hip)
template <int A_MAX>
__global__ __launch_bounds__(256) void step_sokoban_kernel(GymView gv, int W, int H, int render)
{
    __shared__ FrameScratch s_fs[A_MAX == 1 ? 1 : 4];
    const int env = blockIdx.x;
    if (threadIdx.x < 64) sokoban_tick<A_MAX>(gv, env);
    if (!render) return;
    __syncthreads();   // the tick's stores (same CU: same L1) before the frame setup's loads
    if (A_MAX == 1) frame_setup_body<STEP_THREADS, false>(gv, env, W, H, s_fs[0]);
    else {
        const int A = gv.num_agents, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), nw = blockDim.x >> 6;
        for (int a = wave; a < A; a += nw) frame_setup_body<64, true>(gv, env * A + a, W, H, s_fs[wave]);
    }
}

__global__ __launch_bounds__(64) void reset_sokoban_kernel(GymView gv, const SokobanBlob *blobs, int *status, int force_all)
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
__global__ __launch_bounds__(64, MV_STEP_TICKS_WAVES_PER_SIMD) void step_sokoban_ticks_kernel(Args a, int W, int H)
{
    __shared__ FrameScratch s_fs;
    const int env = blockIdx.x;
    for (int j = 0; j < a.n; ++j) {
        const GymView &gv = a.view(j);
        sokoban_tick<1>(gv, env);
        wave_sync();   // the tick's stores before the frame setup's loads (one wave: no barrier needed)
        frame_setup_body<64, true>(gv, env, W, H, s_fs);
    }
}

void launch_step_sokoban_ticks(const GymView *views, int k, hipStream_t stream, int W, int H, hipEvent_t done)
{
    StepTicksArgs8 a8;   // (k <= 8: the views are the launch's arguments, mv_types.h)
    a8.n = k; a8.pad = 0;
    for (int j = 0; j < 8; ++j) a8.gv[j] = views[std::min(j, k - 1)];
    hipExtLaunchKernelGGL(step_sokoban_ticks_kernel<StepTicksArgs8>, dim3(views[0].num_envs), dim3(64), 0, stream, nullptr, done, 0, a8, W, H);
}

void launch_step_sokoban(const GymView &gv, hipStream_t stream, int W, int H, int render)
{
    const dim3 grid(gv.num_envs), block(gv.num_agents == 1 ? STEP_THREADS : 64 * std::min(gv.num_agents, 4));
    if (gv.num_agents == 1) hipLaunchKernelGGL(step_sokoban_kernel<1>, grid, block, 0, stream, gv, W, H, render);
    else hipLaunchKernelGGL(step_sokoban_kernel<MAX_AGENTS>, grid, block, 0, stream, gv, W, H, render);
}

void launch_reset_sokoban(const GymView &gv, const SokobanBlob *blobs, int *status, int force_all, hipStream_t stream)
{
    hipLaunchKernelGGL(reset_sokoban_kernel, dim3(gv.num_envs), dim3(64), 0, stream, gv, blobs, status, force_all);
}

}  // namespace mv
