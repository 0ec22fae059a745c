// megaverse_amd/csrc/mv_step_union.hip -- ONE step launch for several gyms of a job (mv_group, include/megaverse_hip.h).
//
// The reference runs a multi-task job as one MegaverseGym per scenario (megaverse/megaverse_env.py:27-39, make_env_multitask picks
// tasks[task_idx % len(tasks)] per worker); BASELINE.json configs[4] deals the scenarios round-robin over the envs of one batch.  Here every
// scenario keeps its own gym -- own arena, own episode feeder, own refill protocol, its kernels' own data layout -- and only the LAUNCH is
// shared: workgroup b of the union grid belongs to gym s with first[s] <= b < first[s + 1], env b - first[s], and runs that gym's tick
// (the same device functions the per-scenario step kernels call: mv_tick_*.h) and frame setup with that gym's view.  Eight launches, eight
// kernel boundaries and eight sets of stream hand-overs per tick become one; the launch lasts as long as the slowest env of any scenario.
// Registers and LDS are the maxima over the scenarios' ticks (176 VGPRs, ~30 KB): two waves per SIMD, four workgroups per CU -- 1024 envs of
// one agent are still resident at once.
#include <hip/hip_runtime.h>

#include <algorithm>

#include "mv_tick_collect.h"
#include "mv_tick_hex.h"
#include "mv_tick_obstacles.h"
#include "mv_tick_rearrange.h"
#include "mv_tick_sokoban.h"
#include "mv_tick_tower.h"
#include "mv_union.h"

namespace mv {

template <int A_MAX>
__global__ __launch_bounds__(256) void step_union_kernel(UnionStepArgs ua, int W, int H, int render)
{
    __shared__ FrameScratch s_fs[A_MAX == 1 ? 1 : 4];
    __shared__ DepthSortScratch s_ds[A_MAX == 1 ? 1 : 4];   // (long lists: mv_frame.h)
    int s = 0;
#pragma unroll
    for (int i = 1; i < MAX_UNION; ++i)
        if (i < ua.n && (int)blockIdx.x >= ua.first[i]) s = i;
    const GymView &gv = ua.gv[s];
    const int env = (int)blockIdx.x - ua.first[s];
    if (threadIdx.x < 64) {
        switch (gv.scenario) {   // (uniform per workgroup)
        case SCN_TOWER: tick_tower::tower_tick<A_MAX>(gv, env); break;
        case SCN_OBSTACLES:
        case SCN_EMPTY: tick_obstacles::obstacles_tick<A_MAX>(gv, env); break;
        case SCN_COLLECT: tick_collect::collect_tick<A_MAX>(gv, env); break;
        case SCN_REARRANGE: tick_rearrange::rearrange_tick<A_MAX>(gv, env); break;
        case SCN_SOKOBAN: tick_sokoban::sokoban_tick<A_MAX>(gv, env); break;
        default: tick_hex::hex_tick<A_MAX>(gv, env); break;   // SCN_HEX_MEMORY, SCN_HEX_EXPLORE
        }
    }
    if (!render) return;
    __syncthreads();   // the tick's stores (same CU: same L1) before the frame setup's loads
    if (A_MAX == 1) frame_setup_body<STEP_THREADS, false>(gv, env, W, H, s_fs[0], &s_ds[0]);
    else {
        const int A = gv.num_agents, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), nw = blockDim.x >> 6;
        for (int a = wave; a < A; a += nw) frame_setup_body<64, true>(gv, env * A + a, W, H, s_fs[wave], &s_ds[wave]);
    }
}

void launch_step_union(const UnionStepArgs &ua, hipStream_t stream, int W, int H, int render)
{
    const int A = ua.gv[0].num_agents;
    const dim3 grid(ua.first[ua.n]), block(A == 1 ? STEP_THREADS : 64 * std::min(A, 4));
    if (A == 1) hipLaunchKernelGGL(step_union_kernel<1>, grid, block, 0, stream, ua, W, H, render);
    else hipLaunchKernelGGL(step_union_kernel<MAX_AGENTS>, grid, block, 0, stream, ua, W, H, render);
}

}  // namespace mv
