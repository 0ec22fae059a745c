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
#include <hip/hip_ext.h>

#include <algorithm>
#include <cstdlib>

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

// k ticks of every env of every gym of the group, the env's workgroup resident for the whole call (cf. mv_step.hip: step_ticks_kernel): tick, frame setup
// into tick j's slot, tick, ...  Built for the register budget of the other resident multi-tick kernels (they run beside the observation passes of the
// previous call, and what they hold the passes cannot have).
// One wave per env -- except for the gyms with long frame lists (Collect, Hex*: up to 2048 visible primitives): their frame setup is most of their tick, one
// wave per env made their envs the launch's stragglers (57 us per tick where the short-list scenarios need 15-20: measured r08i), so the launch has WAVES waves
// per workgroup, the long-list gyms use them all for the frame setup (wave 0 ticks, the others wait at the barrier) and the other gyms' extra waves leave at
// once.
#ifndef MV_UNION_TICKS_WAVES_PER_SIMD
// the register budget (512 / n): all eight scenarios' ticks in one kernel need ~175 VGPRs; at 128 it spills 1.1 KB per lane into the ticks' inner loops
#define MV_UNION_TICKS_WAVES_PER_SIMD 3
#endif
template <int WAVES>
__global__ __launch_bounds__(64 * WAVES, MV_UNION_TICKS_WAVES_PER_SIMD) void step_union_ticks_kernel(UnionTicksArgs ua, int W, int H)
{
    __shared__ FrameScratch s_fs;
    __shared__ DepthSortScratch s_ds;
    int s = 0;
#pragma unroll
    for (int i = 1; i < MAX_UNION; ++i)
        if (i < ua.n && (int)blockIdx.x >= ua.first[i]) s = i;
    const int env = (int)blockIdx.x - ua.first[s];
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const bool wide = WAVES > 1 && ua.gv[s].vis_stride > VIS_SMALL;   // (uniform over the workgroup)
    if (!wide && wave > 0) return;
    for (int j = 0; j < ua.k; ++j) {
        const GymView gv = tick_view(ua.gv[s], ua.slot_stride[s], j);
        if (wave == 0) {
            switch (gv.scenario) {   // (uniform per workgroup)
            case SCN_TOWER: tick_tower::tower_tick<1>(gv, env); break;
            case SCN_OBSTACLES:
            case SCN_EMPTY: tick_obstacles::obstacles_tick<1>(gv, env); break;
            case SCN_COLLECT: tick_collect::collect_tick<1>(gv, env); break;
            case SCN_REARRANGE: tick_rearrange::rearrange_tick<1>(gv, env); break;
            case SCN_SOKOBAN: tick_sokoban::sokoban_tick<1>(gv, env); break;
            default: tick_hex::hex_tick<1>(gv, env); break;   // SCN_HEX_MEMORY, SCN_HEX_EXPLORE
            }
        }
        if (wide) {
            __syncthreads();   // the tick's stores (same CU: same L1) before the frame setup's loads
            // (ends with a barrier: the next tick starts when every wave is done with the state)
            frame_setup_body<64 * WAVES, false>(gv, env, W, H, s_fs, &s_ds);
        } else {
            wave_sync();   // one wave: no barrier needed
            frame_setup_body<64, true>(gv, env, W, H, s_fs, &s_ds);
        }
    }
}

void launch_step_union_ticks(const UnionTicksArgs &ua0, hipStream_t stream, int W, int H, hipEvent_t done)
{
    bool anyLong = false;
    for (int i = 0; i < ua0.n; ++i) anyLong = anyLong || ua0.gv[i].vis_stride > VIS_SMALL;
    // The gyms with long frame lists LAST, whatever the group's order.  Workgroups start in the order of their ids, and a long-list env's workgroup is the
    // launch's heaviest (four waves of 168 VGPRs for the whole call): started first they take the chip from the previous call's observation launch, which the
    // short ones beside them leave room for -- Mixed 64 x 64 (eight scenarios x 128 envs), long lists first / in MEGAVERSE8's order / last: 15.3-17.2 /
    // 19.1-19.5 / 19.5-19.6 M obs/s (r10m).  (Which block steps which env is this launch's own business: every env's results go to its own gym's arrays.)
    static const int longFirst = getenv("MV_UNION_LONG_FIRST") ? atoi(getenv("MV_UNION_LONG_FIRST")) : 2;   // 1: long lists first, 2: last, 0: the group's order
    UnionTicksArgs ua = ua0;
    if (anyLong && longFirst) {
        int m = 0, envs = 0;
        for (int pass = 0; pass < 2; ++pass)
            for (int i = 0; i < ua0.n; ++i)
                if ((ua0.gv[i].vis_stride > VIS_SMALL) == (pass == (longFirst == 2 ? 1 : 0))) {
                    ua.gv[m] = ua0.gv[i]; ua.slot_stride[m] = ua0.slot_stride[i]; ua.first[m] = envs;
                    envs += ua0.first[i + 1] - ua0.first[i];
                    ++m;
                }
        for (int i = m; i <= MAX_UNION; ++i) ua.first[i] = envs;
    }
    static const int wideWaves = getenv("MV_UNION_TICKS_WAVES") ? atoi(getenv("MV_UNION_TICKS_WAVES")) : 4;   // (2: measured in r09f)
    if (anyLong && wideWaves == 2) hipExtLaunchKernelGGL(step_union_ticks_kernel<2>, dim3(ua.first[ua.n]), dim3(128), 0, stream, nullptr, done, 0, ua, W, H);
    else if (anyLong) hipExtLaunchKernelGGL(step_union_ticks_kernel<4>, dim3(ua.first[ua.n]), dim3(256), 0, stream, nullptr, done, 0, ua, W, H);
    else hipExtLaunchKernelGGL(step_union_ticks_kernel<1>, dim3(ua.first[ua.n]), dim3(64), 0, stream, nullptr, done, 0, ua, W, H);
}

void launch_step_union(const UnionStepArgs &ua, hipStream_t stream, int W, int H, int render)
{
    const int A = ua.gv[0].num_agents;
    const dim3 grid(ua.first[ua.n]), block(A == 1 ? STEP_THREADS : 64 * std::min(A, 4));
    if (A == 1) hipLaunchKernelGGL(step_union_kernel<1>, grid, block, 0, stream, ua, W, H, render);
    else hipLaunchKernelGGL(step_union_kernel<MAX_AGENTS>, grid, block, 0, stream, ua, W, H, render);
}

}  // namespace mv
