// megaverse_amd/csrc/mv_frame_order.h -- longest-processing-time-first order of the raster pass.
//
// frame_setup_kernel bins every agent frame by estimated raster cost (gv.lpt_bucket); this counting sort turns the bins into
// the permutation the raster kernel walks, most expensive frame first.  It runs as ONE EXTRA WORKGROUP of the step kernel
// (blockIdx.x == num_envs), i.e. concurrently with the physics of the next tick, on the bins of the previous observation
// pass: a frame's cost barely changes from one tick to the next, and any permutation is a correct order.
#pragma once
#include <hip/hip_runtime.h>

#include "mv_types.h"

namespace mv {
namespace {

constexpr int LPT_BUCKETS = 256;

// called by all 64 lanes of one single-wave workgroup
__device__ __forceinline__ void sort_frames_by_cost(const GymView &gv)
{
    __shared__ int s_hist[LPT_BUCKETS], s_start[LPT_BUCKETS];
    const int lane = threadIdx.x, frames = gv.num_envs * gv.num_agents;
    for (int b = lane; b < LPT_BUCKETS; b += 64) s_hist[b] = 0;
    __syncthreads();
    for (int f = lane; f < frames; f += 64) atomicAdd(&s_hist[min(max(gv.lpt_bucket[f], 0), LPT_BUCKETS - 1)], 1);
    __syncthreads();
    if (lane == 0) {   // 256 bins: a serial scan is a few hundred cycles
        int acc = 0;
        for (int b = LPT_BUCKETS - 1; b >= 0; --b) { s_start[b] = acc; acc += s_hist[b]; }
    }
    __syncthreads();
    for (int f = lane; f < frames; f += 64) gv.lpt_order[atomicAdd(&s_start[min(max(gv.lpt_bucket[f], 0), LPT_BUCKETS - 1)], 1)] = f;
}

}  // namespace
}  // namespace mv
