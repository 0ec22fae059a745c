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

// called by all 64 lanes of one single-wave workgroup.  Bins are bytes; a lane fetches 8 x 16 of them per round with
// independent 16-byte loads (the loop is latency-bound: one dependent load per frame made it 50 us for 8192 frames).
__device__ __forceinline__ void sort_frames_by_cost(const GymView &gv)
{
    __shared__ int s_hist[LPT_BUCKETS], s_start[LPT_BUCKETS];
    const int lane = threadIdx.x, frames = gv.num_envs * gv.num_agents;
    const uint4 *bins = reinterpret_cast<const uint4 *>(gv.lpt_bucket);   // the array is padded to a multiple of 4096 bytes
    for (int b = lane; b < LPT_BUCKETS; b += 64) s_hist[b] = 0;
    __syncthreads();
#pragma unroll 1
    for (int pass = 0; pass < 2; ++pass) {
#pragma unroll 1
        for (int base = 0; base < frames; base += 8 * 64 * 16) {
            uint4 v[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const int first = base + (k * 64 + lane) * 16;
                v[k] = first < frames ? bins[first >> 4] : make_uint4(0, 0, 0, 0);
            }
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const int first = base + (k * 64 + lane) * 16;
                const unsigned w[4] = {v[k].x, v[k].y, v[k].z, v[k].w};
#pragma unroll
                for (int j = 0; j < 16; ++j) {
                    const int f = first + j;
                    if (f < frames) {
                        const int bin = (w[j >> 2] >> (8 * (j & 3))) & 255;
                        if (pass == 0) atomicAdd(&s_hist[bin], 1);
                        else gv.lpt_order[atomicAdd(&s_start[bin], 1)] = f;
                    }
                }
            }
        }
        __syncthreads();
        if (pass == 0) {
            if (lane == 0) {   // 256 bins: a serial scan is a few hundred cycles
                int acc = 0;
                for (int b = LPT_BUCKETS - 1; b >= 0; --b) { s_start[b] = acc; acc += s_hist[b]; }
            }
            __syncthreads();
        }
    }
}

}  // namespace
}  // namespace mv
