// megaverse_amd/csrc/mv_raster.h -- host interface of the observation pass (mv_raster.hip)
#pragma once
#include <hip/hip_runtime.h>

#include "mv_types.h"

namespace mv {

// frame setup (+ frame sort, exact mode) + raster of every agent's W x H observation into `obs` on `stream`; `between` (optional) is
// recorded between the setup/sort kernels and the raster kernel; -1 if W/H are too large.  fast = 1: raster_fast_kernel (hardware
// reciprocals, frames looked up from the cost bins in its prologue; pixels within the tolerance DESIGN.md states), 0: the bit-exact
// raster_kernel behind frame_order_kernel.  setup_done = 1: the frame lists were built by the step kernel (mv_frame.h).
// publish (fast kernel only): the step's staged rewards / dones / true objectives (gv.rewards ...) are copied into these public arrays by the
// first workgroups of the raster launch -- ordered, on `stream`, with the observations (pipelined steps, mv_api.hip)
struct PublishTo { float *rewards; uint8_t *done; float *true_objective; };
// done: an event that completes when the pass does.  The fast kernels carry it as the completion signal of their own dispatch packet
// (hipExtLaunchKernelGGL's stop event) -- a separate hipEventRecord is one more packet the queue works off between two passes, ~5 us.
int launch_raster(const GymView &gv, uint32_t *obs, int W, int H, hipStream_t stream, hipEvent_t between = nullptr, int fast = 1, int setup_done = 0,
                  const PublishTo *publish = nullptr, hipEvent_t done = nullptr);

// the fast observation pass of n gyms of one job (frame lists already built by their step kernels) with at most two launches; publish: n
// entries or null; -1 if W / H / n are too large
int launch_raster_union(const GymView *views, uint32_t *const *obs, const PublishTo *publish, int n,
                        int W, int H, hipStream_t stream, hipEvent_t between = nullptr,
                        hipEvent_t done = nullptr);

// the fast observation passes of k ticks of ONE gym (a batched call; views[j] / obs[j] / publish[j] of tick j, the observation slabs distinct) with one
// launch: the next tick's expensive frames run in the tail of the previous tick's pass; 0: launched, 1: not applicable (the caller launches tick by tick)
int launch_raster_batch(const GymView *views, uint32_t *const *obs, const PublishTo *publish, int k,
                        int W, int H, hipStream_t stream, hipEvent_t done = nullptr);

// the fast observation passes of k ticks of n gyms of one job (a batched group call; views / obs / publish tick-major: [j * n + i]) with ONE launch.
// raster_union_batch_applicable: decided before the step launch (its frame setups leave the clearing of the cost histograms to the passes);
// launch_raster_union_batch: 0 launched, 1 not applicable, -1 / -2 bad sizes / views that are not one hand-over slot apart per tick
bool raster_union_batch_applicable(int k, int n, int W, int H);
int launch_raster_union_batch(const GymView *views, uint32_t *const *obs, const PublishTo *publish, int k,
                              int n, int W, int H, hipStream_t stream, hipEvent_t done = nullptr);

}  // namespace mv
