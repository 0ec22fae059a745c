// megaverse_amd/csrc/mv_union.h -- launch arguments of the union kernels: several gyms' views side by side, and which workgroups are whose
#pragma once
#include <hip/hip_runtime.h>

#include "mv_types.h"

namespace mv {

enum : int { MAX_UNION = 8 };   // gyms per group (the reference's multi-task set has eight scenarios, megaverse_env.py:18-21)
enum : int { MAX_GROUP_TICKS = 8 };   // ticks of one batched group call's two launches (k x n output pointers travel as kernel arguments)

struct UnionStepArgs {
    int32_t n;                      // gyms
    int32_t first[MAX_UNION + 1];   // first workgroup (= env) of gym s; first[n] = total
    GymView gv[MAX_UNION];
};

static_assert(sizeof(UnionStepArgs) + 16 <= 4096, "UnionStepArgs + (W, H, render) must fit the 4 KB kernel-argument segment");

// k consecutive ticks of every gym of a group with ONE launch (step_union_ticks_kernel): tick 0's view of every gym; tick j's differs from it in its
// hand-over slot -- ten buffers, all `slot_stride` bytes further per tick (mv_api.hip carves a gym's slots out of its arena one after the other) --, its
// action index and its cost histogram (consecutive, modulo their number): derived in the kernel (mv_types.h: tick_view), not passed (k x n views do not fit the
// 4 KB of kernel arguments).
struct UnionTicksArgs {
    int32_t n, k;
    int32_t first[MAX_UNION + 1];
    int64_t slot_stride[MAX_UNION];
    GymView gv[MAX_UNION];
};
static_assert(sizeof(UnionTicksArgs) + 16 <= 4096, "UnionTicksArgs + (W, H) must fit the 4 KB kernel-argument segment");

// done: completed by the launch's own dispatch packet   // (one agent per env)
void launch_step_union_ticks(const UnionTicksArgs &ua, hipStream_t stream, int W, int H, hipEvent_t done = nullptr);

void launch_step_union(const UnionStepArgs &ua, hipStream_t stream, int W, int H, int render);

}  // namespace mv
