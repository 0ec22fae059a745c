// megaverse_amd/csrc/mv_union.h -- launch arguments of the union kernels: several gyms' views side by side, and which workgroups are whose
#pragma once
#include <hip/hip_runtime.h>

#include "mv_types.h"

namespace mv {

enum : int { MAX_UNION = 8 };   // gyms per group (the reference's multi-task set has eight scenarios, megaverse_env.py:18-21)

struct UnionStepArgs {
    int32_t n;                      // gyms
    int32_t first[MAX_UNION + 1];   // first workgroup (= env) of gym s; first[n] = total
    GymView gv[MAX_UNION];
};

static_assert(sizeof(UnionStepArgs) + 16 <= 4096, "UnionStepArgs + (W, H, render) must fit the 4 KB kernel-argument segment");

void launch_step_union(const UnionStepArgs &ua, hipStream_t stream, int W, int H, int render);

}  // namespace mv
