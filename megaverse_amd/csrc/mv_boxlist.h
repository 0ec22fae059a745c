// megaverse_amd/csrc/mv_boxlist.h -- wave-level helpers shared by the step kernels that keep the scene as lists
// (Obstacles, Collect): movable boxes two per lane, 128-bit column occupancy masks, reward bookkeeping.
//   ObjectStackingComponent queries   scenarios/include/scenarios/component_object_stacking.hpp:58-168
//   Scenario::rewardAgent/rewardTeam  env/include/env/scenario.hpp:259-307
#pragma once
#include <hip/hip_runtime.h>

#include "mv_math.h"
#include "mv_types.h"

namespace mv {
namespace {

struct Objs {          // lane l owns movable box l (k = 0) and, for l < 16, box 64 + l (k = 1)
    int x[2], y[2], z[2], state[2];
    bool valid[2];
};

struct Bits128 {       // bit (y + 32) for y in [-32, 95]
    unsigned long long lo, hi;
};
__device__ __forceinline__ unsigned long long low_bits(int n)   // n in [0, 64]: the n lowest bits
{
    return n >= 64 ? ~0ull : (1ull << n) - 1ull;
}
__device__ __forceinline__ void set_range(Bits128 &b, int y0, int y1)   // [y0, y1)
{
    const int lo = min(max(y0 + 32, 0), 128), hi = min(max(y1 + 32, 0), 128);
    if (lo >= hi) return;
    b.lo |= low_bits(min(hi, 64)) & ~low_bits(min(lo, 64));
    b.hi |= low_bits(max(hi - 64, 0)) & ~low_bits(max(lo - 64, 0));
}
// Where does a box dropped into cell y of a column come to rest?  (ObjectStackingComponent's "descend until the cell below is
// solid or holds an object, not below y = -30", component_object_stacking.hpp:92-105) -- loop-free: the highest occupied
// cell below y decides.  `occ` = solid | objects of the column.
__device__ __forceinline__ int drop_height(const Bits128 &occ, int y)
{
    const int idx = min(y + 32, 128);   // bit index of the start cell; everything above the window is free
    if (idx <= 0) return y;
    unsigned long long m;
    if (idx > 64) {
        m = occ.hi & low_bits(idx - 64);
        if (m) return (64 + (63 - __clzll((long long)m)) + 1) - 32;
        m = occ.lo;
    } else m = occ.lo & low_bits(idx);
    if (m) return ((63 - __clzll((long long)m)) + 1) - 32;
    return min(y, -30);
}
__device__ __forceinline__ bool test(const Bits128 &b, int y)
{
    const int i = y + 32;
    if (i < 0 || i >= 128) return false;
    return i < 64 ? ((b.lo >> i) & 1ull) : ((b.hi >> (i - 64)) & 1ull);
}
// (wave_or_u64: mv_math.h, the DPP reduction)

// placed movable boxes of column (x, z)
__device__ __forceinline__ Bits128 column_objects(const Objs &o, int x, int z)
{
    Bits128 m{0ull, 0ull};
#pragma unroll
    for (int k = 0; k < 2; ++k)
        if (o.valid[k] && o.state[k] == 0 && o.x[k] == x && o.z[k] == z) set_range(m, o.y[k], o.y[k] + 1);
    m.lo = wave_or_u64(m.lo); m.hi = wave_or_u64(m.hi);
    return m;
}

__device__ __forceinline__ int object_at(const Objs &o, int x, int y, int z)
{
    const bool h0 = o.valid[0] && o.state[0] == 0 && o.x[0] == x && o.y[0] == y && o.z[0] == z;
    const bool h1 = o.valid[1] && o.state[1] == 0 && o.x[1] == x && o.y[1] == y && o.z[1] == z;
    const unsigned long long m0 = __ballot(h0), m1 = __ballot(h1);
    if (m0) return __ffsll((long long)m0) - 1;
    if (m1) return 64 + (__ffsll((long long)m1) - 1);
    return -1;
}

// (Scenario::rewardAgent / rewardTeam live in mv_agents.h: the agent records of an env are in LDS during a tick)

__device__ __forceinline__ void voxel_of(V3 p, int out[3])
{
    out[0] = (int)floorf(p.x); out[1] = (int)floorf(p.y); out[2] = (int)floorf(p.z);
}

}  // namespace
}  // namespace mv
