// megaverse_amd/csrc/mv_rearrange.h -- the fixed geometry of the Rearrange scenario, shared by its step kernel and the raster.
//   reference: src/libs/scenarios/src/scenario_rearrange.cpp:203-299, include/scenarios/scenario_rearrange.hpp:130-131
#pragma once
#include <hip/hip_runtime.h>

#include "mv_math.h"
#include "mv_types.h"

namespace mv {
namespace {

constexpr int RE_LEFT_X = 5, RE_LEFT_Y = 2, RE_LEFT_Z = 5, RE_RIGHT_X = 13, RE_RIGHT_Y = 2, RE_RIGHT_Z = 5;   // leftCenter, rightCenter
constexpr int ROOM_L = 19, ROOM_W = 14;                                                                       // RearrangePlatform::init

// drawable scale of an item: scales[shape] * objSize (0.45) (:205-212)
__device__ __forceinline__ V3 item_draw_scale(int shape)
{
    const float s = 0.45f;
    if (shape == SHAPE_CAPSULE) return v3(0.8f * s, 0.5f * s, 0.8f * s);
    if (shape == SHAPE_CYLINDER) return v3(0.9f * s, 2.0f * s, 0.9f * s);
    return v3(1.0f * s, 1.0f * s, 1.0f * s);
}
// half extents of its collision box: the drawable scale times the collision scale (cylinder (1, 0.5, 1), capsule (1, 2, 1), :231-236)
__device__ __forceinline__ V3 item_collision_half(int shape)
{
    const V3 d = item_draw_scale(shape);
    if (shape == SHAPE_CAPSULE) return v3(d.x * 1.0f, d.y * 2.0f, d.z * 1.0f);
    if (shape == SHAPE_CYLINDER) return v3(d.x * 1.0f, d.y * 0.5f, d.z * 1.0f);
    return d;
}

// the 9 static colliding boxes (:285-298), centre +- half extents: k = 0 raised floor, 1-4 left pedestal, 5-8 right pedestal
__device__ __forceinline__ unsigned static_box(int k, V3 &lo, V3 &hi)
{
    V3 half, c;
    unsigned color = 0x555555u;
    if (k == 0) { half = v3(8.35f, 0.5f, 5.65f); c = v3(9.5f + 0.0f, 0.0f + 1.0f, 7.0f + 0.0f); }
    else {
        const bool left = k <= 4;
        const int j = left ? k - 1 : k - 5;
        const V3 base = left ? v3(float(RE_LEFT_X), float(RE_LEFT_Y), float(RE_LEFT_Z)) : v3(float(RE_RIGHT_X), float(RE_RIGHT_Y), float(RE_RIGHT_Z));
        half = j == 1 ? v3(1.5f, 0.5f, 1.5f) : v3(3.0f, 0.5f, 3.0f);
        const float dx = j <= 1 ? 0.5f : left ? (j == 2 ? 1.0f : 1.5f) : (j == 2 ? 0.0f : -0.5f);
        const float dy = j == 0 ? -0.5f : j == 1 ? -0.45f : j == 2 ? -0.66f : -0.82f;
        const float dz = j <= 1 ? 0.5f : j == 2 ? 1.0f : 1.5f;
        c = v3(base.x + dx, base.y + dy, base.z + dz);
        if (j != 1) color = left ? 0xffffffu : 0x2eb5d0u;   // LAYOUT_DEFAULT / BLUE; the middle slab is DARK_GREY
    }
    lo = v3(c.x - half.x, c.y - half.y, c.z - half.z);
    hi = v3(c.x + half.x, c.y + half.y, c.z + half.z);
    return color;
}

}  // namespace
}  // namespace mv
