// megaverse_amd/csrc/mv_raster.hip -- first-person observation pass, written straight into the
// [N*A][H][W][4] RGBA8 observation slab in HBM.
//
// Replaces MagnumEnvRenderer::Impl::{reset,drawAgent,draw}
//   (reference: src/libs/magnum_rendering/src/magnum_env_renderer.cpp:158-340), i.e. per agent:
//   clear (0,0,0) + depth test + back-face culling + instanced Phong over the env's drawables
//   (layout slabs layout_utils.cpp:17-50, terrain slab :53-68, movable boxes
//   component_object_stacking.hpp:170-198, agent body/eyes/time bar scenario_default.hpp:99-170),
//   camera of env_renderer.hpp:34-38 / agent.cpp:33, then glReadPixels.  No GL here: every pixel
//   is one primary ray against the env's convex primitives, which picks the same front surface a
//   depth-tested rasteriser does.
//
// Mapping: one 256-thread workgroup per agent frame.
//   prologue (once per frame, all in LDS):
//     * agent cameras; per-column / per-row ray terms (the pixel -> ray arithmetic is separable);
//     * the frame's primitive list (<=128), each with its bounds relative to the ray origin of
//       its frame and its two Phong colour terms;
//     * frame-level visibility: every primitive's conservative screen rectangle from its 8 projected
//       corners; invisible ones are dropped, the few that cross the camera plane ("straddlers": the
//       floor and walls of the room the camera stands in) are kept apart; order-free compaction.
//   per 16x4-pixel tile (one wavefront):
//     * culling: rectangle-bounded primitives = 4 integer compares, one or two per lane; straddlers =
//       one (primitive, frustum plane) pair per lane; __ballot -> survivor masks;
//     * every lane intersects its pixel-centre ray with the survivors only (mask-bit iteration is
//       wave-uniform, the primitive is an LDS broadcast read), keeps (depth, slot) minimum;
//     * entry face of the winner -> normal, Phong, one RGBA8 dword store per lane.
// Culling is conservative (tile bounds at pixel edges + 1 px slack, rays at pixel centres), and
// equal-depth hits resolve to the lowest slot, so the image equals brute force over all primitives
// in slot order - which is what the CPU oracle does and the parity tests compare bit for bit.
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <math.h>
#include <stdlib.h>

#include <algorithm>
#include <cstring>
#include <numeric>
#include <vector>
#include <stdio.h>

#include "mv_frame.h"
#include "mv_math.h"
#include "mv_raster.h"
#include "mv_rearrange.h"
#include "mv_types.h"
#include "mv_union.h"

namespace mv {

namespace {

// Slab test against bounds given RELATIVE to the ray origin.  tn/tf per axis are returned through
// tEnter/tExit only; the entry axis is recovered for the winning primitive alone (entry_axis()).
// ANYZERO selects the exact handling of direction components that are exactly 0 (the oracle skips
// such an axis after an inside test); the common case has none and needs no selects.
template <bool ANYZERO>
__device__ __forceinline__ bool ray_box(V3 d, V3 inv, const float *lo, const float *hi, float &t_out)
{
    float tn[3], tf[3];
    const float dd[3] = {d.x, d.y, d.z}, ii[3] = {inv.x, inv.y, inv.z};
    bool outside = false;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const float t1 = lo[k] * ii[k], t2 = hi[k] * ii[k];
        tn[k] = __builtin_fminf(t1, t2);
        tf[k] = __builtin_fmaxf(t1, t2);
        if (ANYZERO && dd[k] == 0.0f) {
            outside |= (lo[k] > 0.0f) || (hi[k] < 0.0f);   // origin outside the slab and never entering it
            tn[k] = -INFINITY; tf[k] = INFINITY;
        }
    }
    const float tEnter = __builtin_fmaxf(__builtin_fmaxf(tn[0], tn[1]), tn[2]);
    const float tExit = __builtin_fminf(__builtin_fminf(tf[0], tf[1]), tf[2]);
    t_out = tEnter;
    return !outside && tEnter <= tExit && tEnter >= NEAR_Z && tEnter <= FAR_Z;
}

// first axis (x, y, z order) whose near-plane crossing equals the entry depth
__device__ __forceinline__ int entry_axis(V3 d, V3 inv, const float *lo, const float *hi, float tEnter)
{
    const float dd[3] = {d.x, d.y, d.z}, ii[3] = {inv.x, inv.y, inv.z};
    int axis = 2;
#pragma unroll
    for (int k = 2; k >= 0; --k) {
        const float tn = __builtin_fminf(lo[k] * ii[k], hi[k] * ii[k]);
        if (dd[k] != 0.0f && tn == tEnter) axis = k;
    }
    return axis;
}

// FAST variants (raster_fast_kernel): v_rcp_f32 / v_sqrt_f32 / v_rsq_f32 (1 ulp) instead of the correctly rounded sequences
template <bool FAST> __device__ __forceinline__ float f_inv(float x) { return FAST ? __builtin_amdgcn_rcpf(x) : 1.0f / x; }
template <bool FAST> __device__ __forceinline__ float f_div(float a, float b) { return FAST ? a * __builtin_amdgcn_rcpf(b) : a / b; }
template <bool FAST> __device__ __forceinline__ float f_sqrt(float x) { return FAST ? __builtin_amdgcn_sqrtf(x) : sqrtf(x); }
template <bool FAST> __device__ __forceinline__ float f_rsqrt(float x) { return FAST ? __builtin_amdgcn_rsqf(x) : 1.0f / sqrtf(x); }

template <bool FAST = false>
__device__ __forceinline__ bool ray_capsule(V3 o, V3 d, V3 c, float r, float hl, float &t_out, V3 &n_out)
{
    bool hit = false;
    float best = INFINITY; V3 bn = v3(0, 0, 0);
    const float ox = o.x - c.x, oz = o.z - c.z;
    const float A = d.x * d.x + d.z * d.z;
    if (A > 0.0f) {
        const float B = ox * d.x + oz * d.z;
        const float C = (ox * ox + oz * oz) - r * r;
        const float disc = B * B - A * C;
        if (disc >= 0.0f) {
            const float t = f_div<FAST>(-B - f_sqrt<FAST>(disc), A);
            const float y = o.y + t * d.y;
            if (t >= NEAR_Z && t <= FAR_Z && y >= c.y - hl && y <= c.y + hl) {
                hit = true; best = t;
                const float inv = f_inv<FAST>(r);
                bn = v3((ox + t * d.x) * inv, 0.0f, (oz + t * d.z) * inv);
            }
        }
    }
#pragma unroll
    for (int s = 0; s < 2; ++s) {
        const float cy = s == 0 ? c.y - hl : c.y + hl;
        const float oy = o.y - cy;
        const float A3 = (d.x * d.x + d.y * d.y) + d.z * d.z;
        const float B3 = (ox * d.x + oy * d.y) + oz * d.z;
        const float C3 = ((ox * ox + oy * oy) + oz * oz) - r * r;
        const float disc = B3 * B3 - A3 * C3;
        if (disc >= 0.0f) {
            const float t = f_div<FAST>(-B3 - f_sqrt<FAST>(disc), A3);
            const float y = o.y + t * d.y;
            const bool capSide = s == 0 ? (y <= cy) : (y >= cy);
            if (t >= NEAR_Z && t <= FAR_Z && capSide && t < best) {
                hit = true; best = t;
                const float inv = f_inv<FAST>(r);
                bn = v3((ox + t * d.x) * inv, (oy + t * d.y) * inv, (oz + t * d.z) * inv);
            }
        }
    }
    if (hit) { t_out = best; n_out = bn; }
    return hit;
}

// Unit capped cylinder (Primitives::cylinderSolid(.., halfLength 0.5, CapEnds), rendering/src/render_utils.cpp:30): radius 1, |y| <= hl
template <bool FAST = false>
__device__ __forceinline__ bool ray_cylinder_unit(V3 o, V3 d, float hl, float &t_out, V3 &n_out)
{
    bool hit = false;
    float best = INFINITY; V3 bn = v3(0, 0, 0);
    const float A = d.x * d.x + d.z * d.z;
    if (A > 0.0f) {
        const float B = o.x * d.x + o.z * d.z;
        const float C = (o.x * o.x + o.z * o.z) - 1.0f;
        const float disc = B * B - A * C;
        if (disc >= 0.0f) {
            const float t = f_div<FAST>(-B - f_sqrt<FAST>(disc), A);
            const float y = o.y + t * d.y;
            if (t >= NEAR_Z && t <= FAR_Z && y >= -hl && y <= hl) { hit = true; best = t; bn = v3(o.x + t * d.x, 0.0f, o.z + t * d.z); }
        }
    }
#pragma unroll
    for (int s = 0; s < 2; ++s) {   // caps: entered from outside only (back faces are culled)
        const float cy = s == 0 ? -hl : hl;
        const bool entering = s == 0 ? (d.y > 0.0f && o.y < cy) : (d.y < 0.0f && o.y > cy);
        if (entering) {
            const float t = f_div<FAST>(cy - o.y, d.y);
            const float x = o.x + t * d.x, z = o.z + t * d.z;
            if (t >= NEAR_Z && t <= FAR_Z && x * x + z * z <= 1.0f && t < best) { hit = true; best = t; bn = v3(0.0f, s == 0 ? -1.0f : 1.0f, 0.0f); }
        }
    }
    if (hit) { t_out = best; n_out = bn; }
    return hit;
}

// Scaled shapes: the ray goes into the shape's unit space (diagonal scale: t is unchanged), the unit-space normal comes back
// through the inverse-transpose scale.  `rel` = shape centre minus the ray origin, both in the primitive's frame.
template <bool FAST = false>
__device__ __forceinline__ bool ray_scaled_shape(int kind, V3 d, V3 rel, V3 scale, float &t_out, V3 &n_out)
{
    const V3 oo = v3(f_div<FAST>(0.0f - rel.x, scale.x), f_div<FAST>(0.0f - rel.y, scale.y), f_div<FAST>(0.0f - rel.z, scale.z));
    const V3 dd = v3(f_div<FAST>(d.x, scale.x), f_div<FAST>(d.y, scale.y), f_div<FAST>(d.z, scale.z));
    V3 nl = v3(0, 0, 0);
    bool hit;
    if (kind == PRIM_CYLINDER_S) hit = ray_cylinder_unit<FAST>(oo, dd, 0.5f, t_out, nl);
    else hit = ray_capsule<FAST>(oo, dd, v3(0, 0, 0), 1.0f, kind == PRIM_CAPSULE_S ? 1.0f : 0.0f, t_out, nl);
    if (!hit) return false;
    V3 n = v3(f_div<FAST>(nl.x, scale.x), f_div<FAST>(nl.y, scale.y), f_div<FAST>(nl.z, scale.z));
    n = n * f_rsqrt<FAST>(len2(n));
    n_out = n;
    return true;
}

// open cone (no base cap), apex a, axis +-y; only its outside is visible (back-face culling).  Diamonds of the
// Obstacles scenarios: two of these base to base (layout_utils.cpp:114-126)
template <bool FAST = false>
__device__ __forceinline__ bool ray_cone(V3 o, V3 d, V3 a, float r, float h, float dirSign, float &t_out, V3 &n_out)
{
    const float k = f_div<FAST>(r, h) * f_div<FAST>(r, h);
    const float ox = o.x - a.x, oz = o.z - a.z;
    const float s0 = dirSign * (a.y - o.y);
    const float ds = -dirSign * d.y;
    const float A = (d.x * d.x + d.z * d.z) - k * (ds * ds);
    const float B = (ox * d.x + oz * d.z) - k * (s0 * ds);
    const float C = (ox * ox + oz * oz) - k * (s0 * s0);
    if (A == 0.0f) return false;
    const float disc = B * B - A * C;
    if (!(disc >= 0.0f)) return false;
    const float sq = f_sqrt<FAST>(disc);
    const float invA = FAST ? __builtin_amdgcn_rcpf(A) : 0.0f;
    bool hit = false;
    float best = INFINITY; V3 bn = v3(0, 0, 0);
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const float t = FAST ? (i == 0 ? (-B - sq) : (-B + sq)) * invA : (i == 0 ? (-B - sq) : (-B + sq)) / A;
        const float s = s0 + t * ds;
        if (!(t >= NEAR_Z && t <= FAR_Z && s >= 0.0f && s <= h && t < best)) continue;
        const float px = ox + t * d.x, pz = oz + t * d.z;
        V3 n = v3(px, dirSign * (k * s), pz);
        if (!(dot(n, d) < 0.0f)) continue;
        const float l2 = len2(n);
        if (!(l2 > 0.0f)) continue;
        n = n * f_rsqrt<FAST>(l2);
        hit = true; best = t; bn = n;
    }
    if (hit) { t_out = best; n_out = bn; }
    return hit;
}

__device__ __forceinline__ float pow300(float x)
{
    const float x2 = x * x, x4 = x2 * x2, x8 = x4 * x4, x16 = x8 * x8, x32 = x16 * x16, x64 = x32 * x32, x128 = x64 * x64,
                x256 = x128 * x128;
    return ((x256 * x32) * x8) * x4;
}

__device__ __forceinline__ unsigned to_u8(float v)
{
    v = fmin_sel(fmax_sel(v, 0.0f), 1.0f);
    return (unsigned)(int)(v * 255.0f + 0.5f);   // v >= 0: truncation == floor
}

__device__ __forceinline__ V3 safe_inv(V3 d)
{
    return v3(d.x == 0.0f ? 0.0f : 1.0f / d.x, d.y == 0.0f ? 0.0f : 1.0f / d.y, d.z == 0.0f ? 0.0f : 1.0f / d.z);
}


}  // namespace


// ---- pass 1 as a kernel of its own (mv_reset, mv_render, hires): one workgroup per frame (mv_frame.h)
__global__ __launch_bounds__(256) void frame_setup_kernel(GymView gv, int W, int H)
{
    __shared__ FrameScratch s_fs;
    __shared__ DepthSortScratch s_ds;
    frame_setup_body<256, false>(gv, blockIdx.x, W, H, s_fs, &s_ds);
}

// ---- pass 1b, one workgroup: counting sort of the frames by cost bin, most expensive first.  (A "last workgroup of
// frame_setup_kernel does it" variant was slower: its device-scope fences write back every XCD's L2, 22 us vs 6 us.)
__global__ __launch_bounds__(1024) void frame_order_kernel(GymView gv, int frames, int *order)
{
    __shared__ int s_hist[LPT_BUCKETS], s_start[LPT_BUCKETS], s_wsum[LPT_BUCKETS / 64];
    const int tid = threadIdx.x;
    if (tid < LPT_BUCKETS) s_hist[tid] = 0;
    __syncthreads();
    for (int f = tid; f < frames; f += 1024) atomicAdd(&s_hist[gv.lpt_bucket[f]], 1);
    __syncthreads();
    {   // start of every bin in descending cost order: exclusive prefix sum over r = 255 - bin, one thread per bin (a serial scan by one
        // thread was most of this kernel's 5 us)
        const int lane = tid & 63, wave = tid >> 6;
        const int b = LPT_BUCKETS - 1 - tid;
        const int h = tid < LPT_BUCKETS ? s_hist[b] : 0;
        int x = h;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const int y = __shfl_up(x, off, 64);
            if (lane >= off) x += y;
        }
        if (lane == 63 && tid < LPT_BUCKETS) s_wsum[wave] = x;
        __syncthreads();
        if (tid < LPT_BUCKETS) {
            int base = 0;
            for (int w = 0; w < wave; ++w) base += s_wsum[w];
            s_start[b] = base + x - h;
        }
    }
    __syncthreads();
    for (int f = tid; f < frames; f += 1024) order[atomicAdd(&s_start[gv.lpt_bucket[f]], 1)] = f;
}

template <int MAXVIS, bool SHAPES>   // SHAPES: the frame may hold scaled spheres / capsules / cylinders (Rearrange)
__global__ __launch_bounds__(256, MAXVIS <= 256 && !SHAPES ? MV_RASTER_WAVES : MAXVIS <= 256 ? 5 : MAXVIS <= 1024 ? 2
                             : 1) void raster_kernel(GymView gv, uint32_t *obs, int W, int H, int split, const int *order)
{
    constexpr int ROUNDS = MAXVIS / 64;
    extern __shared__ __attribute__((aligned(16))) unsigned char s_dyn[];   // column/row ray tables
    __shared__ Prim s_vis[MAXVIS];       // this frame's visible list
    __shared__ short4 s_rect[MAXVIS];    // x0,x1,y0,y1 (pixels)
    __shared__ CamL s_cam[MAX_CAMS];
    __shared__ float s_lutA[256], s_lutD[256];   // colour byte -> AMB * c / 255 and (DIF * c / 255) * LCOL

    float4 *s_col = reinterpret_cast<float4 *>(s_dyn);   // per column i: (dc.x, c00*dc.x, c10*dc.x, c20*dc.x)
    float4 *s_row = s_col + W;                            // per row j:    (dc.y, c01*dc.y, c11*dc.y, c21*dc.y)
    float *s_colinv = reinterpret_cast<float *>(s_row + H);   // 1/dc.x per column (0 where dc.x == 0)
    float *s_rowinv = s_colinv + W;                            // 1/dc.y per row

    const int A = gv.num_agents;
    // `split` workgroups share one frame (interleaved tiles): frames differ up to 10x in cost, smaller work
    // units let the dispatcher level the load across CUs.  Workgroup ids are dealt round-robin over the 8 XCDs;
    // the parts of one frame are given ids that are congruent mod 8 so that they share one XCD's L2 (the frame's
    // list is read `split` times, neighbouring tiles write neighbouring lines).
    int frame, part;
    {
        const int per = 8 * split, group = blockIdx.x / per, r = blockIdx.x - group * per;
        frame = group * 8 + (r & 7); part = r >> 3;
        const int frames = gridDim.x / split;
        if (group * 8 + 8 > frames) { const int b = blockIdx.x - group * per; const int nf = frames - group * 8; frame = group * 8 + b % nf; part = b / nf; }
    }
    frame = order[frame];   // `frame` so far was a position in the cost-sorted order (most expensive first)
    const int env = frame / A, viewer = frame - env * A;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;

    // ---- frames of reference: the cameras (and, Hex scenarios, the wall orientations) as the frame setup left them in the frame's header --
    // the same values it built the list with, and nothing here reads simulator state that a later tick may already be changing (mv_api.hip)
    if (tid < MAX_CAMS) {
        const float *o = reinterpret_cast<const float *>(gv.vis_hdr + (size_t)frame * FRAME_HDR_BYTES) + FH_CAM + FH_CAM_STRIDE * tid;
        CamL cam;
        cam.eye[0] = o[0]; cam.eye[1] = o[1]; cam.eye[2] = o[2];
#pragma unroll
        for (int q = 0; q < 9; ++q) cam.c[q] = o[3 + q];
        cam.origin[0] = o[12]; cam.origin[1] = o[13]; cam.origin[2] = o[14];
        s_cam[tid] = cam;
    }
    __syncthreads();
    {   // separable ray terms: world dir = (c_k0*dc.x + c_k1*dc.y) + c_k2*(-1), dc = (xn*TAN, yn*TAN_Y, -1)
        const float *c = s_cam[viewer].c;
        for (int i = tid; i < W; i += 256) {
            const float dcx = (((float(i) + 0.5f) / float(W)) * 2.0f - 1.0f) * TAN_HALF_FOV;
            s_col[i] = make_float4(dcx, c[0] * dcx, c[3] * dcx, c[6] * dcx);
            s_colinv[i] = dcx == 0.0f ? 0.0f : 1.0f / dcx;
        }
        for (int j = tid; j < H; j += 256) {
            const float dcy = (((float(j) + 0.5f) / float(H)) * 2.0f - 1.0f) * TAN_HALF_FOV_Y;
            s_row[j] = make_float4(dcy, c[1] * dcy, c[4] * dcy, c[7] * dcy);
            s_rowinv[j] = dcy == 0.0f ? 0.0f : 1.0f / dcy;
        }
    }

    // ---- this frame's visible list (frame_setup_kernel) and the colour tables
    const int nVis = min(gv.vis_count[frame], (int)MAXVIS);
    {
        const uint4 *src = reinterpret_cast<const uint4 *>(reinterpret_cast<const Prim *>(gv.vis_prims) + (size_t)frame * gv.vis_stride);
        uint4 *dst = reinterpret_cast<uint4 *>(s_vis);
        for (int i = tid; i < nVis * 2; i += 256) dst[i] = src[i];
        const short4 *rs = reinterpret_cast<const short4 *>(gv.vis_rects) + (size_t)frame * gv.vis_stride;
        for (int i = tid; i < nVis; i += 256) s_rect[i] = rs[i];
        const float AMB = float(0x55) / 255.0f, DIF = float(0xbb) / 255.0f, LCOL = float(0xaa) / 255.0f;
        const float c = float(tid) / 255.0f;
        s_lutA[tid] = AMB * c;
        s_lutD[tid] = (DIF * c) * LCOL;
    }
    __syncthreads();   // cameras (incl. origin), ray tables, list, colour tables complete

    const CamL &cam = s_cam[viewer];
    const V3 eye = v3(cam.eye[0], cam.eye[1], cam.eye[2]);
    const int tilesX = (W + TILE_W - 1) / TILE_W, tilesY = (H + TILE_H * PPL - 1) / (TILE_H * PPL);
    const int numTiles = tilesX * tilesY;
    const float LIGHT[3] = {0.0f, 4.0f, 2.0f};
    uint32_t *out = obs + (size_t)frame * W * H;

    const float nzm[3] = {-cam.c[2], -cam.c[5], -cam.c[8]};   // c_k2 * (-1)

    // Every lane traces PPL pixels of a 16 x (4 PPL) tile: the rows py, py + 4, ...  (PPL > 1 shares the column terms, the
    // survivor records and the culling between rays and overlaps their dependency chains -- and lost, see MV_RASTER_PPL.)
    for (int tile = part * 4 + wave; tile < numTiles; tile += 4 * split) {
        const int ty = tile / tilesX, tx = tile - ty * tilesX;
        const int tx0 = tx * TILE_W, ty0 = ty * (TILE_H * PPL);
        const int tx1 = min(tx0 + TILE_W, W) - 1, ty1 = min(ty0 + TILE_H * PPL, H) - 1;

        // ---- tile culling: one primitive per lane per round of 64, four integer compares against its screen rectangle
        unsigned long long mk[ROUNDS];
        unsigned long long anyMask = 0ull;
#pragma unroll
        for (int k = 0; k < ROUNDS; ++k) {
            mk[k] = 0ull;
            if (k * 64 < nVis) {   // wave-uniform
                const int pos = lane + 64 * k;
                bool v = false;
                if (pos < nVis) {
                    const short4 r = s_rect[pos];
                    v = r.x <= tx1 && r.y >= tx0 && r.z <= ty1 && r.w >= ty0;
                }
                mk[k] = __ballot(v);
                anyMask |= mk[k];
            }
        }
        // ---- this lane's pixels and rays
        const int px = tx0 + (lane & (TILE_W - 1)), pyBase = ty0 + (lane / TILE_W);
        if (anyMask == 0ull) {   // nothing can be seen through this tile: clear colour (0,0,0), alpha 255
#pragma unroll
            for (int k = 0; k < PPL; ++k)
                if (px < W && pyBase + TILE_H * k < H) out[(size_t)(pyBase + TILE_H * k) * W + px] = 0xff000000u;
            continue;
        }
        const int pxc = min(px, W - 1);
        const float4 cx = s_col[pxc];
        const float cxinv = s_colinv[pxc];
        int pyc[PPL];
        V3 dc[PPL], dw[PPL], invW[PPL];
        bool zeroLane = false;
#pragma unroll
        for (int k = 0; k < PPL; ++k) {
            pyc[k] = min(pyBase + TILE_H * k, H - 1);
            const float4 ry = s_row[pyc[k]];
            dc[k] = v3(cx.x, ry.x, -1.0f);
            dw[k] = v3((cx.y + ry.y) + nzm[0], (cx.z + ry.z) + nzm[1], (cx.w + ry.w) + nzm[2]);
            invW[k] = safe_inv(dw[k]);
            zeroLane = zeroLane || dw[k].x == 0.0f || dw[k].y == 0.0f || dw[k].z == 0.0f;
        }
        const bool anyZero = __any(zeroLane);

        float best[PPL];
        int bestPos[PPL], bestSlot[PPL];
        V3 capN[PPL];   // normal of the best capsule/cone hit (boxes recompute theirs from the entry axis)
#pragma unroll
        for (int k = 0; k < PPL; ++k) { best[k] = INFINITY; bestPos[k] = -1; bestSlot[k] = 1 << 30; capN[k] = v3(0, 0, 0); }

#pragma unroll
        for (int half = 0; half < ROUNDS; ++half) {
            unsigned long long m = mk[half];
            while (m) {
                const int bit = __ffsll((long long)m) - 1;
                m &= m - 1;
                const int pos = bit + 64 * half;
                const Prim &q = s_vis[pos];
                const int qkind = q.meta & 15, qfr = (q.meta >> 4) & 15;
                const int qslot = (int)(q.meta >> 8);
#pragma unroll
                for (int k = 0; k < PPL; ++k) {
                    float t; V3 n = v3(0, 0, 0);
                    bool hit;
                    if (qkind == PRIM_CAPSULE) {
                        hit = ray_capsule(eye, dw[k], v3(q.lo[0], q.lo[1], q.lo[2]), q.hi[0], q.hi[1], t, n);
                    } else if (qkind == PRIM_CONE) {
                        hit = ray_cone(eye, dw[k], v3(q.lo[0], q.lo[1], q.lo[2]), q.hi[0], q.hi[1], q.hi[2], t, n);
                    } else if (SHAPES && qkind >= PRIM_SPHERE_S) {   // n comes back in the primitive's frame
                        const V3 df = qfr == 0 ? dw[k] : qfr == 1 + viewer ? dc[k] : mat_tmul(s_cam[qfr - 1].c, dw[k]);
                        hit = ray_scaled_shape(qkind, df, v3(q.lo[0], q.lo[1], q.lo[2]), v3(q.hi[0], q.hi[1], q.hi[2]), t, n);
                    } else if (qfr == 0) {
                        hit = anyZero ? ray_box<true>(dw[k], invW[k], q.lo, q.hi, t) : ray_box<false>(dw[k], invW[k], q.lo, q.hi, t);
                    } else if (qfr == 1 + viewer) {
                        hit = ray_box<true>(dc[k], v3(cxinv, s_rowinv[pyc[k]], -1.0f), q.lo, q.hi, t);
                    } else {
                        const V3 dk = mat_tmul(s_cam[qfr - 1].c, dw[k]);
                        hit = ray_box<true>(dk, safe_inv(dk), q.lo, q.hi, t);
                    }
                    // nearest hit; equal depth -> the primitive drawn first (lowest slot), as a strict "<" scan in slot order would
                    if (hit && (t < best[k] || (t == best[k] && qslot < bestSlot[k]))) { best[k] = t; bestPos[k] = pos; bestSlot[k] = qslot; capN[k] = n; }
                }
            }
        }

        // ---- Phong (Magnum Shaders::Phong, uniforms of magnum_env_renderer.cpp:200-203)
#pragma unroll
        for (int k = 0; k < PPL; ++k) {
            unsigned rgba = 0xff000000u;
            if (bestPos[k] >= 0) {
                const Prim &q = s_vis[bestPos[k]];
                const int qkind = q.meta & 15, qfr = (q.meta >> 4) & 15;
                V3 N;
                if (qkind != PRIM_BOX) {
                    if (!SHAPES || qfr == 0) N = mat_tmul(cam.c, capN[k]);
                    else if (qfr == 1 + viewer) N = capN[k];
                    else N = mat_tmul(cam.c, mat_mul(s_cam[qfr - 1].c, capN[k]));
                }
                else {
                    V3 d, inv;
                    if (qfr == 0) { d = dw[k]; inv = invW[k]; }
                    else if (qfr == 1 + viewer) { d = dc[k]; inv = v3(cxinv, s_rowinv[pyc[k]], -1.0f); }
                    else { d = mat_tmul(s_cam[qfr - 1].c, dw[k]); inv = safe_inv(d); }
                    const int axis = entry_axis(d, inv, q.lo, q.hi, best[k]);
                    const float dax = axis == 0 ? d.x : axis == 1 ? d.y : d.z;
                    const float sgn = dax > 0 ? -1.0f : 1.0f;
                    const V3 n = v3(axis == 0 ? sgn : 0.0f, axis == 1 ? sgn : 0.0f, axis == 2 ? sgn : 0.0f);
                    // C^T (sgn e_axis) == sgn * row `axis` of C (adding exact zeros changes nothing but the sign of a zero)
                    if (qfr == 0) N = v3(sgn * (axis == 0 ? cam.c[0] : axis == 1 ? cam.c[3] : cam.c[6]),
                                         sgn * (axis == 0 ? cam.c[1] : axis == 1 ? cam.c[4] : cam.c[7]),
                                         sgn * (axis == 0 ? cam.c[2] : axis == 1 ? cam.c[5] : cam.c[8]));
                    else if (qfr == 1 + viewer) N = n;
                    else N = mat_tmul(cam.c, mat_mul(s_cam[qfr - 1].c, n));
                }
                const V3 P = dc[k] * best[k];
                V3 Ld = v3(LIGHT[0] - P.x, LIGHT[1] - P.y, LIGHT[2] - P.z);
                Ld = Ld * (1.0f / sqrtf(len2(Ld)));
                const float intensity = fmax_sel(0.0f, dot(N, Ld));
                float spec = 0.0f;
                if (intensity > 0.001f) {
                    const float k2 = 2.0f * dot(N, Ld);
                    const V3 R = v3(k2 * N.x - Ld.x, k2 * N.y - Ld.y, k2 * N.z - Ld.z);
                    V3 Vd = v3(-P.x, -P.y, -P.z);
                    // shininess 300: cos^300 is below 1e-20 once cos < 0.86, i.e. far under half an ulp of the
                    // diffuse term it is added to (>= 0.04), so the addition is an exact no-op there.  Only
                    // pixels inside the highlight cone pay for the normalisation and the power.
                    const float vr = dot(Vd, R);
                    if (vr > 0.0f && vr * vr > 0.7225f * len2(Vd)) {   // cos > 0.85 (|R| == 1 up to rounding)
                        Vd = Vd * (1.0f / sqrtf(len2(Vd)));
                        spec = pow300(fmax_sel(0.0f, dot(Vd, R)));
                        spec = fmin_sel(fmax_sel(spec, 0.0f), 1.0f);
                    }
                }
                unsigned ch[3];
#pragma unroll
                for (int c = 0; c < 3; ++c) {
                    const unsigned byte = (q.color >> (16 - 8 * c)) & 255u;
                    ch[c] = to_u8((s_lutA[byte] + s_lutD[byte] * intensity) + spec);
                }
                rgba = ch[0] | (ch[1] << 8) | (ch[2] << 16) | 0xff000000u;
            }
            const int py = pyBase + TILE_H * k;
            if (px < W && py < H) out[(size_t)py * W + px] = rgba;
        }
    }
}

// =====================================================================================================================
// FAST observation pass (the default; DESIGN.md "pixel tolerance").  Same geometry, same rays (the per-pixel direction
// is bit-identical to the exact kernel's), same drawables and depth order -- but
//   * reciprocals / square roots are single v_rcp_f32 / v_rsq_f32 / v_sqrt_f32 (1 ulp) instead of the correctly rounded
//     sequences; a zero direction component needs no special case (lo * inf is +-inf, 0 * inf = NaN loses every min/max);
//   * (depth, list position) is ONE 32-bit key -- the depth's float bits (relative to NEAR_Z's) with the low POS_BITS replaced by the position --
//     so the nearest-hit update is a single v_min_u32 (depths closer than 2^-15 relative resolve to the earlier drawable,
//     like exact ties do);
//   * Phong for boxes works from per-face constants: for a planar face N.(L - P) = N.L - (plane offset), and the plane
//     offset is t * d_axis, so the normal never materialises: ndl = t |d_k| -+ Lrel_k ; |L - P|^2 is a quadratic in t with
//     per-column / per-row coefficients (L . dc = 4 dcy - 2, dc . dc = dcx^2 + dcy^2 + 1); one v_rsq_f32 per pixel;
//   * the highlight is evaluated only where cos > 0.97 (0.97^300 = 1e-4: 0.03 of one 8-bit step) with v_log/v_exp;
//   * colour = byte * (AMB + DIFL * intensity) + 255 * spec, rounded, saturated and packed by v_cvt_pk_u8_f32.
// Structure: `split` workgroups per frame (interleaved tile groups), launched most-expensive-frame-first -- the hardware
// dispatcher is the work queue.  The prologue is a plain copy (frame_setup_kernel leaves cameras, light vectors, world-box
// masks and the count in a per-frame header) with ONE barrier; the world-frame box loop fetches the next record from LDS while it
// intersects the current one; everything wave-uniform is kept in SGPRs explicitly.
// Pixels differ from the exact kernel / the oracle by at most one 8-bit step in ~5e-5 of the pixels, and by more only where a
// silhouette or a depth near-tie falls within rounding of a pixel centre, ~1e-6 (tests/test_fast_pixels_gpu.py).
// =====================================================================================================================
namespace {

#ifndef MV_GLIST_WAVES_NP2
#define MV_GLIST_WAVES_NP2 5   // waves per SIMD the long-list variants are compiled for (two pixels per lane / one)
#endif
#ifndef MV_GLIST_WAVES_NP1
#define MV_GLIST_WAVES_NP1 6
#endif
constexpr int GLIST_WAVES_NP2 = MV_GLIST_WAVES_NP2, GLIST_WAVES_NP1 = MV_GLIST_WAVES_NP1;
#ifndef MV_FAST_PPL_DEFAULT
#define MV_FAST_PPL_DEFAULT 2   // pixels per lane of raster_fast_kernel (MV_FAST_PPL overrides at run time)
#endif
constexpr float SPEC_COS2 = 0.97f * 0.97f;

// The pixel stores of the fast kernels are NON-TEMPORAL: 64 MB of pixels per pass stream through an L2 of 4 MB per XCD that also holds what is read
// again -- the frame lists the passes read, the env state the step kernels beside them work on.  Measured (r06p, -DMV_PIXEL_PLAIN for plain stores):
// TowerBuilding 1024 envs 23.45 -> 24.08 M obs/s, 512 x 4 agents 23.24 -> 23.87, one tick per call 18.72 -> 19.11, ObstaclesHard 512 15.64 -> 15.88.
#ifndef MV_PIXEL_PLAIN
#define PIXEL_STORE(dst, v) __builtin_nontemporal_store((uint32_t)(v), &(dst))
#else
#define PIXEL_STORE(dst, v) ((dst) = (v))
#endif

// Where the short-list pass's pixels go: the frame as a BUFFER (four scalar registers: base, size), a pixel's place in it a 32-bit byte offset -- one
// v_mad_i32_i24 per pixel where the 64-bit address arithmetic of a global store took three vector instructions; `edgeless`: the frame is whole tiles
// (every BASELINE size is), so no pixel needs its "inside the frame?" compares.  (A store beyond the frame's bytes would be dropped by the buffer's range
// check.)
struct PixOut {
    __amdgpu_buffer_rsrc_t rsrc;
    int W, H, W4;
    bool edgeless;
};
#ifndef MV_PIXEL_PLAIN
constexpr int PIXEL_AUX = 2;   // nt
#else
constexpr int PIXEL_AUX = 0;
#endif
__device__ __forceinline__ void put_px(const PixOut &po, int px, int py, unsigned rgba)
{
    const unsigned off = (unsigned)(__mul24(py, po.W4) + px * 4);
    if (__builtin_expect(po.edgeless, 1)) __builtin_amdgcn_raw_buffer_store_b32(rgba, po.rsrc, off, 0, PIXEL_AUX);
    else {
        asm volatile("" ::: "memory");   // (keeps the two stores apart: merged, the compares run for every pixel and their result is or-ed with `edgeless`)
        if (px < po.W && py < po.H) __builtin_amdgcn_raw_buffer_store_b32(rgba, po.rsrc, off, 0, PIXEL_AUX);
    }
}

struct FastArgs {   // what raster_fast_kernel needs of the GymView (fewer live SGPRs than the whole view)
    const unsigned char *vis_hdr;
    const Prim *vis_prims;
    const short4 *vis_rects;
    const int *hist, *list;   // this pass's cost histogram [256][LPT_SUBS] and the frame lists per bin and sub-list (mv_frame.h)
    int num_agents, vis_stride, frames;
    // pipelined steps: the step's staged outputs -> the public arrays, by the first ceil(pub_n / 256) workgroups (pub_n = 0: nothing to publish)
    const float *stage_rewards, *stage_true;
    const uint8_t *stage_done;
    float *pub_rewards, *pub_true;
    uint8_t *pub_done;
    int pub_n;
#ifdef MV_RASTER_TIMING
    unsigned long long *rdbg;   // instrumented builds: per wave, clock marks of the phases (dumped at exit)
#endif
    // the one-launch passes of a batched call: a pass's cost histogram is cleared by the pass itself -- every workgroup counts itself in once it has
    // looked its frame up, the one that completes the count (wg_total) zeroes the histogram and the counter -- so that the step launch of the call
    // that reuses it finds it clean without a fill kernel in front of it (mv_api.hip: take_hist).  nullptr: not this launch's business.
    int *hist_done;
    int wg_total;
    // d > 0: the cheapest 1/d of the frames -- the LAST workgroups of the launch -- are cut into tail_split pieces instead of the launch's split
    // (launch_raster)
    int tail_div, tail_split;
    // 1: tiles that one face of one world box covers take the planar path (planar_tile / overlay_tile); 0: every tile takes the general one (MV_PLANAR=0,
    // comparisons); 2: no overlay_tile
    int planar;
};

// true_objective is only ever recorded by a finishing env (vector_env.cpp:96-101): the others keep the value of their last episode
__device__ __forceinline__ void fast_publish(const FastArgs &fa, int blk)
{
    if (fa.pub_n == 0) return;
    const int i = blk * (int)blockDim.x + (int)threadIdx.x;
    if (i >= fa.pub_n) return;
    fa.pub_rewards[i] = fa.stage_rewards[i];
    const int e = i / fa.num_agents;
    if (fa.stage_done[e]) fa.pub_true[i] = fa.stage_true[i];
    if (i < fa.pub_n / fa.num_agents) fa.pub_done[i] = fa.stage_done[i];
}

// a wave-uniform 64-bit value that came through LDS, back into SGPRs (so that loops over its bits are scalar loops)
__device__ __forceinline__ unsigned long long uniform_u64(unsigned long long v)
{
    const unsigned lo = (unsigned)__builtin_amdgcn_readfirstlane((unsigned)v), hi = (unsigned)__builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
    return ((unsigned long long)hi << 32) | lo;   // (the builtin returns int: no sign extension of the low half)
}

__device__ __forceinline__ float uniform_f32(float v)   // (the builtin takes an int: pass the BITS, not the value)
{
    return __uint_as_float(__builtin_amdgcn_readfirstlane(__float_as_uint(v)));
}

// Rarely used per-frame constants are read from LDS where they are needed instead of living in registers across the tile loop.
// The pointer is laundered so that hipcc does not hoist the loads -- as an LDS (address space 3) pointer: laundering a generic
// pointer turns the loads into flat_load, whose s_waitcnt vmcnt(0) would also wait for the previous tile's pixel store.
typedef __attribute__((address_space(3))) const float lds_float;
__device__ __forceinline__ lds_float *local_lds(const float *p)
{
    lds_float *q = (lds_float *)p;
    asm volatile("" : "+v"(q));
    return q;
}
__device__ __forceinline__ V3 lds_tmul(lds_float *m, V3 v)   // mat_tmul with the matrix in LDS
{
    return v3((m[0] * v.x + m[3] * v.y) + m[6] * v.z, (m[1] * v.x + m[4] * v.y) + m[7] * v.z, (m[2] * v.x + m[5] * v.y) + m[8] * v.z);
}
__device__ __forceinline__ V3 lds_mul(lds_float *m, V3 v)
{
    return v3((m[0] * v.x + m[1] * v.y) + m[2] * v.z, (m[3] * v.x + m[4] * v.y) + m[5] * v.z, (m[6] * v.x + m[7] * v.y) + m[8] * v.z);
}

// Depth keys.  A hit's key is (bits(t) - bits(NEAR_Z)) with the low bits replaced by the list position: for NEAR_Z <= t <= FAR_Z it is
// at most KEY_FAR, and anything else -- t < NEAR_Z (the subtraction wraps), negative, NaN, beyond FAR_Z -- is larger: the range test of
// the exact kernel costs nothing per primitive, only one compare per pixel at the end.
constexpr unsigned KEY_NEAR = 0x3c23d70au;   // bits of NEAR_Z = 0.01f
constexpr unsigned KEY_FAR = 0x42f00000u - KEY_NEAR;   // bits of FAR_Z = 120.0f, relative
static_assert(NEAR_Z == 0.01f && FAR_Z == 120.0f, "update KEY_NEAR / KEY_FAR");

// depthMask: ~POS_MASK, handed in IN A VECTOR REGISTER (box_run launders it): gfx950's VOP3 encoding takes one scalar operand and no literal, so "(x & literal)
// | scalar" is two instructions, "(x & vector) | scalar" is one v_and_or_b32 -- 19 instead of 20 per box and pixel in the pass's innermost loop
template <unsigned POS_MASK>
__device__ __forceinline__ unsigned box_key(V3 inv, const float4 lo, const float4 hi, int pos, unsigned depthMask = ~POS_MASK)
{
    const float t1x = lo.x * inv.x, t2x = hi.x * inv.x, t1y = lo.y * inv.y, t2y = hi.y * inv.y, t1z = lo.z * inv.z, t2z = hi.z * inv.z;
    const float tn = __builtin_fmaxf(__builtin_fmaxf(__builtin_fminf(t1x, t2x), __builtin_fminf(t1y, t2y)), __builtin_fminf(t1z, t2z));
    const float tf = __builtin_fminf(__builtin_fminf(__builtin_fmaxf(t1x, t2x), __builtin_fmaxf(t1y, t2y)), __builtin_fmaxf(t1z, t2z));
    const unsigned key = ((__float_as_uint(tn) - KEY_NEAR) & depthMask) | (unsigned)pos;
    return tn <= tf ? key : ~0u;
}
template <unsigned POS_MASK>
__device__ __forceinline__ unsigned hit_key(bool hit, float t, int pos)   // other shapes: their intersectors apply the range test themselves
{
    return hit ? (((__float_as_uint(t) - KEY_NEAR) & ~POS_MASK) | (unsigned)pos) : ~0u;
}

// entry depth of a box given relative to the ray origin; +inf/NaN semantics make zero direction components harmless
__device__ __forceinline__ bool fast_box(V3 inv, const float4 lo, const float4 hi, float &t_out)
{
    const float t1x = lo.x * inv.x, t2x = hi.x * inv.x, t1y = lo.y * inv.y, t2y = hi.y * inv.y, t1z = lo.z * inv.z, t2z = hi.z * inv.z;
    const float tn = __builtin_fmaxf(__builtin_fmaxf(__builtin_fminf(t1x, t2x), __builtin_fminf(t1y, t2y)), __builtin_fminf(t1z, t2z));
    const float tf = __builtin_fminf(__builtin_fminf(__builtin_fmaxf(t1x, t2x), __builtin_fmaxf(t1y, t2y)), __builtin_fmaxf(t1z, t2z));
    t_out = tn;
    return tn <= tf && tn >= NEAR_Z && tn <= FAR_Z;
}

}  // namespace

namespace {

// one primitive that is not an axis-aligned box of the world frame -- a camera-attached box, a capsule, a cone, a scaled shape -- against this
// lane's ray: its depth key (or ~0u), and for the curved ones the normal in the primitive's frame
// a box that lives in another frame of reference than the world's -- the viewer's own camera (its time bar, the object it carries) or another camera (another
// agent's visor and bar) -- against this lane's ray: the entry depth, range-tested
__device__ __forceinline__ bool other_box(const float4 lo, const float4 hi, int qfr, const float *s_hdr, int viewer, V3 dw, float dcx, float dcy, float &t)
{
    const V3 df = qfr == 1 + viewer ? v3(dcx, dcy, -1.0f) : lds_tmul(local_lds(s_hdr + FH_CAM + FH_CAM_STRIDE * (qfr - 1) + 3), dw);
    return fast_box(v3(__builtin_amdgcn_rcpf(df.x), __builtin_amdgcn_rcpf(df.y), __builtin_amdgcn_rcpf(df.z)), lo, hi, t);
}

template <bool SHAPES>
__device__ __forceinline__ bool other_rec(const float4 lo, const float4 hi, const float *s_hdr, const float *camv,
                                          int viewer, V3 dw, float dcx, float dcy, float &t, V3 &n)
{
    const unsigned meta = __builtin_amdgcn_readfirstlane(__float_as_uint(lo.w));
    const int qkind = meta & 15, qfr = (meta >> 4) & 15;
    t = 0.0f;
    n = v3(0, 0, 0);
    bool hit;
    if (qkind == PRIM_BOX) hit = other_box(lo, hi, qfr, s_hdr, viewer, dw, dcx, dcy, t);
    else {
        lds_float *ce = local_lds(camv);   // (read here, not kept in registers across the tile loop)
        const V3 eye = v3(ce[0], ce[1], ce[2]);
        if (qkind == PRIM_CAPSULE) hit = ray_capsule<true>(eye, dw, v3(lo.x, lo.y, lo.z), hi.x, hi.y, t, n);
        else if (qkind == PRIM_CONE) hit = ray_cone<true>(eye, dw, v3(lo.x, lo.y, lo.z), hi.x, hi.y, hi.z, t, n);
        else if (SHAPES) {
            const V3 df = qfr == 0 ? dw : qfr == 1 + viewer ? v3(dcx, dcy, -1.0f) : lds_tmul(local_lds(s_hdr + FH_CAM + FH_CAM_STRIDE * (qfr - 1) + 3), dw);
            hit = ray_scaled_shape<true>(qkind, df, v3(lo.x, lo.y, lo.z), v3(hi.x, hi.y, hi.z), t, n);
        } else hit = false;
    }
    return hit && t >= NEAR_Z && t <= FAR_Z;
}

template <bool SHAPES, unsigned POS_MASK>
__device__ __forceinline__ unsigned fast_other(int pos, const float4 *s_vis, const float *s_hdr,
                                               const float *camv, int viewer, V3 dw, float dcx, float dcy, V3 &n)
{
    float t;
    const bool hit = other_rec<SHAPES>(s_vis[2 * pos], s_vis[2 * pos + 1], s_hdr, camv, viewer, dw, dcx, dcy, t, n);
    return hit_key<POS_MASK>(hit, t, pos);
}

// The part of Phong every kind of hit shares: depth t along the pixel's ray, N . (L - P) and N . (-P) (both unnormalised in (L - P) / P), the
// pixel's dc . dc and L . dc, the colour's bytes as floats -> RGBA8.  (One function for the general and the planar-tile path: the same
// operations in the same order, so a tile drawn by either has the same bytes.)
// SPEC = false: the caller has PROVED that no pixel it shades this way can lie inside the highlight cone (classify_tiles: highlight bound) -- the test
// below would fail for every one of them and spec255 stay 0: the same bytes without the seven instructions of V . R.
template <bool SPEC = true>
__device__ __forceinline__ unsigned phong_tail(float t, float ndl, float nv, float a2, float ldc, float cr, float cg, float cb)
{
    // |L - P|^2 = |L|^2 - 2 t (L . dc) + t^2 (dc . dc)
    const float ta = t * a2;
    const float len2LP = __builtin_fmaf(t, ta - 2.0f * ldc, 20.0f);
    const float rs = __builtin_amdgcn_rsqf(len2LP);
    const float intensity = __builtin_fmaxf(0.0f, ndl * rs);
    float spec255 = 0.0f;
    if (SPEC && intensity > 0.001f) {
        // V . R with V = -P unnormalised: 2 (N . Ld)(N . -P) + Ld . P
        const float vr = __builtin_fmaf(2.0f * intensity, nv, (t * (ldc - ta)) * rs);
        const float p2 = t * ta;   // |P|^2
        if (vr > 0.0f && vr * vr > SPEC_COS2 * p2) {
            const float cosv = __builtin_fminf(vr * __builtin_amdgcn_rsqf(p2), 1.0f);
            spec255 = 255.0f * __builtin_amdgcn_exp2f(300.0f * __builtin_amdgcn_logf(cosv));
        }
    }
    const float AMB = float(0x55) / 255.0f, DIFL = (float(0xbb) / 255.0f) * (float(0xaa) / 255.0f);
    const float sc = __builtin_fmaf(DIFL, intensity, AMB);
    const float r8 = __builtin_fmaf(cr, sc, spec255), g8 = __builtin_fmaf(cg, sc, spec255), b8 = __builtin_fmaf(cb, sc, spec255);
    // v_cvt_pk_u8_f32: round to nearest (even on ties; the exact kernel rounds ties up: they do not occur), saturate, insert the byte
    return __builtin_amdgcn_cvt_pk_u8_f32(b8, 2, __builtin_amdgcn_cvt_pk_u8_f32(g8, 1, __builtin_amdgcn_cvt_pk_u8_f32(r8, 0, 0xff000000u)));
}

// Phong (Magnum Shaders::Phong, uniforms of magnum_env_renderer.cpp:200-203) for the winning hit of a pixel; 0xff000000 when there is none
// the winning primitive's record (lo, hi) and -- for the curved kinds -- the depth and normal kept with the hit -> the pixel's colour
template <bool SHAPES>
__device__ __forceinline__ unsigned shade_rec(const float4 lo, const float4 hi, float tkey, V3 bn, const float *s_hdr, const float *camv,
                                              int viewer, V3 dw, V3 inv, float dcx, float dcy, float a2, float ldc)
{
    unsigned rgba;
    {
        const V3 dc = v3(dcx, dcy, -1.0f);
        const unsigned meta = __float_as_uint(lo.w), color = __float_as_uint(hi.w);
        const int qkind = meta & 15, qfr = (meta >> 4) & 15;
        float t, ndl, nv;   // depth; N . (L - P) and N . (-P), both unnormalised in (L - P) / P
        if (qkind == PRIM_BOX) {
            V3 d = dw, iv = inv;
            if (qfr != 0) {
                d = qfr == 1 + viewer ? dc : lds_tmul(local_lds(s_hdr + FH_CAM + FH_CAM_STRIDE * (qfr - 1) + 3), dw);
                iv = v3(__builtin_amdgcn_rcpf(d.x), __builtin_amdgcn_rcpf(d.y), __builtin_amdgcn_rcpf(d.z));
            }
            const float tnx = __builtin_fminf(lo.x * iv.x, hi.x * iv.x), tny = __builtin_fminf(lo.y * iv.y, hi.y * iv.y),
                        tnz = __builtin_fminf(lo.z * iv.z, hi.z * iv.z);
            t = __builtin_fmaxf(__builtin_fmaxf(tnx, tny), tnz);
            const int axis = tnx == t ? 0 : tny == t ? 1 : 2;   // first axis whose near-plane crossing is the entry depth
            const float dk = axis == 0 ? d.x : axis == 1 ? d.y : d.z;
            const float lk = s_hdr[FH_LREL + 4 * qfr + axis];
            nv = t * __builtin_fabsf(dk);                                                      // plane offset along the outward normal
            ndl = nv - __uint_as_float(__float_as_uint(lk) ^ (__float_as_uint(dk) & 0x80000000u));   // sgn (Lrel_k - t d_k), sgn = -sign(d_k)
        } else {   // capsules, cones, scaled shapes: the normal was kept with the hit (depth: the key's)
            t = tkey;
            V3 N;
            lds_float *cc = local_lds(camv + 3);
            if (!SHAPES || qkind < PRIM_SPHERE_S || qfr == 0) N = lds_tmul(cc, bn);
            else if (qfr == 1 + viewer) N = bn;
            else N = lds_tmul(cc, lds_mul(local_lds(s_hdr + FH_CAM + FH_CAM_STRIDE * (qfr - 1) + 3), bn));
            const V3 P = dc * t;
            nv = -dot(N, P);
            ndl = dot(N, v3(0.0f - P.x, 4.0f - P.y, 2.0f - P.z));
        }
        rgba = phong_tail(t, ndl, nv, a2, ldc, float((color >> 16) & 255u), float((color >> 8) & 255u), float(color & 255u));
    }
    return rgba;
}

template <bool SHAPES, unsigned POS_MASK>
__device__ __forceinline__ unsigned fast_shade(unsigned best, V3 bn, const float4 *s_vis, const float *s_hdr, const float *camv,
                                               int viewer, V3 dw, V3 inv, float dcx, float dcy, float a2, float ldc)
{
    unsigned rgba = 0xff000000u;
    if (best <= (KEY_FAR | POS_MASK)) {   // a hit between the near and the far plane
        const int pos = (int)(best & POS_MASK);
        rgba = shade_rec<SHAPES>(s_vis[2 * pos], s_vis[2 * pos + 1], __uint_as_float((best & ~POS_MASK) + KEY_NEAR),
                                 bn, s_hdr, camv, viewer, dw, inv, dcx, dcy, a2, ldc);
    }
    return rgba;
}

struct FastFrame { int frame, part, viewer, nVis, split; };

// the frames at the END of the cost order -- the cheapest 1/div of them, a multiple of 8 -- that a launch cuts finer than the others (launch_raster:
// fine-grained tail)
__host__ __device__ inline int tail_frames(int frames, int div) { return (frames / div) & ~7; }

// The fast kernels' prologue: which frame is this workgroup's, then copies (header, list, rectangles) + the separable ray tables; ends with the
// one barrier.  Workgroup ids are dealt round-robin over the 8 XCDs; the `split` parts of one frame get ids that are congruent mod 8 so that they
// share one XCD's L2 (the frame's list is read `split` times, neighbouring tiles write neighbouring lines).
// GLIST: the records stay in global memory (the tile loop fetches the surviving ones with scalar loads); LDS gets the rectangles and one class
// byte per primitive instead: 0 a box in the world frame, 1..3 a box in hex wall frame 0..2, 4 anything else
template <int MAXVIS, bool GLIST = false, int NT = 256>   // NT: threads of the workgroup (256, or 512: a whole frame per workgroup, raster_fast_body)
__device__ __forceinline__ FastFrame fast_prologue(const FastArgs &fa, int blk, int W, int H, int split,
                                                   float4 *s_vis, short4 *s_rect, float *s_hdr, float4 *s_col,
                                                   float4 *s_row, float2 *s_rowq, float *s_colq, unsigned char *s_cls = nullptr)
{
    const int A = fa.num_agents;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    int position, part;
    {
        int b = blk, first = 0, frames = fa.frames;   // the segment of the cost order this workgroup belongs to: its first position, its length, its split
        if (fa.tail_div) {   // the cheapest frames / tail_div frames, last in the cost order, in tail_split pieces each (tail_frames)
            const int q = tail_frames(frames, fa.tail_div), head = frames - q;
            if (b < split * head) frames = head;
            else { b -= split * head; first = head; split = fa.tail_split; frames = q; }
        }
        const int per = 8 * split, group = b / per, r = b - group * per;   // b: this workgroup's index within its segment of its gym's part of the grid
        position = group * 8 + (r & 7); part = r >> 3;
        if (group * 8 + 8 > frames) { const int nf = frames - group * 8; position = group * 8 + r % nf; part = r / nf; }
        position += first;
    }
    // position -> frame, most expensive frames first: the frame setup left every frame in the list of its cost bin; prefix-sum the 256 bin
    // counts (bin 255 first) and take entry (position - start) of the bin whose range holds `position`
    __shared__ int s_wsum[4], s_frame, s_lastWG;
    {
        static_assert(LPT_SUBS == 4, "the bin's counters are read as one int4");
        // (one thread per cost bin; a 512-thread workgroup's other waves only keep the barriers company)
        const bool binThread = NT == 256 || tid < LPT_BUCKETS;
        // the bin's LPT_SUBS counters
        const int4 c0 = binThread ? *reinterpret_cast<const int4 *>(fa.hist + (LPT_BUCKETS - 1 - tid) * LPT_SUBS) : make_int4(0, 0, 0, 0);
        const int h = (c0.x + c0.y) + (c0.z + c0.w);
        int x = h;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const int y = __shfl_up(x, off, 64);
            if (lane >= off) x += y;
        }
        if (lane == 63 && binThread) s_wsum[wave] = x;
        __syncthreads();
        int base = 0;
        for (int w = 0; w < wave && w < 4; ++w) base += s_wsum[w];
        const int end = base + x, start = end - h;
        if (binThread && position >= start && position < end) {   // the bin's lists one after the other
            const int cs[LPT_SUBS] = {c0.x, c0.y, c0.z, c0.w};
            int off = position - start, sub = 0;
#pragma unroll
            for (int q = 0; q < LPT_SUBS - 1; ++q)
                if (sub == q && off >= cs[q]) { off -= cs[q]; sub = q + 1; }
            s_frame = fa.list[(size_t)((LPT_BUCKETS - 1 - tid) * LPT_SUBS + sub) * lpt_sub_capacity(fa.frames) + off];
        }
        __syncthreads();
    }
    // Every thread's loads from the histogram have RETURNED here -- their values went into the prefix sums and the look-up above, on this side of the
    // barriers (which are also compiler fences: no load can sink below them) -- so the count may be relaxed: the workgroup that completes it zeroes the
    // histogram strictly after every reader's data arrived.  (A release / acquire pair at agent scope would be the textbook form; on gfx950 it is a
    // buffer_wbl2 + buffer_inv per workgroup -- an L2 write-back in the middle of 8192 workgroups' pixel stores -- for an ordering the data dependence
    // already gives.)  The zeroing itself is ordered against the next writer -- the frame setups of a later step launch -- by the kernel boundary.
#ifdef MV_HIST_FENCE   // (the textbook form, for the A/B measurement DESIGN.md 3.3 quotes: release / acquire at agent scope on the count)
    if (tid == 0) s_lastWG = fa.hist_done != nullptr && __hip_atomic_fetch_add(fa.hist_done, 1, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT) == fa.wg_total - 1;
#else
    if (tid == 0) s_lastWG = fa.hist_done != nullptr && __hip_atomic_fetch_add(fa.hist_done, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == fa.wg_total - 1;
#endif
    const int frame = __builtin_amdgcn_readfirstlane(s_frame);
    const int viewer = frame % A;
    const float *gh = reinterpret_cast<const float *>(fa.vis_hdr + (size_t)frame * FRAME_HDR_BYTES);
    const int nVis = __builtin_amdgcn_readfirstlane(min(__float_as_int(gh[FH_COUNT]), (int)MAXVIS));

    // ---- prologue: copies + the separable ray tables, one barrier
    if (tid < FH_FLOATS) s_hdr[tid] = gh[tid];
    {
        const float4 *src = reinterpret_cast<const float4 *>(fa.vis_prims + (size_t)frame * fa.vis_stride);
        if (!GLIST) for (int i = tid; i < nVis * 2; i += NT) s_vis[i] = src[i];
        else
            for (int i = tid; i < nVis; i += NT) {
                const unsigned ml = __float_as_uint(reinterpret_cast<const float *>(src)[8 * i + 3]);   // kind | frame << 4 | slot << 8
                const unsigned fl = (ml >> 4) & 15u;
                s_cls[i] = (ml & 15u) != (unsigned)PRIM_BOX ? 4 : fl == 0u ? 0 : fl > (unsigned)MAX_AGENTS ? (unsigned char)(fl - (unsigned)MAX_AGENTS) : 4;
            }
        const short4 *rs = fa.vis_rects + (size_t)frame * fa.vis_stride;
        for (int i = tid; i < nVis; i += NT) s_rect[i] = rs[i];
        const float *c = gh + FH_CAM + FH_CAM_STRIDE * viewer + 3;   // (same arithmetic as the exact kernel: rays are bit-identical)
        for (int i = tid; i < W; i += NT) {
            const float dcx = (((float(i) + 0.5f) / float(W)) * 2.0f - 1.0f) * TAN_HALF_FOV;
            s_col[i] = make_float4(dcx, c[0] * dcx, c[3] * dcx, c[6] * dcx);
            s_colq[i] = dcx * dcx;
        }
        for (int j = tid; j < H; j += NT) {
            const float dcy = (((float(j) + 0.5f) / float(H)) * 2.0f - 1.0f) * TAN_HALF_FOV_Y;
            s_row[j] = make_float4(dcy, c[1] * dcy, c[4] * dcy, c[7] * dcy);
            s_rowq[j] = make_float2(dcy * dcy + 1.0f, 4.0f * dcy - 2.0f);
        }
    }
    __syncthreads();
    if (s_lastWG) {   // the pass's last workgroup: nobody reads the histogram any more
        int *h = const_cast<int *>(fa.hist);
        for (int i = tid; i < LPT_BUCKETS * LPT_SUBS; i += NT) h[i] = 0;
        if (tid == 0) __hip_atomic_store(fa.hist_done, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }

    return FastFrame{frame, part, viewer, nVis, split};
}

// box_key where the SIGN of every component of the ray's direction is known, the same for all pixels of the tile (SGN bit a set: component a is negative --
// classify_tiles finds that out per tile, from the tile's corner rays): the near plane of axis a is lo_a (hi_a if negative), so the six min / max of the
// general form are not needed -- the products are the same ones, picked by the compiler instead of by v_min / v_max: 13 instead of 19 instructions
template <unsigned POS_MASK, int SGN>
__device__ __forceinline__ unsigned box_key_signed(V3 inv, const float4 lo, const float4 hi, int pos, unsigned depthMask)
{
    const float nx = (SGN & 1) ? hi.x : lo.x, fx = (SGN & 1) ? lo.x : hi.x, ny = (SGN & 2) ? hi.y : lo.y, fy = (SGN & 2)
                      ? lo.y : hi.y, nz = (SGN & 4) ? hi.z : lo.z, fz = (SGN & 4) ? lo.z : hi.z;
    const float tn = __builtin_fmaxf(__builtin_fmaxf(nx * inv.x, ny * inv.y), nz * inv.z);
    const float tf = __builtin_fminf(__builtin_fminf(fx * inv.x, fy * inv.y), fz * inv.z);
    const unsigned key = ((__float_as_uint(tn) - KEY_NEAR) & depthMask) | (unsigned)pos;
    return tn <= tf ? key : ~0u;
}

// the boxes of one frame of reference among list positions 64 k .. 64 k + 63 (mask m) against this lane's NP rays, given each ray's inverse
// direction in that frame: the next record is fetched from LDS while the current one is intersected (with NP = 2 a record fetched once serves
// two pixels, and the two slab tests are independent instruction chains)
template <unsigned POS_MASK, int NP, int SGN = -1>   // SGN >= 0: the signs of the rays' components, uniform over the tile (box_key_signed)
__device__ __forceinline__ void box_run(unsigned long long m, int k, const V3 (&inv)[NP], const float4 *s_vis, unsigned (&best)[NP])
{
    if (!m) return;
    int p0 = __ffsll((long long)m) - 1 + 64 * k, p1 = 0;
    m &= m - 1;
    float4 lo0 = s_vis[2 * p0], hi0 = s_vis[2 * p0 + 1], lo1 = lo0, hi1 = hi0;
    unsigned depthMask = ~POS_MASK;
    asm volatile("" : "+v"(depthMask));   // (a vector register: box_key)
    for (;;) {
        bool more = m != 0ull;
        if (more) { p1 = __ffsll((long long)m) - 1 + 64 * k; m &= m - 1; lo1 = s_vis[2 * p1]; hi1 = s_vis[2 * p1 + 1]; }
#pragma unroll
        for (int j = 0; j < NP; ++j) best[j] = min(best[j], SGN >= 0 ? box_key_signed<POS_MASK, (SGN >= 0 ? SGN : 0)>(inv[j], lo0, hi0, p0, depthMask)
             : box_key<POS_MASK>(inv[j], lo0, hi0, p0, depthMask));
        if (!more) break;
        more = m != 0ull;
        if (more) { p0 = __ffsll((long long)m) - 1 + 64 * k; m &= m - 1; lo0 = s_vis[2 * p0]; hi0 = s_vis[2 * p0 + 1]; }
#pragma unroll
        for (int j = 0; j < NP; ++j) best[j] = min(best[j], SGN >= 0 ? box_key_signed<POS_MASK, (SGN >= 0 ? SGN : 0)>(inv[j], lo1, hi1, p1, depthMask)
             : box_key<POS_MASK>(inv[j], lo1, hi1, p1, depthMask));
        if (!more) break;
    }
}

}  // namespace

// ---- planar tiles ------------------------------------------------------------------------------------------------------------------------
// Half of a TowerBuilding frame's non-empty tiles show ONE face of ONE axis-aligned world box (floor, a wall, the side of a near box) and
// nothing else -- and the general path still pays a full ray (three v_rcp_f32), a slab test per culling survivor and the entry-axis selects
// of the shading for every pixel of them.  So the workgroup CLASSIFIES its tiles before it draws any (classify_tiles, when the frame's list is
// one culling round): for a ray o + t d that enters a box through its face on axis k (plane offset p_k from the eye, the eye outside) the
// hit lies inside the face's edge on axis m at bound b_m iff
//     g = sign(p_k) (p_k d_m - b_m d_k) >= 0      (lo edge; the opposite sign for the hi edge),
// and if all four hold the face is front-facing (behind the eye at most one of an axis' two edge conditions can hold); g < 0 for one edge means
// "not through this face" whatever the facing.  d is an AFFINE function of the pixel (mv_frame.h: dw = col[i] + row[j] - c2), so every g is
// an edge function a i + b j + c of the pixel coordinates -- the trivial accept / trivial reject of a tile against a convex polygon:
// its minimum over a tile is its value at one corner, min and max differ by a per-edge constant.
//   * sixteen lanes per box compute its (up to) twelve edge functions -- three candidate faces (those the eye is outside of) x four edges --
//     with a margin two orders of magnitude above the rounding of either arithmetic (PLANAR_MARGIN, ~1 % of a pixel) folded in;
//   * then ONE LANE PER TILE tests its tile against them: two v_fma_f32 and two compares per edge (a wave per tile, with a lane per edge and the
//     verdict assembled from ballots, was built first and measured: 19.8 M instead of 10.7 M scalar instructions per launch, 60.8 us
//     instead of 53.7 -- profiles/r04a_wave_per_tile_*); the four waves share the list out among them and merge per tile in LDS:
//     the boxes (and other primitives) the tile's pixels can hit at all -- which replaces the rectangle culling of the tile loop --
//     and whether one face covers the tile.
// A tile with one possible hit whose face covers it is drawn by planar_tile(): one v_rcp_f32 per pixel, t = p_k / d_k is the same product
// the slab test forms, the face constants are wave-uniform, and the shared phong_tail() makes the bytes identical to the general path's
// (tests/test_fast_pixels_gpu.py: test_planar_tiles_change_no_byte); a tile nothing can be hit through is cleared; the others take
// the general path with the refined survivor mask.
// Reference for what is drawn: magnum_env_renderer.cpp:288-330 (depth-tested, back-face-culled boxes), :200-203 (Phong uniforms).
constexpr float PLANAR_MARGIN = 2e-4f;
// a classified tile's class (s_tile[u].w, classify_tiles)
enum : unsigned { TC_EMPTY = 0, TC_PLANAR = 1, TC_PLANAR_SPEC = 2, TC_OVERLAY = 3, TC_GENERAL = 4 };
constexpr int CLS_MAX_TILES = 128;   // tiles of one workgroup that can be classified (LDS: 16 B each); more (hires frames): the general path throughout

// s_tile[u] of the workgroup's u-th tile (u = 4 j + w is tile (j split + part) 4 + w of the frame, the tile loop's order): x, y = the list positions
// (< 64) its pixels can hit, z = axis + 1 of the covering face when a world box's face covers the tile (only ever read when exactly one bit
// of the mask is set).  s_line[64 wave + 16 s + 4 f + e]: edge e of face f of the s-th box of the wave's current round as (a, b, c_in, span):
// inside at every pixel of the tile <=> a x0 + b y0 + c_in >= 0 at the tile's first pixel (x0, y0); outside at every pixel <=> that + span <= 0.
// Called by all 256 threads after the prologue's barrier (s_vis, s_rect, the header are in LDS); ends with a barrier.
// The tile loop's ORDER (raster_fast_body): the classification also files every tile under its class -- s_cnt[c] tiles of category c (0 general, 1 overlay, 2
// planar with the highlight test, 3 planar, 4 empty), the entry's category, run length and place within its category in s_tile[u].z (the covering face it held
// moved into the class word) and its first pixel in s_txy[u] (x | y << 16) -- so that the workgroup draws its tiles most expensive class first from a compact
// list, and clears the empty ones -- 43 % of a TowerBuilding frame's tiles (r07d census) -- with all its threads at once instead of handing them out one by
// one.
template <int TH, int NT>
__device__ __forceinline__ void classify_tiles(uint4 *s_tile, float4 *s_line, const float4 *s_vis, const short4 *s_rect,
                                               const float *s_hdr, const float *camv, int nVis, unsigned long long wb0,
                                               int W, int H, int part, int split, int tilesX, int numTiles,
                                                       int perWG, bool overlayOn, int *s_cnt, unsigned *s_txy, bool bulkClear)
{
    constexpr int NW = NT / 64;        // waves of the workgroup
    constexpr int PPR = 16 / NW;       // list positions per wave and round (256 edge-function slots in all: 4 with four waves, 2 with eight)
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    if (tid < CLS_MAX_TILES) s_tile[tid] = make_uint4(0u, 0u, 0u, 0u);
    if (tid < 8) s_cnt[tid] = 0;
    // the ray's world components as affine functions of the pixel: d_a(i, j) = A_a i + B_a j + C_a  (dc = (((i + .5) / W) 2 - 1) TAN, ..., -1)
    const float sx = 2.0f * TAN_HALF_FOV / float(W), ox = (1.0f / float(W) - 1.0f) * TAN_HALF_FOV;
    const float sy = 2.0f * TAN_HALF_FOV_Y / float(H), oy = (1.0f / float(H) - 1.0f) * TAN_HALF_FOV_Y;
    // this lane as an edge task: lane 16 s + q, q = 4 f + e: face axis k = f, edge axis m one of the two others (e >> 1), hi bound (e & 1)
    const int q = lane & 15, es = lane >> 4;
    const int ek = min(q >> 2, 2);
    int em = ek + 1 + ((q >> 1) & 1);
    em = em >= 3 ? em - 3 : em;
    const bool ehi = q & 1, eactive = q < 12;
    const float Ak = camv[3 + 3 * ek] * sx, Bk = camv[4 + 3 * ek] * sy, Ck = (camv[3 + 3 * ek] * ox + camv[4 + 3 * ek] * oy) - camv[5 + 3 * ek];
    const float Am = camv[3 + 3 * em] * sx, Bm = camv[4 + 3 * em] * sy, Cm = (camv[3 + 3 * em] * ox + camv[4 + 3 * em] * oy) - camv[5 + 3 * em];
    float4 *myLines = s_line + 16 * PPR * wave;
    constexpr float WX = float(TILE_W - 1), WY = float(TH - 1);
    __syncthreads();   // s_tile cleared
    // this lane as a tile: one per chunk of 64 local tiles (a whole 128-tile frame per workgroup: two)
    constexpr int CH = CLS_MAX_TILES / 64;
    int tX0[CH], tY0[CH], tX1[CH], tY1[CH];
    bool tvalid[CH];
    unsigned mlo[CH], mhi[CH], cover[CH];
#pragma unroll
    for (int c = 0; c < CH; ++c) {
        const int u = 64 * c + lane;
        const int tile = ((u >> 2) * split + part) * 4 + (u & 3);
        tvalid[c] = u < perWG && tile < numTiles;
        const int tyi = tile / tilesX, txi = tile - tyi * tilesX;
        tX0[c] = txi * TILE_W; tY0[c] = tyi * TH;
        tX1[c] = min(tX0[c] + TILE_W, W) - 1; tY1[c] = min(tY0[c] + TH, H) - 1;
        mlo[c] = mhi[c] = cover[c] = 0u;
    }
    for (int base = PPR * wave; base < nVis; base += 16) {   // the waves share the list out: PPR positions per wave and round
        // ---- the edge functions of this round's boxes (sixteen lanes per box)
        const int P = min(base + min(es, PPR - 1), nVis - 1);
        const bool isBox = es < PPR && base + es < nVis && ((wb0 >> P) & 1ull) != 0ull;
        const float4 lo4 = s_vis[2 * P], hi4 = s_vis[2 * P + 1];
        const float lok = ek == 0 ? lo4.x : ek == 1 ? lo4.y : lo4.z, hik = ek == 0 ? hi4.x : ek == 1 ? hi4.y : hi4.z;
        const float lom = em == 0 ? lo4.x : em == 1 ? lo4.y : lo4.z, him = em == 0 ? hi4.x : em == 1 ? hi4.y : hi4.z;
        const float bnd = ehi ? him : lom;
        const bool cand = isBox && eactive && (lok > 0.0f || hik < 0.0f);   // the eye is outside the box along k: the face towards it can be entered
        const float pk = lok > 0.0f ? lok : hik;
        // a covering face is drawn without a range test per pixel: its depth |hit| / |d| lies in [|p| / 1.69, |hit|_1] (|d| in [1, 1.69])
        const float far1 = (__builtin_fmaxf(__builtin_fabsf(lo4.x), __builtin_fabsf(hi4.x)) + __builtin_fmaxf(__builtin_fabsf(lo4.y), __builtin_fabsf(hi4.y))) +
                           __builtin_fmaxf(__builtin_fabsf(lo4.z), __builtin_fabsf(hi4.z));
        const bool coverOK = cand && __builtin_fabsf(pk) >= 4.0f * NEAR_Z && far1 <= 0.5f * FAR_Z;
        const float sg = ((pk > 0.0f) != ehi) ? 1.0f : -1.0f;   // sign(p), reversed for the hi edge
        const float ea = sg * (pk * Am - bnd * Ak), eb = sg * (pk * Bm - bnd * Bk), ec = sg * (pk * Cm - bnd * Ck);
        const float mg = PLANAR_MARGIN * (__builtin_fabsf(pk) + __builtin_fabsf(bnd));
        const float cin = ((ec + __builtin_fminf(0.0f, ea * WX)) + __builtin_fminf(0.0f, eb * WY)) - mg;
        const float span = (__builtin_fabsf(ea) * WX + __builtin_fabsf(eb) * WY) + 2.0f * mg;
        // (no such face: never inside, always outside)
        if (es < PPR) myLines[lane] = cand ? make_float4(ea, eb, cin, span) : make_float4(0.0f, 0.0f, -1.0f, 0.0f);
        const unsigned long long candMask = __ballot(cand), coverMask = __ballot(coverOK);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");   // (a wave's LDS operations execute in order: only the compiler has to keep them so)
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        // ---- every tile against them
#pragma unroll
        for (int sl = 0; sl < PPR; ++sl) {
            const int Q = base + sl;
            if (Q >= nVis) break;
            const uint2 rr = *reinterpret_cast<const uint2 *>(&s_rect[Q]);   // x0 | x1 << 16, y0 | y1 << 16
            const bool worldBox = (wb0 >> Q) & 1ull;
#pragma unroll
            for (int c = 0; c < CH; ++c) {
                if (64 * c >= perWG) break;
                const bool ov = tvalid[c] & ((int)(rr.x & 0xffffu) <= tX1[c]) & ((int)(rr.x >> 16) >= tX0[c])
                                             & ((int)(rr.y & 0xffffu) <= tY1[c]) & ((int)(rr.y >> 16) >= tY0[c]);
                if (!__any(ov)) continue;
                bool hit = ov;
                unsigned cov = 0u;
                if (worldBox) {   // an axis-aligned box of the world frame: through which face, if any?
                    const float fx0 = float(tX0[c]), fy0 = float(tY0[c]);
                    bool missAll = true;
#pragma unroll
                    for (int f = 0; f < 3; ++f) {
                        if (!((candMask >> (16 * sl + 4 * f)) & 1ull)) continue;   // (no face towards the eye on this axis)
                        bool ins = true, mis = false;
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            const float4 L = myLines[16 * sl + 4 * f + e];
                            const float g = __builtin_fmaf(L.x, fx0, __builtin_fmaf(L.y, fy0, L.z));
                            ins = ins && g >= 0.0f;
                            mis = mis || g + L.w <= 0.0f;
                        }
                        if (ins && ((coverMask >> (16 * sl + 4 * f)) & 1ull)) cov = (unsigned)f + 1u;
                        missAll = missAll && mis;
                    }
                    hit = ov && !missAll;
                }
                if (hit) {
                    if (Q < 32) mlo[c] |= 1u << Q; else mhi[c] |= 1u << (Q - 32);
                    cover[c] |= cov;
                }
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");   // (the next round's lines are written after this round's were read)
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }
#pragma unroll
    for (int c = 0; c < CH; ++c)
        if (tvalid[c]) {
            const int u = 64 * c + lane;
            if (mlo[c]) atomicOr(&s_tile[u].x, mlo[c]);
            if (mhi[c]) atomicOr(&s_tile[u].y, mhi[c]);
            if (cover[c]) atomicOr(&s_tile[u].z, cover[c]);
        }
    __syncthreads();
    // ---- every tile's class, ONE word the tile loop dispatches on (s_tile[u].w): TC_EMPTY nothing can be hit; TC_PLANAR / TC_PLANAR_SPEC one face of one
    // world box covers the tile and nothing else can be hit (without / with the highlight test); TC_OVERLAY the same with boxes of other frames of reference
    // in front of it or not (overlay_tile); TC_GENERAL everything else.  Covered classes carry the face: axis << 3 | list position << 5.
    // The highlight bound: Shaders::Phong's highlight (magnum_env_renderer.cpp:200-203, shininess 300) is evaluated where cos(V, R) > 0.97 (phong_tail).
    // On a planar face, with L' the light mirrored in the face's plane, R at a point P of the face is the direction from L' to P, V the direction from P to
    // the eye E: the angle between them is the exterior angle at P of the triangle E P L', the sum of its interior angles at E and at L' -- so it is at
    // least the angle at E, the angle between the pixel's ray and the direction from the eye to L'.  If that angle at the tile's centre exceeds acos(0.965)
    // (15.2 degrees: 1.1 degrees of margin over acos(0.97) for the rounding of either side) plus the tile's angular radius (two rays through points a, b of
    // the plane z = -1: sin(angle) <= |a - b|), no pixel of the tile passes phong_tail's test: the planar path leaves the seven instructions of V . R out
    // (phong_tail<false>), the bytes stay what they were.  One lane per tile, one wave per 64 tiles: ~40 instructions per frame.
    {
        const float hx = 0.5f * WX * sx, hy = 0.5f * WY * sy;
        const float r2 = hx * hx + hy * hy;                                    // (tile radius)^2 on the plane z = -1 = sin^2 of the bound on its angular radius
        // cos(acos(0.965) + asin(r)), rounded towards the larger angle
        const float cosT = 0.965f * __builtin_amdgcn_sqrtf(__builtin_fmaxf(1.0f - r2, 0.0f)) - 0.2623f * __builtin_amdgcn_sqrtf(r2);
        const bool usable = r2 < 0.5f && cosT > 0.0f;
        const float cosT2 = cosT * cosT;
        const float L0 = s_hdr[FH_LREL + 0], L1 = s_hdr[FH_LREL + 1], L2 = s_hdr[FH_LREL + 2];   // the light relative to the eye, world axes (frame 0)
        // list positions that hold anything but a box (capsules, cones, scaled shapes): a covered tile with one of those among its candidates stays general
        const unsigned long long nonBox = __ballot(lane < nVis && (__float_as_uint(s_vis[2 * min(lane, max(nVis - 1, 0))].w) & 15u) != (unsigned)PRIM_BOX);
#pragma unroll
        for (int c = 0; c < CH; ++c) {
            if (wave != (c % NW) || !tvalid[c]) continue;
            const int u = 64 * c + lane;
            const uint4 tc = s_tile[u];
            const unsigned long long m = ((unsigned long long)tc.y << 32) | tc.x, wbm = m & wb0, others = m & ~wb0;
            unsigned info = TC_GENERAL;
            if (m == 0ull) info = TC_EMPTY;
            else if (wbm != 0ull && (wbm & (wbm - 1ull)) == 0ull && tc.z != 0u) {   // one world box, and one of its faces covers the tile
                const int pos = __ffsll((long long)wbm) - 1, k = (int)tc.z - 1;
                if (others == 0ull) {
                    bool maybe = true;   // may a pixel of the tile lie in the highlight cone?
                    if (usable) {
                        const float lok = reinterpret_cast<const float *>(s_vis)[8 * pos + k], hik = reinterpret_cast<const float *>(s_vis)[8 * pos + 4 + k];
                        const float plane = lok > 0.0f ? lok : hik;
                        const float m0 = k == 0 ? 2.0f * plane - L0 : L0, m1 = k == 1 ? 2.0f * plane - L1 : L1, m2 = k == 2 ? 2.0f * plane - L2 : L2;   // L'
                        const float xc = float(tX0[c]) + 0.5f * WX, yc = float(tY0[c]) + 0.5f * WY;
                        const float dcx = sx * xc + ox, dcy = sy * yc + oy;
                        const float d0 = (camv[3] * dcx + camv[4] * dcy) - camv[5], d1 = (camv[6] * dcx + camv[7] * dcy) - camv[8],
                                          d2 = (camv[9] * dcx + camv[10] * dcy) - camv[11];
                        const float dot = (d0 * m0 + d1 * m1) + d2 * m2;
                        const float dd = (d0 * d0 + d1 * d1) + d2 * d2, mm = (m0 * m0 + m1 * m1) + m2 * m2;
                        maybe = dot > 0.0f && dot * dot >= cosT2 * (dd * mm);
                    }
                    info = (maybe ? (unsigned)TC_PLANAR_SPEC : (unsigned)TC_PLANAR) | ((unsigned)k << 3) | ((unsigned)pos << 5);
                } else if ((others & nonBox) == 0ull && overlayOn) info = (unsigned)TC_OVERLAY | ((unsigned)k << 3) | ((unsigned)pos << 5);
            }
            if (info == TC_GENERAL) {
                // The signs of the rays' world components over the tile (bits 11..14: 8 | sign bits where every component keeps its sign; general_tile,
                // box_key_signed).  A component is affine in the pixel: it keeps the sign -- and the magnitude, far above either arithmetic's rounding -- of
                // the tile's four corner rays if those agree.
                const float xa = float(tX0[c]), xb = float(tX1[c]), ya = float(tY0[c]), yb = float(tY1[c]);
                unsigned sg = 8u;
#pragma unroll
                for (int a = 0; a < 3; ++a) {
                    const float Aa = camv[3 + 3 * a] * sx, Ba = camv[4 + 3 * a] * sy, Ca = (camv[3 + 3 * a] * ox + camv[4 + 3 * a] * oy) - camv[5 + 3 * a];
                    const float v00 = __builtin_fmaf(Aa, xa, __builtin_fmaf(Ba, ya, Ca)), v10 = __builtin_fmaf(Aa, xb, __builtin_fmaf(Ba, ya, Ca));
                    const float v01 = __builtin_fmaf(Aa, xa, __builtin_fmaf(Ba, yb, Ca)), v11 = __builtin_fmaf(Aa, xb, __builtin_fmaf(Ba, yb, Ca));
                    const float lo4 = __builtin_fminf(__builtin_fminf(v00, v10), __builtin_fminf(v01, v11)),
                                                      hi4 = __builtin_fmaxf(__builtin_fmaxf(v00, v10), __builtin_fmaxf(v01, v11));
                    if (hi4 < -1e-4f) sg |= 1u << a;
                    else if (!(lo4 > 1e-4f)) sg = 0u;
                    if (sg == 0u) break;
                }
                info |= sg << 11;
            }
            s_tile[u].w = info;
            const unsigned tclass = info & 7u, cat = tclass == TC_GENERAL ? 0u : tclass == TC_OVERLAY ? 1u
                    : tclass == TC_PLANAR_SPEC ? 2u : tclass == TC_PLANAR ? 3u : 4u;
            // RUNS: neighbours in a tile row that one face of one box covers alike (the same class word), or that are empty, are ONE entry of the drawing order
            // -- at most eight tiles (a tile row of a 128-pixel frame): the run's uniform set-up (the face's plane, colour, light term), its rows' ray terms
            // and its hand-out are paid once.  The
            // valid tiles of a chunk are its first lanes (every lane here is one); a run's tiles are neighbours in one tile row.
            const bool runs = tclass == TC_PLANAR || tclass == TC_PLANAR_SPEC || (tclass == TC_EMPTY && !bulkClear);
            // (the previous lane's tile is the left neighbour only where the workgroup owns whole tile rows: a frame cut into `split` pieces deals its tiles
            // out in fours)
            const unsigned txy = (unsigned)tX0[c] | ((unsigned)tY0[c] << 16);
            const unsigned prevInfo = (unsigned)__shfl_up((int)info, 1, 64), prevTxy = (unsigned)__shfl_up((int)txy, 1, 64);
            const bool natHead = !runs || lane == 0 || prevInfo != info || prevTxy + (unsigned)TILE_W != txy;
            const unsigned long long nat = __ballot(natHead), below = nat & ((2ull << lane) - 1ull);   // (bit `lane` and lower; lane 0 is a head: never empty)
            const int posInRun = lane - (63 - __clzll((long long)below));
            const bool head = natHead || (posInRun & 7) == 0;
            const unsigned long long heads = __ballot(head), valid = __ballot(true), above = lane < 63 ? heads >> (lane + 1) : 0ull;
            const int nextHead = above ? lane + __ffsll((long long)above) : 64;
            const int len = min(nextHead, (int)__popcll(valid)) - lane;                                  // (1 .. 8)
            // cat | run length << 3 | place within the category << 7; 7: not a run's first tile, not listed.  (The covering face this word held is in the class
            // word now.)
            s_tile[u].z = head ? cat | ((unsigned)len << 3) | ((unsigned)atomicAdd(&s_cnt[cat], 1) << 7) : 7u;
            s_txy[u] = txy;
        }
        __syncthreads();
    }
}

// the pixels of `n` neighbouring tiles of one tile row (first pixel (tx0, ty0)) that the face of axis k (the one towards the eye) of the world box at list
// position `pos` covers: the face's constants and the rows' ray terms once, per tile the columns' terms and the pixels -- planar_tile's arithmetic, operation
// for operation
template <int NP, bool SPEC>
__device__ __forceinline__ void planar_run(int pos, int k, int n, const float4 *s_vis, const float *s_hdr, const float4 *s_col, const float4 *s_row,
                                           const float2 *s_rowq, const float *s_colq, float nzk, int tx0, int ty0, int lane, const PixOut &po)
{
    const int W = po.W, H = po.H;
    const float lok = uniform_f32(reinterpret_cast<const float *>(s_vis)[8 * pos + k]),
                                  hik = uniform_f32(reinterpret_cast<const float *>(s_vis)[8 * pos + 4 + k]);
    const float plane = lok > 0.0f ? lok : hik;   // the face towards the eye
    const unsigned color = (unsigned)__builtin_amdgcn_readfirstlane(__float_as_uint(reinterpret_cast<const float *>(s_vis)[8 * pos + 7]));
    const float cr = float((color >> 16) & 255u), cg = float((color >> 8) & 255u), cb = float(color & 255u);
    const float lk = uniform_f32(s_hdr[FH_LREL + k]);
    const float lks = plane > 0.0f ? lk : 0.0f - lk;
    int lpx = lane;
    asm volatile("" : "+v"(lpx));   // (this lane's place in the tile is formed here, per run: kept across the tile loop it is spilled at seven waves per SIMD)
    const int lx = lpx & (TILE_W - 1), py0 = ty0 + lpx / TILE_W;
    float rowk[NP];
    float2 rq[NP];
#pragma unroll
    for (int j = 0; j < NP; ++j) {
        const int pyc = min(py0 + TILE_H * j, H - 1);
        rowk[j] = reinterpret_cast<const float *>(s_row)[4 * pyc + 1 + k];
        rq[j] = s_rowq[pyc];
    }
#pragma unroll 1
    for (int q = 0; q < n; ++q) {
        const int px = tx0 + TILE_W * q + lx, pxc = min(px, W - 1);
        const float colk = reinterpret_cast<const float *>(s_col)[4 * pxc + 1 + k];
        const float cq = s_colq[pxc];
#pragma unroll
        for (int j = 0; j < NP; ++j) {
            const float dk = (colk + rowk[j]) + nzk;              // the ray's component along k: the same sum the general path forms
            const float t = plane * __builtin_amdgcn_rcpf(dk);    // == min(lo_k inv_k, hi_k inv_k) of the slab test
            const float nv = t * __builtin_fabsf(dk);
            const unsigned rgba = phong_tail<SPEC>(t, nv - lks, nv, cq + rq[j].x, rq[j].y, cr, cg, cb);
            put_px(po, px, py0 + TILE_H * j, rgba);
        }
    }
}

// the pixels of a tile that the face of axis k (the one towards the eye) of the world box at list position `pos` covers
template <int NP, bool SPEC>
__device__ __forceinline__ void planar_tile(int pos, int k, const float4 *s_vis, const float *s_hdr, const float4 *s_col, const float4 *s_row,
                                            const float2 *s_rowq, const float *s_colq, float nzk, int px, int py0, const PixOut &po)
{
    const int W = po.W, H = po.H;
    const float lok = uniform_f32(reinterpret_cast<const float *>(s_vis)[8 * pos + k]),
                                  hik = uniform_f32(reinterpret_cast<const float *>(s_vis)[8 * pos + 4 + k]);
    const float plane = lok > 0.0f ? lok : hik;   // the face towards the eye
    const unsigned color = (unsigned)__builtin_amdgcn_readfirstlane(__float_as_uint(reinterpret_cast<const float *>(s_vis)[8 * pos + 7]));
    const float cr = float((color >> 16) & 255u), cg = float((color >> 8) & 255u), cb = float(color & 255u);
    const float lk = uniform_f32(s_hdr[FH_LREL + k]);   // the light along k, relative to the eye (frame 0: world axes)
    const float lks = plane > 0.0f ? lk : 0.0f - lk;    // sgn (Lrel_k - t d_k) = t |d_k| - (d_k < 0 ? -Lrel_k : Lrel_k), and d_k has the plane offset's sign
    const int pxc = min(px, W - 1);
    const float colk = reinterpret_cast<const float *>(s_col)[4 * pxc + 1 + k];
    const float cq = s_colq[pxc];
#pragma unroll
    for (int j = 0; j < NP; ++j) {
        const int py = py0 + TILE_H * j, pyc = min(py, H - 1);
        const float rowk = reinterpret_cast<const float *>(s_row)[4 * pyc + 1 + k];
        const float2 rq = s_rowq[pyc];
        const float dk = (colk + rowk) + nzk;                 // the ray's component along k: the same sum the general path forms
        const float t = plane * __builtin_amdgcn_rcpf(dk);    // == min(lo_k inv_k, hi_k inv_k) of the slab test
        const float nv = t * __builtin_fabsf(dk);
        const unsigned rgba = phong_tail<SPEC>(t, nv - lks, nv, cq + rq.x, rq.y, cr, cg, cb);
        put_px(po, px, py, rgba);
    }
}

// a tile nothing can be seen through: the clear colour (0, 0, 0), alpha 255 -- 16 pixels of a row are 64 bytes: four lanes per row write 16 bytes each
// (a quarter of the store instructions of one dword per pixel) where the frame's rows allow it
typedef unsigned v4u_t __attribute__((ext_vector_type(4)));
template <int NP>
__device__ __forceinline__ void clear_tile(const PixOut &po, int tx0, int ty0, int lane)
{
    constexpr int TH = TILE_H * NP;
    // (lane and colour are laundered: what is derived from them is formed here, per tile -- hoisted out of the tile loop it costs the kernel registers it does
    // not have at seven waves per SIMD)
    unsigned c = 0xff000000u;
    asm volatile("" : "+v"(lane), "+v"(c));
    if ((po.W & 3) == 0 && tx0 + TILE_W <= po.W) {   // (uniform; the frame's base is 16-byte aligned: raster_fast_body checks)
        const int row = ty0 + (lane >> 2), col = tx0 + 4 * (lane & 3);
        const v4u_t v = {c, c, c, c};
        if (lane < 4 * TH && row < po.H) __builtin_amdgcn_raw_buffer_store_b128(v, po.rsrc, (unsigned)(__mul24(row, po.W4) + col * 4), 0, PIXEL_AUX);
        return;
    }
    const int px = tx0 + (lane & (TILE_W - 1)), py0 = ty0 + lane / TILE_W;
#pragma unroll
    for (int j = 0; j < NP; ++j) {
        const int py = py0 + TILE_H * j;
        if (px < po.W && py < po.H) __builtin_amdgcn_raw_buffer_store_b32(c, po.rsrc, (unsigned)(__mul24(py, po.W4) + px * 4), 0, PIXEL_AUX);
    }
}

// A covered tile with company: ONE world box whose face covers the tile (as in planar_tile) and, in front of it or not, boxes of other frames of reference
// whose rectangles meet the tile -- the viewer's time bar along the bottom row of tiles of every frame (scenario_default.hpp:137-145,164-169), the object it
// carries (component_object_stacking.hpp:146-152).  The general path would set up the whole ray (three v_rcp_f32), run the slab test of the world box and
// recover its entry axis per pixel; here the face's depth comes as in planar_tile -- the same product the slab test forms, so the same depth key -- the other
// boxes are intersected by the very function the general path uses (other_box), the nearest wins by the same key comparison, and the pixel is shaded by
// phong_tail with the face's constants or, where another box won, by the general path's fast_shade: the same bytes (test_planar_tiles_change_no_byte).
template <bool SHAPES, unsigned POS_MASK, int NP>
__device__ __forceinline__ void overlay_tile(int posA, int k, unsigned long long rest, const float4 *s_vis,
                                             const float *s_hdr, const float *camv, int viewer, const float4 *s_col,
                                             const float4 *s_row, const float2 *s_rowq, const float *s_colq,
                                                     float nzm0, float nzm1, float nzm2, int px, int py0, const PixOut &po)
{
    const int W = po.W, H = po.H;
    const float lok = uniform_f32(reinterpret_cast<const float *>(s_vis)[8 * posA + k]),
                                  hik = uniform_f32(reinterpret_cast<const float *>(s_vis)[8 * posA + 4 + k]);
    const float plane = lok > 0.0f ? lok : hik;
    const unsigned color = (unsigned)__builtin_amdgcn_readfirstlane(__float_as_uint(reinterpret_cast<const float *>(s_vis)[8 * posA + 7]));
    const float cr = float((color >> 16) & 255u), cg = float((color >> 8) & 255u), cb = float(color & 255u);
    const float lk = uniform_f32(s_hdr[FH_LREL + k]);
    const float lks = plane > 0.0f ? lk : 0.0f - lk;
    const int pxc = min(px, W - 1);
    const float4 cx = s_col[pxc];
    const float cq = s_colq[pxc];
#pragma unroll
    for (int j = 0; j < NP; ++j) {
        const int py = py0 + TILE_H * j, pyc = min(py, H - 1);
        const float4 ry = s_row[pyc];
        const float2 rq = s_rowq[pyc];
        const V3 dw = v3((cx.y + ry.y) + nzm0, (cx.z + ry.z) + nzm1, (cx.w + ry.w) + nzm2);
        const float dk = k == 0 ? dw.x : k == 1 ? dw.y : dw.z;
        const float t = plane * __builtin_amdgcn_rcpf(dk);
        // box_key of the covering face: it is hit (classify_tiles), its entry depth is this product
        const unsigned keyA = ((__float_as_uint(t) - KEY_NEAR) & ~POS_MASK) | (unsigned)posA;
        unsigned best = keyA;
        unsigned long long m = rest;
        while (m) {
            const int pos = __ffsll((long long)m) - 1;
            m &= m - 1;
            const float4 lo = s_vis[2 * pos], hi = s_vis[2 * pos + 1];
            const int qfr = (int)((__builtin_amdgcn_readfirstlane(__float_as_uint(lo.w)) >> 4) & 15);
            float tb;
            const bool hit = other_box(lo, hi, qfr, s_hdr, viewer, dw, cx.x, ry.x, tb);
            best = min(best, hit_key<POS_MASK>(hit, tb, pos));
        }
        const float a2 = cq + rq.x, ldc = rq.y;
        unsigned rgba;
        if (best == keyA) {
            const float nv = t * __builtin_fabsf(dk);
            rgba = phong_tail<true>(t, nv - lks, nv, a2, ldc, cr, cg, cb);
        } else rgba = fast_shade<SHAPES, POS_MASK>(best, v3(0.0f, 0.0f, 0.0f), s_vis, s_hdr, camv, viewer, dw, v3(0.0f, 0.0f, 0.0f), cx.x, ry.x, a2, ldc);
        put_px(po, px, py, rgba);
    }
}

// The general path for a tile of a classified frame: its list is ONE culling round (at most 64 primitives) and the tile's candidates -- mv0, not empty -- came
// out of the classification, so the rays are set up at once and there is no loop over rounds (the same arithmetic, in the same order, as the loop in
// raster_fast_body).
template <bool SHAPES, unsigned POS_MASK, int NP>
__device__ __forceinline__ void general_tile(unsigned long long mv0, unsigned long long wb0, unsigned signs, const float4 *s_vis,
                                             const float *s_hdr, const float *camv, int viewer, const float4 *s_col,
                                             const float4 *s_row, const float2 *s_rowq, const float *s_colq,
                                                     float nzm0, float nzm1, float nzm2, int px, int py0, const PixOut &po)
{
    const int W = po.W, H = po.H;
    const int pxc = min(px, W - 1);
    const float4 cx = s_col[pxc];
    const float cq = s_colq[pxc];
    const float dcx = cx.x;
    V3 dw[NP], inv[NP], bn[NP];
    float dcy[NP], a2[NP], ldc[NP];
    unsigned best[NP];
#pragma unroll
    for (int j = 0; j < NP; ++j) {
        const int pyc = min(py0 + TILE_H * j, H - 1);
        const float4 ry = s_row[pyc];
        const float2 rq = s_rowq[pyc];
        dcy[j] = ry.x;
        dw[j] = v3((cx.y + ry.y) + nzm0, (cx.z + ry.z) + nzm1, (cx.w + ry.w) + nzm2);
        inv[j] = v3(__builtin_amdgcn_rcpf(dw[j].x), __builtin_amdgcn_rcpf(dw[j].y), __builtin_amdgcn_rcpf(dw[j].z));
        a2[j] = cq + rq.x; ldc[j] = rq.y;
        bn[j] = v3(0.0f, 0.0f, 0.0f);
        best[j] = ~0u;
    }
    // signs: 8 | the sign bits of the rays' world components where each is the same over the whole tile (classify_tiles), else 0
    const unsigned long long wbm = mv0 & wb0;
    switch (signs) {   // (uniform)
    case 8: box_run<POS_MASK, NP, 0>(wbm, 0, inv, s_vis, best); break;
    case 9: box_run<POS_MASK, NP, 1>(wbm, 0, inv, s_vis, best); break;
    case 10: box_run<POS_MASK, NP, 2>(wbm, 0, inv, s_vis, best); break;
    case 11: box_run<POS_MASK, NP, 3>(wbm, 0, inv, s_vis, best); break;
    case 12: box_run<POS_MASK, NP, 4>(wbm, 0, inv, s_vis, best); break;
    case 13: box_run<POS_MASK, NP, 5>(wbm, 0, inv, s_vis, best); break;
    case 14: box_run<POS_MASK, NP, 6>(wbm, 0, inv, s_vis, best); break;
    case 15: box_run<POS_MASK, NP, 7>(wbm, 0, inv, s_vis, best); break;
    default: box_run<POS_MASK, NP>(wbm, 0, inv, s_vis, best); break;
    }
    unsigned long long rest = mv0 & ~wb0;
    while (rest) {
        const int pos = __ffsll((long long)rest) - 1;
        rest &= rest - 1;
#pragma unroll
        for (int j = 0; j < NP; ++j) {
            V3 n = v3(0, 0, 0);
            const unsigned key = fast_other<SHAPES, POS_MASK>(pos, s_vis, s_hdr, camv, viewer, dw[j], dcx, dcy[j], n);
            if (key < best[j]) { best[j] = key; bn[j] = n; }
        }
    }
#pragma unroll
    for (int j = 0; j < NP; ++j) {
        const unsigned rgba = fast_shade<SHAPES, POS_MASK>(best[j], bn[j], s_vis, s_hdr, camv, viewer, dw[j], inv[j], dcx, dcy[j], a2[j], ldc[j]);
        put_px(po, px, py0 + TILE_H * j, rgba);
    }
}

// HEXF (Hex scenarios): most primitives are boxes in one of the three wall frames (header records 8..10: rotations about Y by +30, -30, 90
// degrees around the world origin).  The ray's inverse direction in each of them is set up once per pixel -- four v_rcp_f32, the 90 degree
// frame only permutes the world one -- and their boxes run through the same prefetching loop as the world's; which list positions hold a box
// of which frame is found by the wave itself while culling (one LDS read per lane and round) instead of coming from the header.
// (Measured and rejected for these long lists, ~700 primitives per frame: a two-level variant -- a wave culls the list against a block of 2 x 2
// tiles once and its tiles only look at the survivors -- 164 / 187 us against 170 / 189 us at 128 x 128 and slower at 64 x 64: the time
// goes into intersecting the ~19 boxes that survive per tile, not into the rectangle tests.)
// NP (pixels per lane): a wave's tile is 16 x (4 NP) pixels, lane (lx, ly) traces rows ty0 + ly, ty0 + ly + 4, ...  With NP = 2 the culling
// ballots, the scalar loop control, the LDS fetch of every surviving record and the column terms of the ray are paid once per two pixels,
// and the two pixels' dependency chains (rcp -> slab test -> key, shading) interleave; per-pixel arithmetic is unchanged, and culling being
// conservative the pixels are identical to NP = 1 (tests/test_fast_pixels_gpu.py: test_pixels_per_lane_variants_agree).
// LDS of the two bodies below comes from their kernel (one buffer, carved here), so that a kernel which holds both (raster_union_all_kernel)
// pays for the larger of the two, not for their sum
constexpr int fast_lds_bytes(int maxvis) { return 40 * maxvis + 4 * FH_FLOATS; }    // records 32 B + rectangles 8 B per primitive, frame header
constexpr int glist_lds_bytes(int maxvis) { return 9 * maxvis + 4 * FH_FLOATS; }    // rectangles 8 B + class 1 B per primitive, frame header

#ifdef MV_RASTER_TIMING
#define RT_COUNT(i, n) do { const unsigned long long n_ = (unsigned long long)(n); if (fa.rdbg && lane == 0) atomicAdd(fa.rdbg + (size_t)16384 * 4 * 8 + (i), n_); } while (0)   /* census of the tile loop (wave-uniform events) */
#else
#define RT_COUNT(i, n) do { } while (0)
#endif
// a classified tile -- or, LISTED, an entry of the drawing order: a run of up to eight planar / empty neighbours of a tile row -- by its class (classify_tiles)
template <bool SHAPES, unsigned POS_MASK, int NP, bool LISTED>
__device__ __forceinline__ void classified_tile(const FastArgs &fa, const int u, const int tx0, const int ty0, const uint4 *s_tile,
                                                const float4 *s_vis, const float *s_hdr, const float *camv, int viewer,
                                            const float4 *s_col, const float4 *s_row, const float2 *s_rowq,
                                                    const float *s_colq, float nzm0, float nzm1, float nzm2, unsigned long long wb0,
                                            int lane, const PixOut &po)
{
    const unsigned info = (unsigned)__builtin_amdgcn_readfirstlane(s_tile[u].w), tclass = info & 7u;
    const int run = LISTED ? (int)(((unsigned)__builtin_amdgcn_readfirstlane(s_tile[u].z) >> 3) & 15u) : 1;
    RT_COUNT(0, run);                                 // classified tiles
    if (tclass == TC_EMPTY) {   // nothing: the clear colour
        RT_COUNT(1, run);
        for (int q = 0; q < run; ++q) clear_tile<NP>(po, tx0 + TILE_W * q, ty0, lane);
        return;
    }
    if (LISTED && tclass <= TC_PLANAR_SPEC) {   // a run of tiles one face covers
        const int k = (int)((info >> 3) & 3u), posA = (int)(info >> 5);
        const float nzk = k == 0 ? nzm0 : k == 1 ? nzm1 : nzm2;
        RT_COUNT(2, run);
        RT_COUNT(12, tclass == TC_PLANAR ? run : 0);
        // (no pixel of these tiles lies in the highlight cone)
        if (tclass == TC_PLANAR) planar_run<NP, false>(posA, k, run, s_vis, s_hdr, s_col, s_row, s_rowq, s_colq, nzk, tx0, ty0, lane, po);
        else planar_run<NP, true>(posA, k, run, s_vis, s_hdr, s_col, s_row, s_rowq, s_colq, nzk, tx0, ty0, lane, po);
        return;
    }
    int lpx = lane;
    // (this lane's place in the tile is formed here, per tile: kept across the loop it is spilled at seven waves per SIMD, and the reload's s_waitcnt vmcnt(0)
    // also waits for the previous tile's pixel stores)
    asm volatile("" : "+v"(lpx));
    const int px = tx0 + (lpx & (TILE_W - 1)), py0 = ty0 + lpx / TILE_W;
    if (tclass <= TC_OVERLAY) {   // one world box, and one of its faces covers the tile
#if defined(MV_RASTER_DEBUG_SKIP) && MV_RASTER_DEBUG_SKIP == 3   // (measurement builds: covered tiles cost nothing)
        return;
#endif
        const int k = (int)((info >> 3) & 3u), posA = (int)(info >> 5);
        const float nzk = k == 0 ? nzm0 : k == 1 ? nzm1 : nzm2;
        RT_COUNT(2, tclass != TC_OVERLAY);
        RT_COUNT(12, tclass == TC_PLANAR);
        RT_COUNT(13, tclass == TC_OVERLAY);
        // (no pixel of the tile lies in the highlight cone)
        if (tclass == TC_PLANAR) planar_tile<NP, false>(posA, k, s_vis, s_hdr, s_col, s_row, s_rowq, s_colq, nzk, px, py0, po);
        else if (tclass == TC_PLANAR_SPEC) planar_tile<NP, true>(posA, k, s_vis, s_hdr, s_col, s_row, s_rowq, s_colq, nzk, px, py0, po);
        else {   // ... and boxes of other frames of reference (the time bar, a carried object)
            const uint2 tm = *reinterpret_cast<const uint2 *>(&s_tile[u]);
            const unsigned long long others = (((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane(tm.y) << 32)
                                               | (unsigned)__builtin_amdgcn_readfirstlane(tm.x)) & ~wb0;
            overlay_tile<SHAPES, POS_MASK, NP>(posA, k, others, s_vis, s_hdr, camv, viewer, s_col, s_row, s_rowq, s_colq, nzm0, nzm1, nzm2, px, py0, po);
        }
        return;
    }
#if defined(MV_RASTER_DEBUG_SKIP) && MV_RASTER_DEBUG_SKIP == 2   // (measurement builds: general tiles cost nothing)
    return;
#endif
    const uint2 tm = *reinterpret_cast<const uint2 *>(&s_tile[u]);
    const unsigned long long mv0 = ((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane(tm.y) << 32) | (unsigned)__builtin_amdgcn_readfirstlane(tm.x);
    RT_COUNT(4, 1);
    RT_COUNT(5, __popcll(mv0 & wb0));
    RT_COUNT(6, __popcll(mv0 & ~wb0));
    general_tile<SHAPES, POS_MASK, NP>(mv0, wb0, fa.planar == 3 ? 0u : (info >> 11) & 15u, s_vis, s_hdr, camv,
                                       viewer, s_col, s_row, s_rowq, s_colq, nzm0, nzm1, nzm2, px, py0, po);
}

template <int MAXVIS, bool SHAPES, bool HEXF, int NP, bool CLS = true, int NT = 256>   // CLS: with the tile classification
__device__ __forceinline__ void raster_fast_body(const FastArgs &fa, uint32_t *obs, int W, int H, int split, int blk, unsigned char *lds)
{
    constexpr unsigned POS_MASK = MAXVIS - 1;
    constexpr int TH = TILE_H * NP;           // tile height
    extern __shared__ __attribute__((aligned(16))) unsigned char s_dyn[];
    float4 *s_vis = reinterpret_cast<float4 *>(lds);                  // [2 * MAXVIS] Prim records as (lo, meta) (hi, colour)
    short4 *s_rect = reinterpret_cast<short4 *>(lds + 32 * MAXVIS);   // [MAXVIS]
    float *s_hdr = reinterpret_cast<float *>(lds + 40 * MAXVIS);      // [FH_FLOATS]
    static_assert(MAXVIS / 64 <= 16, "world-box masks: 16 x 64 positions");

    float4 *s_col = reinterpret_cast<float4 *>(s_dyn);   // per column i: (dc.x, c00*dc.x, c10*dc.x, c20*dc.x)
    float4 *s_row = s_col + W;                            // per row j:    (dc.y, c01*dc.y, c11*dc.y, c21*dc.y)
    float2 *s_rowq = reinterpret_cast<float2 *>(s_row + H);   // (dc.y^2 + 1, L . dc = 4 dc.y - 2)
    float *s_colq = reinterpret_cast<float *>(s_rowq + H);    // dc.x^2   (last: keeps every table naturally aligned for odd W)

    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
#ifdef MV_RASTER_TIMING
    unsigned long long rt_[6];
    rt_[0] = __builtin_amdgcn_s_memtime(); rt_[5] = __builtin_amdgcn_s_memrealtime();
#define RT_MARK(i) rt_[i] = __builtin_amdgcn_s_memtime()
#else
#define RT_MARK(i) do { } while (0)
#endif
    fast_publish(fa, blk);
    const FastFrame ff = fast_prologue<MAXVIS, false, NT>(fa, blk, W, H, split, s_vis, s_rect, s_hdr, s_col, s_row, s_rowq, s_colq);
    RT_MARK(1);
    const int frame = ff.frame, part = ff.part, viewer = ff.viewer, nVis = ff.nVis;
    split = ff.split;   // (fine-grained tail: this workgroup's frame may be cut into more pieces than the launch's nominal number)

    const float *camv = s_hdr + FH_CAM + FH_CAM_STRIDE * viewer;   // eye(3) c(9) origin(3)
    const float nzm0 = uniform_f32(-camv[3 + 2]), nzm1 = uniform_f32(-camv[3 + 5]), nzm2 = uniform_f32(-camv[3 + 8]);
    uint32_t *out = obs + (size_t)frame * W * H;
    const int tilesX = (W + TILE_W - 1) / TILE_W, tilesY = (H + TH - 1) / TH;
    const int numTiles = tilesX * tilesY;
    PixOut po;
    po.rsrc = __builtin_amdgcn_make_buffer_rsrc(out, /*stride*/ 0, W * H * 4, 0x00020000);
    po.W = W; po.H = H; po.W4 = 4 * W;
    po.edgeless = (W % TILE_W) == 0 && (H % TH) == 0;

    // What the first round of 64 list positions needs is the same for every tile of the frame: this lane's primitive's rectangle and the
    // world-box mask stay in registers (two VGPRs, two SGPRs) instead of costing two dependent LDS round trips per tile (most frames of the
    // short-list scenarios hold fewer than 64 visible primitives: their only round)
    // (the rectangle only in the one-pixel build: the two-pixel one has no VGPRs to spare at 7 waves per SIMD)
    uint2 rr0 = make_uint2(0u, 0u);
    if (NP == 1) rr0 = *reinterpret_cast<const uint2 *>(&s_rect[min(lane, max(nVis - 1, 0))]);
    const unsigned long long wb0 = uniform_u64(*reinterpret_cast<const unsigned long long *>(s_hdr + FH_WB));

    // planar tiles (classify_tiles): where the frame's whole list is one culling round and the workgroup's share of the tiles fits the table
    constexpr bool PLANAR = !HEXF && CLS;
    // (the tables live in the part of the record buffer a classified frame -- at most 64 visible primitives -- leaves unused: records 64 .. 255)
    static_assert(!PLANAR || (MAXVIS == 256 && CLS_MAX_TILES == 128), "classification tables: 256 edge functions + 128 tiles = 192 records");
    float4 *s_line = s_vis + 2 * 64;
    uint4 *s_tile = reinterpret_cast<uint4 *>(s_vis + 2 * 64 + 256);
    const int perWG = (numTiles - part * 4 + 4 * split - 1) / (4 * split) * 4;   // this workgroup's tiles, rounded up to four per turn of its waves
    // (uniform over the workgroup; few tiles do not repay the pass over the list)
    const bool cls = PLANAR && fa.planar && nVis <= 64 && perWG <= CLS_MAX_TILES && perWG >= 32;
    __shared__ int s_next;   // the tile loop's hand-out counter (below)
    __shared__ int s_cnt[8];                                  // classified frames: tiles per category (classify_tiles)
    __shared__ unsigned s_txy[CLS_MAX_TILES];                 // ... every tile's first pixel, x | y << 16
    __shared__ unsigned short s_order[CLS_MAX_TILES], s_empty[CLS_MAX_TILES];   // ... the tiles in drawing order (most expensive class first); the empty ones
    if (tid == 0) s_next = NT / 64;
    int nList = 0;   // classified: tiles in s_order
#ifndef MV_TILE_LIST
#define MV_TILE_LIST 1   // (0: the tiles in frame order, the empty ones handed out like the others -- the A/B of r09c)
#endif
    constexpr bool LISTED = PLANAR && MV_TILE_LIST != 0;
    // whole tiles, rows of whole 16-byte groups: the empty tiles are cleared by all threads together (below) and stay out of the list
    const bool bulk = LISTED && po.edgeless && (W & 3) == 0;
    if (PLANAR && cls && !LISTED) classify_tiles<TH, NT>(s_tile, s_line, s_vis, s_rect, s_hdr, camv, nVis, wb0, W, H,
        part, split, tilesX, numTiles, perWG, fa.planar != 2, s_cnt, s_txy, true);
    else if (PLANAR && cls) {
        // (ends with a barrier)
        classify_tiles<TH, NT>(s_tile, s_line, s_vis, s_rect, s_hdr, camv, nVis, wb0, W, H, part, split,
                               tilesX, numTiles, perWG, fa.planar != 2, s_cnt, s_txy, bulk);
        const int c0 = s_cnt[0], c1 = s_cnt[1], c2 = s_cnt[2], c3 = s_cnt[3], nE = s_cnt[4];
        nList = __builtin_amdgcn_readfirstlane(c0 + c1 + c2 + c3 + (bulk ? 0 : nE));
        if (tid < CLS_MAX_TILES) {
            const int tile = ((tid >> 2) * split + part) * 4 + (tid & 3);
            if (tid < perWG && tile < numTiles) {
                const unsigned z = s_tile[tid].z, cat = z & 7u, place = z >> 7;
                const int base = cat == 0u ? 0 : cat == 1u ? c0 : cat == 2u ? c0 + c1 : cat == 3u ? c0 + c1 + c2 : c0 + c1 + c2 + c3;
                if (cat == 4u && bulk) s_empty[place] = (unsigned short)tid;
                else if (cat != 7u) s_order[base + (int)place] = (unsigned short)tid;   // (7: inside a run, drawn with the run's first tile)
            }
        }
        __syncthreads();
        if (bulk) {   // a tile is TH rows of four 16-byte groups: TH * 4 threads per tile, NT / (TH * 4) tiles per turn
            constexpr int CPT = TH * 4;
            static_assert(NT % CPT == 0, "threads per empty tile");
            const int sub = tid % CPT;
            unsigned c = 0xff000000u;
            asm volatile("" : "+v"(c));
            const v4u_t v = {c, c, c, c};
            const unsigned inoff = (unsigned)__mul24(sub >> 2, po.W4) + (unsigned)(sub & 3) * 16u;
            for (int e = tid / CPT; e < nE; e += NT / CPT) {
                const unsigned txy = s_txy[s_empty[e]];
                __builtin_amdgcn_raw_buffer_store_b128(v, po.rsrc, (unsigned)__mul24((int)(txy >> 16), po.W4) + (txy & 0xffffu) * 4u + inoff, 0, PIXEL_AUX);
            }
            if (wave == 0) { RT_COUNT(0, nE); RT_COUNT(1, nE); }
        }
    } else __syncthreads();
    RT_MARK(2);
// (measurement builds: the pass's fixed cost -- prologue + classification -- alone; pixels are NOT drawn)
#if defined(MV_RASTER_DEBUG_SKIP) && MV_RASTER_DEBUG_SKIP == 1
    return;
#endif

    // The workgroup's tiles are handed out one at a time (an LDS counter; the next index is requested while the current tile is drawn): a wave
    // that always drew the same tile column of its frame -- tile index = wave mod 4 -- lived as long as the most crowded column, and the
    // workgroup's other three waited for it 4-5 us on average, up to 17 (r04e).  u = 4 j + w is tile (j split + part) 4 + w of the frame.
    int unext = 0;
    // classified frames, LISTED: the entries of the drawing order (most expensive class first), one at a time; a tile's first pixel comes from the table (no
    // tile arithmetic, and none of its scalars alive in this loop: at seven waves per SIMD the loop below keeps the register file full)
    if (LISTED && cls)
        for (int it = wave; it < nList; it = __builtin_amdgcn_readfirstlane(unext)) {
            const int u = __builtin_amdgcn_readfirstlane((int)s_order[it]);
            if (lane == 0) unext = atomicAdd(&s_next, 1);
            const unsigned txy = (unsigned)__builtin_amdgcn_readfirstlane(s_txy[u]);
            classified_tile<SHAPES, POS_MASK, NP, LISTED>(fa, u, (int)(txy & 0xffffu), (int)(txy >> 16), s_tile, s_vis, s_hdr,
                                                          camv, viewer, s_col, s_row, s_rowq, s_colq, nzm0, nzm1, nzm2, wb0, lane, po);
        }
    // tile / tilesX == umulhi(tile, ceil(2^32 / tilesX)) while tile * tilesX < 2^32
    const unsigned tilesXinv = (unsigned)((0x100000000ull + (unsigned)tilesX - 1u) / (unsigned)tilesX);
    for (int u = wave; !(LISTED && cls); u = __builtin_amdgcn_readfirstlane(unext)) {
        const int tile = ((u >> 2) * split + part) * 4 + (u & 3);
        if (tile >= numTiles) break;   // (u grows with every request: every later tile of this wave is out of range, too)
        if (lane == 0) unext = atomicAdd(&s_next, 1);
        const int ty = (int)__umulhi((unsigned)tile, tilesXinv), tx = tile - ty * tilesX;
        const int tx0 = tx * TILE_W, ty0 = ty * TH;
        const int tx1 = min(tx0 + TILE_W, W) - 1, ty1 = min(ty0 + TH, H) - 1;
        if (PLANAR && !LISTED && cls) {   // (MV_TILE_LIST=0: the classified tiles in frame order)
            classified_tile<SHAPES, POS_MASK, NP, LISTED>(fa, u, tx0, ty0, s_tile, s_vis, s_hdr, camv, viewer,
                                                          s_col, s_row, s_rowq, s_colq, nzm0, nzm1, nzm2, wb0, lane, po);
            continue;
        }
        // ---- which primitives can this tile's pixels hit?  (first round of 64 list positions)
        unsigned long long mv0;
        int lpx = lane;
        asm volatile("" : "+v"(lpx));
        const int px = tx0 + (lpx & (TILE_W - 1)), py0 = ty0 + lpx / TILE_W;
        const int pxc = min(px, W - 1);
        {   // tile culling: one primitive per lane, four integer compares against its screen rectangle
            int l2 = lane;
            asm volatile("" : "+v"(l2));   // (the rectangle's address is formed here, per tile: kept across the loop it was spilled at seven waves per SIMD)
            const int cpos = min(l2, max(nVis - 1, 0));
            const uint2 rr = NP == 1 ? rr0 : *reinterpret_cast<const uint2 *>(&s_rect[cpos]);   // one 8-byte read; x0 | x1 << 16, y0 | y1 << 16 (all >= 0)
            const bool v = (lane < nVis) & ((int)(rr.x & 0xffffu) <= tx1) & ((int)(rr.x >> 16) >= tx0)
                            & ((int)(rr.y & 0xffffu) <= ty1) & ((int)(rr.y >> 16) >= ty0);
            mv0 = __ballot(v);
            RT_COUNT(3, 1);                                   // unclassified tiles
        }
        RT_COUNT(4, 1);                                       // tiles on the general path
        RT_COUNT(5, __popcll(mv0 & wb0));                     // their slab tests (first round)
        RT_COUNT(6, __popcll(mv0 & ~wb0));                    // their other primitives
        V3 dw[NP], inv[NP];
        V3 ih0[NP], ih1[NP], ih2[NP];   // HEXF: the ray's inverse direction in wall frames 0, 1, 2
        float dcx = 0.0f, dcy[NP], a2[NP], ldc[NP];
        unsigned best[NP];
        V3 bn[NP];                       // normal of the best hit when it is not a box (boxes recover theirs from the entry axis)
#pragma unroll
        for (int j = 0; j < NP; ++j) {
            dw[j] = inv[j] = bn[j] = v3(0, 0, 0);
            ih0[j] = ih1[j] = ih2[j] = v3(0, 0, 0);
            dcy[j] = a2[j] = ldc[j] = 0.0f;
            best[j] = ~0u;
        }
        bool rayReady = false;   // wave-uniform: the rays are set up when the first primitive survives the culling
#pragma unroll 1
        for (int k = 0; k * 64 < nVis; ++k) {
            // ---- tile culling (the first round's: above)
            const int cpos = min(lane + 64 * k, nVis - 1);
            unsigned long long mvis = mv0;
            bool v = (mv0 >> lane) & 1ull;
            if (k > 0) {
                const uint2 rr = *reinterpret_cast<const uint2 *>(&s_rect[cpos]);
                v = (lane + 64 * k < nVis) & ((int)(rr.x & 0xffffu) <= tx1) & ((int)(rr.x >> 16) >= tx0)
                     & ((int)(rr.y & 0xffffu) <= ty1) & ((int)(rr.y >> 16) >= ty0);
                mvis = __ballot(v);
            }
            if (mvis == 0ull) continue;
            if (!rayReady) {
                rayReady = true;
                const float4 cx = s_col[pxc];
                const float cq = s_colq[pxc];
                dcx = cx.x;
#pragma unroll
                for (int j = 0; j < NP; ++j) {
                    const int pyc = min(py0 + TILE_H * j, H - 1);
                    const float4 ry = s_row[pyc];
                    const float2 rq = s_rowq[pyc];
                    dcy[j] = ry.x;
                    dw[j] = v3((cx.y + ry.y) + nzm0, (cx.z + ry.z) + nzm1, (cx.w + ry.w) + nzm2);
                    inv[j] = v3(__builtin_amdgcn_rcpf(dw[j].x), __builtin_amdgcn_rcpf(dw[j].y), __builtin_amdgcn_rcpf(dw[j].z));
                    a2[j] = cq + rq.x; ldc[j] = rq.y;
                    if (HEXF) {   // (same products and sums as mat_tmul with the frame's matrix: c x + (-s) z, s x + c z)
                        const float cx8 = 0.8660254f * dw[j].x, cz8 = 0.8660254f * dw[j].z, hx5 = 0.5f * dw[j].x, hz5 = 0.5f * dw[j].z;
                        ih0[j] = v3(__builtin_amdgcn_rcpf(cx8 - hz5), inv[j].y, __builtin_amdgcn_rcpf(hx5 + cz8));
                        ih1[j] = v3(__builtin_amdgcn_rcpf(cx8 + hz5), inv[j].y, __builtin_amdgcn_rcpf(cz8 - hx5));
                        ih2[j] = v3(0.0f - inv[j].z, inv[j].y, inv[j].x);   // 90 degrees: (x, z) -> (-z, x)
                    }
                }
            }
            unsigned long long rest;
            if (HEXF) {
                const unsigned ml = __float_as_uint(s_vis[2 * cpos].w);   // this lane's primitive: kind | frame << 4 | slot << 8
                const bool box = v && (ml & 15u) == (unsigned)PRIM_BOX;
                const unsigned fl = (ml >> 4) & 15u;
                const unsigned long long m0 = __ballot(box && fl == 0u), m1 = __ballot(box && fl == (unsigned)MAX_AGENTS + 1u),
                                         m2 = __ballot(box && fl == (unsigned)MAX_AGENTS + 2u), m3 = __ballot(box && fl == (unsigned)MAX_AGENTS + 3u);
                box_run<POS_MASK, NP>(m0, k, inv, s_vis, best);
                box_run<POS_MASK, NP>(m1, k, ih0, s_vis, best);
                box_run<POS_MASK, NP>(m2, k, ih1, s_vis, best);
                box_run<POS_MASK, NP>(m3, k, ih2, s_vis, best);
                rest = mvis & ~(m0 | m1 | m2 | m3);
            } else {
                // ---- world-frame boxes (a bit mask from the frame header): the next record is fetched from LDS while the current one is intersected
                const unsigned long long wb = k == 0 ? wb0 : uniform_u64(*reinterpret_cast<const unsigned long long *>(s_hdr + FH_WB + 2 * k));
                box_run<POS_MASK, NP>(mvis & wb, k, inv, s_vis, best);
                rest = mvis & ~wb;
            }
            // ---- everything else: camera-attached boxes, capsules, cones, scaled shapes
            while (rest) {
                const int pos = __ffsll((long long)rest) - 1 + 64 * k;
                rest &= rest - 1;
#pragma unroll
                for (int j = 0; j < NP; ++j) {
                    V3 n = v3(0, 0, 0);
                    const unsigned key = fast_other<SHAPES, POS_MASK>(pos, s_vis, s_hdr, camv, viewer, dw[j], dcx, dcy[j], n);
                    if (key < best[j]) { best[j] = key; bn[j] = n; }
                }
            }
        }
#pragma unroll
        for (int j = 0; j < NP; ++j) {
            RT_COUNT(7, __ballot(best[j] <= (KEY_FAR | POS_MASK)) != 0ull);                     // (wave, pixel row) pairs that shade at all
            RT_COUNT(8, __popcll(__ballot(best[j] <= (KEY_FAR | POS_MASK))));                    // pixels with a hit on the general path
            // ... that shade something that is not a world box
            RT_COUNT(9, __ballot(best[j] <= (KEY_FAR | POS_MASK) && !((wb0 >> (best[j] & 63u)) & 1ull)) != 0ull);
            const unsigned rgba = fast_shade<SHAPES, POS_MASK>(best[j], bn[j], s_vis, s_hdr, camv, viewer, dw[j], inv[j], dcx, dcy[j], a2[j], ldc[j]);
            put_px(po, px, py0 + TILE_H * j, rgba);
        }
    }
#ifdef MV_RASTER_TIMING
    if (fa.rdbg && lane == 0 && blk < 16384 / (NT / 256)) {
        unsigned long long *o = fa.rdbg + ((size_t)blk * (NT / 64) + wave) * 8;
        o[0] = rt_[0]; o[1] = rt_[1]; o[2] = rt_[2]; o[3] = __builtin_amdgcn_s_memtime(); o[4] = rt_[5]; o[5] = __builtin_amdgcn_s_memrealtime();
        o[6] = (unsigned long long)nVis | ((unsigned long long)(cls ? 1 : 0) << 32); o[7] = (unsigned long long)ff.frame;
    }
#endif
}

// ---- the long-list variant ("global list"): Collect and the Hex scenarios ---------------------------------------------------------------------------
// A frame of these scenarios holds hundreds of visible primitives (a Hex maze seen from its rim: more than a thousand).  Keeping their 32-byte
// records in LDS cost 32 KB per workgroup -- three waves per SIMD, and a hard cap of 1024.  Here LDS holds only what the CULLING needs, 8 bytes of
// screen rectangle and one class byte per primitive (18 KB for 2048 of them): the records stay where the frame setup wrote them, and the few that
// survive a tile's culling are fetched through the SCALAR cache -- their list position is wave-uniform, a record is eight SGPRs that feed the
// slab test's multiplies directly (no VGPRs, no LDS bandwidth, no prologue copy of the list).  Twice the occupancy hides the longer fetch.
// The nearest hit is kept as a 64-bit (depth bits, list position) pair compared as one unsigned: full 24-bit depth precision at any list
// length (the short-list variants pack both into 32 bits), exact ties resolve to the earlier drawable like everywhere else.
typedef __attribute__((address_space(4))) const float cfloat;   // constant address space: loads from uniform addresses become s_load
__device__ __forceinline__ float4 rec4(cfloat *p, int i) { return make_float4(p[4 * i], p[4 * i + 1], p[4 * i + 2], p[4 * i + 3]); }   // (one s_load_dwordx4)

#ifndef MV_GLIST_TILE_W
#define MV_GLIST_TILE_W 16
#endif
struct Key2 { unsigned d, p; };   // depth bits minus KEY_NEAR (> KEY_FAR: not a hit), list position

__device__ __forceinline__ void key2_min(Key2 &best, bool valid, unsigned d, unsigned p)
{
    const unsigned long long a = ((unsigned long long)d << 32) | p, b = ((unsigned long long)best.d << 32) | best.p;
    const bool less = valid && a < b;
    best.d = less ? d : best.d;
    best.p = less ? p : best.p;
}

// What breaks a depth tie is the drawable's SLOT (the reference's draw order), not its position in the list: the two orders are the same in a list kept
// as found, but a long list may be in depth classes (mv_frame.h: DepthSortScratch).  The low 11 bits carry the position (< VIS_XL = 2048), for the shading.
static_assert(VIS_XL <= 2048, "tie_key: 11 bits of list position");
__device__ __forceinline__ unsigned tie_key(const float4 lo, int pos) { return ((__float_as_uint(lo.w) >> 8) << 11) | (unsigned)pos; }

template <int NP>
__device__ __forceinline__ void box_test_g(const V3 (&inv)[NP], const float4 lo, const float4 hi, int pos, Key2 (&best)[NP])
{
#pragma unroll
    for (int j = 0; j < NP; ++j) {
        const float t1x = lo.x * inv[j].x, t2x = hi.x * inv[j].x, t1y = lo.y * inv[j].y, t2y = hi.y * inv[j].y, t1z = lo.z * inv[j].z, t2z = hi.z * inv[j].z;
        const float tn = __builtin_fmaxf(__builtin_fmaxf(__builtin_fminf(t1x, t2x), __builtin_fminf(t1y, t2y)), __builtin_fminf(t1z, t2z));
        const float tf = __builtin_fminf(__builtin_fminf(__builtin_fmaxf(t1x, t2x), __builtin_fmaxf(t1y, t2y)), __builtin_fmaxf(t1z, t2z));
        key2_min(best[j], tn <= tf, __float_as_uint(tn) - KEY_NEAR, tie_key(lo, pos));
    }
}

// the boxes of one frame of reference among list positions 64 k .. 64 k + 63 (mask m): the next record is requested while the current one is tested
template <int NP>
__device__ __forceinline__ void box_run_g(unsigned long long m, int k, const V3 (&inv)[NP], cfloat *cp, Key2 (&best)[NP])
{
    if (!m) return;
    int p0 = __ffsll((long long)m) - 1 + 64 * k, p1 = 0;
    m &= m - 1;
    float4 lo0 = rec4(cp, 2 * p0), hi0 = rec4(cp, 2 * p0 + 1), lo1 = lo0, hi1 = hi0;
    for (;;) {
        bool more = m != 0ull;
        if (more) { p1 = __ffsll((long long)m) - 1 + 64 * k; m &= m - 1; lo1 = rec4(cp, 2 * p1); hi1 = rec4(cp, 2 * p1 + 1); }
        box_test_g<NP>(inv, lo0, hi0, p0, best);
        if (!more) break;
        more = m != 0ull;
        if (more) { p0 = __ffsll((long long)m) - 1 + 64 * k; m &= m - 1; lo0 = rec4(cp, 2 * p0); hi0 = rec4(cp, 2 * p0 + 1); }
        box_test_g<NP>(inv, lo1, hi1, p1, best);
        if (!more) break;
    }
}

// one round of a tile's culling: lane i holds list position 64 k + i -- its rectangle against the tile's; -> the mask of the positions that meet it
__device__ __forceinline__ unsigned long long cull_round_g(const short4 *s_rect, int lane, int k, int nVis, int tx0, int tx1, int ty0, int ty1, int &cpos, bool &v)
{
    cpos = min(lane + 64 * k, nVis - 1);
    const uint2 rr = *reinterpret_cast<const uint2 *>(&s_rect[cpos]);   // x0 | x1 << 16, y0 | y1 << 16
    v = (lane + 64 * k < nVis) & ((int)(rr.x & 0xffffu) <= tx1) & ((int)(rr.x >> 16) >= tx0) & ((int)(rr.y & 0xffffu) <= ty1) & ((int)(rr.y >> 16) >= ty0);
    return __ballot(v);
}

template <int MAXVIS, bool SHAPES, bool HEXF, int NP>
__device__ __forceinline__ void raster_glist_body(const FastArgs &fa, uint32_t *obs, int W, int H, int split, int blk, unsigned char *lds)
{
    // the long-list pass's tile: GT_W x GT_H pixels of one wave (x NP rows of them); 32 x 2: every row a wave stores is one whole 128-byte line
    constexpr int GT_W = MV_GLIST_TILE_W, GT_H = 64 / GT_W, TH = GT_H * NP;
    extern __shared__ __attribute__((aligned(16))) unsigned char s_dyn[];
    short4 *s_rect = reinterpret_cast<short4 *>(lds);                 // [MAXVIS]
    float *s_hdr = reinterpret_cast<float *>(lds + 8 * MAXVIS);       // [FH_FLOATS]
    unsigned char *s_cls = lds + 8 * MAXVIS + 4 * FH_FLOATS;          // [MAXVIS]

    float4 *s_col = reinterpret_cast<float4 *>(s_dyn);
    float4 *s_row = s_col + W;
    float2 *s_rowq = reinterpret_cast<float2 *>(s_row + H);
    float *s_colq = reinterpret_cast<float *>(s_rowq + H);

    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    fast_publish(fa, blk);
    const FastFrame ff = fast_prologue<MAXVIS, true>(fa, blk, W, H, split, nullptr, s_rect, s_hdr, s_col, s_row, s_rowq, s_colq, s_cls);
    const int frame = ff.frame, part = ff.part, viewer = ff.viewer, nVis = ff.nVis;
    split = ff.split;   // (fine-grained tail: this workgroup's frame may be cut into more pieces than the launch's nominal number)
    const float4 *gp = reinterpret_cast<const float4 *>(fa.vis_prims + (size_t)frame * fa.vis_stride);   // this frame's records
    cfloat *cp = (cfloat *)gp;

    const float *camv = s_hdr + FH_CAM + FH_CAM_STRIDE * viewer;
    const float nzm0 = uniform_f32(-camv[3 + 2]), nzm1 = uniform_f32(-camv[3 + 5]), nzm2 = uniform_f32(-camv[3 + 8]);
    uint32_t *out = obs + (size_t)frame * W * H;
    const int tilesX = (W + GT_W - 1) / GT_W, tilesY = (H + TH - 1) / TH;
    const int numTiles = tilesX * tilesY;
    const int lx = lane & (GT_W - 1), ly = lane / GT_W;

    // the workgroup's tiles are handed out one at a time, as in raster_fast_body (a wave that always drew the same tile column lived as long as the
    // most crowded one); u = 4 j + w is tile (j split + part) 4 + w of the frame
    __shared__ int s_next;
    if (tid == 0) s_next = 4;
    __syncthreads();
    // (the frame setup says so in the header)
    const bool depthSorted = (unsigned)__builtin_amdgcn_readfirstlane(__float_as_uint(s_hdr[FH_WB + 31])) == DEPTH_SORTED_MARK;
    const unsigned tilesXinv = (unsigned)((0x100000000ull + (unsigned)tilesX - 1u) / (unsigned)tilesX);
    int unext = 0;
    for (int u = wave; ; u = __builtin_amdgcn_readfirstlane(unext)) {
        const int tile = ((u >> 2) * split + part) * 4 + (u & 3);
        if (tile >= numTiles) break;
        if (lane == 0) unext = atomicAdd(&s_next, 1);
        const int ty = (int)__umulhi((unsigned)tile, tilesXinv), tx = tile - ty * tilesX;
        const int tx0 = tx * GT_W, ty0 = ty * TH;
        const int tx1 = min(tx0 + GT_W, W) - 1, ty1 = min(ty0 + TH, H) - 1;
        const int px = tx0 + lx, py0 = ty0 + ly;
        const int pxc = min(px, W - 1);
#ifdef MV_RASTER_TIMING
        int metTotal = 0, metWorld = 0;
#endif
        RT_COUNT(10, 1);                  // tiles of the long-list pass
        RT_COUNT(12, (nVis + 63) / 64);   // rounds their lists have
        // ---- the rounds of the list up to the first one with a primitive on this tile: there is nothing to set up before (a third of a Hex frame's tiles,
        // half of a Collect frame's, meet nothing at all: they are cleared here)
        int k = 0, cpos = 0;
        bool v = false;
        unsigned long long mvis = 0ull;
#pragma unroll 1
        for (; k * 64 < nVis; ++k) {
            RT_COUNT(11, 1);   // rounds of the list walked
            mvis = cull_round_g(s_rect, lane, k, nVis, tx0, tx1, ty0, ty1, cpos, v);
#if defined(MV_GLIST_DEBUG_SKIP) && MV_GLIST_DEBUG_SKIP == 3   // (measurement builds: the list is culled, nothing else)
            mvis = 0ull;
#endif
            if (mvis) break;
        }
        if (mvis == 0ull) {
            RT_COUNT(14, 1);   // tiles that met nothing
#pragma unroll
            for (int j = 0; j < NP; ++j) {
                const int py = py0 + GT_H * j;
                if (px < W && py < H) PIXEL_STORE(out[(unsigned)(py * W + px)], 0xff000000u);
            }
            continue;
        }
        // ---- the pixels' rays
        V3 dw[NP], inv[NP], ih0[NP], ih1[NP], ih2[NP], bn[NP];
        float dcx, dcy[NP], a2[NP], ldc[NP];
        Key2 best[NP];
        {
            const float4 cx = s_col[pxc];
            const float cq = s_colq[pxc];
            dcx = cx.x;
#pragma unroll
            for (int j = 0; j < NP; ++j) {
                const int pyc = min(py0 + GT_H * j, H - 1);
                const float4 ry = s_row[pyc];
                const float2 rq = s_rowq[pyc];
                dcy[j] = ry.x;
                dw[j] = v3((cx.y + ry.y) + nzm0, (cx.z + ry.z) + nzm1, (cx.w + ry.w) + nzm2);
                inv[j] = v3(__builtin_amdgcn_rcpf(dw[j].x), __builtin_amdgcn_rcpf(dw[j].y), __builtin_amdgcn_rcpf(dw[j].z));
                a2[j] = cq + rq.x; ldc[j] = rq.y;
                if (HEXF) {   // (same products and sums as mat_tmul with the frame's matrix: c x + (-s) z, s x + c z)
                    const float cx8 = 0.8660254f * dw[j].x, cz8 = 0.8660254f * dw[j].z, hx5 = 0.5f * dw[j].x, hz5 = 0.5f * dw[j].z;
                    ih0[j] = v3(__builtin_amdgcn_rcpf(cx8 - hz5), inv[j].y, __builtin_amdgcn_rcpf(hx5 + cz8));
                    ih1[j] = v3(__builtin_amdgcn_rcpf(cx8 + hz5), inv[j].y, __builtin_amdgcn_rcpf(cz8 - hx5));
                    ih2[j] = v3(0.0f - inv[j].z, inv[j].y, inv[j].x);   // 90 degrees: (x, z) -> (-z, x)
                } else ih0[j] = ih1[j] = ih2[j] = inv[j];
                bn[j] = v3(0, 0, 0);
                best[j].d = ~0u; best[j].p = 0u;
            }
        }
        // ---- the rounds with primitives on this tile
        bool more = true;
#pragma unroll 1
        while (more) {
            RT_COUNT(13, __popcll(mvis));   // primitives whose rectangle meets the tile
#ifdef MV_RASTER_TIMING
            metTotal += __popcll(mvis); metWorld += __popcll(__ballot(v && s_cls[cpos] == 0u));
#endif
#if defined(MV_GLIST_DEBUG_SKIP) && MV_GLIST_DEBUG_SKIP == 2   // (measurement builds: culled, the rays set up, nothing tested)
#pragma unroll
            for (int j = 0; j < NP; ++j)
                best[j].p += __float_as_uint(inv[j].x + inv[j].y + inv[j].z + ih0[j].x + ih0[j].z + ih1[j].x + ih1[j].z + a2[j] + ldc[j]);
#else
            const unsigned cl = s_cls[cpos];
            const unsigned long long m0 = __ballot(v && cl == 0u);
            unsigned long long boxes = m0;
            box_run_g<NP>(m0, k, inv, cp, best);
            if (HEXF) {
                const unsigned long long m1 = __ballot(v && cl == 1u), m2 = __ballot(v && cl == 2u), m3 = __ballot(v && cl == 3u);
                box_run_g<NP>(m1, k, ih0, cp, best);
                box_run_g<NP>(m2, k, ih1, cp, best);
                box_run_g<NP>(m3, k, ih2, cp, best);
                boxes |= m1 | m2 | m3;
            }
            unsigned long long rest = mvis & ~boxes;   // camera-attached boxes, capsules, cones, scaled shapes (and, not HEXF, wall-frame boxes)
            RT_COUNT(4, __popcll(boxes));   // slab tests
            RT_COUNT(5, __popcll(rest));    // other primitives tested
            while (rest) {
                const int pos = __ffsll((long long)rest) - 1 + 64 * k;
                rest &= rest - 1;
                const float4 lo = rec4(cp, 2 * pos), hi = rec4(cp, 2 * pos + 1);
#pragma unroll
                for (int j = 0; j < NP; ++j) {
                    V3 n = v3(0, 0, 0);
                    float t;
                    const bool hit = other_rec<SHAPES>(lo, hi, s_hdr, camv, viewer, dw[j], dcx, dcy[j], t, n);
                    const unsigned d = __float_as_uint(t) - KEY_NEAR;
                    const unsigned tk = tie_key(lo, pos);
                    const bool less = hit && (d < best[j].d || (d == best[j].d && tk < best[j].p));
                    if (less) { best[j].d = d; best[j].p = tk; bn[j] = n; }
                }
            }
#endif
            // the next round with a primitive on this tile
            more = false;
#pragma unroll 1
            for (++k; k * 64 < nVis; ++k) {
                // A list in depth classes (DepthSortScratch).  The header's word for the round: the depth class it begins with.  A tile whose every pixel
                // holds a hit nearer than the class's floor is done with the list: everything from here on is hidden behind what has been found.
                if (depthSorted && k < 31) {
                    const unsigned wd = (unsigned)__builtin_amdgcn_readfirstlane(__float_as_uint(s_hdr[FH_WB + k]));
                    // (depth_class_floor, with a margin far above the rounding of the corners' depths)
                    const float floorD = __uint_as_float(((wd & 63u) + (120u << 2)) << 21) * (1.0f - 1e-4f);
                    const unsigned bound = floorD > NEAR_Z ? __float_as_uint(floorD) - KEY_NEAR : 0u;          // in the units of the depth keys; 0: never stop here
                    bool covered = true;
#pragma unroll
                    for (int j = 0; j < NP; ++j) covered = covered && (best[j].d < bound || px >= W || py0 + GT_H * j >= H);
                    if (__all(covered)) { RT_COUNT(15, 1); break; }
                }
                RT_COUNT(11, 1);   // rounds of the list walked
                mvis = cull_round_g(s_rect, lane, k, nVis, tx0, tx1, ty0, ty1, cpos, v);
                if (mvis) { more = true; break; }
            }
        }
#ifdef MV_RASTER_TIMING
        RT_COUNT(9, metTotal == 1 && metWorld == 1);       // ... one world-frame box and nothing else
        RT_COUNT(8, metTotal == 2);
        RT_COUNT(7, metTotal >= 3 && metTotal <= 5);
        RT_COUNT(6, metTotal >= 6);
#endif
#pragma unroll
        for (int j = 0; j < NP; ++j) {
            unsigned rgba = 0xff000000u;
#if defined(MV_GLIST_DEBUG_SKIP) && MV_GLIST_DEBUG_SKIP >= 1   // (measurement builds: nothing shaded)
            rgba = best[j].d ^ best[j].p;
            if (false)
#else
            if (best[j].d <= KEY_FAR)
#endif
            {   // a hit between the near and the far plane: the winner's record, one 32-byte read per lane
                const unsigned wpos = best[j].p & 2047u;   // (tie_key: the position in its low bits)
                const float4 lo = gp[2 * wpos], hi = gp[2 * wpos + 1];
                rgba = shade_rec<SHAPES>(lo, hi, __uint_as_float(best[j].d + KEY_NEAR), bn[j], s_hdr, camv, viewer, dw[j], inv[j], dcx, dcy[j], a2[j], ldc[j]);
            }
            RT_COUNT(3, __popcll(__ballot(best[j].d <= KEY_FAR)));   // pixels with a hit
            const int py = py0 + GT_H * j;
            if (px < W && py < H) PIXEL_STORE(out[(unsigned)(py * W + px)], rgba);
        }
    }
}

template <int MAXVIS, bool SHAPES, int WAVES, bool HEXF, int NP>
__global__ __launch_bounds__(256, WAVES) void raster_glist_kernel(FastArgs fa, uint32_t *obs, int W, int H, int split)
{
    __shared__ __attribute__((aligned(16))) unsigned char s_buf[glist_lds_bytes(MAXVIS)];
    raster_glist_body<MAXVIS, SHAPES, HEXF, NP>(fa, obs, W, H, split, (int)blockIdx.x, s_buf);
}

// WAVES: waves per SIMD the variant is compiled for (register budget 512 / WAVES)
template <int MAXVIS, bool SHAPES, int WAVES, bool HEXF = false, int NP = 1, int NT = 256>
__global__ __launch_bounds__(NT, WAVES) void raster_fast_kernel(FastArgs fa, uint32_t *obs, int W, int H, int split)
{
    __shared__ __attribute__((aligned(16))) unsigned char s_buf[fast_lds_bytes(MAXVIS)];
    raster_fast_body<MAXVIS, SHAPES, HEXF, NP, true, NT>(fa, obs, W, H, split, (int)blockIdx.x, s_buf);
}

// The observation pass of several gyms of one job with one launch (mv_group): workgroup b belongs to gym s with first[s] <= b < first[s + 1]
// and is workgroup b - first[s] of that gym's own grid -- its frame lists, cost bins, header slab, observation slab.  One variant serves
// every gym of the launch: the caller groups the gyms by the variant that can draw them (below).
struct UnionRasterArgs {
    int32_t n;
    int32_t first[MAX_UNION + 1];
    uint32_t *obs[MAX_UNION];
    FastArgs fa[MAX_UNION];
};

template <int MAXVIS, bool SHAPES, int WAVES, bool HEXF, int NP>
__global__ __launch_bounds__(256, WAVES) void raster_fast_union_kernel(UnionRasterArgs ua, int W, int H, int split)
{
    int s = 0;
#pragma unroll
    for (int i = 1; i < MAX_UNION; ++i)
        if (i < ua.n && (int)blockIdx.x >= ua.first[i]) s = i;
    __shared__ __attribute__((aligned(16))) unsigned char s_buf[fast_lds_bytes(MAXVIS)];
    raster_fast_body<MAXVIS, SHAPES, HEXF, NP>(ua.fa[s], ua.obs[s], W, H, split, (int)blockIdx.x - ua.first[s], s_buf);
}

// The observation passes of the k ticks of ONE batched call (mv_step_n) with one launch: workgroups are dealt tick by tick, every tick in its own cost order
// (its own bins, lists, header slab, observation slab).  A launch is throughput-bound while the chip is full and has a tail while its expensive frames
// finish -- half of a launch's waves are done after 32 of its 47 us (r04g); here the next tick's expensive frames start in that tail, and there is one
// tail per call instead of one per tick.  The step's staged outputs of ALL k ticks are published by the first workgroups, element by element in tick
// order (true_objective is only ever written by a finishing env: what a later tick does not touch keeps the earlier tick's value, as with one launch
// per tick).  Arguments: tick 0's FastArgs; tick j's differ from them in the hand-over slot (slot_stride bytes further per tick, mv_types.h: tick_view),
// the cost histogram (consecutive, modulo their number) and where the outputs go (passed per tick: rings wrap) -- k sets of FastArgs do not fit the
// 4 KB of kernel arguments beyond k = 8, and k is up to MAX_STEP_TICKS = 16.
struct TicksRasterArgs {
    int32_t k, per_tick;                // ticks, workgroups of one tick's pass
    int32_t hists, parity0;             // the gym's cost histograms, the one tick 0's pass draws from
    int64_t slot_stride;
    const int *hist_base;               // histogram 0
    int *done_base;                     // "workgroups that have looked their frame up" counter of histogram 0 (null: the passes do not clear their histograms)
    uint32_t *obs[MAX_STEP_TICKS];
    float *pub_rewards[MAX_STEP_TICKS];
    uint8_t *pub_done[MAX_STEP_TICKS];
    FastArgs fa;                        // tick 0's
};

__device__ __forceinline__ FastArgs ticks_raster_args(const TicksRasterArgs &a, int t)
{
    FastArgs f = a.fa;
    const int64_t d = a.slot_stride * t;
    f.vis_hdr += d;
    f.vis_prims = reinterpret_cast<const Prim *>(reinterpret_cast<const unsigned char *>(f.vis_prims) + d);
    f.vis_rects = reinterpret_cast<const short4 *>(reinterpret_cast<const unsigned char *>(f.vis_rects) + d);
    f.list = reinterpret_cast<const int *>(reinterpret_cast<const unsigned char *>(f.list) + d);
    f.stage_rewards = reinterpret_cast<const float *>(reinterpret_cast<const unsigned char *>(f.stage_rewards) + d);
    f.stage_true = reinterpret_cast<const float *>(reinterpret_cast<const unsigned char *>(f.stage_true) + d);
    f.stage_done += d;
    const int par = (a.parity0 + t) % a.hists;
    f.hist = a.hist_base + par * (LPT_BUCKETS * LPT_SUBS);
    f.hist_done = a.done_base ? a.done_base + par : nullptr;
    f.pub_rewards = a.pub_rewards[t];
    f.pub_done = a.pub_done[t];
    return f;
}

template <int MAXVIS, bool SHAPES, int WAVES, int NP, int NT = 256>
__global__ __launch_bounds__(NT, WAVES) void raster_fast_batch_kernel(TicksRasterArgs a, int W, int H, int split)
{
    int j = 0, r = (int)blockIdx.x;
    while (r >= a.per_tick) { r -= a.per_tick; ++j; }   // (at most k - 1 scalar iterations)
    if (j == 0)
        for (int t = 0; t < a.k; ++t) fast_publish(ticks_raster_args(a, t), r);   // (per_tick = frames x split >= the frames / 256 workgroups this takes)
    __shared__ __attribute__((aligned(16))) unsigned char s_buf[fast_lds_bytes(MAXVIS)];
    FastArgs fa = ticks_raster_args(a, j);
    fa.pub_n = 0;
    raster_fast_body<MAXVIS, SHAPES, false, NP, true, NT>(fa, a.obs[j], W, H, split, r, s_buf);
}

// (the long-list variant of raster_fast_batch_kernel: the k passes of one batched call of a Collect / Hex gym)
template <int MAXVIS, bool SHAPES, int WAVES, bool HEXF, int NP>
__global__ __launch_bounds__(256, WAVES) void raster_glist_batch_kernel(TicksRasterArgs a, int W, int H, int split)
{
    int j = 0, r = (int)blockIdx.x;
    while (r >= a.per_tick) { r -= a.per_tick; ++j; }
    if (j == 0)
        for (int t = 0; t < a.k; ++t) fast_publish(ticks_raster_args(a, t), r);
    __shared__ __attribute__((aligned(16))) unsigned char s_buf[glist_lds_bytes(MAXVIS)];
    FastArgs fa = ticks_raster_args(a, j);
    fa.pub_n = 0;
    raster_glist_body<MAXVIS, SHAPES, HEXF, NP>(fa, a.obs[j], W, H, split, r, s_buf);
}

template <int MAXVIS, bool SHAPES, int WAVES, bool HEXF, int NP>
__global__ __launch_bounds__(256, WAVES) void raster_glist_union_kernel(UnionRasterArgs ua, int W, int H, int split)
{
    int s = 0;
#pragma unroll
    for (int i = 1; i < MAX_UNION; ++i)
        if (i < ua.n && (int)blockIdx.x >= ua.first[i]) s = i;
    __shared__ __attribute__((aligned(16))) unsigned char s_buf[glist_lds_bytes(MAXVIS)];
    raster_glist_body<MAXVIS, SHAPES, HEXF, NP>(ua.fa[s], ua.obs[s], W, H, split, (int)blockIdx.x - ua.first[s], s_buf);
}

// Both list lengths in ONE launch: the short-list body (records in LDS) for the gyms with up to 256 visible primitives per frame, the long-list
// body (records through the scalar cache) for the others; which one a workgroup runs is decided by its gym (`large` bit per gym).  Each of a
// Mixed group's two launches holds too few frames to fill the chip and lasts as long as its heaviest frame: one launch fills it once and has
// one tail.  The workgroups of the long-list gyms come first (the expensive frames).  split: per list length (a long-list frame is cut into
// more pieces).
struct UnionRasterAllArgs {
    UnionRasterArgs u;
    int32_t large[MAX_UNION];
    int32_t split_small, split_large;
};

// The short-list gyms' frames of a union launch take the tile classification where a frame is large enough to repay it -- two pixels per lane, 8192 pixels and
// more -- and not below: r09f, Mixed (eight scenarios) without / with, M obs/s: 128 x 128 11.0 / 12.8; 64 x 64 17.3 / 16.7 (the classified body's code and
// registers cost the small frames' passes 6 % even where no frame is classified).
#ifndef MV_UNION_CLS
#define MV_UNION_CLS (NPS >= 2)
#endif
template <int WAVES, int NPS>
__global__ __launch_bounds__(256, WAVES) void raster_union_all_kernel(UnionRasterAllArgs a, int W, int H)
{
    int s = 0;
#pragma unroll
    for (int i = 1; i < MAX_UNION; ++i)
        if (i < a.u.n && (int)blockIdx.x >= a.u.first[i]) s = i;
    constexpr int LDS = fast_lds_bytes(VIS_SMALL) > glist_lds_bytes(VIS_XL) ? fast_lds_bytes(VIS_SMALL) : glist_lds_bytes(VIS_XL);
    __shared__ __attribute__((aligned(16))) unsigned char s_buf[LDS];
    const int blk = (int)blockIdx.x - a.u.first[s];
    if (a.large[s]) raster_glist_body<VIS_XL, true, true, 1>(a.u.fa[s], a.u.obs[s], W, H, a.split_large, blk, s_buf);
    // (no tile classification: this kernel's LDS is the long-list body's, and occupancy is what it lives on)
    else raster_fast_body<VIS_SMALL, true, false, NPS, MV_UNION_CLS>(a.u.fa[s], a.u.obs[s], W, H, a.split_small, blk, s_buf);
}

// The observation passes of the k ticks of ONE batched group call (mv_group_step: n gyms -- scenarios -- x k ticks) with one launch: what
// raster_fast_batch_kernel does for one gym and raster_union_all_kernel for one tick, together.  Workgroups are dealt tick by tick (tick j's passes start
// in the tail of tick j - 1's), inside a tick gym by gym, the long-list gyms first.  k x n FastArgs do not fit the 4 KB of kernel arguments: tick 0's are
// passed, tick j's differ from them in the hand-over slot (slot_stride bytes further per tick: mv_union.h, tick_view), the cost histogram (consecutive,
// modulo their number) and where the outputs go (passed per tick: rings wrap).
struct UnionBatchArgs {
    int32_t n, k, per_tick;             // gyms, ticks, workgroups of one tick's passes
    int32_t first[MAX_UNION + 1];       // first workgroup of gym s within a tick; first[n] = per_tick
    int32_t large[MAX_UNION];
    int32_t split_small, split_large;
    int32_t hists[MAX_UNION], parity0[MAX_UNION];   // gym s: its cost histograms, the one tick 0's pass draws from
    int64_t slot_stride[MAX_UNION];
    const int *hist_base[MAX_UNION];    // histogram 0 of gym s
    int *done_base[MAX_UNION];          // "workgroups that have looked their frame up" counter of histogram 0 (null: the pass does not clear its histogram)
    uint32_t *obs[MAX_GROUP_TICKS][MAX_UNION];
    float *pub_rewards[MAX_GROUP_TICKS][MAX_UNION];
    uint8_t *pub_done[MAX_GROUP_TICKS][MAX_UNION];
    FastArgs fa[MAX_UNION];             // tick 0's
};
static_assert(sizeof(UnionBatchArgs) + 16 <= 4096, "UnionBatchArgs + (W, H) must fit the 4 KB kernel-argument segment");

__device__ __forceinline__ FastArgs union_batch_args(const UnionBatchArgs &a, int s, int t)
{
    FastArgs f = a.fa[s];
    const int64_t d = a.slot_stride[s] * t;
    f.vis_hdr += d;
    f.vis_prims = reinterpret_cast<const Prim *>(reinterpret_cast<const unsigned char *>(f.vis_prims) + d);
    f.vis_rects = reinterpret_cast<const short4 *>(reinterpret_cast<const unsigned char *>(f.vis_rects) + d);
    f.list = reinterpret_cast<const int *>(reinterpret_cast<const unsigned char *>(f.list) + d);
    f.stage_rewards = reinterpret_cast<const float *>(reinterpret_cast<const unsigned char *>(f.stage_rewards) + d);
    f.stage_true = reinterpret_cast<const float *>(reinterpret_cast<const unsigned char *>(f.stage_true) + d);
    f.stage_done += d;
    const int par = (a.parity0[s] + t) % a.hists[s];
    f.hist = a.hist_base[s] + par * (LPT_BUCKETS * LPT_SUBS);
    f.hist_done = a.done_base[s] ? a.done_base[s] + par : nullptr;
    f.pub_rewards = a.pub_rewards[t][s];
    f.pub_done = a.pub_done[t][s];
    return f;
}

template <int WAVES, int NPS>
__global__ __launch_bounds__(256, WAVES) void raster_union_batch_kernel(UnionBatchArgs a, int W, int H)
{
    int j = 0, r = (int)blockIdx.x;
    while (r >= a.per_tick) { r -= a.per_tick; ++j; }   // (at most k - 1 scalar iterations)
    int s = 0;
#pragma unroll
    for (int i = 1; i < MAX_UNION; ++i)
        if (i < a.n && r >= a.first[i]) s = i;
    const int blk = r - a.first[s];
    // the staged outputs of ALL k ticks of gym s are published by the first workgroups of its tick-0 pass, element by element in tick order (what a later
    // tick does not touch -- true_objective of an env that did not finish -- keeps the earlier tick's value, as with one launch per tick)
    if (j == 0)
        for (int t = 0; t < a.k; ++t) fast_publish(union_batch_args(a, s, t), blk);
    constexpr int LDS = fast_lds_bytes(VIS_SMALL) > glist_lds_bytes(VIS_XL) ? fast_lds_bytes(VIS_SMALL) : glist_lds_bytes(VIS_XL);
    __shared__ __attribute__((aligned(16))) unsigned char s_buf[LDS];
    FastArgs fa = union_batch_args(a, s, j);
    fa.pub_n = 0;
    if (a.large[s]) raster_glist_body<VIS_XL, true, true, 1>(fa, a.obs[j][s], W, H, a.split_large, blk, s_buf);
    else raster_fast_body<VIS_SMALL, true, false, NPS, MV_UNION_CLS>(fa, a.obs[j][s], W, H, a.split_small, blk, s_buf);
}

#ifdef MV_RASTER_TIMING
static unsigned long long *g_rdbg = nullptr;
static void rdbg_dump()
{   // the LAST launch's marks: phase cycles per wave, workgroup life times, when the workgroups ended relative to the launch's first start
    if (!g_rdbg) return;
    (void)hipDeviceSynchronize();
    {   // the census of the tile loop over all launches (RT_COUNT)
        unsigned long long c[16];
        const bool got = hipMemcpy(c, g_rdbg + (size_t)16384 * 4 * 8, sizeof(c), hipMemcpyDeviceToHost) == hipSuccess;
        if (got && c[10])
            fprintf(stderr, "long-list census (all launches): tiles %llu, rounds of 64 list positions walked %.2f of %.2f per tile, early stops (every pixel "
                    "nearer than the next class) in %.3f of the tiles, "
                            "%.1f primitives met per tile; tiles that met nothing %.3f, one world-frame box only %.3f, two primitives %.3f, 3-5: %.3f, 6 and "
                                    "more: %.3f\n",
                    c[10], double(c[11]) / c[10], double(c[12]) / c[10], double(c[15]) / c[10], double(c[13]) / c[10],
                                  double(c[14]) / c[10], double(c[9]) / c[10], double(c[8]) / c[10],
                    double(c[7]) / c[10], double(c[6]) / c[10]);
        if (got && c[10])
            fprintf(stderr, "long-list census, tests: %.2f slab tests and %.2f other primitives per tile (%.2f / %.2f per tile that meets anything), %.1f of a tile's "
                    "pixels hit something\n", double(c[4]) / c[10], double(c[5]) / c[10], double(c[4]) / double(c[10] - c[14]), double(c[5]) / double(c[10] - c[14]),
                    double(c[3]) / c[10]);
        if (got && c[4] && !c[10])
            fprintf(stderr, "raster census (all launches): classified tiles %llu (empty %llu, planar %llu of which %llu without the highlight test, covered "
                    "+ overlay %llu), unclassified %llu, general-path tiles %llu: "
                            "slab tests %.2f and other primitives %.2f per tile\n",
                    c[0], c[1], c[2], c[12], c[13], c[3], c[4], double(c[5]) / c[4], double(c[6]) / c[4]);
    }
    std::vector<unsigned long long> h((size_t)16384 * 4 * 8);
    if (hipMemcpy(h.data(), g_rdbg, h.size() * 8, hipMemcpyDeviceToHost) != hipSuccess) return;
    double sumP = 0, sumC = 0, sumT = 0; size_t n = 0;
    unsigned long long r0 = ~0ull, r1 = 0;
    std::vector<double> life, endAt;
    for (size_t w = 0; w < (size_t)16384 * 4; ++w) {
        const unsigned long long *o = &h[w * 8];
        if (!o[0]) continue;
        sumP += double(o[1] - o[0]); sumC += double(o[2] - o[1]); sumT += double(o[3] - o[2]); ++n;
        r0 = std::min(r0, o[4]); r1 = std::max(r1, o[5]);
    }
    if (!n) return;
    for (size_t w = 0; w < (size_t)16384 * 4; ++w) {
        const unsigned long long *o = &h[w * 8];
        if (!o[0]) continue;
        life.push_back(double(o[5] - o[4]) * 0.01); endAt.push_back(double(o[5] - r0) * 0.01);   // s_memrealtime: 100 MHz
    }
    std::vector<double> spread;   // per workgroup: last wave's end - first wave's end
    for (size_t g = 0; g < 16384; ++g) {
        unsigned long long lo = ~0ull, hi = 0; int k = 0;
        for (int w = 0; w < 4; ++w) { const unsigned long long *o = &h[(g * 4 + w) * 8]; if (o[0]) { lo = std::min(lo, o[5]); hi = std::max(hi, o[5]); ++k; } }
        if (k == 4) spread.push_back(double(hi - lo) * 0.01);
    }
    std::sort(spread.begin(), spread.end());
    if (!spread.empty()) fprintf(stderr, "raster timing: within a workgroup, last wave's end - first wave's end (us): mean %.1f p50 %.1f p90 %.1f max %.1f\n",
                                 std::accumulate(spread.begin(), spread.end(), 0.0) / spread.size(),
                                                 spread[spread.size() / 2], spread[spread.size() * 9 / 10], spread.back());
    {   // the longest-lived waves: where in the cost order was their workgroup (blk = its index in the launch), how long is their frame's list?
        std::vector<std::pair<double, size_t>> byLife;
        for (size_t w = 0; w < (size_t)16384 * 4; ++w) { const unsigned long long *o = &h[w * 8]; if (o[0]) byLife.push_back({double(o[5] - o[4]) * 0.01, w}); }
        std::sort(byLife.begin(), byLife.end(), [](auto &a, auto &b) { return a.first > b.first; });
        for (size_t q = 0; q < std::min<size_t>(byLife.size(), 12); q += 1) {
            const size_t w = byLife[q].second; const unsigned long long *o = &h[w * 8];
            fprintf(stderr, "raster timing: long wave #%zu: life %.1f us (start +%.1f us), workgroup %zu of the launch, frame %llu, nVis %llu, classified "
                    "%llu, cycles prologue %llu cls %llu tiles %llu\n", q,
                    byLife[q].first, double(o[4] - r0) * 0.01, w / 4, o[7], o[6] & 0xffffffffull, o[6] >> 32, o[1] - o[0], o[2] - o[1], o[3] - o[2]);
        }
        // mean life by decile of the launch order
        const size_t nw = byLife.size();
        std::vector<double> sum(10, 0.0); std::vector<int> cnt(10, 0);
        size_t maxw = 0; for (auto &p : byLife) maxw = std::max(maxw, p.second);
        for (auto &p : byLife) { const int d = (int)std::min<size_t>(9, p.second * 10 / (maxw + 1)); sum[d] += p.first; ++cnt[d]; }
        fprintf(stderr, "raster timing: mean wave life by decile of the launch order (us):");
        for (int d = 0; d < 10; ++d) fprintf(stderr, " %.1f", cnt[d] ? sum[d] / cnt[d] : 0.0);
        fprintf(stderr, "  (%zu waves)\n", nw);
        {   // the phases by decile (k cycles), the start offsets, the list lengths
            std::vector<double> pp(10, 0.0), pc(10, 0.0), pt(10, 0.0), ps(10, 0.0), pn(10, 0.0);
            for (auto &p : byLife) {
                const int d = (int)std::min<size_t>(9, p.second * 10 / (maxw + 1)); const unsigned long long *o = &h[p.second * 8];
                pp[d] += double(o[1] - o[0]); pc[d] += double(o[2] - o[1]); pt[d] += double(o[3] - o[2]);
                                ps[d] += double(o[4] - r0) * 0.01; pn[d] += double(o[6] & 0xffffffffull);
            }
            fprintf(stderr, "raster timing: by decile: prologue kcyc");
            for (int d = 0; d < 10; ++d) fprintf(stderr, " %.1f", cnt[d] ? pp[d] / cnt[d] / 1e3 : 0.0);
            fprintf(stderr, " | classification kcyc");
            for (int d = 0; d < 10; ++d) fprintf(stderr, " %.1f", cnt[d] ? pc[d] / cnt[d] / 1e3 : 0.0);
            fprintf(stderr, " | tiles kcyc");
            for (int d = 0; d < 10; ++d) fprintf(stderr, " %.1f", cnt[d] ? pt[d] / cnt[d] / 1e3 : 0.0);
            fprintf(stderr, " | start us");
            for (int d = 0; d < 10; ++d) fprintf(stderr, " %.1f", cnt[d] ? ps[d] / cnt[d] : 0.0);
            fprintf(stderr, " | nVis");
            for (int d = 0; d < 10; ++d) fprintf(stderr, " %.0f", cnt[d] ? pn[d] / cnt[d] : 0.0);
            fprintf(stderr, "\n");
        }
    }
    std::sort(life.begin(), life.end()); std::sort(endAt.begin(), endAt.end());
    auto pct = [](const std::vector<double> &v, double p) { return v[std::min(v.size() - 1, (size_t)(p * v.size()))]; };
    fprintf(stderr, "raster timing (last launch, %zu waves): cycles per wave: prologue %.0f, classification %.0f, tile loop %.0f | wave life us: mean %.1f "
            "p10 %.1f p50 %.1f p90 %.1f max %.1f | "
                    "launch %.1f us; waves ended by (us): 10%% %.1f, 25%% %.1f, 50%% %.1f, 75%% %.1f, 90%% %.1f, 99%% %.1f\n",
            n, sumP / n, sumC / n, sumT / n, std::accumulate(life.begin(), life.end(), 0.0) / life.size(),
                                                             pct(life, 0.1), pct(life, 0.5), pct(life, 0.9), life.back(),
            double(r1 - r0) * 0.01, pct(endAt, 0.1), pct(endAt, 0.25), pct(endAt, 0.5), pct(endAt, 0.75), pct(endAt, 0.9), pct(endAt, 0.99));
}
#endif

// a pass whose frame setup did not clear the next pass's cost histogram (a multi-tick step launch: GymView::lpt_no_clear) clears its own when its
// last workgroup has looked its frame up (FastArgs::hist_done); `workgroups`: the launch's grid
static void self_clear(struct FastArgs &fa, const GymView &gv, int workgroups);

static FastArgs fast_args_of(const GymView &gv, const PublishTo *publish)
{
    const int frames = gv.num_envs * gv.num_agents;
    FastArgs fa;
    fa.vis_hdr = gv.vis_hdr; fa.vis_prims = reinterpret_cast<const Prim *>(gv.vis_prims); fa.vis_rects = reinterpret_cast<const short4 *>(gv.vis_rects);
    fa.hist = gv.lpt_hist + gv.lpt_parity * (LPT_BUCKETS * LPT_SUBS); fa.list = gv.lpt_list; fa.num_agents = gv.num_agents;
                                             fa.vis_stride = gv.vis_stride; fa.frames = frames;
    fa.stage_rewards = gv.rewards; fa.stage_true = gv.true_objective; fa.stage_done = gv.done;
    fa.pub_rewards = publish ? publish->rewards : nullptr; fa.pub_true = publish ? publish->true_objective
            : nullptr; fa.pub_done = publish ? publish->done : nullptr;
    fa.pub_n = publish ? frames : 0;
    const char *pe = getenv("MV_PLANAR");   // (read at every launch: the two paths are compared within one process by tests/test_fast_pixels_gpu.py)
    fa.planar = pe && *pe ? atoi(pe) : 1;   // (2: classified, but without overlay_tile; 3: without the sign-specialised slab tests -- comparisons)
    fa.tail_div = 0; fa.tail_split = 0;
    fa.hist_done = nullptr; fa.wg_total = 0;
#ifdef MV_RASTER_TIMING
    if (!g_rdbg && hipMalloc((void **)&g_rdbg, (size_t)(16384 * 4 * 8 + 64) * 8) == hipSuccess) { (void)hipMemset(g_rdbg,
        0, (size_t)(16384 * 4 * 8 + 64) * 8); atexit(rdbg_dump); }
    fa.rdbg = g_rdbg;
#endif
    return fa;
}

static void self_clear(FastArgs &fa, const GymView &gv, int workgroups)
{
    if (!gv.lpt_no_clear) return;
    fa.hist_done = gv.lpt_hist + (size_t)gv.lpt_hists * (LPT_BUCKETS * LPT_SUBS) + gv.lpt_parity;   // (the counters lie behind the histograms)
    fa.wg_total = workgroups;
}

// Pixels per lane of the fast kernels (tile 16 x 4 NP): two from 8192 pixels per frame up; below that (64 x 64) the larger tiles cull worse than
// the shared work saves (Mixed 64 x 64: 9.7 M obs/s with one, 7.7 M with two; Collect 13.6 / 13.1, HexMemory 8.2 / 8.4).  MV_FAST_PPL = 1 | 2
// overrides (read at every launch: the variants are compared within one process by tests/test_fast_pixels_gpu.py).
// The long-list variants (Collect, Hex*) always take one: their two-pixel builds need 78-95 VGPRs, and with the records coming through the
// scalar cache occupancy is worth more (Collect 128 x 128: 97.6 us with one, 146 with two; HexMemory 155 / 171).
// (In the one-launch passes of a batched call -- `batch` -- the long lists take two as well: eight passes' worth of workgroups keep the chip full at the lower
// occupancy, and half as many tiles pay the per-tile work.  r07g/h, one / two pixels per lane: HexMemory 8.0 / 9.2 M obs/s, HexExplore 8.5 / 10.1, Collect 12.2
// / 14.1, Collect 128 x 72 15.3 / 16.8; one pass per launch: HexMemory 6.9 / 6.95, Collect 11.2 / 9.3.)
static int fast_pixels_per_lane(int W, int H, bool longList = false, bool batch = false)
{
    const char *e = getenv("MV_FAST_PPL");
    if (e && *e) return atoi(e) >= 2 ? 2 : 1;
    return (!longList || batch) && MV_FAST_PPL_DEFAULT >= 2 && W * H >= 8192 ? 2 : 1;
}

// Workgroups per frame of the fast kernels.  One pixel per lane: 4 (r02 sweeps: 4 and 8 best at 1024 frames).  Two pixels per lane (half as many
// tiles, a prologue per workgroup): measured on 1024 frames of 128 x 128, 2: 58.0 us, 4: 61.3, 8: 69.7.  With FEW frames in a launch (a Mixed
// group's two launches: 384 / 640 frames) a launch lasts as long as its slowest frame -- kernel traces: 128 HexMemory frames of 64 x 64 take
// 43 us, 1024 of them 84 -- so the frame is cut into more pieces: enough workgroups to fill the chip about twice (4096 / 2048), at most 16 / 8,
// at least one (long lists) / two tiles per wave.
// ONE workgroup per frame where the launch holds workgroups for several rounds of the chip anyway -- the k passes of a batched call in one
// launch (`batch`, frames = k x the pass's), or 4096 frames and more in a single pass: the prologue and the tile classification are paid once
// per frame instead of twice, and the frames that start later fill in behind the early ones (r06d/r06j, TowerBuilding 128 x 128, two -> one:
// 1024 envs x 8 ticks 21.9 -> 23.3 M obs/s, 512 x 8: 17.9 -> 20.4, 512 x 4 agents x 8: 21.5 -> 23.3; single passes of 4096 frames 20.7 -> 22.3,
// of 2048 frames 19.1 -> 18.4, of 1024: 50 -> 72 us -- 1024 workgroups are half a round).
static int fast_split(int W, int H, int np, int frames, bool longList = false, bool batch = false)
{
    const int ftiles = ((W + TILE_W - 1) / TILE_W) * ((H + TILE_H * np - 1) / (TILE_H * np));
    // (one pixel per lane -- frames below 8192 pixels -- in the one-launch passes of a batched call: whole frames per workgroup as well; r09f, Mixed 64 x 64,
    // four / one: 17.3 / 19.4 M obs/s)
    const int want = ((np >= 2 ? 2048 : 4096) + frames - 1) / std::max(frames, 1), lo = (batch || frames >= 4096) && np >= 2
                      && !longList ? 1 : batch && !longList ? 1 : np >= 2 ? 2 : 4, hi = np >= 2 ? 8 : 16;
    int split = lo;
    while (split < hi && split < want) split <<= 1;
    while (split > 1 && ftiles < 4 * split * (longList ? 1 : 2)) split >>= 1;
    return split;
}

// Fast observation pass of n gyms (already set up by their step kernels) with at most two launches: the gyms whose frames hold up to 256
// visible primitives (TowerBuilding, the Obstacles family, Empty, Sokoban, Rearrange: the small variant with scaled shapes) and the ones with
// up to 1024 (Collect, HexMemory, HexExplore: the large variant with the wall-frame box runs; a Collect frame has no wall-frame boxes and
// takes the same path as in its own variant: its world boxes through the box runs, its cones through the general loop).  Pixels are the ones
// each gym's own launch produces, byte for byte (same per-pixel arithmetic; tests/test_multitask_gpu.py).
// a launch whose completion signal is `done`, carried by its dispatch packet (null: a plain launch)
template <class K, class... A>
static void launch_done(K kernel, dim3 grid, dim3 block, size_t dyn, hipStream_t stream, hipEvent_t done, A... args)
{
    if (done) hipExtLaunchKernelGGL(kernel, grid, block, dyn, stream, nullptr, done, 0, args...);
    else hipLaunchKernelGGL(kernel, grid, block, dyn, stream, args...);
}

int launch_raster_union(const GymView *views, uint32_t *const *obs, const PublishTo *publish, int n, int W,
                        int H, hipStream_t stream, hipEvent_t between, hipEvent_t done)
{
    if (W > MAX_W || H > MAX_H || n > MAX_UNION) return -1;
    const size_t dyn = (size_t)(W + H) * sizeof(float4) + (size_t)W * sizeof(float) + (size_t)H * sizeof(float2);
    const int np = fast_pixels_per_lane(W, H);
    int unionFrames[2] = {0, 0};
    for (int i = 0; i < n; ++i) unionFrames[views[i].vis_stride > VIS_SMALL ? 1 : 0] += views[i].num_envs * views[i].num_agents;
    if (between) (void)hipEventRecord(between, stream);
    if (unionFrames[0] > 0 && unionFrames[1] > 0) {   // both list lengths present: one launch for all gyms (raster_union_all_kernel)
        UnionRasterAllArgs a;
        a.u.n = 0;
        a.split_small = fast_split(W, H, np, unionFrames[0] + unionFrames[1], false);
        a.split_large = fast_split(W, H, 1, unionFrames[0] + unionFrames[1], true);
        int wgs = 0;
        for (int large = 1; large >= 0; --large)
            for (int i = 0; i < n; ++i) {
                if ((views[i].vis_stride > VIS_SMALL) != (large != 0)) continue;
                a.u.first[a.u.n] = wgs;
                a.u.obs[a.u.n] = obs[i];
                a.u.fa[a.u.n] = fast_args_of(views[i], publish ? &publish[i] : nullptr);
                a.large[a.u.n] = large;
                wgs += views[i].num_envs * views[i].num_agents * (large ? a.split_large : a.split_small);
                ++a.u.n;
            }
        for (int i = a.u.n; i <= MAX_UNION; ++i) a.u.first[i] = wgs;
        for (int i = a.u.n; i < MAX_UNION; ++i) a.large[i] = 0;
        if (np == 2) launch_done(raster_union_all_kernel<6, 2>, dim3(wgs), dim3(256), dyn, stream, done, a, W, H);
        else launch_done(raster_union_all_kernel<7, 1>, dim3(wgs), dim3(256), dyn, stream, done, a, W, H);
        return 0;
    }
    for (int large = 1; large >= 0; --large) {   // the expensive frames first
        UnionRasterArgs ua;
        ua.n = 0;
        int wgs = 0;
        const int lnp = large ? fast_pixels_per_lane(W, H, true) : np;
        const int split = fast_split(W, H, lnp, unionFrames[large], large != 0);
        for (int i = 0; i < n; ++i) {
            const bool isLarge = views[i].vis_stride > VIS_SMALL;
            if (isLarge != (large != 0)) continue;
            ua.first[ua.n] = wgs;
            ua.obs[ua.n] = obs[i];
            ua.fa[ua.n] = fast_args_of(views[i], publish ? &publish[i] : nullptr);
            wgs += views[i].num_envs * views[i].num_agents * split;
            ++ua.n;
        }
        if (!ua.n) continue;
        for (int i = ua.n; i <= MAX_UNION; ++i) ua.first[i] = wgs;
        if (large) {
            if (lnp == 2) hipLaunchKernelGGL((raster_glist_union_kernel<VIS_XL, true, GLIST_WAVES_NP2, true, 2>),
                dim3(wgs), dim3(256), dyn, stream, ua, W, H, split);
            else hipLaunchKernelGGL((raster_glist_union_kernel<VIS_XL, true, GLIST_WAVES_NP1, true, 1>), dim3(wgs), dim3(256), dyn, stream, ua, W, H, split);
        } else {
            if (lnp == 2) hipLaunchKernelGGL((raster_fast_union_kernel<VIS_SMALL, true, 6, false, 2>), dim3(wgs), dim3(256), dyn, stream, ua, W, H, split);
            else hipLaunchKernelGGL((raster_fast_union_kernel<VIS_SMALL, true, 8, false, 1>), dim3(wgs), dim3(256), dyn, stream, ua, W, H, split);
        }
    }
    if (done) (void)hipEventRecord(done, stream);   // (one or two launches: recorded behind them)
    return 0;
}

// the fast observation passes of k ticks of one gym (views[j]: the slot tick j's step kernel filled; obs[j] / publish[j]: where tick j's
// outputs go, which must differ from tick to tick -- an output ring -- for the observations) with one launch; 1: not this gym (long lists, hires
// sizes, k out of range): the caller launches tick by tick
// the fast observation passes of k ticks of n gyms (a batched group call: views / obs / publish tick-major, [j * n + i]) with ONE launch
// (raster_union_batch_kernel); 1: not applicable -- decide with raster_union_batch_applicable BEFORE the step launch (its frame setups then leave the
// clearing of the cost histograms to the passes)
bool raster_union_batch_applicable(int k, int n, int W, int H)
{
    static const int off = getenv("MV_RASTER_BATCH") && atoi(getenv("MV_RASTER_BATCH")) == 0;
    return !off && W <= MAX_W && H <= MAX_H && k >= 2 && k <= MAX_GROUP_TICKS && n >= 1 && n <= MAX_UNION;
}

int launch_raster_union_batch(const GymView *views, uint32_t *const *obs, const PublishTo *publish,
                              int k, int n, int W, int H, hipStream_t stream, hipEvent_t done)
{
    if (!raster_union_batch_applicable(k, n, W, H)) return 1;
    const size_t dyn = (size_t)(W + H) * sizeof(float4) + (size_t)W * sizeof(float) + (size_t)H * sizeof(float2);
    const int np = fast_pixels_per_lane(W, H);
    int unionFrames[2] = {0, 0};
    for (int i = 0; i < n; ++i) unionFrames[views[i].vis_stride > VIS_SMALL ? 1 : 0] += views[i].num_envs * views[i].num_agents;
    UnionBatchArgs a;
    std::memset(&a, 0, sizeof a);
    a.n = 0; a.k = k;
    a.split_small = fast_split(W, H, np, (unionFrames[0] + unionFrames[1]) * k, false, true);
    a.split_large = fast_split(W, H, 1, (unionFrames[0] + unionFrames[1]) * k, true, true);
    if (const char *e = getenv("MV_UNION_SPLIT_SMALL")) a.split_small = std::max(1, atoi(e));   // (experiments)
    if (const char *e = getenv("MV_UNION_SPLIT_LARGE")) a.split_large = std::max(1, atoi(e));
    int wgs = 0;
    for (int large = 1; large >= 0; --large)   // the expensive frames first
        for (int i = 0; i < n; ++i) {
            const GymView &v0 = views[i];
            if ((v0.vis_stride > VIS_SMALL) != (large != 0)) continue;
            const int q = a.n++;
            const int frames = v0.num_envs * v0.num_agents, split = large ? a.split_large : a.split_small;
            a.first[q] = wgs;
            a.large[q] = large;
            a.fa[q] = fast_args_of(v0, publish ? &publish[i] : nullptr);
            a.fa[q].wg_total = frames * split;
            a.hists[q] = v0.lpt_hists; a.parity0[q] = v0.lpt_parity;
            a.hist_base[q] = v0.lpt_hist;
            a.done_base[q] = v0.lpt_no_clear ? v0.lpt_hist + (size_t)v0.lpt_hists * (LPT_BUCKETS * LPT_SUBS) : nullptr;
            a.slot_stride[q] = k > 1 ? (int64_t)(reinterpret_cast<const unsigned char *>(views[n + i].vis_prims) - reinterpret_cast<const unsigned char *>(v0.vis_prims)) : 0;
            for (int j = 0; j < k; ++j) {   // (tick j's view is tick 0's, one slot further per tick: mv_union.h)
                const GymView &vj = views[(size_t)j * n + i], want = tick_view(v0, a.slot_stride[q], j);
                if (vj.vis_prims != want.vis_prims || vj.vis_hdr != want.vis_hdr || vj.lpt_list != want.lpt_list
                    || vj.rewards != want.rewards || vj.done != want.done ||
                    vj.true_objective != want.true_objective || vj.lpt_parity != want.lpt_parity || vj.lpt_no_clear != v0.lpt_no_clear)
                    return -2;
                a.obs[j][q] = obs[(size_t)j * n + i];
                a.pub_rewards[j][q] = publish ? publish[(size_t)j * n + i].rewards : nullptr;
                a.pub_done[j][q] = publish ? publish[(size_t)j * n + i].done : nullptr;
            }
            wgs += frames * split;
        }
    for (int i = a.n; i <= MAX_UNION; ++i) a.first[i] = wgs;
    a.per_tick = wgs;
    if (np == 2) launch_done(raster_union_batch_kernel<6, 2>, dim3(wgs * k), dim3(256), dyn, stream, done, a, W, H);
    else launch_done(raster_union_batch_kernel<7, 1>, dim3(wgs * k), dim3(256), dyn, stream, done, a, W, H);
    return 0;
}

int launch_raster_batch(const GymView *views, uint32_t *const *obs, const PublishTo *publish, int k, int W, int H, hipStream_t stream, hipEvent_t done)
{
    if (W > MAX_W || H > MAX_H || k < 2 || k > MAX_STEP_TICKS) return 1;
    const GymView &gv = views[0];
    static const int off = getenv("MV_RASTER_BATCH") && atoi(getenv("MV_RASTER_BATCH")) == 0;
    if (off) return 1;
    const size_t dyn = (size_t)(W + H) * sizeof(float4) + (size_t)W * sizeof(float) + (size_t)H * sizeof(float2);
    const int frames = gv.num_envs * gv.num_agents;
    const bool hexScen = gv.scenario == SCN_HEX_MEMORY || gv.scenario == SCN_HEX_EXPLORE, longList = gv.vis_stride > VIS_SMALL || hexScen;
    // (the k passes fill the chip together -- later passes' workgroups start as earlier ones end -- so the frame is cut for k x frames of them:
    // 512 TowerBuilding frames, 8 ticks per call: 16.2 M obs/s with two workgroups per frame, 14.0 M with the four a single pass of 512 frames takes)
    const int np = longList ? fast_pixels_per_lane(W, H, true, true) : fast_pixels_per_lane(W, H);
    const int split = fast_split(W, H, np, frames * k, longList, true);
    int64_t slotStride = 0;
    if (!slot_stride_of(views, k, slotStride)) return 1;   // (views that are not one hand-over slot apart per tick: tick by tick)
    TicksRasterArgs a;
    std::memset(&a, 0, sizeof a);
    a.k = k; a.per_tick = frames * split;
    a.hists = gv.lpt_hists; a.parity0 = gv.lpt_parity;
    a.slot_stride = slotStride;
    a.hist_base = gv.lpt_hist;
    a.done_base = gv.lpt_no_clear ? gv.lpt_hist + (size_t)gv.lpt_hists * (LPT_BUCKETS * LPT_SUBS) : nullptr;
    a.fa = fast_args_of(gv, publish ? &publish[0] : nullptr);
    a.fa.wg_total = frames * split;
    for (int j = 0; j < k; ++j) {
        a.obs[j] = obs[j];
        a.pub_rewards[j] = publish ? publish[j].rewards : nullptr;
        a.pub_done[j] = publish ? publish[j].done : nullptr;
    }
    const dim3 grid(k * frames * split), block(256);
    if (longList) {   // the long-list variants (records through the scalar cache)
        if (np == 2) {
            if (hexScen) launch_done(raster_glist_batch_kernel<VIS_XL, true, GLIST_WAVES_NP2, true, 2>, grid, block, dyn, stream, done, a, W, H, split);
            else launch_done(raster_glist_batch_kernel<VIS_LARGE, false, GLIST_WAVES_NP2, false, 2>, grid, block, dyn, stream, done, a, W, H, split);
        } else {
            if (hexScen) launch_done(raster_glist_batch_kernel<VIS_XL, true, GLIST_WAVES_NP1, true, 1>, grid, block, dyn, stream, done, a, W, H, split);
            else launch_done(raster_glist_batch_kernel<VIS_LARGE, false, GLIST_WAVES_NP1, false, 1>, grid, block, dyn, stream, done, a, W, H, split);
        }
        return 0;
    }
    const bool shapes = gv.scenario == SCN_REARRANGE;
    if (np == 2) {
        if (shapes) launch_done(raster_fast_batch_kernel<VIS_SMALL, true, 6, 2>, grid, block, dyn, stream, done, a, W, H, split);
        else launch_done(raster_fast_batch_kernel<VIS_SMALL, false, 7, 2>, grid, block, dyn, stream, done, a, W, H, split);
    } else {
        if (shapes) launch_done(raster_fast_batch_kernel<VIS_SMALL, true, 8, 1>, grid, block, dyn, stream, done, a, W, H, split);
        else launch_done(raster_fast_batch_kernel<VIS_SMALL, false, 8, 1>, grid, block, dyn, stream, done, a, W, H, split);
    }
    return 0;
}

int launch_raster(const GymView &gv, uint32_t *obs, int W, int H, hipStream_t stream, hipEvent_t between,
                  int fast, int setup_done, const PublishTo *publish, hipEvent_t done)
{
    if (W > MAX_W || H > MAX_H) return -1;
    const int tiles = ((W + TILE_W - 1) / TILE_W) * ((H + TILE_H * PPL - 1) / (TILE_H * PPL));
    const int frames = gv.num_envs * gv.num_agents;
    if (!setup_done) hipLaunchKernelGGL(frame_setup_kernel, dim3(frames), dim3(256), 0, stream, gv, W, H);
    if (!fast) hipLaunchKernelGGL(frame_order_kernel, dim3(1), dim3(1024), 0, stream, gv, frames, gv.lpt_order);   // (fast: no sort kernel)
    if (between) (void)hipEventRecord(between, stream);
    if (fast) {
        const size_t dyn = (size_t)(W + H) * sizeof(float4) + (size_t)W * sizeof(float) + (size_t)H * sizeof(float2);
        const int np = fast_pixels_per_lane(W, H, gv.vis_stride > VIS_SMALL);
        const int split = fast_split(W, H, np, frames, gv.vis_stride > VIS_SMALL);
        // variants: <= 256 visible primitives, + scaled shapes (Rearrange), <= 1024 through the scalar cache (Collect), <= 2048 + scaled shapes + wall frames
        // (Hex*); the short-list ones are built for 8 waves per SIMD with one pixel per lane (64 VGPRs), for 7 with two (72; with the scaled shapes: 6, 80)
        using KernelFn = void (*)(FastArgs, uint32_t *, int, int, int);
        FastArgs fa = fast_args_of(gv, publish);
        // (Collect, measured and rejected: a 1024-entry launch for the frames above 256 visible primitives + a 256-entry launch for the rest,
        // 75 + 69 us against 107 us for the single 1024-entry launch: each launch pays its own tail, and the cones, not occupancy, dominate)
        const bool hexScen = gv.scenario == SCN_HEX_MEMORY || gv.scenario == SCN_HEX_EXPLORE;
        KernelFn fn;
        if (np == 2)
            fn = hexScen ? raster_glist_kernel<VIS_XL, true, GLIST_WAVES_NP2, true, 2> : gv.vis_stride > VIS_SMALL
                    ? raster_glist_kernel<VIS_LARGE, false, GLIST_WAVES_NP2, false, 2>
               : gv.scenario == SCN_REARRANGE ? raster_fast_kernel<VIS_SMALL, true, 6, false, 2> : raster_fast_kernel<VIS_SMALL, false, 7, false, 2>;
        else
            fn = hexScen ? raster_glist_kernel<VIS_XL, true, GLIST_WAVES_NP1, true, 1> : gv.vis_stride > VIS_SMALL
                    ? raster_glist_kernel<VIS_LARGE, false, GLIST_WAVES_NP1, false, 1>
               : gv.scenario == SCN_REARRANGE ? raster_fast_kernel<VIS_SMALL, true, 8> : raster_fast_kernel<VIS_SMALL, false, 8>;
        const int ftiles = ((W + TILE_W - 1) / TILE_W) * ((H + TILE_H * np - 1) / (TILE_H * np));
        // Fine-grained tail.  The SIMD's arbiter serves its OLDEST wave first: workgroups finish roughly in launch order whatever they cost (wave life
        // by decile of the launch order, frames in random order: 18 us for the first tenth, 35 us for the eighth, all started within 0.3 us -- r05b),
        // and a launch ends with its youngest workgroups finishing alone, a few waves per SIMD, far from filling it (the last 4 k of a SIMD's 20 k
        // vector instructions take 22 of a launch's 49 us).  So the frames at the END of the cost order -- the cheapest -- are cut into more, smaller
        // workgroups that keep arriving while the big early ones drain: the cheapest eighth in eight pieces each, 49.6 -> 48.3 us alone (r05d; a quarter
        // in four: 48.5; a sixteenth in eight: 49.4).  (Built, measured and removed, DESIGN.md 0b.3: a graded split -- the expensive eighth in four
        // pieces --, the frame's cost as the last pass's classification found it fed back into the cost bins, wave priorities by remaining work, a whole
        // frame per eight-wave workgroup.)
        constexpr int TAIL_DIV = 8, TAIL_SPLIT = 8;
        // (split 2: a launch that fills the chip; 512 frames in four pieces each: 12.0 M obs/s with the tail cut finer, 12.4 without, r05i)
        if (split <= 2 && gv.vis_stride <= VIS_SMALL && ftiles >= 16 * TAIL_SPLIT && tail_frames(frames, TAIL_DIV) > 0) {
            const int q = tail_frames(frames, TAIL_DIV);
            fa.tail_div = TAIL_DIV; fa.tail_split = TAIL_SPLIT;
            self_clear(fa, gv, (frames - q) * split + q * TAIL_SPLIT);
            launch_done(fn, dim3((frames - q) * split + q * TAIL_SPLIT), dim3(256), dyn, stream, done, fa, obs, W, H, split);
            return 0;
        }
        self_clear(fa, gv, frames * split);
        launch_done(fn, dim3(frames * split), dim3(256), dyn, stream, done, fa, obs, W, H, split);
        return 0;
    }
    const size_t dyn = (size_t)(W + H) * sizeof(float4) + (size_t)(W + H) * sizeof(float);
    int split = 4;
    while (split > 1 && tiles < 4 * split * 2) split >>= 1;   // keep at least two tiles per wave
    const dim3 grid(frames * split), block(256);
    if (gv.scenario == SCN_HEX_MEMORY || gv.scenario == SCN_HEX_EXPLORE) hipLaunchKernelGGL((raster_kernel<VIS_XL, true>),
        grid, block, dyn, stream, gv, obs, W, H, split, gv.lpt_order);
    else if (gv.vis_stride > VIS_SMALL) hipLaunchKernelGGL((raster_kernel<VIS_LARGE, false>), grid, block, dyn, stream, gv, obs, W, H, split, gv.lpt_order);
    else if (gv.scenario == SCN_REARRANGE) hipLaunchKernelGGL((raster_kernel<VIS_SMALL, true>), grid, block, dyn, stream, gv, obs, W, H, split, gv.lpt_order);
    else hipLaunchKernelGGL((raster_kernel<VIS_SMALL, false>), grid, block, dyn, stream, gv, obs, W, H, split, gv.lpt_order);
    if (done) (void)hipEventRecord(done, stream);
    return 0;
}

}  // namespace mv
