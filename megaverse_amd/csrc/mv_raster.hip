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
// Mapping: one 256-thread workgroup per agent frame.  The frame's primitive list (<=128 x 32 B)
// and all agent cameras are staged once in LDS.  Each wavefront walks 16x4-pixel tiles; per tile
// the 64 lanes first cull the primitive list against the tile's frustum (one or two primitives per
// lane, __ballot -> 2 x 64-bit survivor masks), then every lane intersects its pixel's ray with
// just the survivors, iterating the mask bits so the loop is wave-uniform and the primitive comes
// from an LDS broadcast read.  Culling is conservative (tile bounds at pixel edges, rays at pixel
// centres), so the image is identical to brute force over all primitives.
#include <hip/hip_runtime.h>
#include <math.h>

#include "mv_math.h"
#include "mv_types.h"

namespace mv {

namespace {

constexpr float TAN_HALF_FOV = 1.19175359f;                       // tan(100deg / 2), env_renderer.hpp:36
constexpr float TAN_HALF_FOV_Y = 1.19175359f / (128.0f / 72.0f);  // aspect 128/72 is baked into the projection
constexpr float NEAR_Z = 0.01f, FAR_Z = 120.0f;
constexpr float OBJ_HALF = 0.39f, CARRY_SCALE = 0.78f;
constexpr int TILE_W = 16, TILE_H = 4;
constexpr int MAX_PRIMS = 128;

__constant__ unsigned AGENT_COLORS[7] = {0xffdd3c, 0x3bb372, 0x2eb5d0, 0xffb400, 0xd468ee, 0x222222, 0xff0000};

enum : int { PRIM_NONE = 0, PRIM_BOX = 1, PRIM_CAPSULE = 2 };

struct alignas(16) Prim {   // 32 B
    float lo[3]; uint32_t color;
    float hi[3]; int32_t meta;   // kind | frame << 8 ; frame 0 = world axes, 1+k = camera frame of agent k
};

struct CamL {
    float eye[3];
    float c[9];       // row-major 3x3, columns = camera right/up/back in world
    float origin[3];  // viewer eye expressed in this camera's frame
};

__device__ __forceinline__ V3 mat_mul(const float *m, V3 v)
{
    return v3((m[0] * v.x + m[1] * v.y) + m[2] * v.z, (m[3] * v.x + m[4] * v.y) + m[5] * v.z, (m[6] * v.x + m[7] * v.y) + m[8] * v.z);
}
__device__ __forceinline__ V3 mat_tmul(const float *m, V3 v)
{
    return v3((m[0] * v.x + m[3] * v.y) + m[6] * v.z, (m[1] * v.x + m[4] * v.y) + m[7] * v.z, (m[2] * v.x + m[5] * v.y) + m[8] * v.z);
}

// entry point of the ray o + t*d into [lo,hi]; inv = 1/d per axis.  Back faces are culled, the
// hit must lie in [NEAR_Z, FAR_Z] (t is view depth because d.z == -1 in camera space).
__device__ __forceinline__ bool ray_box(V3 o, V3 d, V3 inv, const float *lo, const float *hi, float &t_out, int &axis_out)
{
    float tEnter = -INFINITY, tExit = INFINITY;
    int axis = -1;
    const float oo[3] = {o.x, o.y, o.z}, dd[3] = {d.x, d.y, d.z}, ii[3] = {inv.x, inv.y, inv.z};
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        if (dd[k] == 0.0f) {
            if (oo[k] < lo[k] || oo[k] > hi[k]) return false;
        } else {
            const float t1 = (lo[k] - oo[k]) * ii[k], t2 = (hi[k] - oo[k]) * ii[k];
            const float tn = fmin_sel(t1, t2), tf = fmax_sel(t1, t2);
            if (tn > tEnter) { tEnter = tn; axis = k; }
            tExit = fmin_sel(tExit, tf);
        }
    }
    if (axis < 0 || tEnter > tExit || tEnter < NEAR_Z || tEnter > FAR_Z) return false;
    t_out = tEnter;
    axis_out = axis;
    return true;
}

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
            const float t = (-B - sqrtf(disc)) / A;
            const float y = o.y + t * d.y;
            if (t >= NEAR_Z && t <= FAR_Z && y >= c.y - hl && y <= c.y + hl) {
                hit = true; best = t;
                const float inv = 1.0f / r;
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
            const float t = (-B3 - sqrtf(disc)) / A3;
            const float y = o.y + t * d.y;
            const bool capSide = s == 0 ? (y <= cy) : (y >= cy);
            if (t >= NEAR_Z && t <= FAR_Z && capSide && t < best) {
                hit = true; best = t;
                const float inv = 1.0f / r;
                bn = v3((ox + t * d.x) * inv, (oy + t * d.y) * inv, (oz + t * d.z) * inv);
            }
        }
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
    return (unsigned)(int)floorf(v * 255.0f + 0.5f);
}

__device__ __forceinline__ V3 safe_inv(V3 d)
{
    return v3(d.x == 0.0f ? 0.0f : 1.0f / d.x, d.y == 0.0f ? 0.0f : 1.0f / d.y, d.z == 0.0f ? 0.0f : 1.0f / d.z);
}

// max over the box corners of n . (corner - e)
__device__ __forceinline__ float support(V3 n, V3 e, const float *lo, const float *hi)
{
    const float ax = fmaxf(n.x * (lo[0] - e.x), n.x * (hi[0] - e.x));
    const float ay = fmaxf(n.y * (lo[1] - e.y), n.y * (hi[1] - e.y));
    const float az = fmaxf(n.z * (lo[2] - e.z), n.z * (hi[2] - e.z));
    return ax + ay + az;
}

}  // namespace

__global__ __launch_bounds__(256) void raster_kernel(GymView gv, uint32_t *obs, int W, int H)
{
    __shared__ Prim s_prim[MAX_PRIMS];
    __shared__ CamL s_cam[MAX_AGENTS];

    const int A = gv.num_agents;
    const int frame = blockIdx.x;
    const int env = frame / A, viewer = frame - env * A;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;

    const EnvHeader *hdr = gv.hdr + env;
    const AgentState *agents = gv.agents + (size_t)env * A;

    // ---- stage cameras
    if (tid < A) {
        const AgentState a = agents[tid];
        CamL cam;
        cam.eye[0] = a.pos[0]; cam.eye[1] = (a.pos[1] + 0.05f) + 0.41f; cam.eye[2] = a.pos[2];
        float sp, cp;
        sincos_poly(a.pitch, sp, cp);
        cam.c[0] = a.m00; cam.c[1] = a.m02 * sp; cam.c[2] = a.m02 * cp;
        cam.c[3] = 0.0f;  cam.c[4] = cp;         cam.c[5] = -sp;
        cam.c[6] = a.m20; cam.c[7] = a.m22 * sp; cam.c[8] = a.m22 * cp;
        cam.origin[0] = cam.origin[1] = cam.origin[2] = 0.0f;
        s_cam[tid] = cam;
    }
    __syncthreads();
    if (tid < A) {
        const V3 ev = v3(s_cam[viewer].eye[0], s_cam[viewer].eye[1], s_cam[viewer].eye[2]);
        const V3 ek = v3(s_cam[tid].eye[0], s_cam[tid].eye[1], s_cam[tid].eye[2]);
        const V3 o = mat_tmul(s_cam[tid].c, ev - ek);
        s_cam[tid].origin[0] = o.x; s_cam[tid].origin[1] = o.y; s_cam[tid].origin[2] = o.z;
    }

    // ---- stage the primitive list (slot order == draw order used for depth ties)
    if (tid < MAX_PRIMS) {
        Prim p;
        p.lo[0] = p.lo[1] = p.lo[2] = p.hi[0] = p.hi[1] = p.hi[2] = 0.0f; p.color = 0; p.meta = PRIM_NONE;
        if (tid < MAX_BOXES) {
            if (tid < hdr->num_boxes) {
                const LayoutBox b = gv.boxes[(size_t)env * MAX_BOXES + tid];
                if (b.type & VX_OPAQUE) {
                    p.meta = PRIM_BOX;
                    p.lo[0] = float(b.min[0]); p.lo[1] = float(b.min[1]); p.lo[2] = float(b.min[2]);
                    p.hi[0] = float(b.max[0]); p.hi[1] = float(b.max[1]); p.hi[2] = float(b.max[2]);
                    p.color = (unsigned)(b.slot == 0 ? hdr->layout_color : hdr->wall_color);
                }
            }
        } else if (tid == MAX_BOXES) {   // building-zone slab
            p.meta = PRIM_BOX;
            p.lo[0] = float(hdr->bz[0]); p.lo[1] = 1.0f; p.lo[2] = float(hdr->bz[2]);
            p.hi[0] = float(hdr->bz[1]); p.hi[1] = 1.0f + 0.05f; p.hi[2] = float(hdr->bz[3]);
            p.color = 0x555555u;
        } else if (tid < MAX_BOXES + 1 + MAX_OBJECTS) {
            const int j = tid - (MAX_BOXES + 1);
            if (j < hdr->num_objects) {
                const MovableObject o = gv.objects[(size_t)env * MAX_OBJECTS + j];
                p.color = 0xadd8e6u;
                if (o.state == 0) {
                    p.meta = PRIM_BOX;
                    const float cx = float(o.x) + 0.5f, cy = float(o.y) + 0.5f, cz = float(o.z) + 0.5f;
                    p.lo[0] = cx - OBJ_HALF; p.lo[1] = cy - OBJ_HALF; p.lo[2] = cz - OBJ_HALF;
                    p.hi[0] = cx + OBJ_HALF; p.hi[1] = cy + OBJ_HALF; p.hi[2] = cz + OBJ_HALF;
                } else {
                    p.meta = PRIM_BOX | ((int)o.state << 8);
                    const float hh = OBJ_HALF * CARRY_SCALE;
                    const float cx = 0.0f, cy = -0.44f + -0.3f, cz = -1.0f;
                    p.lo[0] = cx - hh; p.lo[1] = cy - hh; p.lo[2] = cz - hh;
                    p.hi[0] = cx + hh; p.hi[1] = cy + hh; p.hi[2] = cz + hh;
                }
            }
        } else {
            const int q = tid - (MAX_BOXES + 1 + MAX_OBJECTS);
            const int k = q / 3, part = q - 3 * k;
            if (k < A) {
                if (part == 0 && k != viewer) {
                    const AgentState a = agents[k];
                    p.meta = PRIM_CAPSULE;
                    p.lo[0] = a.pos[0]; p.lo[1] = (a.pos[1] + 0.05f) + 0.09f; p.lo[2] = a.pos[2];
                    p.hi[0] = 0.35f; p.hi[1] = 0.36f; p.hi[2] = 0.0f;
                    p.color = AGENT_COLORS[k % 7];
                } else if (part == 1 && k != viewer) {
                    p.meta = PRIM_BOX | ((1 + k) << 8);
                    p.lo[0] = -0.25f; p.lo[1] = -0.12f; p.lo[2] = -0.19f - 0.2f;
                    p.hi[0] = 0.25f; p.hi[1] = 0.12f; p.hi[2] = -0.19f + 0.2f;
                    p.color = 0x2c3e50u;
                } else if (part == 2) {
                    const float bw = hdr->bar_half_width;
                    p.meta = PRIM_BOX | ((1 + k) << 8);
                    p.lo[0] = -bw; p.lo[1] = -0.131f - 0.0015f; p.lo[2] = -0.2f - 0.001f;
                    p.hi[0] = bw; p.hi[1] = -0.131f + 0.0015f; p.hi[2] = -0.2f + 0.001f;
                    p.color = 0x2eb5d0u;
                }
            }
        }
        s_prim[tid] = p;
    }
    __syncthreads();

    const CamL &cam = s_cam[viewer];
    const V3 eye = v3(cam.eye[0], cam.eye[1], cam.eye[2]);
    const int tilesX = (W + TILE_W - 1) / TILE_W, tilesY = (H + TILE_H - 1) / TILE_H;
    const int numTiles = tilesX * tilesY;
    const float LIGHT[3] = {0.0f, 4.0f, 2.0f};
    const float AMB = float(0x55) / 255.0f, DIF = float(0xbb) / 255.0f, LCOL = float(0xaa) / 255.0f;
    uint32_t *out = obs + (size_t)frame * W * H;

    for (int tile = wave; tile < numTiles; tile += 4) {
        const int ty = tile / tilesX, tx = tile - ty * tilesX;

        // ---- conservative tile-vs-primitive culling, 2 primitives per lane
        // tile frustum in camera space: x in [x0,x1]*w, y in [y0,y1]*w for depth w = -z >= NEAR
        const float x0 = ((float(tx * TILE_W) / float(W)) * 2.0f - 1.0f) * TAN_HALF_FOV;
        const float x1 = ((float(min(tx * TILE_W + TILE_W, W)) / float(W)) * 2.0f - 1.0f) * TAN_HALF_FOV;
        const float y0 = ((float(ty * TILE_H) / float(H)) * 2.0f - 1.0f) * TAN_HALF_FOV_Y;
        const float y1 = ((float(min(ty * TILE_H + TILE_H, H)) / float(H)) * 2.0f - 1.0f) * TAN_HALF_FOV_Y;
        const V3 planeC[5] = {v3(1.0f, 0.0f, x0), v3(-1.0f, 0.0f, -x1), v3(0.0f, 1.0f, y0), v3(0.0f, -1.0f, -y1), v3(0.0f, 0.0f, -1.0f)};
        V3 planeW[5];
#pragma unroll
        for (int q = 0; q < 5; ++q) planeW[q] = mat_mul(cam.c, planeC[q]);

        bool keep[2];
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const Prim &p = s_prim[lane + 64 * k];
            const int kind = p.meta & 255, fr = p.meta >> 8;
            bool vis = kind != PRIM_NONE;
            if (vis) {
                float lo[3] = {p.lo[0], p.lo[1], p.lo[2]}, hi[3] = {p.hi[0], p.hi[1], p.hi[2]};
                if (kind == PRIM_CAPSULE) {
                    const float r = p.hi[0], hl = p.hi[1];
                    lo[0] = p.lo[0] - r; lo[1] = p.lo[1] - (hl + r); lo[2] = p.lo[2] - r;
                    hi[0] = p.lo[0] + r; hi[1] = p.lo[1] + (hl + r); hi[2] = p.lo[2] + r;
                }
#pragma unroll
                for (int q = 0; q < 5; ++q) {
                    V3 n, e;
                    if (fr == 0) { n = planeW[q]; e = eye; }
                    else if (fr == 1 + viewer) { n = planeC[q]; e = v3(0.0f, 0.0f, 0.0f); }
                    else { n = mat_tmul(s_cam[fr - 1].c, planeW[q]); e = v3(s_cam[fr - 1].origin[0], s_cam[fr - 1].origin[1], s_cam[fr - 1].origin[2]); }
                    const float s = support(n, e, lo, hi);
                    const float bound = (q == 4) ? NEAR_Z * 0.5f : -1e-4f;
                    if (s < bound) vis = false;
                }
            }
            keep[k] = vis;
        }
        unsigned long long m0 = __ballot(keep[0]), m1 = __ballot(keep[1]);

        // ---- per-pixel rays
        const int px = tx * TILE_W + (lane & (TILE_W - 1)), py = ty * TILE_H + (lane / TILE_W);
        const float xn = ((float(px) + 0.5f) / float(W)) * 2.0f - 1.0f;
        const float yn = ((float(py) + 0.5f) / float(H)) * 2.0f - 1.0f;
        const V3 dc = v3(xn * TAN_HALF_FOV, yn * TAN_HALF_FOV_Y, -1.0f);
        const V3 dw = mat_mul(cam.c, dc);
        const V3 invW = safe_inv(dw), invC = safe_inv(dc);

        float best = INFINITY;
        V3 bestN = v3(0, 0, 0);     // in the winning primitive's own frame
        int bestFrame = 0;
        unsigned bestColor = 0;
        bool any = false;

#pragma unroll 1
        for (int half = 0; half < 2; ++half) {
            unsigned long long m = half ? m1 : m0;
            while (m) {
                const int bit = __ffsll((long long)m) - 1;
                m &= m - 1;
                const Prim &p = s_prim[bit + 64 * half];
                const int kind = p.meta & 255, fr = p.meta >> 8;
                float t; V3 n = v3(0, 0, 0);
                bool hit;
                if (kind == PRIM_CAPSULE) {
                    hit = ray_capsule(eye, dw, v3(p.lo[0], p.lo[1], p.lo[2]), p.hi[0], p.hi[1], t, n);
                } else {
                    int axis = 0;
                    float dax;
                    if (fr == 0) {
                        hit = ray_box(eye, dw, invW, p.lo, p.hi, t, axis);
                        dax = axis == 0 ? dw.x : axis == 1 ? dw.y : dw.z;
                    } else if (fr == 1 + viewer) {
                        hit = ray_box(v3(0.0f, 0.0f, 0.0f), dc, invC, p.lo, p.hi, t, axis);
                        dax = axis == 0 ? dc.x : axis == 1 ? dc.y : dc.z;
                    } else {
                        const CamL &ck = s_cam[fr - 1];
                        const V3 dk = mat_tmul(ck.c, dw);
                        hit = ray_box(v3(ck.origin[0], ck.origin[1], ck.origin[2]), dk, safe_inv(dk), p.lo, p.hi, t, axis);
                        dax = axis == 0 ? dk.x : axis == 1 ? dk.y : dk.z;
                    }
                    const float sgn = dax > 0 ? -1.0f : 1.0f;
                    n = v3(axis == 0 ? sgn : 0.0f, axis == 1 ? sgn : 0.0f, axis == 2 ? sgn : 0.0f);
                }
                if (hit && t < best) { best = t; bestN = n; bestFrame = fr; bestColor = p.color; any = true; }
            }
        }

        // ---- Phong (Magnum Shaders::Phong, uniforms of magnum_env_renderer.cpp:200-203)
        unsigned rgba = 0xff000000u;
        if (any) {
            V3 N;
            if (bestFrame == 0) N = mat_tmul(cam.c, bestN);
            else if (bestFrame == 1 + viewer) N = bestN;
            else N = mat_tmul(cam.c, mat_mul(s_cam[bestFrame - 1].c, bestN));
            const V3 P = dc * best;
            V3 Ld = v3(LIGHT[0] - P.x, LIGHT[1] - P.y, LIGHT[2] - P.z);
            Ld = Ld * (1.0f / sqrtf(len2(Ld)));
            const float intensity = fmax_sel(0.0f, dot(N, Ld));
            float spec = 0.0f;
            if (intensity > 0.001f) {
                const float k2 = 2.0f * dot(N, Ld);
                const V3 R = v3(k2 * N.x - Ld.x, k2 * N.y - Ld.y, k2 * N.z - Ld.z);
                V3 Vd = v3(-P.x, -P.y, -P.z);
                Vd = Vd * (1.0f / sqrtf(len2(Vd)));
                spec = pow300(fmax_sel(0.0f, dot(Vd, R)));
                spec = fmin_sel(fmax_sel(spec, 0.0f), 1.0f);
            }
            const float col[3] = {float((bestColor >> 16) & 255) / 255.0f, float((bestColor >> 8) & 255) / 255.0f,
                                  float(bestColor & 255) / 255.0f};
            unsigned ch[3];
#pragma unroll
            for (int c = 0; c < 3; ++c) ch[c] = to_u8((AMB * col[c] + (DIF * col[c]) * LCOL * intensity) + spec);
            rgba = ch[0] | (ch[1] << 8) | (ch[2] << 16) | 0xff000000u;
        }
        if (px < W && py < H) out[(size_t)py * W + px] = rgba;
    }
}

void launch_raster(const GymView &gv, uint32_t *obs, int W, int H, hipStream_t stream)
{
    hipLaunchKernelGGL(raster_kernel, dim3(gv.num_envs * gv.num_agents), dim3(256), 0, stream, gv, obs, W, H);
}

}  // namespace mv
