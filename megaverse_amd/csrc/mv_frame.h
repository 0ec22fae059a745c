// megaverse_amd/csrc/mv_frame.h -- frame setup of the observation pass: which primitives can a camera see, and where on the screen?
//
// Shared by mv_raster.hip (frame_setup_kernel: mv_reset / mv_render / hires) and by the step kernels, which run it as the second
// half of their workgroup: one workgroup per env, wave 0 runs the tick (physics + scenario logic + auto-reset), then all of its
// waves build the env's frame lists -- one launch less per step, and the frame setup of the quick envs hides the slow envs' tick.
// Reference for the drawables and the camera: magnum_env_renderer.cpp:158-340, env_renderer.hpp:34-38, layout_utils.cpp:17-126,
// component_object_stacking.hpp:170-198, scenario_default.hpp:99-170 (see mv_raster.hip for the pass as a whole).
#pragma once
#include <hip/hip_runtime.h>
#include <math.h>

#include "mv_math.h"
#include "mv_rearrange.h"
#include "mv_types.h"

namespace mv {
namespace {

#ifndef MV_STEP_THREADS
#define MV_STEP_THREADS 128
#endif
constexpr int STEP_THREADS = MV_STEP_THREADS;   // workgroup size of the fused step + frame setup kernels (see mv_step.hip)

constexpr float TAN_HALF_FOV = 1.19175359f;                       // tan(100deg / 2), env_renderer.hpp:36
constexpr float TAN_HALF_FOV_Y = 1.19175359f / (128.0f / 72.0f);  // aspect 128/72 is baked into the projection
constexpr float NEAR_Z = 0.01f, FAR_Z = 120.0f;
constexpr float OBJ_HALF = 0.39f, CARRY_SCALE = 0.78f;
constexpr int TILE_W = 16, TILE_H = 4;
#ifndef MV_RASTER_PPL
#define MV_RASTER_PPL 1   // 2 was measured: 114 us vs 99 us (bigger tiles keep more survivors, registers cost occupancy)
#endif
constexpr int PPL = MV_RASTER_PPL;   // pixels per lane in the raster kernel: a wave's tile is 16 x (4 PPL)
#ifndef MV_RASTER_WAVES
#define MV_RASTER_WAVES 7   // waves per SIMD the small variant is compiled for (register budget 512 / n)
#endif
constexpr int VIS_SMALL = 256, VIS_LARGE = 1024, VIS_XL = 2048;   // visible primitives kept per frame (TowerBuilding <= 121 slots, Obstacles <= 280;
                                                                  // Collect up to ~1300; a Hex maze seen from its rim: > 1024 of its <= 2146 slots)
// (LPT_BUCKETS, LPT_SUBS, lpt_sub_capacity: mv_types.h)
constexpr float CLIP_W = 0.005f;       // NEAR_Z / 2: boxes are clipped against this depth before projecting
constexpr int MAX_W = 1024, MAX_H = 1024;

__constant__ unsigned AGENT_COLORS[7] = {0xffdd3c, 0x3bb372, 0x2eb5d0, 0xffb400, 0xd468ee, 0x222222, 0xff0000};

enum : int { PRIM_NONE = 0, PRIM_BOX = 1, PRIM_CAPSULE = 2, PRIM_CONE = 3,
             PRIM_SPHERE_S = 4, PRIM_CAPSULE_S = 5, PRIM_CYLINDER_S = 6 };   // unit sphere / capsule (r 1, hl 1) / capped cylinder (r 1, hl 0.5),
                                                                              // scaled by hi, centred at lo in the primitive's frame

struct alignas(16) Prim {   // 32 B, one visible primitive of a frame (written by frame_setup_kernel, read by raster_kernel)
    float lo[3]; uint32_t meta;   // box: bounds minus the ray origin of its frame; capsule: centre (world); cone: apex (world)
    float hi[3]; uint32_t color;  //        capsule: (radius, halfLen, 0); cone: (base radius, height, +1 apex up / -1 apex down)
};                                // meta = kind | frame << 4 | slot << 8 ; frame 0 = world axes, 1+k = camera frame of agent k

// Per-frame header written by frame_setup_kernel for raster_fast_kernel (so that its prologue is a plain copy): floats
//   [16 k ..] frame of reference k -- camera of agent k for k < MAX_AGENTS, hex wall orientation k - MAX_AGENTS after that --: eye(3)
//   c(9) origin(3), and in the spare 16th float of record 0 the visible count (int bits)   [FH_LREL + 4 f ..] light position relative
//   to the viewer's eye in the axes of frame f (0 world, 1 + k record k)   [FH_WB + 2 r ..] 64-bit mask of list positions 64 r .. 64 r + 63
//   that hold an axis-aligned box in the world frame
enum : int { FH_CAM = 0, FH_CAM_STRIDE = 16, FH_COUNT = 15, FH_LREL = FH_CAM + FH_CAM_STRIDE * MAX_CAMS, FH_WB = FH_LREL + 4 * (1 + MAX_CAMS),
             FH_FLOATS = FH_WB + 2 * 16 };
static_assert(FH_FLOATS * 4 <= FRAME_HDR_BYTES && FH_FLOATS <= 256, "frame header: one float per thread of the raster prologue, one slot");

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

// Conservative screen rectangle of a box: the projection of its part in front of the plane w = CLIP_W (camera
// depth; nothing nearer than NEAR_Z = 2 CLIP_W can be hit).  Fully in front: the 8 projected corners.  Crossing
// the plane (the floor under the viewer, a wall beside it): the corners in front plus the points where the 12
// edges pierce the plane -- the convex hull of those is the clipped box, so its projection is bounded by theirs.
//   returns 0: nothing in front of the plane, 1: rect valid (pixels, one pixel of slack on every side)
// The rectangle only ever CULLS (both raster kernels intersect every primitive whose rectangle meets a tile; the pixels do not depend on it
// beyond that), and it carries a pixel of slack: its up to twenty reciprocals are the hardware's 1-ulp v_rcp_f32, not correctly rounded
// divides (each a chain of ~12 dependent instructions on the step kernel's critical path).
__device__ __forceinline__ int screen_rect(const float *blo, const float *bhi, int fr, const CamL *cams, int viewer, int W, int H,
                                           int rect[4], float *min_depth = nullptr)   // min_depth: the smallest camera depth of the box's corners
{
    const CamL &cv = cams[viewer];
    float cx[8], cy[8], cw[8];
    float wmin = INFINITY, wmax = -INFINITY;
    // The eight corners in the viewer's camera space: the min corner and the three edge vectors go through the frame's transform once, the
    // corners are sums of those (an affine map of a box: 51 operations instead of 8 x 18 -- or 8 x 54 under divergence, when the lanes of a wave hold
    // primitives of different frames).  The rectangle is a conservative bound with a pixel of slack: the last bits of the corners do not matter.
    V3 base = v3(blo[0], blo[1], blo[2]);
    V3 ex = v3(bhi[0] - blo[0], 0.0f, 0.0f), ey = v3(0.0f, bhi[1] - blo[1], 0.0f), ez = v3(0.0f, 0.0f, bhi[2] - blo[2]);
    if (fr != 1 + viewer) {   // (the viewer's own frame: already camera space)
        if (fr != 0) {        // another camera's / a wall's frame -> world
            const CamL &ck = cams[fr - 1];
            base = mat_mul(ck.c, base) + v3(ck.eye[0], ck.eye[1], ck.eye[2]);
            ex = mat_mul(ck.c, ex); ey = mat_mul(ck.c, ey); ez = mat_mul(ck.c, ez);
        }
        base = mat_tmul(cv.c, base - v3(cv.eye[0], cv.eye[1], cv.eye[2]));
        ex = mat_tmul(cv.c, ex); ey = mat_tmul(cv.c, ey); ez = mat_tmul(cv.c, ez);
    }
    {
        const V3 c0 = base, c1 = base + ex, c2 = base + ey, c3 = c1 + ey;
        const V3 cs[8] = {c0, c1, c2, c3, c0 + ez, c1 + ez, c2 + ez, c3 + ez};
#pragma unroll
        for (int c = 0; c < 8; ++c) {
            cx[c] = cs[c].x; cy[c] = cs[c].y; cw[c] = -cs[c].z;
            wmin = fminf(wmin, cw[c]); wmax = fmaxf(wmax, cw[c]);
        }
    }
    if (wmax < CLIP_W) return 0;
    if (min_depth) *min_depth = wmin;
    float xmin = INFINITY, xmax = -INFINITY, ymin = INFINITY, ymax = -INFINITY;
#pragma unroll
    for (int c = 0; c < 8; ++c)
        if (cw[c] >= CLIP_W) {
            const float iw = __builtin_amdgcn_rcpf(cw[c]);
            const float xn = cx[c] * iw, yn = cy[c] * iw;
            xmin = fminf(xmin, xn); xmax = fmaxf(xmax, xn); ymin = fminf(ymin, yn); ymax = fmaxf(ymax, yn);
        }
    if (wmin < CLIP_W) {
#pragma unroll
        for (int axis = 0; axis < 3; ++axis)
#pragma unroll
            for (int c = 0; c < 8; ++c) {
                const int d = c | (1 << axis);
                if (d == c) continue;
                if ((cw[c] >= CLIP_W) != (cw[d] >= CLIP_W)) {
                    const float t = (CLIP_W - cw[c]) * __builtin_amdgcn_rcpf(cw[d] - cw[c]);
                    const float xn = (cx[c] + t * (cx[d] - cx[c])) * (1.0f / CLIP_W), yn = (cy[c] + t * (cy[d] - cy[c])) * (1.0f / CLIP_W);
                    xmin = fminf(xmin, xn); xmax = fmaxf(xmax, xn); ymin = fminf(ymin, yn); ymax = fmaxf(ymax, yn);
                }
            }
    }
    xmin *= 1.0f / TAN_HALF_FOV; xmax *= 1.0f / TAN_HALF_FOV; ymin *= 1.0f / TAN_HALF_FOV_Y; ymax *= 1.0f / TAN_HALF_FOV_Y;
    // pixel i is covered when its centre (i + 0.5) lies inside; one pixel of slack on every side (clamped first: a
    // point just in front of the plane projects to 1e3 .. 1e6 screen widths)
    xmin = fmaxf(xmin, -4.0f); xmax = fminf(xmax, 4.0f); ymin = fmaxf(ymin, -4.0f); ymax = fminf(ymax, 4.0f);
    const float fx0 = (xmin * 0.5f + 0.5f) * float(W) - 1.5f, fx1 = (xmax * 0.5f + 0.5f) * float(W) + 0.5f;
    const float fy0 = (ymin * 0.5f + 0.5f) * float(H) - 1.5f, fy1 = (ymax * 0.5f + 0.5f) * float(H) + 0.5f;
    if (fx1 < 0.0f || fy1 < 0.0f || fx0 > float(W) || fy0 > float(H)) return 0;
    rect[0] = (int)floorf(fmaxf(fx0, 0.0f)); rect[1] = (int)ceilf(fminf(fx1, float(W - 1)));
    rect[2] = (int)floorf(fmaxf(fy0, 0.0f)); rect[3] = (int)ceilf(fminf(fy1, float(H - 1)));
    return 1;
}

// Hex scenarios: wall orientation k as a frame of reference without an eye: columns of c = its axes in the world, Ry(r) =
// [[c, 0, s], [0, 1, 0], [-s, 0, c]] with (cos, sin) = (0.8660254, 0.5), (0.8660254, -0.5), (0, 1) (mv_gen_hex.cpp, mv_physics.h)
__device__ __forceinline__ CamL hex_frame(int k)
{
    const float c = k == 2 ? 0.0f : 0.8660254f, s = k == 0 ? 0.5f : k == 1 ? -0.5f : 1.0f;
    CamL f;
    f.eye[0] = f.eye[1] = f.eye[2] = 0.0f;
    f.c[0] = c; f.c[1] = 0.0f; f.c[2] = s;
    f.c[3] = 0.0f; f.c[4] = 1.0f; f.c[5] = 0.0f;
    f.c[6] = -s; f.c[7] = 0.0f; f.c[8] = c;
    f.origin[0] = f.origin[1] = f.origin[2] = 0.0f;
    return f;
}

// LDS scratch of one frame setup in flight (one per workgroup, or one per wave when every wave of a workgroup sets up its own frame)
struct FrameScratch {
    CamL cam[MAX_CAMS];      // agent cameras, then (Hex scenarios) the three wall orientations as eye-less frames
    int cost;                // tiles x primitives the raster pass will have to look at (scheduling estimate)
    int cnt[8];              // [parity*4 + wave]: visible primitives found by each wave this round
    unsigned wbits[32];      // world-frame-box bit of every list position (frame header, raster_fast_kernel)
};

// ---- pass 1: which primitives can this camera see, and where on the screen?  THREADS (64, 128, 256) threads work on one frame:
// a whole workgroup (WAVE_LOCAL = false, barriers are __syncthreads()), or -- THREADS = 64, WAVE_LOCAL = true -- one wavefront of a
// workgroup whose other waves are busy with other frames (ordering points are wave_sync()).
#ifdef MV_TICK_TIMING
#define MV_TF(k)                                                                                              \
    do {                                                                                                      \
        const unsigned long long now_ = __builtin_amdgcn_s_memtime();                                         \
        if (gv.dbg && threadIdx.x == 0) gv.dbg[(size_t)(frame / gv.num_agents) * 64 + 56 + (k)] += now_ - tf_last_;   \
        tf_last_ = now_;                                                                                      \
    } while (0)
#else
#define MV_TF(k) do { } while (0)
#endif
// Long lists in depth classes.  A Hex frame holds ~700 visible primitives -- every wall of the maze inside the view frustum -- and all but the
// few nearest are hidden behind those: the observation pass culled and intersected its way through the whole list for every tile (11 rounds of 64
// positions, ~19 boxes surviving a tile's culling) to find out.  A ray's parameter t IS the camera depth of the hit (dc = (x, y, -1)), so the
// smallest camera depth of a primitive's bounding-box corners bounds every t it can produce from below.  The frame setup files each visible
// primitive under a depth class (64 classes, four per octave from 2^-7 up), first into a scratch copy of the list in the order found, then --
// one counting pass -- into the list proper class by class, nearest first, and leaves in the header the lower bound of the class each round of 64
// positions begins with (FH_WB + r: the long-list kernels do not use the world-box masks kept there for the short ones; FH_WB + 31: a marker that
// the list is in this order).  The pass stops walking the list where every pixel of the tile already holds a hit nearer than the next round's
// bound.  The pixels do not change: the winner is the minimum over (depth, slot), whatever the order (raster_glist_body).  Fast pixel mode only --
// the exact kernel resolves depth ties by list position, which must stay the drawables' order.
struct DepthSortScratch {
    unsigned char bin[VIS_XL];   // depth class of every position of the list as found
    int count[64];               // primitives per class, then the classes' write cursors
    int first[64];               // first position of every class in the list proper
};
constexpr unsigned DEPTH_SORTED_MARK = 0x44505354u;
__device__ __forceinline__ int depth_class(float d)   // d >= NEAR_Z: 0 .. 63, four classes per octave from 2^-7 (everything from 2^9 up: 63)
{
    return min(63, max(0, (int)(__float_as_uint(d) >> 21) - (120 << 2)));
}
__device__ __forceinline__ float depth_class_floor(int c) { return __uint_as_float((unsigned)(c + (120 << 2)) << 21); }

// PIPE (the software-pipelined multi-tick step kernels, mv_step.hip: step_ticks_pipe_kernel): this wave sets tick j's frame up WHILE the env's tick wave
// computes tick j + 1; everything read from the simulator state (header, agents, box / object / reward records) is loaded before one workgroup
// barrier -- in the last round of slots, behind its record loads --, which the tick wave meets before it writes tick j + 1's state back.
template <int THREADS, bool WAVE_LOCAL, bool PIPE = false>
__device__ __forceinline__ void frame_setup_body(const GymView &gv, const int frame, const int W, const int H, FrameScratch &fs, DepthSortScratch *ds = nullptr)
{
    static_assert(!WAVE_LOCAL || THREADS == 64, "a wave-local frame setup is one wavefront");
    auto sync = [] { if (WAVE_LOCAL) wave_sync(); else __syncthreads(); };
#ifdef MV_TICK_TIMING
    unsigned long long tf_last_ = __builtin_amdgcn_s_memtime();
#endif
    CamL *const s_cam = fs.cam;
    int &s_cost = fs.cost;
    int *const s_cnt = fs.cnt;
    unsigned *const s_wbits = fs.wbits;

    const int A = gv.num_agents;
    const int env = frame / A, viewer = frame - env * A;
    const int tid = WAVE_LOCAL ? (int)(threadIdx.x & 63) : (int)threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const EnvHeader *hdr = gv.hdr + env;
    const AgentState *agents = gv.agents + (size_t)env * A;
    const int maxVis = gv.vis_stride;
    if (tid == 0) s_cost = 0;
    if (tid < 32) s_wbits[tid] = 0u;
    int myCost = 0;
    Prim *vis = reinterpret_cast<Prim *>(gv.vis_prims) + (size_t)frame * maxVis;
    short4 *rects = reinterpret_cast<short4 *>(gv.vis_rects) + (size_t)frame * maxVis;
    const bool sorted = ds != nullptr && gv.depth_sort != 0 && gv.sort_scratch != nullptr;   // (uniform over the launch)
    Prim *visFound = vis;       // where the list goes as it is found: the list proper, or its scratch copy
    short4 *rectsFound = rects;
    if (sorted) {
        uint8_t *sc = gv.sort_scratch + (size_t)frame * maxVis * (sizeof(Prim) + sizeof(short4));
        visFound = reinterpret_cast<Prim *>(sc);
        rectsFound = reinterpret_cast<short4 *>(sc + (size_t)maxVis * sizeof(Prim));
        for (int i = tid; i < 64; i += THREADS) ds->count[i] = 0;
    }

    // ---- cameras
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
    const int scen = hdr->scenario;
    const bool hex = scen == SCN_HEX_MEMORY || scen == SCN_HEX_EXPLORE;
    if (hex && tid >= MAX_AGENTS && tid < MAX_CAMS) s_cam[tid] = hex_frame(tid - MAX_AGENTS);
    sync();
    if (tid < A || (hex && tid >= MAX_AGENTS && tid < MAX_CAMS)) {
        const V3 ev = v3(s_cam[viewer].eye[0], s_cam[viewer].eye[1], s_cam[viewer].eye[2]);
        const V3 ek = v3(s_cam[tid].eye[0], s_cam[tid].eye[1], s_cam[tid].eye[2]);
        const V3 o = mat_tmul(s_cam[tid].c, ev - ek);
        s_cam[tid].origin[0] = o.x; s_cam[tid].origin[1] = o.y; s_cam[tid].origin[2] = o.z;
    }
    MV_TF(0);   // cameras
    // ---- primitive slots, packed (slot order == the order the reference emits drawables == depth-tie order):
    //   layout slabs | terrain slabs (TowerBuilding: the building zone; Rearrange: static boxes + target items) | movable boxes / items
    //   | 2 cones per diamond | 3 per agent
    const int nLayout = hdr->num_boxes;
    const bool rearrange = scen == SCN_REARRANGE;   // its "terrain" slots: 9 static boxes, then the target arrangement's items
    const bool sokoban = scen == SCN_SOKOBAN;       // its "terrain" slots: one per level cell (wall cap / goal pad / nothing), x-major
    const int sokoW = sokoban ? max(hdr->W, 1) : 1;
    const int slotTerrain = nLayout, nTerrainSlots = scen == SCN_TOWER ? 1 : rearrange ? NUM_STATIC + hdr->num_terrain
                                                   : sokoban ? hdr->L * sokoW : hex ? 0 : hdr->num_terrain;
    const int slotObjects = slotTerrain + nTerrainSlots;
    // (Hex*: layout slots = the maze's boxes, no terrain / movable boxes, three slots per collectable: a pillar is three cylinders)
    const int slotRewards = slotObjects + hdr->num_objects, nRewardSlots = scen == SCN_TOWER ? 0 : hex ? 3 * hdr->num_rewards : 2 * hdr->num_rewards;
    const int slotAgents = slotRewards + nRewardSlots;
    const int numSlots = slotAgents + 3 * A;
    const LayoutBox *gboxes = gv.boxes + (size_t)env * gv.box_stride;

    // Each round classifies THREADS slots and appends the visible ones to the LDS list (order-free: depth ties are
    // resolved on the slot id).
    int nVis = 0;   // wave-uniform running total
    constexpr int NW = THREADS / 64;   // waves
    for (int rd = 0; rd * THREADS < numSlots; ++rd) {
        float ext[3] = {0, 0, 0};   // scaled shapes: half extents of the bounding box
        const int slot = tid + THREADS * rd;
        int kind = PRIM_NONE, fr = 0;
        unsigned color = 0;
        float lo[3] = {0, 0, 0}, hi[3] = {0, 0, 0};
        if (slot < numSlots) {
            if (slot < nLayout && hex) {   // floor / wall / edging / landmark: a box in the world or in a wall frame (records 8..10)
                const HexRec b = gv.hex_boxes[(size_t)env * HEX_MAX_BOXES + slot];
                kind = PRIM_BOX;
                fr = (b.meta & 15) == 0 ? 0 : MAX_AGENTS + (b.meta & 15);
                lo[0] = b.a[0]; lo[1] = b.a[1]; lo[2] = b.a[2]; hi[0] = b.b[0]; hi[1] = b.b[1]; hi[2] = b.b[2];
                color = (unsigned)b.color;
            } else if (slot < nLayout) {
                const LayoutBox b = gboxes[slot];
                if (b.type & VX_OPAQUE) {
                    kind = PRIM_BOX;
                    lo[0] = float(b.min[0]); lo[1] = float(b.min[1]); lo[2] = float(b.min[2]);
                    hi[0] = float(b.max[0]); hi[1] = float(b.max[1]); hi[2] = float(b.max[2]);
                    if (sokoban) {   // addBoundingBoxes scales by the voxel size, 2 (layout_utils.cpp:22-34, scenario_sokoban.cpp:104-120)
                        lo[0] *= 2.0f; lo[1] *= 2.0f; lo[2] *= 2.0f; hi[0] *= 2.0f; hi[1] *= 2.0f; hi[2] *= 2.0f;
                    }
                    color = (unsigned)(b.slot == 0 ? hdr->layout_color : hdr->wall_color);
                }
            } else if (slot < slotObjects) {
                if (rearrange) {
                    const int j = slot - slotTerrain;
                    if (j < NUM_STATIC) {   // raised floor + pedestals (addStaticCollidingBox)
                        V3 blo3, bhi3;
                        color = static_box(j, blo3, bhi3);
                        kind = PRIM_BOX;
                        lo[0] = blo3.x; lo[1] = blo3.y; lo[2] = blo3.z; hi[0] = bhi3.x; hi[1] = bhi3.y; hi[2] = bhi3.z;
                    } else {                // the target arrangement on the left pedestal
                        const ArrangementItem it = gv.items[(size_t)env * MAX_ITEMS + (j - NUM_STATIC)];
                        const V3 sc = item_draw_scale(it.shape);
                        const float cx = float(it.off[0] + RE_LEFT_X) + 0.5f, cy = float(it.off[1] + RE_LEFT_Y)
                                               + 0.5f, cz = float(it.off[2] + RE_LEFT_Z) + 0.5f;
                        color = (unsigned)it.color;
                        if (it.shape == SHAPE_BOX) {
                            kind = PRIM_BOX;
                            lo[0] = cx - sc.x; lo[1] = cy - sc.y; lo[2] = cz - sc.z; hi[0] = cx + sc.x; hi[1] = cy + sc.y; hi[2] = cz + sc.z;
                        } else {
                            kind = it.shape == SHAPE_SPHERE ? PRIM_SPHERE_S : it.shape == SHAPE_CAPSULE ? PRIM_CAPSULE_S : PRIM_CYLINDER_S;
                            lo[0] = cx; lo[1] = cy; lo[2] = cz; hi[0] = sc.x; hi[1] = sc.y; hi[2] = sc.z;
                        }
                    }
                } else if (sokoban) {   // wall caps (0.7 high: the walls themselves are not drawn) and goal pads, scenario_sokoban.cpp:243-273
                    const int j = slot - slotTerrain, cxi = j / sokoW, czi = j - cxi * sokoW;
                    const int t = cxi < SOKO_DIM && czi < SOKO_DIM ? (int)gv.soko_cells[(size_t)env * (SOKO_DIM * SOKO_DIM) + cxi * SOKO_DIM + czi] : 0;
                    if (t) {
                        const float hh = t == SOKO_WALL ? 0.35f : 0.025f;
                        const float cx = 2.0f * float(cxi) + 2.0f / 2, cy = 2.0f + hh, cz = 2.0f * float(czi) + 2.0f / 2;
                        kind = PRIM_BOX;
                        lo[0] = cx - 1.0f; lo[1] = cy - hh; lo[2] = cz - 1.0f; hi[0] = cx + 1.0f; hi[1] = cy + hh; hi[2] = cz + 1.0f;
                        color = t == SOKO_WALL ? 0xffa770u : 0x50c878u;   // LIGHT_ORANGE / LIGHT_GREEN
                    }
                } else if (scen == SCN_TOWER) {   // building-zone slab (layout_utils.cpp:53-68)
                    kind = PRIM_BOX;
                    lo[0] = float(hdr->bz[0]); lo[1] = 1.0f; lo[2] = float(hdr->bz[2]);
                    hi[0] = float(hdr->bz[1]); hi[1] = 1.0f + 0.05f; hi[2] = float(hdr->bz[3]);
                    color = 0x555555u;
                } else {                   // exit pad / lava: 0.05-thick slab on the box's floor
                    const TerrainBox t = gv.terrain[(size_t)env * MAX_TERRAIN + (slot - slotTerrain)];
                    kind = PRIM_BOX;
                    lo[0] = float(t.min[0]); lo[1] = float(t.min[1]); lo[2] = float(t.min[2]);
                    hi[0] = float(t.max[0]); hi[1] = float(t.min[1]) + 0.05f; hi[2] = float(t.max[2]);
                    color = t.type == TERRAIN_EXIT ? 0x50c878u : 0xff0000u;   // platforms.hpp:47-56
                }
            } else if (slot < slotRewards) {
                const MovableObject o = gv.objects[(size_t)env * MAX_OBJECTS + (slot - slotObjects)];
                color = 0xadd8e6u;
                kind = PRIM_BOX;
                if (rearrange) {   // the movable copy of item (slot - slotObjects): standing on the right pedestal or carried
                    const ArrangementItem it = gv.items[(size_t)env * MAX_ITEMS + (slot - slotObjects)];
                    V3 sc = item_draw_scale(it.shape);
                    float cx = float(o.x) + 0.5f, cy = float(o.y) + 0.5f, cz = float(o.z) + 0.5f;
                    if (o.state > 0) {
                        fr = (int)o.state;
                        cx = 0.0f; cy = -0.44f + -0.3f; cz = -1.0f;
                        sc = v3(sc.x * CARRY_SCALE, sc.y * CARRY_SCALE, sc.z * CARRY_SCALE);
                    }
                    color = (unsigned)it.color;
                    if (it.shape == SHAPE_BOX) {
                        lo[0] = cx - sc.x; lo[1] = cy - sc.y; lo[2] = cz - sc.z; hi[0] = cx + sc.x; hi[1] = cy + sc.y; hi[2] = cz + sc.z;
                    } else {
                        kind = it.shape == SHAPE_SPHERE ? PRIM_SPHERE_S : it.shape == SHAPE_CAPSULE ? PRIM_CAPSULE_S : PRIM_CYLINDER_S;
                        lo[0] = cx; lo[1] = cy; lo[2] = cz; hi[0] = sc.x; hi[1] = sc.y; hi[2] = sc.z;
                    }
                } else if (sokoban) {   // the pushable boxes, scenario_sokoban.cpp:275-293: half extents (0.8, 0.36, 0.8) at (x + 0.5, y + 0.2, z + 0.5) voxels
                    const float sx = (2.0f / 2) * 0.8f, sy = 0.45f * 0.8f;
                    const float cx = (float(o.x) + 0.5f) * 2.0f, cy = (float(o.y) + 0.2f) * 2.0f, cz = (float(o.z) + 0.5f) * 2.0f;
                    color = 0x3a7fa6u;   // DARK_BLUE
                    lo[0] = cx - sx; lo[1] = cy - sy; lo[2] = cz - sx; hi[0] = cx + sx; hi[1] = cy + sy; hi[2] = cz + sx;
                } else if (o.state <= 0) {
                    const float cx = float(o.x) + 0.5f, cy = float(o.y) + 0.5f, cz = float(o.z) + 0.5f;
                    lo[0] = cx - OBJ_HALF; lo[1] = cy - OBJ_HALF; lo[2] = cz - OBJ_HALF;
                    hi[0] = cx + OBJ_HALF; hi[1] = cy + OBJ_HALF; hi[2] = cz + OBJ_HALF;
                } else {
                    fr = (int)o.state;
                    const float hh = OBJ_HALF * CARRY_SCALE;
                    const float cx = 0.0f, cy = -0.44f + -0.3f, cz = -1.0f;
                    lo[0] = cx - hh; lo[1] = cy - hh; lo[2] = cz - hh;
                    hi[0] = cx + hh; hi[1] = cy + hh; hi[2] = cz + hh;
                }
            } else if (slot < slotAgents && hex) {   // collectables: addSphere / addPillar / addDiamond, layout_utils.cpp:85-126
                const int q = slot - slotRewards, j = q / 3, part = q - 3 * j;
                const HexRec o = gv.hex_objs[(size_t)env * HEX_MAX_OBJS + j];
                const int shape = o.meta & 15;
                color = (unsigned)o.color;
                if (shape == HEX_SPHERE) {
                    if (part == 0) { kind = PRIM_SPHERE_S; lo[0] = o.a[0]; lo[1] = o.a[1]; lo[2] = o.a[2]; hi[0] = o.b[0]; hi[1] = o.b[1]; hi[2] = o.b[2]; }
                } else if (shape == HEX_PILLAR) {   // a cylinder and two caps (1.2 x as wide, 0.15 high) at +-0.47 of its scale
                    const float capY = 0.47f * o.b[1];
                    kind = PRIM_CYLINDER_S;
                    lo[0] = o.a[0]; lo[2] = o.a[2];
                    lo[1] = part == 0 ? o.a[1] : part == 1 ? o.a[1] + capY : o.a[1] - capY;
                    hi[0] = part == 0 ? o.b[0] : o.b[0] * 1.2f; hi[1] = part == 0 ? o.b[1] : 0.15f; hi[2] = part == 0 ? o.b[2] : o.b[2] * 1.2f;
                } else if (part < 2) {              // two cones base to base
                    kind = PRIM_CONE;
                    lo[0] = o.a[0]; lo[2] = o.a[2];
                    lo[1] = part == 0 ? o.a[1] + 0.5f * o.b[1] : o.a[1] - 1.5f * o.b[1];
                    hi[0] = o.b[0]; hi[1] = o.b[1]; hi[2] = part == 0 ? 1.0f : -1.0f;
                }
            } else if (slot < slotAgents) {   // diamonds: addDiamond, layout_utils.cpp:114-126
                const int j = (slot - slotRewards) >> 1, part = (slot - slotRewards) & 1;
                const MovableObject r = gv.rewards_obj[(size_t)env * gv.reward_stride + j];
                if (r.state != 0) {
                    // Obstacles (scenario_obstacles.cpp:254): scale (0.17, 0.45, 0.17) * 0.8 at y + 0.7, green;
                    // Collect (scenario_collect.cpp:192,208): unscaled at y + 0.8, green (+1) or red (-1)
                    const bool collect = scen == SCN_COLLECT;
                    const float sx = collect ? 0.17f : 0.17f * 0.8f, sy = collect ? 0.45f : 0.45f * 0.8f;
                    const float cx = float(r.x) + 0.5f, cy = float(r.y) + (collect ? 0.8f : 0.7f), cz = float(r.z) + 0.5f;
                    kind = PRIM_CONE;
                    color = r.state == 2 ? 0xff0000u : 0x3bb372u;
                    lo[0] = cx; lo[2] = cz;
                    lo[1] = part == 0 ? cy + 0.5f * sy : cy - 1.5f * sy;
                    hi[0] = sx; hi[1] = sy; hi[2] = part == 0 ? 1.0f : -1.0f;
                }
            } else {
                const int q = slot - slotAgents;
                const int k = q / 3, part = q - 3 * k;
                if (part == 0 && k != viewer) {
                    const AgentState a = agents[k];
                    kind = PRIM_CAPSULE;
                    lo[0] = a.pos[0]; lo[1] = (a.pos[1] + 0.05f) + 0.09f; lo[2] = a.pos[2];
                    hi[0] = 0.35f; hi[1] = 0.36f; hi[2] = 0.0f;
                    color = AGENT_COLORS[k % 7];
                } else if (part == 1 && k != viewer) {
                    kind = PRIM_BOX; fr = 1 + k;
                    lo[0] = -0.25f; lo[1] = -0.12f; lo[2] = -0.19f - 0.2f;
                    hi[0] = 0.25f; hi[1] = 0.12f; hi[2] = -0.19f + 0.2f;
                    color = 0x2c3e50u;
                } else if (part == 2) {
                    const float bw = hdr->bar_half_width;
                    kind = PRIM_BOX; fr = 1 + k;
                    lo[0] = -bw; lo[1] = -0.131f - 0.0015f; lo[2] = -0.2f - 0.001f;
                    hi[0] = bw; hi[1] = -0.131f + 0.0015f; hi[2] = -0.2f + 0.001f;
                    color = 0x2eb5d0u;
                }
            }
        }
        MV_TF(1);   // the slots' records
        // the state is read (the loads have returned: the barrier's fences): the tick wave may write the next tick's
        if (PIPE && (rd + 1) * THREADS >= numSlots) __syncthreads();
        // frame-level visibility
        int cls = 0;
        int rect[4] = {0, 0, 0, 0};
        float minDepth = NEAR_Z;
        if (kind != PRIM_NONE) {
            float blo[3] = {lo[0], lo[1], lo[2]}, bhi[3] = {hi[0], hi[1], hi[2]};
            if (kind == PRIM_CAPSULE) {
                const float r = hi[0], hl = hi[1];
                blo[0] = lo[0] - r; blo[1] = lo[1] - (hl + r); blo[2] = lo[2] - r;
                bhi[0] = lo[0] + r; bhi[1] = lo[1] + (hl + r); bhi[2] = lo[2] + r;
            } else if (kind >= PRIM_SPHERE_S) {
                ext[0] = hi[0]; ext[1] = kind == PRIM_CAPSULE_S ? hi[1] * 2.0f : kind == PRIM_CYLINDER_S ? hi[1] * 0.5f : hi[1]; ext[2] = hi[2];
                blo[0] = lo[0] - ext[0]; blo[1] = lo[1] - ext[1]; blo[2] = lo[2] - ext[2];
                bhi[0] = lo[0] + ext[0]; bhi[1] = lo[1] + ext[1]; bhi[2] = lo[2] + ext[2];
            } else if (kind == PRIM_CONE) {
                const float r = hi[0], h = hi[1];
                blo[0] = lo[0] - r; blo[2] = lo[2] - r; bhi[0] = lo[0] + r; bhi[2] = lo[2] + r;
                blo[1] = hi[2] > 0.0f ? lo[1] - h : lo[1];
                bhi[1] = hi[2] > 0.0f ? lo[1] : lo[1] + h;
            }
            cls = screen_rect(blo, bhi, fr, s_cam, viewer, W, H, rect, &minDepth);
        }
        MV_TF(2);   // screen rectangles
        const unsigned long long mV = __ballot(cls != 0);
        int *cnt = s_cnt + (rd & 1) * 4;   // double-buffered: one barrier per round
        if (lane == 0) cnt[wave] = __popcll(mV);
        sync();
        int pos = nVis, tot = 0;
#pragma unroll
        for (int q = 0; q < NW; ++q) {
            if (q < wave) pos += cnt[q];
            tot += cnt[q];
        }
        nVis += tot;
        pos += __popcll(mV & ((1ull << lane) - 1ull));
        MV_TF(3);   // list positions (barrier)
        if (cls != 0 && pos >= maxVis) atomicOr(&gv.episode_status[gv.num_envs + 1], (int)ST_VISIBLE);   // dropped -- and reported by mv_step
        if (cls != 0 && pos < maxVis) {   // at most vis_stride visible primitives per frame
            Prim p;
            p.meta = (uint32_t)(kind | (fr << 4) | (slot << 8));
            p.color = color;
            if (kind == PRIM_BOX || kind >= PRIM_SPHERE_S) {   // bounds (scaled shapes: centre) relative to the ray origin of the primitive's frame
                V3 o = v3(0.0f, 0.0f, 0.0f);
                if (fr == 0) o = v3(s_cam[viewer].eye[0], s_cam[viewer].eye[1], s_cam[viewer].eye[2]);
                else if (fr != 1 + viewer) o = v3(s_cam[fr - 1].origin[0], s_cam[fr - 1].origin[1], s_cam[fr - 1].origin[2]);
                p.lo[0] = lo[0] - o.x; p.lo[1] = lo[1] - o.y; p.lo[2] = lo[2] - o.z;
                if (kind == PRIM_BOX) { p.hi[0] = hi[0] - o.x; p.hi[1] = hi[1] - o.y; p.hi[2] = hi[2] - o.z; }
                else { p.hi[0] = hi[0]; p.hi[1] = hi[1]; p.hi[2] = hi[2]; }
            } else {
                p.lo[0] = lo[0]; p.lo[1] = lo[1]; p.lo[2] = lo[2];
                p.hi[0] = hi[0]; p.hi[1] = hi[1]; p.hi[2] = hi[2];
            }
            visFound[pos] = p;
            rectsFound[pos] = make_short4((short)rect[0], (short)rect[1], (short)rect[2], (short)rect[3]);
            if (sorted) {
                const int dc = depth_class(fmaxf(minDepth, NEAR_Z));
                ds->bin[pos] = (unsigned char)dc;
                atomicAdd(&ds->count[dc], 1);
            }
            if (kind == PRIM_BOX && fr == 0 && pos < 1024) atomicOr(&s_wbits[pos >> 5], 1u << (pos & 31));   // (the header's masks: the short-list raster)
            myCost += ((rect[1] / TILE_W) - (rect[0] / TILE_W) + 1) * ((rect[3] / TILE_H) - (rect[2] / TILE_H) + 1);
        }
    }
    MV_TF(4);   // records and rectangles written
    if (myCost) atomicAdd(&s_cost, myCost);
    sync();
    float *const fhs = reinterpret_cast<float *>(gv.vis_hdr + (size_t)frame * FRAME_HDR_BYTES);
    if (sorted) {   // the list as found -> the list proper, class by class
        const int n = min(nVis, maxVis);
        if (tid < 64) {   // (one wave: the classes' first positions = exclusive prefix sum of their counts)
            const int h = ds->count[tid];
            int x = h;
#pragma unroll
            for (int off = 1; off < 64; off <<= 1) {
                const int y = __shfl_up(x, off, 64);
                if (lane >= off) x += y;
            }
            ds->first[tid] = x - h;
            ds->count[tid] = x - h;   // the class's write cursor
        }
        sync();
        for (int i = tid; i < n; i += THREADS) {
            const int place = atomicAdd(&ds->count[ds->bin[i]], 1);
            vis[place] = visFound[i];
            rects[place] = rectsFound[i];
        }
        // What the pass needs to know before it walks round r (positions 64 r ..): the class that round begins with -- nothing from there on is
        // nearer than its floor.  (Also built and measured, r07e: the rectangle that holds everything from round r on, so that a tile it does not
        // meet could stop as well -- no gain on top of the depth bound, and 10 us more frame setup per tick.)
        if (tid < 32) {
            const int p0 = 64 * tid;
            int c = 0;
            for (int q = 1; q < 64; ++q)
                if (ds->first[q] <= p0) c = q;   // (first[] is non-decreasing; an empty class shares its first position with the next one)
            fhs[FH_WB + tid] = __uint_as_float(tid == 31 ? DEPTH_SORTED_MARK : (p0 < n ? (unsigned)c : 0u));
        }
    }
    // The frame appends itself to its cost bin with a RETURNING atomic (its place in the bin's list), a round trip to L2 that nothing else in this
    // kernel waits for: issued first, the header's stores go out while it is in flight, the dependent store comes last.
    int binPlace = 0, bin = 0;
    if (tid == 0) {
        gv.vis_count[frame] = min(nVis, maxVis);
        // longest-processing-time-first scheduling of the raster pass: frames are binned by estimated cost, the raster
        // kernel takes them from the most expensive bin down (frames differ several-fold in cost; starting the heavy
        // ones first keeps the tail of the launch short); frame_order_kernel turns the bins into a permutation
        const int tiles = ((W + TILE_W - 1) / TILE_W) * ((H + TILE_H - 1) / TILE_H);
        bin = min(LPT_BUCKETS - 1, (16 * s_cost) / max(tiles, 1));   // 1/16 of "one primitive on every tile"
        gv.lpt_bucket[frame] = bin;   // (exact raster kernel: frame_order_kernel sorts on these)
        // fast raster kernel: the frame appends itself to its bin; every raster workgroup prefix-sums the 256 bin counts in its prologue
        // and looks its frame up -- no sort kernel, no launch boundary.  The order inside a bin is whatever the atomics made it: it only
        // schedules.  gv.lpt_hists histograms rotate: this pass's raster reads one while the setups of the next steps -- which may run
        // concurrently, on the simulation stream (mv_api.hip) -- fill the following ones, each clearing the one after its own.
        binPlace = atomicAdd(&gv.lpt_hist[(gv.lpt_parity * LPT_BUCKETS + bin) * LPT_SUBS + (frame & (LPT_SUBS - 1))], 1);
    }
    {   // frame header
        float *fh = reinterpret_cast<float *>(gv.vis_hdr + (size_t)frame * FRAME_HDR_BYTES);
        if (tid == 0) fh[FH_COUNT] = __int_as_float(min(nVis, maxVis));
        if (tid < A || (hex && tid >= MAX_AGENTS && tid < MAX_CAMS)) {
            const CamL &cm = s_cam[tid];
            float *o = fh + FH_CAM + FH_CAM_STRIDE * tid;
            o[0] = cm.eye[0]; o[1] = cm.eye[1]; o[2] = cm.eye[2];
#pragma unroll
            for (int q = 0; q < 9; ++q) o[3 + q] = cm.c[q];
            o[12] = cm.origin[0]; o[13] = cm.origin[1]; o[14] = cm.origin[2];   // (o[15] of record 0 is the count)
        }
        // light (0,4,2) camera-relative (magnum_env_renderer.cpp:201) in the axes of every frame a box can live in
        if (tid >= 32 && (tid <= 32 + A || (hex && tid > 32 + MAX_AGENTS && tid <= 32 + MAX_CAMS))) {
            const int f = tid - 32;
            const V3 lw = mat_mul(s_cam[viewer].c, v3(0.0f, 4.0f, 2.0f));
            const V3 l = f == 0 ? lw : f == 1 + viewer ? v3(0.0f, 4.0f, 2.0f) : mat_tmul(s_cam[f - 1].c, lw);
            float *o = fh + FH_LREL + 4 * f;
            o[0] = l.x; o[1] = l.y; o[2] = l.z; o[3] = 0.0f;
        }
        // (sorted: the rounds' depth bounds, above; word 31 -- positions 992 .. 1023, which no short list has -- is the marker's place)
        if (tid < 32 && !sorted) fh[FH_WB + tid] = __uint_as_float(tid == 31 ? 0u : s_wbits[tid]);
    }
    if (tid == 0) gv.lpt_list[(size_t)(bin * LPT_SUBS + (frame & (LPT_SUBS - 1))) * lpt_sub_capacity(gv.num_envs * gv.num_agents) + binPlace] = frame;
    // the histogram the NEXT pass fills: last read by a raster pass as many passes ago as there are slots, which this step waited for
    if (frame == 0 && !gv.lpt_no_clear)
        for (int i = tid; i < LPT_BUCKETS * LPT_SUBS; i += THREADS) gv.lpt_hist[((gv.lpt_parity + 1) % gv.lpt_hists) * (LPT_BUCKETS * LPT_SUBS) + i] = 0;
    MV_TF(5);   // header, cost bin
    sync();   // the LDS scratch above is reused by the next frame of this workgroup (fused step + setup kernels)
}

}  // namespace
}  // namespace mv
