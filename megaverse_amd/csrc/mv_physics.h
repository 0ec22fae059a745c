// megaverse_amd/csrc/mv_physics.h -- kinematic character physics shared by the step kernels.
//
// Replaces KinematicCharacterController::{setAcceleration, preStep, playerStep, stepUp, stepForwardAndStrafe,
// stepDown, recoverFromPenetration, updateTargetPositionBasedOnCollision}
//   (reference: src/libs/env/src/kinematic_character_controller.cpp:156-442,519-602,753-792) and the Bullet
//   2.89 calls they make (ghost convexSweepTest, contact manifolds; third party, restated as conservative
//   advancement on exact closest points -- DESIGN.md "physics model").
// One wavefront owns one env; every lane holds NC colliders in VGPRs (slot = lane + 64 * k, k < NC); a sweep
// is "each lane casts against its colliders, a 64-bit (fraction, slot) wave-min picks the winner".
#pragma once
#include <hip/hip_runtime.h>
#include <float.h>

#include "mv_math.h"
#include "mv_types.h"

namespace mv {
namespace {

// constants: see SURVEY.md appendix A for the reference line of each
constexpr float DT = 1.0f / 15.0f;
constexpr float CAP_R = 0.33f;
constexpr float CAP_HH = 1.05f * 0.5f;
constexpr float STEP_HEIGHT = 0.2f;
constexpr float GRAVITY = 1.4f * 9.8f;
constexpr float FALL_SPEED = 55.0f;
constexpr float MAX_H_SPEED = 4.5f, MAX_AIR_SPEED = 1.0f, NORMAL_DECEL = 15.0f;
constexpr float MAX_ACCEL = 35.0f + 15.0f, MAX_AIR_ACCEL = 3.0f, EXCEED_DECEL = (35.0f + 15.0f) * 2;
constexpr float MAX_PEN_DEPTH = 0.041f;
constexpr float MAX_SLOPE_COS = 0.70710678f;
constexpr float ALLOWED_CCD_PEN = 0.04f;
constexpr float CAST_RADIUS = 0.001f;
constexpr int CAST_MAX_ITER = 64;
constexpr float SIMD_EPS = FLT_EPSILON;
constexpr float ROTATE_RAD = 3.5f, ROTATE_X_RAD = 1.5f;
constexpr float OBJ_COLL_HALF = 0.39f * 1.15f;
constexpr float OBJ_COLL_YOFF = -0.05f;

struct Col {
    int kind;   // 0 none, 1 box (bounds already grown by CAP_HH in y), 2 vertical capsule, 3 + k (OBB builds only): box in hex wall frame k
    V3 lo, hi;
};

// Hex scenarios: the three directions a honeycomb's walls run in, as rotations about Y (mv_gen_hex.cpp); frame k: local = Ry^T world
__device__ __forceinline__ V3 hex_to_local(int k, V3 p)
{
    const float c = k == 2 ? 0.0f : 0.8660254f, s = k == 0 ? 0.5f : k == 1 ? -0.5f : 1.0f;
    return v3(c * p.x - s * p.z, p.y, s * p.x + c * p.z);
}
__device__ __forceinline__ V3 hex_to_world(int k, V3 p)
{
    const float c = k == 2 ? 0.0f : 0.8660254f, s = k == 0 ? 0.5f : k == 1 ? -0.5f : 1.0f;
    return v3(c * p.x + s * p.z, p.y, c * p.z - s * p.x);
}

struct Closest {
    float dist;
    V3 n;
};

__device__ __forceinline__ Closest closest_box(V3 p, V3 lo, V3 hi, float r)
{
    const float qx = fmin_sel(fmax_sel(p.x, lo.x), hi.x);
    const float qy = fmin_sel(fmax_sel(p.y, lo.y), hi.y);
    const float qz = fmin_sel(fmax_sel(p.z, lo.z), hi.z);
    const V3 v = v3(p.x - qx, p.y - qy, p.z - qz);
    const float d2 = len2(v);
    Closest c;
    if (d2 > 0.0f) {
        const float d = sqrtf(d2);
        const float inv = 1.0f / d;
        c.dist = d - r;
        c.n = v * inv;
    } else {
        float m = p.x - lo.x; V3 n = v3(-1, 0, 0);
        float t = hi.x - p.x; if (t < m) { m = t; n = v3(1, 0, 0); }
        t = p.y - lo.y; if (t < m) { m = t; n = v3(0, -1, 0); }
        t = hi.y - p.y; if (t < m) { m = t; n = v3(0, 1, 0); }
        t = p.z - lo.z; if (t < m) { m = t; n = v3(0, 0, -1); }
        t = hi.z - p.z; if (t < m) { m = t; n = v3(0, 0, 1); }
        c.dist = -m - r;
        c.n = n;
    }
    return c;
}

__device__ __forceinline__ Closest closest_capsule(V3 p, V3 centre, float halfLen, float r)
{
    const float qy = fmin_sel(fmax_sel(p.y, centre.y - halfLen), centre.y + halfLen);
    const V3 v = v3(p.x - centre.x, p.y - qy, p.z - centre.z);
    const float d2 = len2(v);
    Closest c;
    if (d2 > 1e-12f) {
        const float d = sqrtf(d2);
        const float inv = 1.0f / d;
        c.dist = d - r;
        c.n = v * inv;
    } else {
        c.dist = -r;
        c.n = v3(1, 0, 0);
    }
    return c;
}

template <bool OBB = false>
__device__ __forceinline__ Closest closest(const Col &col, V3 p)
{
    if (col.kind == 1) return closest_box(p, col.lo, col.hi, CAP_R);
    if (OBB && col.kind >= 3) {   // rotated about Y only: the capsule is vertical in the wall's frame too
        Closest c = closest_box(hex_to_local(col.kind - 3, p), col.lo, col.hi, CAP_R);
        c.n = hex_to_world(col.kind - 3, c.n);
        return c;
    }
    return closest_capsule(p, col.lo, col.hi.x, 2 * CAP_R);
}

// The same closest-point query without the normalisation: v = p minus the collider's closest point (NOT unit), d = |v|, dist = the signed
// surface distance.  Centre inside the (grown) box / on the capsule's axis: v = the unit exit normal, d = 1.  closest().n == v * (1 / d).
struct Raw {
    V3 v;
    float d, dist;
};

__device__ __forceinline__ Raw raw_box(V3 p, V3 lo, V3 hi, float r)
{
    const float qx = fmin_sel(fmax_sel(p.x, lo.x), hi.x);
    const float qy = fmin_sel(fmax_sel(p.y, lo.y), hi.y);
    const float qz = fmin_sel(fmax_sel(p.z, lo.z), hi.z);
    const V3 v = v3(p.x - qx, p.y - qy, p.z - qz);
    const float d2 = len2(v);
    Raw c;
    if (d2 > 0.0f) {
        c.d = sqrtf(d2);
        c.dist = c.d - r;
        c.v = v;
    } else {
        float m = p.x - lo.x; V3 n = v3(-1, 0, 0);
        float t = hi.x - p.x; if (t < m) { m = t; n = v3(1, 0, 0); }
        t = p.y - lo.y; if (t < m) { m = t; n = v3(0, -1, 0); }
        t = hi.y - p.y; if (t < m) { m = t; n = v3(0, 1, 0); }
        t = p.z - lo.z; if (t < m) { m = t; n = v3(0, 0, -1); }
        t = hi.z - p.z; if (t < m) { m = t; n = v3(0, 0, 1); }
        c.d = 1.0f;
        c.dist = -m - r;
        c.v = n;
    }
    return c;
}

__device__ __forceinline__ Raw raw_capsule(V3 p, V3 centre, float halfLen, float r)
{
    const float qy = fmin_sel(fmax_sel(p.y, centre.y - halfLen), centre.y + halfLen);
    const V3 v = v3(p.x - centre.x, p.y - qy, p.z - centre.z);
    const float d2 = len2(v);
    Raw c;
    if (d2 > 1e-12f) {
        c.d = sqrtf(d2);
        c.dist = c.d - r;
        c.v = v;
    } else {
        c.d = 1.0f;
        c.dist = -r;
        c.v = v3(1, 0, 0);
    }
    return c;
}

// conservative advancement of the capsule along d against one collider ([3P] btContinuousConvexCollision::calcTimeOfImpact for a
// translating shape with exact closest points).  Bullet advances by dist / (-(d . n)) with the unit normal n = v / |v|; here the same
// quotient is formed as (dist |v|) / (-(d . v)): ONE correctly rounded divide per iteration instead of two (1 / |v| and the quotient),
// the normal is normalised once, on a hit.  A box in a hex wall frame (OBB builds) is cast in that frame: start and direction are rotated
// in once, the normal is rotated back once.  (The CPU restatement the parity tests compare with forms the same operations in the same order.)
#ifdef MV_TICK_TIMING
__shared__ unsigned s_cast_dbg[8];
#define MV_CAST_ITERS(n) do { if (iters) *iters = (n); } while (0)
#else
#define MV_CAST_ITERS(n) do { } while (0)
#endif
template <bool OBB = false>
__device__ __forceinline__ bool convex_cast(const Col &col, V3 p, V3 d, float &fraction, V3 &normal, int *iters = nullptr)
{
    const bool boxLike = col.kind != 2;
    const int fr = col.kind - 3;
    if (OBB && col.kind >= 3) { p = hex_to_local(fr, p); d = hex_to_local(fr, d); }
    float lambda = 0.0f, lastLambda = 0.0f;
    int numIter = 0;
    Raw c = boxLike ? raw_box(p, col.lo, col.hi, CAP_R) : raw_capsule(p, col.lo, col.hi.x, 2 * CAP_R);
    float dist = c.dist + ALLOWED_CCD_PEN;
    float proj = -dot(d, c.v);   // |v| times Bullet's projected velocity
    if (proj <= SIMD_EPS * c.d) return false;
    while (dist > CAST_RADIUS) {
        proj = -dot(d, c.v);
        if (proj <= SIMD_EPS * c.d) return false;
        lambda = lambda + (dist * c.d) / proj;
        if (lambda > 1.0f) return false;
        if (lambda < 0.0f) return false;
        if (lambda <= lastLambda) return false;
        lastLambda = lambda;
        const V3 x = v3(p.x + lambda * d.x, p.y + lambda * d.y, p.z + lambda * d.z);
        c = boxLike ? raw_box(x, col.lo, col.hi, CAP_R) : raw_capsule(x, col.lo, col.hi.x, 2 * CAP_R);
        dist = c.dist + ALLOWED_CCD_PEN;
        MV_CAST_ITERS(numIter + 1);
        if (++numIter > CAST_MAX_ITER) return false;
    }
    fraction = lambda;
    const float inv = 1.0f / c.d;
    V3 n = c.v * inv;
    if (OBB && col.kind >= 3) n = hex_to_world(fr, n);
    normal = n;
    return true;
}

// Can the cast against a box-like collider hit at all?  A hit needs the capsule surface within CAST_RADIUS - ALLOWED_CCD_PEN (< 0: 0.039 deep)
// of the collider somewhere on the path, i.e. the centre within CAP_R - 0.039 of the (grown) box: impossible when the path's bounding box,
// widened by the full CAP_R, misses the box.  Exact (it only ever skips casts that return false), and it keeps the correctly rounded sqrt and
// divide of a first iteration away from the many colliders that are nowhere near the agent.
template <bool OBB = false>
__device__ __forceinline__ bool cast_can_hit(const Col &col, V3 p, V3 d)
{
    if (col.kind == 2) return true;
    if (OBB && col.kind >= 3) { p = hex_to_local(col.kind - 3, p); d = hex_to_local(col.kind - 3, d); }
    const V3 q = v3(p.x + d.x, p.y + d.y, p.z + d.z);
    return fmin_sel(p.x, q.x) - CAP_R <= col.hi.x && fmax_sel(p.x, q.x) + CAP_R >= col.lo.x &&
           fmin_sel(p.y, q.y) - CAP_R <= col.hi.y && fmax_sel(p.y, q.y) + CAP_R >= col.lo.y &&
           fmin_sel(p.z, q.z) - CAP_R <= col.hi.z && fmax_sel(p.z, q.z) + CAP_R >= col.lo.z;
}

// closest accepted hit over all colliders of the wave; ties resolved towards the lowest slot
template <int NC, bool OBB = false>
__device__ __forceinline__ bool sweep(const Col (&col)[NC], V3 from, V3 to, V3 up, float minSlopeDot, float &fraction,
                                      V3 &normal)
{
    const V3 d = to - from;
    unsigned fb[NC];   // hit fraction bits of this lane's collider k (a fraction is in [0, 1): its bits order like the value), ~0: no accepted hit
    V3 nn[NC];
#pragma unroll
    for (int k = 0; k < NC; ++k) { nn[k] = v3(0, 0, 0); fb[k] = ~0u; }
#ifdef MV_TICK_TIMING
    int its[NC];
#pragma unroll
    for (int k = 0; k < NC; ++k) its[k] = 0;
#define MV_ITS_ARG , &its[k]
#else
#define MV_ITS_ARG
#endif
#pragma unroll
    for (int k = 0; k < NC; ++k) {
        if (col[k].kind != 0 && cast_can_hit<OBB>(col[k], from, d)) {
            float f; V3 n;
            if (convex_cast<OBB>(col[k], from, d, f, n MV_ITS_ARG) && (len2(n) > 0.0001f) && (f < 1.0f) && !(dot(up, n) < minSlopeDot)) {
                fb[k] = __float_as_uint(f);
                nn[k] = n;
            }
        }
    }
#undef MV_ITS_ARG
#ifdef MV_TICK_TIMING
    {   // s_cast_dbg: [0] sweeps, [1] sum over sweeps of the per-slot longest casts (what the wave iterates today), [2] sum over sweeps of the longest
        // per-lane total (what it would iterate with the lanes' casts queued), [3] the longest single cast, [4] casts started
        unsigned serial = 0, lanesum = 0, started = 0;
#pragma unroll
        for (int k = 0; k < NC; ++k) { serial += ~wave_min_u32(~(unsigned)its[k]); lanesum += (unsigned)its[k]; started += (unsigned)__popcll(__ballot(its[k] > 0)); }
        const unsigned queued = ~wave_min_u32(~lanesum);
        unsigned longest = 0;
#pragma unroll
        for (int k = 0; k < NC; ++k) longest = max(longest, ~wave_min_u32(~(unsigned)its[k]));
        if (lane_id() == 0) { s_cast_dbg[0] += 1; s_cast_dbg[1] += serial; s_cast_dbg[2] += queued; s_cast_dbg[3] = max(s_cast_dbg[3],
            longest); s_cast_dbg[4] += started; }
    }
#endif
    // the closest hit of the wave; ties resolve towards the lowest slot (slot = lane + 64 k: the serial order of the reference's callback):
    // smallest fraction first (one DPP reduction), then the first k that holds it, then the first lane of that k (ballots)
    unsigned mine = fb[0];
#pragma unroll
    for (int k = 1; k < NC; ++k) mine = min(mine, fb[k]);
    const unsigned fmin = wave_min_u32(mine);
    if (fmin == ~0u) { fraction = 1.0f; return false; }
    int which = 0, src = 0;
#pragma unroll
    for (int k = NC - 1; k >= 0; --k) {
        const unsigned long long m = __ballot(fb[k] == fmin);
        if (m != 0ull) { which = k; src = __ffsll((long long)m) - 1; }
    }
    V3 cand = nn[0];
#pragma unroll
    for (int k = 1; k < NC; ++k)
        if (which == k) cand = nn[k];
    normal = v3(bcast_f(cand.x, src), bcast_f(cand.y, src), bcast_f(cand.z, src));
    fraction = __uint_as_float(fmin);
    return true;
}

// push out of the first (lowest slot) collider that is penetrated deeper than MAX_PEN_DEPTH.  Every lane only needs the signed distance of
// its colliders; the contact normal is normalised for the winner alone (one divide per call; same values as closest().n)
template <int NC, bool OBB = false>
__device__ __forceinline__ bool recover_from_penetration(const Col (&col)[NC], V3 &pos)
{
    Raw c[NC];
    bool pen[NC];
#pragma unroll
    for (int k = 0; k < NC; ++k) {
        pen[k] = false;
        c[k].dist = 0.0f; c[k].d = 1.0f; c[k].v = v3(0, 0, 0);
        if (col[k].kind == 1) c[k] = raw_box(pos, col[k].lo, col[k].hi, CAP_R);
        else if (OBB && col[k].kind >= 3) c[k] = raw_box(hex_to_local(col[k].kind - 3, pos), col[k].lo, col[k].hi, CAP_R);
        else if (col[k].kind == 2) c[k] = raw_capsule(pos, col[k].lo, col[k].hi.x, 2 * CAP_R);
        if (col[k].kind != 0) pen[k] = c[k].dist < -MAX_PEN_DEPTH;
    }
    // lowest slot first: slot = lane + 64 * k
    int which = -1, src = 0;
#pragma unroll
    for (int k = NC - 1; k >= 0; --k) {
        const unsigned long long m = __ballot(pen[k]);
        if (m != 0ull) { which = k; src = __ffsll((long long)m) - 1; }
    }
    if (which < 0) return false;
    Raw w = c[0];
    int kind = col[0].kind;
#pragma unroll
    for (int k = 1; k < NC; ++k)
        if (which == k) { w = c[k]; kind = col[k].kind; }
    const float dist = bcast_f(w.dist, src), d = bcast_f(w.d, src);
    const V3 v = v3(bcast_f(w.v.x, src), bcast_f(w.v.y, src), bcast_f(w.v.z, src));
    kind = bcast_i(kind, src);
    const float inv = 1.0f / d;
    V3 n = v * inv;
    if (OBB && kind >= 3) n = hex_to_world(kind - 3, n);
    const float push = -dist;
    pos = v3(pos.x + n.x * push, pos.y + n.y * push, pos.z + n.z * push);
    return true;
}

// "while (recover()) { if (++n > 4) break; }" == at most five calls.  Written as straight-line
// predicated code: no wave-level op (ballot/shuffle) sits inside a loop with a data-dependent exit.
template <int NC, bool OBB = false>
__device__ __forceinline__ void recover_up_to_5(const Col (&col)[NC], V3 &pos)
{
    bool more = true;
#pragma unroll 1
    for (int it = 0; it < 5; ++it)   // (a counted loop, not five copies: the tick's code is ~100 KB as it is; `more` is wave-uniform)
        if (more) more = recover_from_penetration<NC, OBB>(col, pos);
}
// the same, telling `seen` every position on the way (player_step's reach: two opposite pushes can leave the capsule where it started -- the
// positions in between were still occupied)
template <int NC, bool OBB, class Seen>
__device__ __forceinline__ void recover_up_to_5(const Col (&col)[NC], V3 &pos, Seen seen)
{
    bool more = true;
#pragma unroll 1
    for (int it = 0; it < 5; ++it)
        if (more) { more = recover_from_penetration<NC, OBB>(col, pos); seen(pos); }
}

__device__ __forceinline__ bool on_ground(const AgentState &a) { return (fabsf(a.vvel) < SIMD_EPS) && (fabsf(a.voffset) < SIMD_EPS); }

__device__ __forceinline__ void set_acceleration(AgentState &a, V3 acc, float dt)
{
    const bool isOnGround = on_ground(a);
    const float accMag = sqrtf(len2(acc));
    const float currMax = isOnGround ? MAX_ACCEL : MAX_AIR_ACCEL;
    if (!(len2(acc) < SIMD_EPS * SIMD_EPS)) {
        const float k = currMax / accMag;
        acc = acc * k;
    }
    if (isOnGround) {
        a.hvx += acc.x * dt;
        a.hvz += acc.z * dt;
        const float speed = sqrtf(a.hvx * a.hvx + a.hvz * a.hvz);
        if (speed > MAX_H_SPEED) {
            const float dv = EXCEED_DECEL * dt;
            const float k = (speed - dv > MAX_H_SPEED) ? (speed - dv) / speed : MAX_H_SPEED / speed;
            a.hvx *= k; a.hvz *= k;
        }
    } else {
        const float speed = sqrtf(a.hvx * a.hvx + a.hvz * a.hvz);
        const float nx = a.hvx + acc.x * dt, nz = a.hvz + acc.z * dt;
        const float newSpeed = sqrtf(nx * nx + nz * nz);
        if (newSpeed <= MAX_AIR_SPEED || newSpeed < speed) { a.hvx = nx; a.hvz = nz; }
    }
}

__device__ __forceinline__ V3 lerp3(V3 a, V3 b, float rt)
{
    const float s = 1.0f - rt;
    return v3(s * a.x + rt * b.x, s * a.y + rt * b.y, s * a.z + rt * b.z);
}

// preStep + playerStep of the controller for one agent
// reach2 (optional): the largest squared HORIZONTAL distance from the start position of any point the controller sweeps between or stands on this
// tick -- every sweep runs between two of the points recorded here, so the capsule's axis never leaves that circle (mv_tick_tower.h: which agents of
// an env can run their controllers at the same time).  Recording it changes nothing else.
template <int NC, bool OBB = false>
__device__ __forceinline__ void player_step(AgentState &a, const Col (&col)[NC], float dt, float *reach2 = nullptr)
{
    V3 cur = v3(a.pos[0], a.pos[1], a.pos[2]);
    V3 target = cur;
    const V3 original = cur;
    const V3 UP = v3(0, 1, 0);
    float r2 = 0.0f;
    auto reached = [&](V3 p) {
        if (reach2) { const float dx = p.x - original.x, dz = p.z - original.z; r2 = fmax_sel(r2, dx * dx + dz * dz); }
    };

    const bool wasOnGround = on_ground(a);
    a.vvel -= GRAVITY * dt;
    if (a.vvel > 0.0f && a.vvel > a.jump_speed) a.vvel = a.jump_speed;
    if (a.vvel < 0.0f && fabsf(a.vvel) > fabsf(FALL_SPEED)) a.vvel = -fabsf(FALL_SPEED);
    a.voffset = a.vvel * dt;

    {   // stepUp
        const float stepHeight = (a.vvel < 0.0f) ? STEP_HEIGHT : 0.0f;
        const V3 start = cur;
        target = v3(cur.x, cur.y + stepHeight + (a.voffset > 0.0f ? a.voffset : 0.0f), cur.z);
        cur = target;
        float f; V3 n;
        if (sweep<NC, OBB>(col, start, target, v3(0, -1, 0), MAX_SLOPE_COS, f, n)) {
            if (dot(n, UP) > 0.0f) {
                a.step_offset = stepHeight * f;
                cur = lerp3(cur, target, f);
            }
            recover_up_to_5<NC, OBB>(col, cur, reached);
            target = cur;
            if (a.voffset > 0) { a.voffset = 0.0f; a.vvel = 0.0f; a.step_offset = STEP_HEIGHT; }
        } else {
            a.step_offset = stepHeight;
            cur = target;
        }
    }
    reached(cur);

    {   // stepForwardAndStrafe
        const V3 hv = v3(a.hvx, 0.0f, a.hvz);
        target = v3(cur.x + hv.x * dt, cur.y + hv.y * dt, cur.z + hv.z * dt);
        reached(target);
        bool active = true;
#pragma unroll 1
        for (int it = 0; it < 10; ++it) {   // "int maxIter = 10; while (maxIter-- > 0)" with breaks -> flag
            if (active) {
                const V3 before = target;
                const V3 negDir = cur - target;
                float f = 1.0f; V3 n = v3(0, 0, 0);
                bool hit = false;
                if (!(cur.x == target.x && cur.y == target.y && cur.z == target.z)) hit = sweep<NC, OBB>(col, cur, target, negDir, 0.0f, f, n);
                if (!hit) active = false;
                else {
                    V3 dir = target - cur;
                    const float movLen = sqrtf(len2(dir));
                    if (movLen > SIMD_EPS) {
                        dir = dir * (1.0f / movLen);
                        const float mag = dot(dir, n);
                        const V3 par = n * mag;
                        const V3 perp = dir - par;
                        target = cur;
                        target = target + perp * movLen;
                        target = target + par * (movLen * f);
                    }
                    V3 cd = target - cur;
                    const float dist2 = len2(cd);
                    if (dist2 > 0.0001f) {
                        cd = cd * (1.0f / sqrtf(dist2));
                        if (dot(cd, hv) <= 0.0f) { target = cur; active = false; }
                    } else { target = cur; active = false; }
                    // An iteration is a pure function of `target` (cur and the colliders do not change in this loop): once it maps the target
                    // onto itself, bit for bit, the remaining iterations would all do the same.  An agent pushing into a wall or a corner gets
                    // there after two or three iterations and would otherwise sweep ten times (2 % of the ticks, i.e. some env of every launch,
                    // and the launch lasts as long as its slowest tick).  The CPU restatement runs the plain loop; the parity tests compare.
                    if (__float_as_uint(target.x) == __float_as_uint(before.x) && __float_as_uint(target.y) == __float_as_uint(before.y) &&
                        __float_as_uint(target.z) == __float_as_uint(before.z))
                        active = false;
                    reached(target);
                }
            }
        }
        cur = target;
    }

    {   // stepDown
        float downVel = (a.vvel < 0.0f) ? -a.vvel : 0.0f;
        if (downVel > 0.0f && downVel > FALL_SPEED && (wasOnGround || !a.was_jumping)) downVel = FALL_SPEED;
        target = v3(target.x, target.y - (a.step_offset + downVel * dt), target.z);
        float f; V3 n;
        if (sweep<NC, OBB>(col, cur, target, UP, MAX_SLOPE_COS, f, n)) {
            cur = lerp3(cur, target, f);
            a.vvel = 0.0f; a.voffset = 0.0f; a.was_jumping = 0;
        } else cur = target;
    }

    reached(cur);
    a.hvx = (cur.x - original.x) / dt;
    a.hvz = (cur.z - original.z) / dt;

    recover_up_to_5<NC, OBB>(col, cur, reached);
    reached(cur);
    if (reach2) *reach2 = r2;
    a.pos[0] = cur.x; a.pos[1] = cur.y; a.pos[2] = cur.z;

    const float speed = sqrtf(a.hvx * a.hvx + a.hvz * a.hvz);
    if (on_ground(a)) {
        if (speed - NORMAL_DECEL * dt < 0) { a.hvx = 0.0f; a.hvz = 0.0f; }
        else { const float k = (speed - NORMAL_DECEL * dt) / speed; a.hvx *= k; a.hvz *= k; }
    }
}


// Env::step's per-agent action block (env.cpp:89-122) + DefaultKinematicAgent look/accelerate/jump (agent.cpp:100-161)
__device__ __forceinline__ void apply_actions(AgentState &a, int ac, float dt, float lookLimit)
{
    V3 fwd = v3(a.m20, 0.0f, -a.m22);
    fwd = fwd * (1.0f / sqrtf(len2(fwd)));
    V3 left = v3(-a.m00, 0.0f, a.m02);
    left = left * (1.0f / sqrtf(len2(left)));
    V3 acc = v3(0, 0, 0);
    if (ac & ACT_FORWARD) acc = acc + fwd;
    else if (ac & ACT_BACKWARD) acc = acc - fwd;
    if (ac & ACT_LEFT) acc = acc + left;
    else if (ac & ACT_RIGHT) acc = acc - left;

    if (ac & (ACT_LOOK_LEFT | ACT_LOOK_RIGHT)) {
        float c, s;
        yaw_matrix(ROTATE_RAD * dt, c, s);
        if (!(ac & ACT_LOOK_LEFT)) s = -s;
        const float n00 = a.m00 * c + a.m02 * (-s), n02 = a.m00 * s + a.m02 * c;
        const float n20 = a.m20 * c + a.m22 * (-s), n22 = a.m20 * s + a.m22 * c;
        a.m00 = n00; a.m02 = n02; a.m20 = n20; a.m22 = n22;
    }
    if (ac & ACT_LOOK_UP) {
        a.pitch += ROTATE_X_RAD * dt;
        a.pitch = fmin_sel(lookLimit, a.pitch);
    } else if (ac & ACT_LOOK_DOWN) {
        a.pitch -= ROTATE_X_RAD * dt * 1.1f;
        a.pitch = fmax_sel(-lookLimit, a.pitch);
    }

    set_acceleration(a, acc, dt);

    if ((ac & ACT_JUMP) && on_ground(a)) {
        a.jump_speed = sqrtf(6.2f * 6.2f);
        a.vvel = a.jump_speed;
        a.was_jumping = 1;
    }

}

struct Cam {
    V3 eye;
    float c[3][3];
};

__device__ __forceinline__ Cam camera_of(const AgentState &a)
{
    Cam cam;
    cam.eye = v3(a.pos[0], (a.pos[1] + 0.05f) + 0.41f, a.pos[2]);
    float sp, cp;
    sincos_poly(a.pitch, sp, cp);
    cam.c[0][0] = a.m00; cam.c[0][1] = a.m02 * sp; cam.c[0][2] = a.m02 * cp;
    cam.c[1][0] = 0.0f;  cam.c[1][1] = cp;         cam.c[1][2] = -sp;
    cam.c[2][0] = a.m20; cam.c[2][1] = a.m22 * sp; cam.c[2][2] = a.m22 * cp;
    return cam;
}

__device__ __forceinline__ V3 cam_to_world(const Cam &cam, V3 v)
{
    return v3((cam.c[0][0] * v.x + cam.c[0][1] * v.y) + cam.c[0][2] * v.z + cam.eye.x,
              (cam.c[1][0] * v.x + cam.c[1][1] * v.y) + cam.c[1][2] * v.z + cam.eye.y,
              (cam.c[2][0] * v.x + cam.c[2][1] * v.y) + cam.c[2][2] * v.z + cam.eye.z);
}


}  // namespace
}  // namespace mv
