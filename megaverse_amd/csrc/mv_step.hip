// megaverse_amd/csrc/mv_step.hip -- one simulation tick for every env: actions -> kinematic
// character physics -> TowerBuilding scenario logic -> timers/done.
//
// Replaces, per env (reference paths relative to src/libs):
//   Env::step                                   env/src/env.cpp:83-152
//   DefaultKinematicAgent look/accelerate/jump  env/src/agent.cpp:100-161
//   KinematicCharacterController::setAcceleration / preStep / playerStep / stepUp /
//     stepForwardAndStrafe / stepDown / recoverFromPenetration / updateTargetPositionBasedOnCollision
//                                               env/src/kinematic_character_controller.cpp:156-442,519-602,753-792
//   Bullet 2.89 ghost convexSweepTest + contact manifolds [third party, not vendored]: restated as
//     conservative advancement on exact closest points (DESIGN.md "physics model")
//   ObjectStackingComponent::step/onInteractAction  scenarios/include/scenarios/component_object_stacking.hpp:45-168
//   FallDetectionComponent::step                scenarios/include/scenarios/component_fall_detection.hpp:33-55
//   TowerBuildingScenario::step + callbacks     scenarios/src/scenario_tower_building.cpp:179-261
//   Scenario::rewardAgent/rewardTeam            env/include/env/scenario.hpp:259-298
//   done bookkeeping of VectorEnv::step         env/src/vector_env.cpp:93-105 (the reset itself: mv_reset.hip)
//
// Mapping: ONE WAVEFRONT PER ENV.  The env's colliders (<=16 layout slabs, <=80 movable boxes,
// <=8 agent capsules) live in VGPRs, two per lane.  A sweep is "every lane casts against its two
// colliders, then a 64-bit (fraction, slot) wave-min picks the winner"; depenetration and object
// lookups are ballot + find-first-lane.  Agents inside an env are order dependent (they collide
// with each other and share voxels) so they run one after another with wave-uniform state.
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
    int kind;   // 0 none, 1 box (bounds already grown by CAP_HH in y), 2 vertical capsule
    V3 lo, hi;
};

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

__device__ __forceinline__ Closest closest(const Col &col, V3 p)
{
    if (col.kind == 1) return closest_box(p, col.lo, col.hi, CAP_R);
    return closest_capsule(p, col.lo, col.hi.x, 2 * CAP_R);
}

// conservative advancement of the capsule along d against one collider
__device__ __forceinline__ bool convex_cast(const Col &col, V3 p, V3 d, float &fraction, V3 &normal)
{
    float lambda = 0.0f, lastLambda = 0.0f;
    int numIter = 0;
    Closest c = closest(col, p);
    float dist = c.dist + ALLOWED_CCD_PEN;
    V3 n = c.n;
    float proj = -dot(d, n);
    if (proj <= SIMD_EPS) return false;
    while (dist > CAST_RADIUS) {
        proj = -dot(d, n);
        if (proj <= SIMD_EPS) return false;
        lambda = lambda + dist / proj;
        if (lambda > 1.0f) return false;
        if (lambda < 0.0f) return false;
        if (lambda <= lastLambda) return false;
        lastLambda = lambda;
        const V3 x = v3(p.x + lambda * d.x, p.y + lambda * d.y, p.z + lambda * d.z);
        c = closest(col, x);
        dist = c.dist + ALLOWED_CCD_PEN;
        n = c.n;
        if (++numIter > CAST_MAX_ITER) return false;
    }
    fraction = lambda;
    normal = n;
    return true;
}

// closest accepted hit over all colliders of the wave; ties resolved towards the lowest slot
__device__ __forceinline__ bool sweep(const Col (&col)[2], V3 from, V3 to, V3 up, float minSlopeDot, float &fraction,
                                      V3 &normal)
{
    const int lane = lane_id();
    const V3 d = to - from;
    unsigned long long key = ~0ull;
    V3 nn[2] = {v3(0, 0, 0), v3(0, 0, 0)};
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        if (col[k].kind != 0) {
            float f; V3 n;
            if (convex_cast(col[k], from, d, f, n) && (len2(n) > 0.0001f) && (f < 1.0f) && !(dot(up, n) < minSlopeDot)) {
                const unsigned long long kk = ((unsigned long long)__float_as_uint(f) << 32) | (unsigned)(lane + 64 * k);
                key = kk < key ? kk : key;
                nn[k] = n;
            }
        }
    }
    const unsigned long long win = wave_min_u64(key);
    if (win == ~0ull) { fraction = 1.0f; return false; }
    const int slot = (int)(win & 0xffffffffu);
    const int src = slot & 63;
    const V3 cand = (slot >> 6) ? nn[1] : nn[0];
    normal = v3(bcast_f(cand.x, src), bcast_f(cand.y, src), bcast_f(cand.z, src));
    fraction = __uint_as_float((unsigned)(win >> 32));
    return true;
}

// push out of the first (lowest slot) collider that is penetrated deeper than MAX_PEN_DEPTH
__device__ __forceinline__ bool recover_from_penetration(const Col (&col)[2], V3 &pos)
{
    Closest c[2];
    bool pen[2];
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        pen[k] = false;
        c[k].dist = 0.0f; c[k].n = v3(0, 0, 0);
        if (col[k].kind != 0) {
            c[k] = closest(col[k], pos);
            pen[k] = c[k].dist < -MAX_PEN_DEPTH;
        }
    }
    const unsigned long long m0 = __ballot(pen[0]), m1 = __ballot(pen[1]);
    if ((m0 | m1) == 0ull) return false;
    const bool second = (m0 == 0ull);
    const int src = __ffsll((long long)(second ? m1 : m0)) - 1;
    const Closest w = second ? c[1] : c[0];
    const float dist = bcast_f(w.dist, src);
    const V3 n = v3(bcast_f(w.n.x, src), bcast_f(w.n.y, src), bcast_f(w.n.z, src));
    const float push = -dist;
    pos = v3(pos.x + n.x * push, pos.y + n.y * push, pos.z + n.z * push);
    return true;
}

// "while (recover()) { if (++n > 4) break; }" == at most five calls.  Written as straight-line
// predicated code: no wave-level op (ballot/shuffle) sits inside a loop with a data-dependent exit.
__device__ __forceinline__ void recover_up_to_5(const Col (&col)[2], V3 &pos)
{
    bool more = true;
#pragma unroll
    for (int it = 0; it < 5; ++it)
        if (more) more = recover_from_penetration(col, pos);
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
__device__ __forceinline__ void player_step(AgentState &a, const Col (&col)[2], float dt)
{
    V3 cur = v3(a.pos[0], a.pos[1], a.pos[2]);
    V3 target = cur;
    const V3 original = cur;
    const V3 UP = v3(0, 1, 0);

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
        if (sweep(col, start, target, v3(0, -1, 0), MAX_SLOPE_COS, f, n)) {
            if (dot(n, UP) > 0.0f) {
                a.step_offset = stepHeight * f;
                cur = lerp3(cur, target, f);
            }
            recover_up_to_5(col, cur);
            target = cur;
            if (a.voffset > 0) { a.voffset = 0.0f; a.vvel = 0.0f; a.step_offset = STEP_HEIGHT; }
        } else {
            a.step_offset = stepHeight;
            cur = target;
        }
    }

    {   // stepForwardAndStrafe
        const V3 hv = v3(a.hvx, 0.0f, a.hvz);
        target = v3(cur.x + hv.x * dt, cur.y + hv.y * dt, cur.z + hv.z * dt);
        bool active = true;
#pragma unroll 1
        for (int it = 0; it < 10; ++it) {   // "int maxIter = 10; while (maxIter-- > 0)" with breaks -> flag
            if (active) {
                const V3 negDir = cur - target;
                float f = 1.0f; V3 n = v3(0, 0, 0);
                bool hit = false;
                if (!(cur.x == target.x && cur.y == target.y && cur.z == target.z)) hit = sweep(col, cur, target, negDir, 0.0f, f, n);
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
        if (sweep(col, cur, target, UP, MAX_SLOPE_COS, f, n)) {
            cur = lerp3(cur, target, f);
            a.vvel = 0.0f; a.voffset = 0.0f; a.was_jumping = 0;
        } else cur = target;
    }

    a.hvx = (cur.x - original.x) / dt;
    a.hvz = (cur.z - original.z) / dt;

    recover_up_to_5(col, cur);
    a.pos[0] = cur.x; a.pos[1] = cur.y; a.pos[2] = cur.z;

    const float speed = sqrtf(a.hvx * a.hvx + a.hvz * a.hvz);
    if (on_ground(a)) {
        if (speed - NORMAL_DECEL * dt < 0) { a.hvx = 0.0f; a.hvz = 0.0f; }
        else { const float k = (speed - NORMAL_DECEL * dt) / speed; a.hvx *= k; a.hvz *= k; }
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

__device__ __forceinline__ float building_reward_coeff(int h)
{
    float res = float(h) * 0.05f;
    const float p = 0.05f * __uint_as_float((unsigned)(127 + h) << 23);
    res += fmin_sel(p, 20.0f);
    return res;
}

// the wave's view of the movable boxes: lane l owns object l-16 (l>=16) and object 48+l (l<32)
struct ObjRegs {
    int x[2], y[2], z[2], state[2];
    bool valid[2];
};

__device__ __forceinline__ int object_at(const ObjRegs &o, int x, int y, int z)
{
    const int lane = lane_id();
    const bool h0 = o.valid[0] && o.state[0] == 0 && o.x[0] == x && o.y[0] == y && o.z[0] == z;
    const bool h1 = o.valid[1] && o.state[1] == 0 && o.x[1] == x && o.y[1] == y && o.z[1] == z;
    const unsigned long long m0 = __ballot(h0), m1 = __ballot(h1);
    (void)lane;
    if (m0) return (__ffsll((long long)m0) - 1) - 16;
    if (m1) return 48 + (__ffsll((long long)m1) - 1);
    return -1;
}

// placed objects of column (x,z) as a bit set over y: bit (y+32), y in [-32,31]
__device__ __forceinline__ unsigned long long column_objects(const ObjRegs &o, int x, int z)
{
    unsigned long long m = 0ull;
#pragma unroll
    for (int k = 0; k < 2; ++k)
        if (o.valid[k] && o.state[k] == 0 && o.x[k] == x && o.z[k] == z && o.y[k] >= -32 && o.y[k] < 32) m |= 1ull << (o.y[k] + 32);
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
        const unsigned lo = (unsigned)__shfl_xor((int)(unsigned)m, off, 64);
        const unsigned hi = (unsigned)__shfl_xor((int)(unsigned)(m >> 32), off, 64);
        m |= ((unsigned long long)hi << 32) | lo;
    }
    return m;
}

// The fields of EnvHeader a tick reads or writes, as scalars.  Copying the whole 128-byte record made
// hipcc keep its array members in an LDS "promoted alloca", which in turn made every wave read the
// workgroup size from the AQL dispatch packet in host memory: 16-20 us of a 35 us kernel.
struct Hdr {
    int num_objects, num_boxes, num_frames, done, highest_tower;
    int bz0, bz1, bz2, bz3;
    float episode_sec, episode_len, bz_reward, bar_half_width, p_vertical_look_limit;
};

__device__ __forceinline__ bool in_zone(const Hdr &h, int x, int z) { return x >= h.bz0 && x < h.bz1 && z >= h.bz2 && z < h.bz3; }

// sum over objects in index order (float addition order is part of the contract)
__device__ __forceinline__ float tower_reward(const Hdr &h, const ObjRegs &o)
{
    float term[2];
#pragma unroll
    for (int k = 0; k < 2; ++k)
        term[k] = (o.valid[k] && o.state[k] == 0 && in_zone(h, o.x[k], o.z[k])) ? building_reward_coeff(o.y[k]) : 0.0f;
    float r = 0.0f;
    for (int i = 0; i < h.num_objects; ++i) {
        const float t = (i < 48) ? bcast_f(term[0], 16 + i) : bcast_f(term[1], i - 48);
        r += t;
    }
    return r;
}

// Scenario::rewardAgent / rewardTeam (scenario.hpp:259-298); fully unrolled so ag[] stays in VGPRs
template <int A_MAX>
__device__ __forceinline__ void reward_agent(AgentState (&ag)[A_MAX], int key, int idx, float mult)
{
#pragma unroll
    for (int i = 0; i < A_MAX; ++i)
        if (i == idx) ag[i].last_reward += ag[i].shaping[key] * mult;
}

template <int A_MAX>
__device__ __forceinline__ void reward_team(AgentState (&ag)[A_MAX], int A, int key, int idx, float mult)
{
#pragma unroll
    for (int i = 0; i < A_MAX; ++i)
        if (i == idx) ag[i].last_reward += ag[i].shaping[key] * (mult * (1 - ag[i].shaping[0]));
#pragma unroll
    for (int i = 0; i < A_MAX; ++i)
        if (i < A) ag[i].last_reward += ag[i].shaping[key] * ag[i].shaping[0] * mult / float(A);
}

__device__ __forceinline__ void voxel_of(V3 p, int out[3])
{
    out[0] = (int)floorf(p.x); out[1] = (int)floorf(p.y); out[2] = (int)floorf(p.z);
}

__device__ __forceinline__ bool in_chunk(int x, int y, int z) { return x >= 0 && x < CX && y >= 0 && y < CY && z >= 0 && z < CZ; }

}  // namespace

template <int A_MAX>
__global__ __launch_bounds__(64) void step_kernel(GymView gv)
{
    const int env = blockIdx.x;
    const int lane = lane_id();
    if (env >= gv.num_envs) return;
    const int A = gv.num_agents;

    const EnvHeader *gh = gv.hdr + env;
    Hdr h;
    h.num_objects = gh->num_objects; h.num_boxes = gh->num_boxes; h.num_frames = gh->num_frames; h.done = gh->done;
    h.highest_tower = gh->highest_tower;
    h.bz0 = gh->bz[0]; h.bz1 = gh->bz[1]; h.bz2 = gh->bz[2]; h.bz3 = gh->bz[3];
    h.episode_sec = gh->episode_sec; h.episode_len = gh->episode_len; h.bz_reward = gh->bz_reward;
    h.bar_half_width = gh->bar_half_width; h.p_vertical_look_limit = gh->p_vertical_look_limit;
    uint8_t *chunk = gv.chunk + (size_t)env * CHUNK_BYTES;
    auto vox = [&](int x, int y, int z) -> unsigned { return in_chunk(x, y, z) ? (unsigned)chunk[(y * CZ + z) * CX + x] : 0u; };

    // ---- wave-resident scene: two colliders + two movable boxes per lane
    Col col[2];
    ObjRegs ob;
    col[0].kind = 0; col[1].kind = 0;
    col[0].lo = col[0].hi = col[1].lo = col[1].hi = v3(0, 0, 0);
    const MovableObject *gobj = gv.objects + (size_t)env * MAX_OBJECTS;
    const int oi[2] = {lane - 16, lane < 32 ? 48 + lane : -1};
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        ob.valid[k] = oi[k] >= 0 && oi[k] < h.num_objects;
        ob.x[k] = ob.y[k] = ob.z[k] = 0; ob.state[k] = 0;
        if (ob.valid[k]) {
            const MovableObject o = gobj[oi[k]];
            ob.x[k] = o.x; ob.y[k] = o.y; ob.z[k] = o.z; ob.state[k] = o.state;
        }
    }
    auto object_collider = [&](int k) {
        if (ob.valid[k] && ob.state[k] == 0) {
            const float cx = float(ob.x[k]) + 0.5f, cy = float(ob.y[k]) + 0.5f + OBJ_COLL_YOFF, cz = float(ob.z[k]) + 0.5f;
            col[k].kind = 1;
            col[k].lo = v3(cx - OBJ_COLL_HALF, (cy - OBJ_COLL_HALF) - CAP_HH, cz - OBJ_COLL_HALF);
            col[k].hi = v3(cx + OBJ_COLL_HALF, (cy + OBJ_COLL_HALF) + CAP_HH, cz + OBJ_COLL_HALF);
        } else col[k].kind = 0;
    };
    if (lane < MAX_BOXES) {
        if (lane < h.num_boxes) {
            const LayoutBox b = gv.boxes[(size_t)env * MAX_BOXES + lane];
            if (b.type & VX_SOLID) {
                col[0].kind = 1;
                col[0].lo = v3(float(b.min[0]), float(b.min[1]) - CAP_HH, float(b.min[2]));
                col[0].hi = v3(float(b.max[0]), float(b.max[1]) + CAP_HH, float(b.max[2]));
            }
        }
    } else object_collider(0);
    if (lane < 32) object_collider(1);

    // ---- agents: wave-uniform copies
    AgentState ag[A_MAX];
    int act[A_MAX];
#pragma unroll
    for (int i = 0; i < A_MAX; ++i)
        if (i < A) {
            ag[i] = gv.agents[(size_t)env * A + i];
            act[i] = gv.actions[(size_t)env * A + i];
            ag[i].last_reward = 0.0f;   // env.cpp:85
        }

    const float dt = DT;

    // ---- actions -> intents (env.cpp:89-122)
#pragma unroll
    for (int i = 0; i < A_MAX; ++i)
        if (i < A) {
            AgentState &a = ag[i];
            const int ac = act[i];
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
                a.pitch = fmin_sel(h.p_vertical_look_limit, a.pitch);
            } else if (ac & ACT_LOOK_DOWN) {
                a.pitch -= ROTATE_X_RAD * dt * 1.1f;
                a.pitch = fmax_sel(-h.p_vertical_look_limit, a.pitch);
            }

            set_acceleration(a, acc, dt);

            if ((ac & ACT_JUMP) && on_ground(a)) {
                a.jump_speed = sqrtf(6.2f * 6.2f);
                a.vvel = a.jump_speed;
                a.was_jumping = 1;
            }
        }

    // ---- physics, agent by agent (controllers run in addAction order, env.cpp:126)
#pragma unroll
    for (int i = 0; i < A_MAX; ++i)
        if (i < A) {
            if (A_MAX > 1 && lane >= 32 && lane < 32 + MAX_AGENTS) {   // capsule colliders of the other agents
                const int j = lane - 32;
                col[1].kind = 0;
#pragma unroll
                for (int q = 0; q < A_MAX; ++q)
                    if (q == j && q < A && q != i) {
                        col[1].kind = 2;
                        col[1].lo = v3(ag[q].pos[0], ag[q].pos[1], ag[q].pos[2]);
                        col[1].hi = v3(2 * CAP_HH, 0.0f, 0.0f);
                    }
            }
            player_step(ag[i], col, dt);
        }

    // ---- scenario step: interact (component_object_stacking.hpp:45-168)
#pragma unroll
    for (int i = 0; i < A_MAX; ++i)
        if (i < A && (act[i] & ACT_INTERACT)) {
            AgentState &a = ag[i];
            const Cam cam = camera_of(a);
            if (a.carrying >= 0) {
                const V3 t = cam_to_world(cam, v3(0.0f, -0.44f + -0.3f, -1.0f));
                int vx[3];
                voxel_of(t, vx);
                bool collidesWithAgent = false;
#pragma unroll
                for (int j = 0; j < A_MAX; ++j)
                    if (j < A && j != i) {
                        int c[3];
                        voxel_of(v3(ag[j].pos[0], ag[j].pos[1] + 0.05f, ag[j].pos[2]), c);
                        if (c[0] == vx[0] && c[1] == vx[1] && c[2] == vx[2]) collidesWithAgent = true;
                    }
                const bool placeable = vx[0] >= 0 && vx[0] < CX && vx[2] >= 0 && vx[2] < CZ && vx[1] < CY;
                const unsigned long long colObj = column_objects(ob, vx[0], vx[2]);   // wave op, outside the loop
                auto has_obj = [&](int y) { return in_chunk(vx[0], y, vx[2]) && ((colObj >> (y + 32)) & 1ull); };
                const bool empty = !(vox(vx[0], vx[1], vx[2]) & VX_SOLID) && !has_obj(vx[1]);
                if (placeable && empty && !collidesWithAgent && in_zone(h, vx[0], vx[2])) {
                    for (;;) {
                        const int by = vx[1] - 1;
                        if (by < -30) break;
                        if ((vox(vx[0], by, vx[2]) & VX_SOLID) || has_obj(by)) break;
                        vx[1] = by;
                    }
                    const int oidx = a.carrying;
#pragma unroll
                    for (int k = 0; k < 2; ++k)
                        if (oi[k] == oidx) { ob.x[k] = vx[0]; ob.y[k] = vx[1]; ob.z[k] = vx[2]; ob.state[k] = 0; }
                    if (lane == 0 && in_chunk(vx[0], vx[1], vx[2])) chunk[(vx[1] * CZ + vx[2]) * CX + vx[0]] |= VX_OBJECT;
                    a.carrying = -1;
                    const float newReward = tower_reward(h, ob);
                    const float delta = newReward - h.bz_reward;
                    h.bz_reward = newReward;
                    reward_team(ag, A, 3, i, delta);
                    h.highest_tower = max(h.highest_tower, vx[1] - 1 + 1);
                }
            } else {
                const V3 pickup = cam_to_world(cam, v3(0.0f, -0.44f, -1.0f));
                int vx[3];
                voxel_of(pickup, vx);
                // maxPickupHeight == 1: try the voxel, then the one above; an object with another one
                // on top of it cannot be taken (component_object_stacking.hpp:131-167)
                const int o0 = object_at(ob, vx[0], vx[1], vx[2]);
                const int o1 = object_at(ob, vx[0], vx[1] + 1, vx[2]);
                const int o2 = object_at(ob, vx[0], vx[1] + 2, vx[2]);
                int oidx = -1, py = vx[1];
                if (o0 >= 0 && o1 < 0) { oidx = o0; py = vx[1]; }
                else if (o1 >= 0 && o2 < 0) { oidx = o1; py = vx[1] + 1; }
                if (oidx >= 0) {
#pragma unroll
                    for (int k = 0; k < 2; ++k)
                        if (oi[k] == oidx) ob.state[k] = 1 + i;
                    if (lane == 0 && in_chunk(vx[0], py, vx[2])) chunk[(py * CZ + vx[2]) * CX + vx[0]] &= (uint8_t)~VX_OBJECT;
                    a.carrying = oidx;
                    if (!a.picked_up) { reward_agent(ag, 1, i, 1); a.picked_up = 1; }
                }
            }
        }

    // ---- fall detection (component_fall_detection.hpp:33-55)
#pragma unroll
    for (int i = 0; i < A_MAX; ++i)
        if (i < A) {
            AgentState &a = ag[i];
            if (a.pos[1] + 0.05f < -20.0f) {
                int p[3] = {a.spawn[0], a.spawn[1], a.spawn[2]};
                while ((vox(p[0], p[1], p[2]) & VX_SOLID) && p[1] < 1000) ++p[1];
                a.pos[0] = float(p[0]) + 0.5f; a.pos[1] = float(p[1]) + 0.5f; a.pos[2] = float(p[2]) + 0.5f;
                a.m00 = 1.0f; a.m02 = 0.0f; a.m20 = 0.0f; a.m22 = 1.0f;
                a.hvx = 0.0f; a.hvz = 0.0f; a.vvel = 0.0f;
            }
        }

    // ---- building-zone visit shaping (scenario_tower_building.cpp:184-198)
#pragma unroll
    for (int i = 0; i < A_MAX; ++i)
        if (i < A) {
            AgentState &a = ag[i];
            if (a.carrying >= 0) {
                int vx[3];
                voxel_of(v3(a.pos[0], a.pos[1] + 0.05f, a.pos[2]), vx);
                if (in_zone(h, vx[0], vx[2]) && !a.visited_zone) {
                    reward_team(ag, A, 2, i, 1);
                    a.visited_zone = 1;
                }
            }
        }

    // ---- timers / done (env.cpp:133-151)
    h.episode_sec += dt;
    h.bar_half_width = fmax_sel(0.0f, (h.episode_len - h.episode_sec) / h.episode_len) * 0.24f;
    if (h.episode_sec >= h.episode_len) h.done = 1;
    ++h.num_frames;

    // ---- write back
    MovableObject *gobjw = gv.objects + (size_t)env * MAX_OBJECTS;
#pragma unroll
    for (int k = 0; k < 2; ++k)
        if (ob.valid[k]) {
            MovableObject o;
            o.x = (int8_t)ob.x[k]; o.y = (int8_t)ob.y[k]; o.z = (int8_t)ob.z[k]; o.state = (int8_t)ob.state[k];
            gobjw[oi[k]] = o;
        }
    if (lane == 0) {
        EnvHeader *wh = gv.hdr + env;
        wh->num_frames = h.num_frames; wh->done = h.done; wh->highest_tower = h.highest_tower;
        wh->episode_sec = h.episode_sec; wh->bz_reward = h.bz_reward; wh->bar_half_width = h.bar_half_width;
        gv.done[env] = (uint8_t)h.done;
    }
#pragma unroll
    for (int i = 0; i < A_MAX; ++i)
        if (i < A && lane == i) {
            ag[i].total_reward += ag[i].last_reward;
            // store the fields a tick can change (everything before `shaping`); writing the whole 128 B record
            // would force the unchanged tail to be carried in scratch memory for the whole kernel
            AgentState *dst = gv.agents + (size_t)env * A + i;
            const AgentState &a = ag[i];
            dst->pos[0] = a.pos[0]; dst->pos[1] = a.pos[1]; dst->pos[2] = a.pos[2];
            dst->m00 = a.m00; dst->m02 = a.m02; dst->m20 = a.m20; dst->m22 = a.m22; dst->pitch = a.pitch;
            dst->hvx = a.hvx; dst->hvz = a.hvz; dst->vvel = a.vvel; dst->voffset = a.voffset;
            dst->step_offset = a.step_offset; dst->jump_speed = a.jump_speed;
            dst->was_jumping = a.was_jumping; dst->carrying = a.carrying; dst->picked_up = a.picked_up; dst->visited_zone = a.visited_zone;
            dst->last_reward = a.last_reward; dst->total_reward = a.total_reward;
            gv.actions[(size_t)env * A + i] = 0;
            gv.rewards[(size_t)env * A + i] = ag[i].last_reward;   // zeroed by the reset kernel if done
            if (h.done) gv.true_objective[(size_t)env * A + i] = float(h.highest_tower);   // vector_env.cpp:97-98
        }
}

void launch_step(const GymView &gv, hipStream_t stream)
{
    const dim3 grid(gv.num_envs), block(64);
    if (gv.num_agents == 1) hipLaunchKernelGGL(step_kernel<1>, grid, block, 0, stream, gv);
    else if (gv.num_agents == 2) hipLaunchKernelGGL(step_kernel<2>, grid, block, 0, stream, gv);
    else if (gv.num_agents <= 4) hipLaunchKernelGGL(step_kernel<4>, grid, block, 0, stream, gv);
    else hipLaunchKernelGGL(step_kernel<8>, grid, block, 0, stream, gv);
}

}  // namespace mv
