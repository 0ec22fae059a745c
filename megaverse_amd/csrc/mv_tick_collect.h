// megaverse_amd/csrc/mv_tick_collect.h -- the collect tick as a device function (namespace mv::tick_collect): shared by the scenario's own step
// kernel (mv_step_collect.hip) and by the union step kernel that steps several gyms with one launch (mv_step_union.hip).
//
// (BASELINE.json configs[4] member).
//
// Replaces, per env (reference paths relative to src/libs):
//   Env::step                                   env/src/env.cpp:83-152            (shared pieces: mv_physics.h)
//   CollectScenario::step / agentFell           scenarios/src/scenario_collect.cpp:163-196,214-218
//   ObjectStackingComponent (default callbacks) scenarios/include/scenarios/component_object_stacking.hpp:45-168
//   FallDetectionComponent                      scenarios/include/scenarios/component_fall_detection.hpp:33-55
//   VectorEnv::step done bookkeeping + Env::reset of finished envs (env/src/vector_env.cpp:93-105)
//
// The landscape is a Perlin heightfield whose merged slabs number from a handful to several hundred, far more
// than fit "a few per lane".  Bullet's broadphase only ever hands the character controller the bodies whose
// AABB meets the swept capsule; the kernel does the same thing once per agent and tick: the wave streams the
// slab list (64 per pass), keeps the slabs that meet a conservative envelope of everything the capsule can do
// this tick, and compacts them -- in list order, which is all that collision tie-breaks depend on -- into an
// LDS candidate list of at most 128 entries (the envelope spans <= 3 x 8 x 3 cells and every candidate owns at
// least one of them, so 128 cannot overflow in a settled scene; an overflow is flagged, never ignored).
// Voxel questions (drop height, teleport cell, solid-or-not) are answered from the heightmap.
#pragma once
#include <hip/hip_runtime.h>

#include <algorithm>

#include "mv_boxlist.h"
#include "mv_actions.h"
#include "mv_agents.h"
#include "mv_frame.h"
#include "mv_math.h"
#include "mv_physics.h"
#include "mv_types.h"

namespace mv {
namespace tick_collect {



constexpr int NC = 2;
constexpr int MAX_CAND = 64 * NC;

// solid cells of heightfield column (x, z): y in [0, top]
__device__ __forceinline__ Bits128 column_solid_hm(const int8_t *hm, int x, int z)
{
    Bits128 m{0ull, 0ull};
    if (x >= 0 && x < HM_DIM && z >= 0 && z < HM_DIM) {
        const int top = hm[x * HM_DIM + z];
        if (top >= 0) set_range(m, 0, top + 1);
    }
    return m;
}

struct Envelope { float lo[3], hi[3]; };   // in capsule-CENTRE space, already widened by the capsule radius and slack

// Everything agent `a` can touch during player_step(): |horizontal move| <= |hv| dt (+ pushes out of penetration),
// up by at most the step height + a jump's vertical offset, down by the step offset + the fall distance.
__device__ __forceinline__ Envelope step_envelope(const AgentState &a, float dt)
{
    const float reach = CAP_R + 0.25f;                       // radius + ccd allowance / cast radius / depenetration slack
    const float hx = fabsf(a.hvx) * dt + 0.35f, hz = fabsf(a.hvz) * dt + 0.35f;
    float vv = a.vvel - GRAVITY * dt;
    vv = fminf(fmaxf(vv, -FALL_SPEED), fmaxf(a.jump_speed, 0.0f));
    const float up = STEP_HEIGHT + fmaxf(vv, 0.0f) * dt + 0.3f;
    const float down = STEP_HEIGHT + fmaxf(-vv, 0.0f) * dt + 0.3f;
    Envelope e;
    e.lo[0] = a.pos[0] - hx - reach; e.hi[0] = a.pos[0] + hx + reach;
    e.lo[2] = a.pos[2] - hz - reach; e.hi[2] = a.pos[2] + hz + reach;
    e.lo[1] = a.pos[1] - down - reach; e.hi[1] = a.pos[1] + up + reach;
    return e;
}
__device__ __forceinline__ bool meets(const Envelope &e, V3 lo, V3 hi)
{
    return lo.x <= e.hi[0] && hi.x >= e.lo[0] && lo.y <= e.hi[1] && hi.y >= e.lo[1] && lo.z <= e.hi[2] && hi.z >= e.lo[2];
}


// Episode swap-in: Env::reset of one env from its resident CollectBlob (called by the env's whole wavefront: by the stand-alone
// reset kernel for mv_reset, and by the tail of the step kernel for the auto-reset of VectorEnv::step, vector_env.cpp:93-105)
__device__ __forceinline__ void swap_in_episode(const GymView &gv, const CollectBlob *blobs, int *status, int env, int force_all)
{
    const int lane = lane_id();
    EnvHeader *gh = gv.hdr + env;
    const int consumed = gh->episodes_consumed;
    const CollectBlob *b = blobs + (size_t)env * gv.spares + consumed % gv.spares;   // ring slot of episode number consumed + 1
    if (b->seq != consumed + 1) {   // the host has not delivered the next episode: must never happen (mv_api.hip keeps one ahead)
        if (lane == 0) { gh->starved |= 1; atomicOr(&status[gv.num_envs + 1], (int)ST_STARVED); }
        return;
    }
    const int A = gv.num_agents;
    const int nb = b->num_boxes;
    const uint4 *src = reinterpret_cast<const uint4 *>(b->boxes);
    uint4 *dst = reinterpret_cast<uint4 *>(gv.boxes + (size_t)env * gv.box_stride);
    for (int i = lane; i < nb * 2; i += 64) dst[i] = src[i];
    const uint4 *hsrc = reinterpret_cast<const uint4 *>(b->heightmap);
    uint4 *hdst = reinterpret_cast<uint4 *>(gv.heightmap + (size_t)env * HM_BYTES);
    for (int i = lane; i < HM_BYTES / 16; i += 64) hdst[i] = hsrc[i];
    for (int i = lane; i < MAX_OBJECTS; i += 64) gv.objects[(size_t)env * MAX_OBJECTS + i] = b->objects[i];
    for (int i = lane; i < COLLECT_MAX_REWARDS; i += 64) gv.rewards_obj[(size_t)env * gv.reward_stride + i] = b->rewards[i];

    for (int k = 0; k < A; ++k) {
        float cs, sn;
        yaw_matrix(b->yaw_frand[k] * 3.14159274f * 2, cs, sn);
        if (lane == 0) {
            AgentState *a = gv.agents + (size_t)env * A + k;
            a->pos[0] = float(b->spawn[k][0]) + 0.5f; a->pos[1] = float(b->spawn[k][1]) + 0.0f + 1.75f; a->pos[2] = float(b->spawn[k][2]) + 0.5f;
            a->m00 = cs; a->m02 = sn; a->m20 = -sn; a->m22 = cs;
            a->pitch = 0.0f; a->hvx = 0.0f; a->hvz = 0.0f; a->vvel = 0.0f; a->voffset = 0.0f; a->step_offset = 0.0f;
            a->jump_speed = 10.0f; a->was_jumping = 0; a->carrying = -1; a->picked_up = 0; a->visited_zone = 0;
            a->spawn[0] = b->spawn[k][0]; a->spawn[1] = b->spawn[k][1]; a->spawn[2] = b->spawn[k][2];
            a->last_reward = 0.0f; a->total_reward = 0.0f;
            gv.rewards[(size_t)env * A + k] = 0.0f;
            gv.actions[(size_t)env * A + k] = 0;
        }
    }
    if (lane == 0) {
        gh->L = b->dim[0]; gh->H = b->dim[1]; gh->W = b->dim[2];
        gh->bz[0] = gh->bz[1] = gh->bz[2] = gh->bz[3] = 0;
        gh->layout_color = b->layout_color; gh->wall_color = b->wall_color; gh->draw_walls = 1;
        gh->num_objects = b->num_objects; gh->num_boxes = nb; gh->num_terrain = 0;
        gh->num_rewards = b->num_rewards; gh->num_platforms = b->num_positive;
        gh->num_frames = 0; gh->done = 0; gh->highest_tower = 0; gh->solved = 0;
        gh->episode_sec = 0.0f; gh->episode_len = b->episode_len; gh->bz_reward = 0.0f; gh->bar_half_width = 0.24f;
        gh->episodes_consumed = consumed + 1;
        status[env] = consumed + 1;              // per-env count, total, error flags: copied to the host after every step
        atomicAdd(&status[gv.num_envs], 1);
        if (force_all) gv.done[env] = 0;
    }
}

template <int A_MAX>
__device__ __forceinline__ void collect_tick(const GymView &gv, const int env)
{
    __shared__ Col s_cand[MAX_CAND];

    const int lane = lane_id();
    if (env >= gv.num_envs) return;
    const int A = gv.num_agents;
    const unsigned long long below = (1ull << lane) - 1ull;

    // ---- header fields as scalars (never copy the record: see mv_step.hip)
    EnvHeader *gh = gv.hdr + env;
    const int numObjects = gh->num_objects, numBoxes = gh->num_boxes, numRewards = gh->num_rewards, numPositive = gh->num_platforms;
    int numFrames = gh->num_frames, done = gh->done, solved = gh->solved, collected = gh->highest_tower, starved = gh->starved;
    float episodeSec = gh->episode_sec;
    const float episodeLen = gh->episode_len, lookLimit = gh->p_vertical_look_limit;
    const LayoutBox *gboxes = gv.boxes + (size_t)env * gv.box_stride;
    const int8_t *hm = gv.heightmap + (size_t)env * HM_BYTES;

    // ---- movable boxes and diamonds: two per lane
    Objs ob;
    const MovableObject *gobj = gv.objects + (size_t)env * MAX_OBJECTS;
    const int oi[2] = {lane, lane < 16 ? 64 + lane : -1};
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        ob.valid[k] = oi[k] >= 0 && oi[k] < numObjects;
        ob.x[k] = ob.y[k] = ob.z[k] = 0; ob.state[k] = 0;
        if (ob.valid[k]) {
            const MovableObject o = gobj[oi[k]];
            ob.x[k] = o.x; ob.y[k] = o.y; ob.z[k] = o.z; ob.state[k] = o.state;
        }
    }
    MovableObject *grew = gv.rewards_obj + (size_t)env * gv.reward_stride;
    const int ri[2] = {lane, lane < COLLECT_MAX_REWARDS - 64 ? 64 + lane : -1};
    int rwx[2], rwy[2], rwz[2], rwState[2];
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        rwx[k] = rwy[k] = rwz[k] = rwState[k] = 0;
        if (ri[k] >= 0 && ri[k] < numRewards) {
            const MovableObject r = grew[ri[k]];
            rwx[k] = r.x; rwy[k] = r.y; rwz[k] = r.z; rwState[k] = r.state;
        }
    }

    // ---- agents: records in LDS (mv_agents.h), one agent's physics fields in registers at a time
    __shared__ AgentState s_ag[A_MAX];
    __shared__ int s_act[A_MAX];
    agents_load(gv, env, A, s_ag, s_act);
    const float dt = DT;

    if (lane < A) {   // actions -> intents: agents are independent here, one lane each
        AgentState a;
        phys_load(a, s_ag[lane]);
        apply_actions(a, s_act[lane], dt, lookLimit);
        phys_store(s_ag[lane], a);
    }
    wave_sync();

    // ---- physics, agent by agent: broadphase into LDS, then the shared controller on two candidates per lane
#pragma unroll 1
    for (int i = 0; i < A; ++i) {
        {
            AgentState a;
            phys_load(a, s_ag[i]);
            const Envelope env_i = step_envelope(a, dt);
            int count = 0;
            for (int base = 0; base < numBoxes; base += 64) {   // slabs, in list order
                const int bi = base + lane;
                bool keep = false;
                V3 lo = v3(0, 0, 0), hi = v3(0, 0, 0);
                if (bi < numBoxes) {
                    const int4 mn = *reinterpret_cast<const int4 *>(&gboxes[bi].min[0]);
                    const int4 mx = *reinterpret_cast<const int4 *>(&gboxes[bi].max[0]);
                    lo = v3(float(mn.x), float(mn.y) - CAP_HH, float(mn.z));
                    hi = v3(float(mx.x), float(mx.y) + CAP_HH, float(mx.z));
                    keep = (mn.w & VX_SOLID) && meets(env_i, lo, hi);
                }
                const unsigned long long m = __ballot(keep);
                const int pos = count + __popcll(m & below);
                if (keep && pos < MAX_CAND) { Col c; c.kind = 1; c.lo = lo; c.hi = hi; s_cand[pos] = c; }
                count += __popcll(m);
            }
#pragma unroll
            for (int k = 0; k < 2; ++k) {   // movable boxes on the ground (carried ones do not collide)
                bool keep = false;
                V3 lo = v3(0, 0, 0), hi = v3(0, 0, 0);
                if (ob.valid[k] && ob.state[k] <= 0) {
                    const float cx = float(ob.x[k]) + 0.5f, cy = float(ob.y[k]) + 0.5f + OBJ_COLL_YOFF, cz = float(ob.z[k]) + 0.5f;
                    lo = v3(cx - OBJ_COLL_HALF, (cy - OBJ_COLL_HALF) - CAP_HH, cz - OBJ_COLL_HALF);
                    hi = v3(cx + OBJ_COLL_HALF, (cy + OBJ_COLL_HALF) + CAP_HH, cz + OBJ_COLL_HALF);
                    keep = meets(env_i, lo, hi);
                }
                const unsigned long long m = __ballot(keep);
                const int pos = count + __popcll(m & below);
                if (keep && pos < MAX_CAND) { Col c; c.kind = 1; c.lo = lo; c.hi = hi; s_cand[pos] = c; }
                count += __popcll(m);
            }
            if (A_MAX > 1) {                // the other agents' capsules, always
                bool keep = false;
                V3 centre = v3(0, 0, 0);
                if (lane < A && lane != i) { keep = true; centre = v3(s_ag[lane].pos[0], s_ag[lane].pos[1], s_ag[lane].pos[2]); }
                const unsigned long long m = __ballot(keep);
                const int pos = count + __popcll(m & below);
                if (keep && pos < MAX_CAND) { Col c; c.kind = 2; c.lo = centre; c.hi = v3(2 * CAP_HH, 0.0f, 0.0f); s_cand[pos] = c; }
                count += __popcll(m);
            }
            if (count > MAX_CAND) { starved |= 2; if (lane == 0) atomicOr(&gv.episode_status[gv.num_envs + 1], (int)ST_CANDIDATES); }
            wave_sync();
            Col col[NC];
#pragma unroll
            for (int k = 0; k < NC; ++k) {
                col[k].kind = 0; col[k].lo = col[k].hi = v3(0, 0, 0);
                if (lane + 64 * k < min(count, (int)MAX_CAND)) col[k] = s_cand[lane + 64 * k];
            }
            wave_sync();
            player_step<NC>(a, col, dt);
            if (lane == 0) phys_store(s_ag[i], a);
            wave_sync();
        }
    }

    // ---- interact: pick up / put down with the default callbacks
#pragma unroll 1
    for (int i = 0; i < A; ++i) {
        if (s_act[i] & ACT_INTERACT) {
            AgentState a;
            phys_load(a, s_ag[i]);
            const int carrying = s_ag[i].carrying;
            const Cam cam = camera_of(a);
            if (carrying >= 0) {
                const V3 t = cam_to_world(cam, v3(0.0f, -0.44f + -0.3f, -1.0f));
                int vx[3];
                voxel_of(t, vx);
                bool collidesWithAgent = false;
                for (int j = 0; j < A; ++j)
                    if (j != i) {
                        int c[3];
                        voxel_of(v3(s_ag[j].pos[0], s_ag[j].pos[1] + 0.05f, s_ag[j].pos[2]), c);
                        if (c[0] == vx[0] && c[1] == vx[1] && c[2] == vx[2]) collidesWithAgent = true;
                    }
                const Bits128 solid = column_solid_hm(hm, vx[0], vx[2]);
                const Bits128 objs = column_objects(ob, vx[0], vx[2]);
                const bool placeable = vx[1] > -120 && vx[1] < 120;
                // a diamond's cell counts as empty (its voxel is not solid and holds no physics object)
                const bool empty = !test(solid, vx[1]) && !test(objs, vx[1]);
                if (placeable && empty && !collidesWithAgent) {
                    Bits128 occ = solid;
                    occ.lo |= objs.lo; occ.hi |= objs.hi;
                    vx[1] = drop_height(occ, vx[1]);
                    const int oidx = carrying;
#pragma unroll
                    for (int k = 0; k < 2; ++k)
                        if (oi[k] == oidx) { ob.x[k] = vx[0]; ob.y[k] = vx[1]; ob.z[k] = vx[2]; ob.state[k] = 0; }
                    if (lane == 0) s_ag[i].carrying = -1;
                }
            } else {
                const V3 pickup = cam_to_world(cam, v3(0.0f, -0.44f, -1.0f));
                int vx[3];
                voxel_of(pickup, vx);
                const int o0 = object_at(ob, vx[0], vx[1], vx[2]);
                const int o1 = object_at(ob, vx[0], vx[1] + 1, vx[2]);
                const int o2 = object_at(ob, vx[0], vx[1] + 2, vx[2]);
                int oidx = -1;
                if (o0 >= 0 && o1 < 0) oidx = o0;
                else if (o1 >= 0 && o2 < 0) oidx = o1;
                if (oidx >= 0) {
#pragma unroll
                    for (int k = 0; k < 2; ++k)
                        if (oi[k] == oidx) ob.state[k] = 1 + i;
                    if (lane == 0) s_ag[i].carrying = oidx;
                }
            }
        }
        wave_sync();
    }

    // ---- fall detection: back above the spawn cell, and a penalty (agentFell)
#pragma unroll 1
    for (int i = 0; i < A; ++i)
        if (s_ag[i].pos[1] + 0.05f < -20.0f) {
            const int sx = s_ag[i].spawn[0], sy = s_ag[i].spawn[1], sz = s_ag[i].spawn[2];
            const Bits128 solid = column_solid_hm(hm, sx, sz);
            int py = sy;
            while (test(solid, py) && py < 1000) ++py;
            wave_sync();
            if (lane == 0) {
                AgentState &a = s_ag[i];
                a.pos[0] = float(sx) + 0.5f; a.pos[1] = float(py) + 0.5f; a.pos[2] = float(sz) + 0.5f;
                a.m00 = 1.0f; a.m02 = 0.0f; a.m20 = 0.0f; a.m22 = 1.0f;
                a.hvx = 0.0f; a.hvz = 0.0f; a.vvel = 0.0f;
            }
            wave_sync();
            reward_agent_lds(s_ag, 2, i, 1);
        }

    // ---- CollectScenario::step: diamonds.  Their cells are distinct, so an agent matches at most one.
#pragma unroll 1
    for (int i = 0; i < A; ++i) {
        {
            int vx[3];
            voxel_of(v3(s_ag[i].pos[0], s_ag[i].pos[1] + 0.05f, s_ag[i].pos[2]), vx);
            int kind = 0;
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                const bool got = rwState[k] != 0 && rwx[k] == vx[0] && rwy[k] == vx[1] && rwz[k] == vx[2];
                const unsigned long long gm = __ballot(got);
                if (gm) kind = __shfl(rwState[k], __ffsll((long long)gm) - 1, 64);
                if (got) rwState[k] = 0;
            }
            if (kind != 0) {
                if (kind == 1) ++collected;
                reward_team_lds(s_ag, A, kind == 1 ? 1 : 2, i, 1);
                if (collected >= numPositive && !solved) {
                    solved = 1;
                    episodeSec = fmax_sel(episodeSec, episodeLen - 0.3f);
                    reward_team_lds(s_ag, A, 3, i, 1);
                }
                // grid.remove(voxel) forgets a movable box that was dropped into the diamond's cell
#pragma unroll
                for (int k = 0; k < 2; ++k)
                    if (ob.valid[k] && ob.state[k] == 0 && ob.x[k] == vx[0] && ob.y[k] == vx[1] && ob.z[k] == vx[2]) ob.state[k] = -1;
            }
        }
    }

    // ---- timers / done
    episodeSec += dt;
    const float bar = fmax_sel(0.0f, (episodeLen - episodeSec) / episodeLen) * 0.24f;
    if (episodeSec >= episodeLen) done = 1;
    ++numFrames;

    // ---- write back
    MovableObject *gobjw = gv.objects + (size_t)env * MAX_OBJECTS;
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        if (ob.valid[k]) {
            MovableObject o;
            o.x = (int8_t)ob.x[k]; o.y = (int8_t)ob.y[k]; o.z = (int8_t)ob.z[k]; o.state = (int8_t)ob.state[k];
            gobjw[oi[k]] = o;
        }
        if (ri[k] >= 0 && ri[k] < numRewards) grew[ri[k]].state = (int8_t)rwState[k];
    }
    if (lane == 0) {
        gh->num_frames = numFrames; gh->done = done; gh->solved = solved; gh->highest_tower = collected; gh->starved = starved;
        gh->episode_sec = episodeSec; gh->bar_half_width = bar;
        gv.done[env] = (uint8_t)done;
    }
    agents_store(gv, env, A, s_ag);
    if (done && lane < A) gv.true_objective[(size_t)env * A + lane] = float(solved);   // scenario_collect.hpp:42

    // ---- the auto-reset of VectorEnv::step: the wave of a finished env swaps the next episode in right here
    if (done) {
        wave_sync();   // one wave per env: orders the stores above before the swap-in's
        swap_in_episode(gv, static_cast<const CollectBlob *>(gv.blobs), gv.episode_status, env, 0);
    }
}

}  // namespace tick_collect
}  // namespace mv
