// megaverse_amd/csrc/mv_tick_hex.h -- the hex tick as a device function (namespace mv::tick_hex): shared by the scenario's own step
// kernel (mv_step_hex.hip) and by the union step kernel that steps several gyms with one launch (mv_step_union.hip).
//
// (members of the reference's multi-task set, scenarios/init.hpp:47-48).
//
// Replaces, per env (reference paths relative to src/libs):
//   Env::step                                   env/src/env.cpp:83-152            (shared pieces: mv_physics.h)
//   HexExploreScenario::step                    scenarios/src/scenario_hex_explore.cpp:43-58
//   HexMemoryScenario::step                     scenarios/src/scenario_hex_memory.cpp:84-127
//   VectorEnv::step done bookkeeping + Env::reset of finished envs (env/src/vector_env.cpp:93-105)
// (no ObjectStackingComponent, no FallDetectionComponent in these scenarios: "interact" does nothing, nobody is put back)
//
// The maze is a list of boxes, each axis-aligned in the world or in one of three frames rotated about Y (mv_gen_hex.cpp): the
// colliding ones -- the floor and up to 294 walls -- are the list's prefix.  As in Collect, the wave streams that prefix once per
// agent and tick (64 per pass), keeps the boxes that meet a conservative envelope of everything the capsule can do this tick, and
// compacts them in list order into an LDS candidate list; the shared controller then works on one candidate per lane.  The
// collectables (<= 128) live two per lane.
#pragma once
#include <hip/hip_runtime.h>

#include <algorithm>

#include "mv_actions.h"
#include "mv_agents.h"
#include "mv_frame.h"
#include "mv_math.h"
#include "mv_physics.h"
#include "mv_types.h"

namespace mv {
namespace tick_hex {



constexpr int NC = 1;
constexpr int MAX_CAND = 64 * NC;

struct Envelope { float c[3], h[3]; };   // centre / half extents in capsule-CENTRE space, widened by the capsule radius and slack

// Everything agent `a` can touch during player_step() (same reasoning as mv_step_collect.hip: step_envelope)
__device__ __forceinline__ Envelope step_envelope(const AgentState &a, float dt)
{
    const float reach = CAP_R + 0.25f;
    const float hx = fabsf(a.hvx) * dt + 0.35f, hz = fabsf(a.hvz) * dt + 0.35f;
    float vv = a.vvel - GRAVITY * dt;
    vv = fminf(fmaxf(vv, -FALL_SPEED), fmaxf(a.jump_speed, 0.0f));
    const float up = STEP_HEIGHT + fmaxf(vv, 0.0f) * dt + 0.3f;
    const float down = STEP_HEIGHT + fmaxf(-vv, 0.0f) * dt + 0.3f;
    Envelope e;
    e.c[0] = a.pos[0]; e.c[2] = a.pos[2]; e.h[0] = hx + reach; e.h[2] = hz + reach;
    e.c[1] = a.pos[1] + 0.5f * (up - down); e.h[1] = 0.5f * (up + down) + reach;
    return e;
}
// does the envelope meet a box given in frame `fr` (0 world, 1 + k wall orientation k)?  The envelope's bounding box in that frame.
__device__ __forceinline__ bool meets(const Envelope &e, int fr, V3 lo, V3 hi)
{
    V3 c = v3(e.c[0], e.c[1], e.c[2]);
    float hx = e.h[0], hz = e.h[2];
    if (fr != 0) {
        c = hex_to_local(fr - 1, c);
        const float ac = fr == 3 ? 0.0f : 0.8660254f, as = fr == 3 ? 1.0f : 0.5f;
        const float rx = ac * e.h[0] + as * e.h[2] + 1e-3f, rz = as * e.h[0] + ac * e.h[2] + 1e-3f;
        hx = rx; hz = rz;
    }
    return lo.x <= c.x + hx && hi.x >= c.x - hx && lo.y <= c.y + e.h[1] && hi.y >= c.y - e.h[1] && lo.z <= c.z + hz && hi.z >= c.z - hz;
}


// Episode swap-in: Env::reset of one env from its resident HexBlob (called by the env's whole wavefront)
__device__ __forceinline__ void swap_in_episode(const GymView &gv, const HexBlob *blobs, int *status, int env, int force_all)
{
    const int lane = lane_id();
    EnvHeader *gh = gv.hdr + env;
    const int consumed = gh->episodes_consumed;
    const HexBlob *b = blobs + (size_t)env * gv.spares + consumed % gv.spares;   // ring slot of episode number consumed + 1
    if (b->seq != consumed + 1) {   // the host has not delivered the next episode (mv_api.hip keeps one ahead)
        if (lane == 0) { gh->starved |= 1; atomicOr(&status[gv.num_envs + 1], (int)ST_STARVED); }
        return;
    }
    const int A = gv.num_agents;
    const int nb = b->num_boxes, no = b->num_objs;
    {
        const uint4 *src = reinterpret_cast<const uint4 *>(b->boxes);
        uint4 *dst = reinterpret_cast<uint4 *>(gv.hex_boxes + (size_t)env * HEX_MAX_BOXES);
        for (int i = lane; i < nb * 2; i += 64) dst[i] = src[i];
        const uint4 *osrc = reinterpret_cast<const uint4 *>(b->objs);
        uint4 *odst = reinterpret_cast<uint4 *>(gv.hex_objs + (size_t)env * HEX_MAX_OBJS);
        for (int i = lane; i < no * 2; i += 64) odst[i] = osrc[i];
    }
    for (int k = 0; k < A; ++k) {
        float cs, sn;
        yaw_matrix(b->yaw[k], cs, sn);
        if (lane == 0) {
            AgentState *a = gv.agents + (size_t)env * A + k;
            const float px = b->spawn[k][0], py = b->spawn[k][1], pz = b->spawn[k][2];
            a->pos[0] = px + 0.5f; a->pos[1] = py + 0.0f + 1.75f; a->pos[2] = pz + 0.5f;   // scenario_default.hpp:89, agent.cpp:45
            a->m00 = cs; a->m02 = sn; a->m20 = -sn; a->m22 = cs;
            a->pitch = 0.0f; a->hvx = 0.0f; a->hvz = 0.0f; a->vvel = 0.0f; a->voffset = 0.0f; a->step_offset = 0.0f;
            a->jump_speed = 10.0f; a->was_jumping = 0; a->carrying = -1; a->picked_up = 0; a->visited_zone = 0;
            a->spawn[0] = (int)floorf(px); a->spawn[1] = (int)floorf(py); a->spawn[2] = (int)floorf(pz);
            a->last_reward = 0.0f; a->total_reward = 0.0f;
            gv.rewards[(size_t)env * A + k] = 0.0f;
            gv.actions[(size_t)env * A + k] = 0;
        }
    }
    if (lane == 0) {
        gh->L = 0; gh->H = 0; gh->W = 0;
        gh->bz[0] = gh->bz[1] = gh->bz[2] = gh->bz[3] = 0;
        gh->layout_color = 0; gh->wall_color = 0; gh->draw_walls = 0;
        gh->num_objects = 0; gh->num_boxes = nb; gh->num_terrain = b->num_colliders;
        gh->num_rewards = no; gh->num_platforms = b->num_good;
        gh->num_frames = 0; gh->done = 0; gh->highest_tower = 0; gh->solved = 0;
        gh->episode_sec = 0.0f; gh->episode_len = b->episode_len; gh->bz_reward = 0.0f; gh->bar_half_width = 0.24f;
        gh->hex_target[0] = b->target[0]; gh->hex_target[1] = b->target[1];
        gh->episodes_consumed = consumed + 1;
        status[env] = consumed + 1;
        atomicAdd(&status[gv.num_envs], 1);
        if (force_all) gv.done[env] = 0;
    }
}

template <int A_MAX>
__device__ __forceinline__ void hex_tick(const GymView &gv, const int env)
{
    __shared__ Col s_cand[MAX_CAND];

    const int lane = lane_id();
    if (env >= gv.num_envs) return;
    const int A = gv.num_agents;
    const unsigned long long below = (1ull << lane) - 1ull;

    EnvHeader *gh = gv.hdr + env;
    const int numCol = gh->num_terrain, numObjs = gh->num_rewards, numGood = gh->num_platforms, scen = gh->scenario;
    int numFrames = gh->num_frames, done = gh->done, solved = gh->solved, collected = gh->highest_tower, starved = gh->starved;
    float episodeSec = gh->episode_sec;
    const float episodeLen = gh->episode_len, lookLimit = gh->p_vertical_look_limit;
    const float targetX = gh->hex_target[0], targetZ = gh->hex_target[1];
    const HexRec *gboxes = gv.hex_boxes + (size_t)env * HEX_MAX_BOXES;
    HexRec *gobjs = gv.hex_objs + (size_t)env * HEX_MAX_OBJS;

    // ---- collectables: two per lane (position + flags; the scale is only drawn)
    float ox[2], oy[2], oz[2];
    int ometa[2];
    bool odirty[2] = {false, false};
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        const int oi = lane + 64 * k;
        ox[k] = oy[k] = oz[k] = 0.0f; ometa[k] = 0;
        if (oi < numObjs) {
            const float4 r = *reinterpret_cast<const float4 *>(&gobjs[oi].a[0]);
            ox[k] = r.x; oy[k] = r.y; oz[k] = r.z; ometa[k] = __float_as_int(r.w);
        }
    }

    __shared__ AgentState s_ag[A_MAX];
    __shared__ int s_act[A_MAX];
    agents_load(gv, env, A, s_ag, s_act);
    const float dt = DT;

    if (lane < A) {   // actions -> intents
        AgentState a;
        phys_load(a, s_ag[lane]);
        apply_actions(a, s_act[lane], dt, lookLimit);
        phys_store(s_ag[lane], a);
    }
    wave_sync();

    // ---- physics, agent by agent: broadphase into LDS, then the shared controller on one candidate per lane
#pragma unroll 1
    for (int i = 0; i < A; ++i) {
        AgentState a;
        phys_load(a, s_ag[i]);
        const Envelope env_i = step_envelope(a, dt);
        int count = 0;
        for (int base = 0; base < numCol; base += 64) {   // floor and walls, in list order
            const int bi = base + lane;
            bool keep = false;
            V3 lo = v3(0, 0, 0), hi = v3(0, 0, 0);
            int fr = 0;
            if (bi < numCol) {
                const float4 ra = *reinterpret_cast<const float4 *>(&gboxes[bi].a[0]);
                const float4 rb = *reinterpret_cast<const float4 *>(&gboxes[bi].b[0]);
                fr = __float_as_int(ra.w) & 15;
                lo = v3(ra.x, ra.y - CAP_HH, ra.z);
                hi = v3(rb.x, rb.y + CAP_HH, rb.z);
                keep = meets(env_i, fr, lo, hi);
            }
            const unsigned long long m = __ballot(keep);
            const int pos = count + __popcll(m & below);
            if (keep && pos < MAX_CAND) { Col c; c.kind = fr == 0 ? 1 : 2 + fr; c.lo = lo; c.hi = hi; s_cand[pos] = c; }
            count += __popcll(m);
        }
        if (A_MAX > 1) {   // the other agents' capsules, always
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
        player_step<NC, true>(a, col, dt);
        if (lane == 0) phys_store(s_ag[i], a);
        wave_sync();
    }

    // ---- scenario logic
    if (scen == SCN_HEX_EXPLORE) {
#pragma unroll 1
        for (int i = 0; i < A; ++i) {
            const V3 d = v3(s_ag[i].pos[0] - targetX, (s_ag[i].pos[1] + 0.05f) - 0.0f, s_ag[i].pos[2] - targetZ);
            const float distance = sqrtf(len2(d));
            if (distance < 1.2f && !solved) {   // (`distance < 1.2` against a double: no float lies between 1.2 and 1.2f)
                solved = 1;
                episodeSec = fmax_sel(episodeSec, episodeLen - 0.3f);   // doneWithTimer()
                reward_team_lds(s_ag, A, 1, i, 1);
                if (lane == 0) {   // rewardObject->translate({1e3, 1e3, 1e3})
                    ox[0] = ox[0] + 1e3f; oy[0] = oy[0] + 1e3f; oz[0] = oz[0] + 1e3f;
                    ometa[0] &= ~256; odirty[0] = true;
                }
            }
        }
    } else {
        if (collected >= numGood && !solved) {
            solved = 1;
            episodeSec = fmax_sel(episodeSec, episodeLen - 0.3f);
        }
#pragma unroll 1
        for (int i = 0; i < A; ++i) {
            const V3 t = v3(s_ag[i].pos[0], s_ag[i].pos[1] + 0.05f, s_ag[i].pos[2]);
            const int vx = (int)floorf(t.x), vy = (int)floorf(t.y), vz = (int)floorf(t.z);
            // the reference walks the 3 x 3 cells around the agent (dx outer, dz inner) and each cell's list in insertion order
            unsigned key[2];
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                key[k] = ~0u;
                const int cx = ((ometa[k] >> 12) & 255) - 128 - vx, cz = ((ometa[k] >> 20) & 255) - 128 - vz;
                if ((ometa[k] & 256) && vy == 0 && cx >= -1 && cx <= 1 && cz >= -1 && cz <= 1) {
                    const V3 d = v3(ox[k] - t.x, oy[k] - t.y, oz[k] - t.z);
                    if (sqrtf(len2(d)) < 1.0f) key[k] = (unsigned)(((cx + 1) * 3 + (cz + 1)) * 128 + lane + 64 * k);
                }
            }
            for (;;) {
                const unsigned m = wave_min_u32(min(key[0], key[1]));
                if (m == ~0u) break;
                const int oi = (int)(m & 127u);
                const int good = (__shfl(oi < 64 ? ometa[0] : ometa[1], oi & 63, 64) >> 4) & 1;
                reward_team_lds(s_ag, A, good ? 1 : 2, i, 1);
                collected += good;
#pragma unroll
                for (int k = 0; k < 2; ++k)
                    if (key[k] == m) {   // it->object->translate({100, 100, 100}): out of everybody's way, still drawn
                        ox[k] = ox[k] + 100.0f; oy[k] = oy[k] + 100.0f; oz[k] = oz[k] + 100.0f;
                        ometa[k] &= ~256; odirty[k] = true; key[k] = ~0u;
                    }
            }
        }
    }

    // ---- timers / done
    episodeSec += dt;
    const float bar = fmax_sel(0.0f, (episodeLen - episodeSec) / episodeLen) * 0.24f;
    if (episodeSec >= episodeLen) done = 1;
    ++numFrames;

    // ---- write back
#pragma unroll
    for (int k = 0; k < 2; ++k)
        if (odirty[k]) *reinterpret_cast<float4 *>(&gobjs[lane + 64 * k].a[0]) = make_float4(ox[k], oy[k], oz[k], __int_as_float(ometa[k]));
    if (lane == 0) {
        gh->num_frames = numFrames; gh->done = done; gh->solved = solved; gh->highest_tower = collected; gh->starved = starved;
        gh->episode_sec = episodeSec; gh->bar_half_width = bar;
        gv.done[env] = (uint8_t)done;
    }
    agents_store(gv, env, A, s_ag);
    if (done && lane < A) gv.true_objective[(size_t)env * A + lane] = float(solved);   // scenario_hex_{memory,explore}.hpp trueObjective

    if (done) {   // the auto-reset of VectorEnv::step
        wave_sync();
        swap_in_episode(gv, static_cast<const HexBlob *>(gv.blobs), gv.episode_status, env, 0);
    }
}

}  // namespace tick_hex
}  // namespace mv
