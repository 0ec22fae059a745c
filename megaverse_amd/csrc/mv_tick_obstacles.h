// megaverse_amd/csrc/mv_tick_obstacles.h -- the obstacles tick as a device function (namespace mv::tick_obstacles): shared by the scenario's own step
// kernel (mv_step_obstacles.hip) and by the union step kernel that steps several gyms with one launch (mv_step_union.hip).
//
// (ObstaclesEasy / Medium / Hard / Walls / Steps / Lava; BASELINE.json configs[2]).
//
// Replaces, per env (reference paths relative to src/libs):
//   Env::step                                   env/src/env.cpp:83-152            (shared pieces: mv_physics.h)
//   ObstaclesScenario::step / agentTouchedLava  scenarios/src/scenario_obstacles.cpp:197-239,268-278
//   ObjectStackingComponent (default callbacks) scenarios/include/scenarios/component_object_stacking.hpp:45-168
//   FallDetectionComponent                      scenarios/include/scenarios/component_fall_detection.hpp:33-55
//   Scenario::rewardTeam/rewardAll/doneWithTimer env/include/env/scenario.hpp:114-117,259-307
//   VectorEnv::step done bookkeeping + Env::reset of finished envs (env/src/vector_env.cpp:93-105): the reset
//   kernel below swaps in the episode the host generator (mv_gen_obstacles.cpp) left resident in HBM.
//
// Same mapping as the TowerBuilding kernel (one wavefront per env, colliders in VGPRs) with four colliders per
// lane: 128 merged layout slabs, 80 movable boxes, 8 agent capsules.  The level is a long chain of platforms,
// so voxel questions ("is this cell solid / lava / exit / holding a diamond?") are answered from the box
// lists with ballots instead of a dense chunk; column occupancy for drops and teleports is a 128-bit wave OR.
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
namespace tick_obstacles {



constexpr int NC = 4;

struct BoxI {
    int min[3], max[3], type;
    bool valid;
};
__device__ __forceinline__ bool contains(const BoxI &b, int x, int y, int z)
{
    return b.valid && x >= b.min[0] && x < b.max[0] && y >= b.min[1] && y < b.max[1] && z >= b.min[2] && z < b.max[2];
}

// solid layout cells of column (x, z)
__device__ __forceinline__ Bits128 column_solid(const BoxI (&lb)[2], int x, int z)
{
    Bits128 m{0ull, 0ull};
#pragma unroll
    for (int k = 0; k < 2; ++k)
        if (lb[k].valid && (lb[k].type & VX_SOLID) && x >= lb[k].min[0] && x < lb[k].max[0] && z >= lb[k].min[2] && z < lb[k].max[2])
            set_range(m, lb[k].min[1], lb[k].max[1]);
    m.lo = wave_or_u64(m.lo); m.hi = wave_or_u64(m.hi);
    return m;
}

// Episode swap-in: Env::reset of one env from its resident EpisodeBlob (called by the env's whole wavefront: by the stand-alone
// reset kernel for mv_reset, and by the tail of the step kernel for the auto-reset of VectorEnv::step, vector_env.cpp:93-105)
__device__ __forceinline__ void swap_in_episode(const GymView &gv, const EpisodeBlob *blobs, int *status, int env, int force_all)
{
    const int lane = lane_id();
    EnvHeader *gh = gv.hdr + env;
    const int consumed = gh->episodes_consumed;
    const EpisodeBlob *b = blobs + (size_t)env * gv.spares + consumed % gv.spares;   // ring slot of episode number consumed + 1
    if (b->seq != consumed + 1) {   // the host has not delivered the next episode: must never happen (mv_api.hip keeps one ahead)
        if (lane == 0) { gh->starved |= 1; atomicOr(&status[gv.num_envs + 1], (int)ST_STARVED); }
        return;
    }
    const int A = gv.num_agents;
    // 128 slabs x 32 B: two 16-byte pieces per slab, 4 per lane
    const uint4 *src = reinterpret_cast<const uint4 *>(b->boxes);
    uint4 *dst = reinterpret_cast<uint4 *>(gv.boxes + (size_t)env * MAX_BOXES);
    for (int i = lane; i < MAX_BOXES * 2; i += 64) dst[i] = src[i];
    if (lane < MAX_TERRAIN * 2)
        reinterpret_cast<uint4 *>(gv.terrain + (size_t)env * MAX_TERRAIN)[lane] = reinterpret_cast<const uint4 *>(b->terrain)[lane];
    for (int i = lane; i < MAX_OBJECTS; i += 64) gv.objects[(size_t)env * MAX_OBJECTS + i] = b->objects[i];
    if (lane < MAX_REWARDS) gv.rewards_obj[(size_t)env * MAX_REWARDS + lane] = b->rewards[lane];

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
        gh->bz[0] = b->org[0]; gh->bz[1] = b->org[1]; gh->bz[2] = b->org[2]; gh->bz[3] = 0;
        gh->layout_color = b->layout_color; gh->wall_color = b->wall_color; gh->draw_walls = b->draw_walls;
        gh->num_objects = b->num_objects; gh->num_boxes = b->num_boxes; gh->num_terrain = b->num_terrain;
        gh->num_rewards = b->num_rewards; gh->num_platforms = b->num_platforms;
        gh->num_frames = 0; gh->done = 0; gh->highest_tower = 0; gh->solved = 0;
        gh->episode_sec = 0.0f; gh->episode_len = b->episode_len; gh->bz_reward = 0.0f; gh->bar_half_width = 0.24f;
        gh->episodes_consumed = consumed + 1;
        status[env] = consumed + 1;              // per-env count, total, error flags: copied to the host after every step
        atomicAdd(&status[gv.num_envs], 1);
        if (force_all) gv.done[env] = 0;
    }
}

// pipe_wait: see tower_tick (mv_tick_tower.h) -- a workgroup barrier before the tick writes anything the previous tick's frame setup may still be reading
template <int A_MAX>
__device__ __forceinline__ void obstacles_tick(const GymView &gv, const int env, const int pipe_wait = 0)
{
    const int lane = lane_id();
    if (env >= gv.num_envs) return;
    const int A = gv.num_agents;

    // ---- header fields as scalars (never copy the record: see mv_step.hip)
    EnvHeader *gh = gv.hdr + env;
    const int numObjects = gh->num_objects, numBoxes = gh->num_boxes, numTerrain = gh->num_terrain, numRewards = gh->num_rewards;
    int numFrames = gh->num_frames, done = gh->done, solved = gh->solved;
    float episodeSec = gh->episode_sec;
    const float episodeLen = gh->episode_len, lookLimit = gh->p_vertical_look_limit;

    // ---- wave-resident scene
    Col col[NC];
    BoxI lb[2];
    Objs ob;
#pragma unroll
    for (int k = 0; k < NC; ++k) { col[k].kind = 0; col[k].lo = col[k].hi = v3(0, 0, 0); }
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        const int bi = lane + 64 * k;
        lb[k].valid = bi < numBoxes;
        lb[k].type = 0;
        lb[k].min[0] = lb[k].min[1] = lb[k].min[2] = lb[k].max[0] = lb[k].max[1] = lb[k].max[2] = 0;
        if (lb[k].valid) {
            const LayoutBox b = gv.boxes[(size_t)env * MAX_BOXES + bi];
            lb[k].min[0] = b.min[0]; lb[k].min[1] = b.min[1]; lb[k].min[2] = b.min[2];
            lb[k].max[0] = b.max[0]; lb[k].max[1] = b.max[1]; lb[k].max[2] = b.max[2];
            lb[k].type = b.type;
            if (b.type & VX_SOLID) {
                col[k].kind = 1;
                col[k].lo = v3(float(b.min[0]), float(b.min[1]) - CAP_HH, float(b.min[2]));
                col[k].hi = v3(float(b.max[0]), float(b.max[1]) + CAP_HH, float(b.max[2]));
            }
        }
    }
    const MovableObject *gobj = gv.objects + (size_t)env * MAX_OBJECTS;
    const int oi[2] = {lane, lane < 16 ? 64 + lane : -1};
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        ob.valid[k] = oi[k] >= 0 && oi[k] < numObjects;
        ob.x[k] = ob.y[k] = ob.z[k] = 0; ob.state[k] = 0;
        if (ob.valid[k]) {
            const MovableObject o = gobj[oi[k]];
            ob.x[k] = o.x; ob.y[k] = o.y; ob.z[k] = o.z; ob.state[k] = o.state;
            if (o.state == 0) {
                const float cx = float(o.x) + 0.5f, cy = float(o.y) + 0.5f + OBJ_COLL_YOFF, cz = float(o.z) + 0.5f;
                col[2 + k].kind = 1;
                col[2 + k].lo = v3(cx - OBJ_COLL_HALF, (cy - OBJ_COLL_HALF) - CAP_HH, cz - OBJ_COLL_HALF);
                col[2 + k].hi = v3(cx + OBJ_COLL_HALF, (cy + OBJ_COLL_HALF) + CAP_HH, cz + OBJ_COLL_HALF);
            }
        }
    }
    // terrain boxes and diamonds: one per lane (lanes 0..15)
    BoxI tb;
    tb.valid = lane < numTerrain; tb.type = 0;
    tb.min[0] = tb.min[1] = tb.min[2] = tb.max[0] = tb.max[1] = tb.max[2] = 0;
    if (tb.valid) {
        const TerrainBox t = gv.terrain[(size_t)env * MAX_TERRAIN + lane];
        tb.min[0] = t.min[0]; tb.min[1] = t.min[1]; tb.min[2] = t.min[2]; tb.max[0] = t.max[0]; tb.max[1] = t.max[1]; tb.max[2] = t.max[2];
        tb.type = t.type;
    }
    int rwx = 0, rwy = 0, rwz = 0, rwActive = 0;
    if (lane < numRewards) {
        const MovableObject r = gv.rewards_obj[(size_t)env * MAX_REWARDS + lane];
        rwx = r.x; rwy = r.y; rwz = r.z; rwActive = r.state;
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

    // ---- physics, agent by agent
#pragma unroll 1
    for (int i = 0; i < A; ++i) {
        if (A_MAX > 1 && lane >= 32 && lane < 32 + MAX_AGENTS) {   // agent capsules: slot 192 + lane, k = 3
            const int j = lane - 32;
            col[3].kind = 0;
            if (j < A && j != i) {
                col[3].kind = 2;
                col[3].lo = v3(s_ag[j].pos[0], s_ag[j].pos[1], s_ag[j].pos[2]);
                col[3].hi = v3(2 * CAP_HH, 0.0f, 0.0f);
            }
        }
        AgentState a;
        phys_load(a, s_ag[i]);
        player_step<NC>(a, col, dt);
        if (lane == 0) phys_store(s_ag[i], a);
        wave_sync();
    }

    // ---- interact: pick up / put down with the default callbacks (anything may be placed anywhere)
#pragma unroll 1
    for (int i = 0; i < A; ++i) {
        if (!(s_act[i] & ACT_INTERACT)) continue;
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
            const Bits128 solid = column_solid(lb, vx[0], vx[2]);
            const Bits128 objs = column_objects(ob, vx[0], vx[2]);
            const bool placeable = vx[1] > -120 && vx[1] < 120;
            const bool solidHere = vx[1] >= 96 || vx[1] < -32 ? false : test(solid, vx[1]);
            const bool empty = !solidHere && !test(objs, vx[1]);
            if (placeable && empty && !collidesWithAgent) {
                Bits128 occ = solid;
                occ.lo |= objs.lo; occ.hi |= objs.hi;
                vx[1] = drop_height(occ, vx[1]);
#pragma unroll
                for (int k = 0; k < 2; ++k)
                    if (oi[k] == carrying) { ob.x[k] = vx[0]; ob.y[k] = vx[1]; ob.z[k] = vx[2]; ob.state[k] = 0; }
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
        wave_sync();
    }
    wave_sync();

    // teleport above the spawn cell (FallDetectionComponent::resetAgent + controller warp); wave-uniform agent index
    auto reset_agent = [&](int i) {
        const int sx = s_ag[i].spawn[0], sy = s_ag[i].spawn[1], sz = s_ag[i].spawn[2];
        const Bits128 solid = column_solid(lb, sx, sz);
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
    };

    if (gv.scenario != SCN_EMPTY) {   // (Empty has no FallDetectionComponent: an agent that walks off the platform keeps falling)
#pragma unroll 1
        for (int i = 0; i < A; ++i)
            if (s_ag[i].pos[1] + 0.05f < -20.0f) reset_agent(i);
    }

    // ---- ObstaclesScenario::step: exit pad, lava, diamonds
    int numAgentsAtExit = 0;
#pragma unroll 1
    for (int i = 0; i < A; ++i) {
        int vx[3];
        voxel_of(v3(s_ag[i].pos[0], s_ag[i].pos[1] + 0.05f, s_ag[i].pos[2]), vx);
        const bool inside = contains(tb, vx[0], vx[1], vx[2]);
        const bool onExit = __ballot(inside && (tb.type & TERRAIN_EXIT)) != 0ull;
        const bool onLava = __ballot(inside && (tb.type & TERRAIN_LAVA)) != 0ull;
        if (onExit) {
            ++numAgentsAtExit;
            if (!s_ag[i].visited_zone) {
                const bool carries = s_ag[i].carrying >= 0;
                wave_sync();
                if (lane == 0) s_ag[i].visited_zone = 1;
                reward_team_lds(s_ag, A, 1, i, 1);
                if (carries) reward_team_lds(s_ag, A, 4, i, 1);
            }
        } else if (onLava) reset_agent(i);
        // diamonds: matched against the cell computed before a lava teleport, like the reference
        const bool got = rwActive && rwx == vx[0] && rwy == vx[1] && rwz == vx[2];
        const unsigned long long gm = __ballot(got);
        if (got) rwActive = 0;
        for (int c = __popcll(gm); c > 0; --c) reward_team_lds(s_ag, A, 3, i, 1);
    }
    if (numAgentsAtExit == A && !solved) {
        solved = 1;
        episodeSec = fmax_sel(episodeSec, episodeLen - 0.3f);
        if (lane < A) s_ag[lane].last_reward += s_ag[lane].shaping[2] * 1;   // rewardAgent(obstaclesAllAgentsAtExit) for every agent
        wave_sync();
    }

    // ---- timers / done
    episodeSec += dt;
    const float bar = fmax_sel(0.0f, (episodeLen - episodeSec) / episodeLen) * 0.24f;
    if (episodeSec >= episodeLen) done = 1;
    ++numFrames;

    // ---- write back
    if (pipe_wait) __syncthreads();
    MovableObject *gobjw = gv.objects + (size_t)env * MAX_OBJECTS;
#pragma unroll
    for (int k = 0; k < 2; ++k)
        if (ob.valid[k]) {
            MovableObject o;
            o.x = (int8_t)ob.x[k]; o.y = (int8_t)ob.y[k]; o.z = (int8_t)ob.z[k]; o.state = (int8_t)ob.state[k];
            gobjw[oi[k]] = o;
        }
    if (lane < numRewards) gv.rewards_obj[(size_t)env * MAX_REWARDS + lane].state = (int8_t)rwActive;
    if (lane == 0) {
        gh->num_frames = numFrames; gh->done = done; gh->solved = solved;
        gh->episode_sec = episodeSec; gh->bar_half_width = bar;
        gv.done[env] = (uint8_t)done;
    }
    agents_store(gv, env, A, s_ag);
    if (done && lane < A) gv.true_objective[(size_t)env * A + lane] = float(solved);   // trueObjective == solved (scenario_obstacles.hpp:34)

    // ---- the auto-reset of VectorEnv::step: the wave of a finished env swaps the next episode in right here
    if (done) {
        wave_sync();   // one wave per env: orders the stores above before the swap-in's
        swap_in_episode(gv, static_cast<const EpisodeBlob *>(gv.blobs), gv.episode_status, env, 0);
    }
}

}  // namespace tick_obstacles
}  // namespace mv
