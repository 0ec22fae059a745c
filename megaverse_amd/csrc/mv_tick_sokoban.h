// megaverse_amd/csrc/mv_tick_sokoban.h -- the sokoban tick as a device function (namespace mv::tick_sokoban): shared by the scenario's own step
// kernel (mv_step_sokoban.hip) and by the union step kernel that steps several gyms with one launch (mv_step_union.hip).
//
//
// Replaces, per env (reference paths relative to src/libs):
//   Env::step                                   env/src/env.cpp:83-152            (shared pieces: mv_physics.h)
//   SokobanScenario::step (push logic)          scenarios/src/scenario_sokoban.cpp:172-236
//   SokobanScenario::addEpisodeDrawables        scenarios/src/scenario_sokoban.cpp:243-294  (collision shape of the boxes :275-293; drawing: mv_frame.h)
//   Scenario::rewardTeam / doneWithTimer        env/include/env/scenario.hpp:114-117,259-307
//   VectorEnv::step done bookkeeping + Env::reset of finished envs (env/src/vector_env.cpp:93-105): the reset kernel below swaps in
//   the episode the host generator (mv_gen_sokoban.cpp: Boxoban level files, per-env shuffled level list) left resident in HBM.
//
// Same mapping as the Obstacles kernel (one wavefront per env, four colliders per lane): <= 128 merged layout slabs (the floor and the
// undrawn two-voxel walls; one voxel is 2 units wide), <= 80 pushable boxes, <= 8 agent capsules.  There is no ObjectStackingComponent
// and no FallDetectionComponent in this scenario: the Interact action pushes the box in front of the agent one cell further when the
// agent stands in the adjacent cell and the cell behind the box is free (no wall, no box, no agent).  Box lookups are ballots over the
// lanes that hold the boxes; wall / goal lookups read the level's 32 x 32 cell map.
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
namespace tick_sokoban {



constexpr int NC = 4;
constexpr float VOXEL = 2.0f;   // SokobanScenario: voxelSize 2 (scenario_sokoban.cpp:104-120)


// Episode swap-in: Env::reset of one env from its resident SokobanBlob (called by the env's whole wavefront)
__device__ __forceinline__ void swap_in_episode(const GymView &gv, const SokobanBlob *blobs, int *status, int env, int force_all)
{
    const int lane = lane_id();
    EnvHeader *gh = gv.hdr + env;
    const int consumed = gh->episodes_consumed;
    const SokobanBlob *b = blobs + (size_t)env * gv.spares + consumed % gv.spares;   // ring slot of episode number consumed + 1
    if (b->seq != consumed + 1) {   // the host has not delivered the next episode (mv_api.hip keeps two ahead): reported, recovered
        if (lane == 0) { gh->starved |= 1; atomicOr(&status[gv.num_envs + 1], (int)ST_STARVED); }
        return;
    }
    const int A = gv.num_agents;
    const uint4 *src = reinterpret_cast<const uint4 *>(b->boxes);
    uint4 *dst = reinterpret_cast<uint4 *>(gv.boxes + (size_t)env * MAX_BOXES);
    for (int i = lane; i < MAX_BOXES * 2; i += 64) dst[i] = src[i];
    for (int i = lane; i < MAX_OBJECTS; i += 64) gv.objects[(size_t)env * MAX_OBJECTS + i] = b->objects[i];
    {
        const uint4 *cs = reinterpret_cast<const uint4 *>(b->cells);
        uint4 *cd = reinterpret_cast<uint4 *>(gv.soko_cells + (size_t)env * (SOKO_DIM * SOKO_DIM));
        for (int i = lane; i < SOKO_DIM * SOKO_DIM / 16; i += 64) cd[i] = cs[i];
    }
    for (int k = 0; k < A; ++k) {
        float cs, sn;
        yaw_matrix(b->yaw_frand[k] * 3.14159274f * 2, cs, sn);
        if (lane == 0) {
            AgentState *a = gv.agents + (size_t)env * A + k;
            const float sx = b->spawn[k][0], sy = b->spawn[k][1], sz = b->spawn[k][2];   // agentStartingPositions: not voxel corners
            a->pos[0] = sx + 0.5f; a->pos[1] = sy + 0.0f + 1.75f; a->pos[2] = sz + 0.5f;
            a->m00 = cs; a->m02 = sn; a->m20 = -sn; a->m22 = cs;
            a->pitch = 0.0f; a->hvx = 0.0f; a->hvz = 0.0f; a->vvel = 0.0f; a->voffset = 0.0f; a->step_offset = 0.0f;
            a->jump_speed = 10.0f; a->was_jumping = 0; a->carrying = -1; a->picked_up = 0; a->visited_zone = 0;
            a->spawn[0] = (int)floorf(sx); a->spawn[1] = (int)floorf(sy); a->spawn[2] = (int)floorf(sz);
            a->last_reward = 0.0f; a->total_reward = 0.0f;
            gv.rewards[(size_t)env * A + k] = 0.0f;
            gv.actions[(size_t)env * A + k] = 0;
        }
    }
    if (lane == 0) {
        gh->L = b->dim[0]; gh->H = b->dim[1]; gh->W = b->dim[2];
        gh->bz[0] = gh->bz[1] = gh->bz[2] = gh->bz[3] = 0;
        gh->layout_color = b->floor_color; gh->wall_color = b->floor_color; gh->draw_walls = 0;
        gh->num_objects = b->num_objects; gh->num_boxes = b->num_boxes; gh->num_terrain = 0; gh->num_rewards = 0; gh->num_platforms = 0;
        gh->num_frames = 0; gh->done = 0; gh->highest_tower = 0; gh->solved = 0;
        gh->episode_sec = 0.0f; gh->episode_len = b->episode_len; gh->bz_reward = 0.0f; gh->bar_half_width = 0.24f;
        gh->episodes_consumed = consumed + 1;
        status[env] = consumed + 1;
        atomicAdd(&status[gv.num_envs], 1);
        if (force_all) gv.done[env] = 0;
    }
}

template <int A_MAX>
__device__ __forceinline__ void sokoban_tick(const GymView &gv, const int env)
{
    const int lane = lane_id();
    if (env >= gv.num_envs) return;
    const int A = gv.num_agents;

    // ---- header fields as scalars (never copy the record: see mv_step.hip)
    EnvHeader *gh = gv.hdr + env;
    const int numObjects = gh->num_objects, numBoxes = gh->num_boxes;
    int numFrames = gh->num_frames, done = gh->done, solved = gh->solved, onGoal = gh->highest_tower;   // numBoxesOnGoal
    float episodeSec = gh->episode_sec;
    const float episodeLen = gh->episode_len, lookLimit = gh->p_vertical_look_limit;
    const uint8_t *cells = gv.soko_cells + (size_t)env * (SOKO_DIM * SOKO_DIM);

    // ---- wave-resident scene: slabs (k = 0, 1), pushable boxes (k = 2: box lane; k = 3, lanes 0..15: box 64 + lane), capsules (k = 3, lanes 32..39)
    Col col[NC];
#pragma unroll
    for (int k = 0; k < NC; ++k) { col[k].kind = 0; col[k].lo = col[k].hi = v3(0, 0, 0); }
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        const int bi = lane + 64 * k;
        if (bi < numBoxes) {
            const LayoutBox b = gv.boxes[(size_t)env * MAX_BOXES + bi];
            if (b.type & VX_SOLID) {   // addBoundingBoxes scales by the voxel size (layout_utils.cpp:22-34)
                col[k].kind = 1;
                col[k].lo = v3(float(b.min[0]) * VOXEL, float(b.min[1]) * VOXEL - CAP_HH, float(b.min[2]) * VOXEL);
                col[k].hi = v3(float(b.max[0]) * VOXEL, float(b.max[1]) * VOXEL + CAP_HH, float(b.max[2]) * VOXEL);
            }
        }
    }
    const MovableObject *gobj = gv.objects + (size_t)env * MAX_OBJECTS;
    const int oi[2] = {lane, lane < 16 ? 64 + lane : -1};
    int ox[2], oy[2], oz[2];
    bool ovalid[2];
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        ovalid[k] = oi[k] >= 0 && oi[k] < numObjects;
        ox[k] = oy[k] = oz[k] = 0;
        if (ovalid[k]) {
            const MovableObject o = gobj[oi[k]];
            ox[k] = o.x; oy[k] = o.y; oz[k] = o.z;
            // scenario_sokoban.cpp:275-293: drawn half extents (0.8, 0.36, 0.8) at (x + 0.5, y + 0.2, z + 0.5) voxels; collision scale (1.15, 3, 1.15), offset
            // (0, 0.6, 0)
            const float sx = (VOXEL / 2) * 0.8f, sy = 0.45f * 0.8f;
            const float cx = (float(o.x) + 0.5f) * VOXEL, cy = (float(o.y) + 0.2f) * VOXEL + 0.6f, cz = (float(o.z) + 0.5f) * VOXEL;
            col[2 + k].kind = 1;
            col[2 + k].lo = v3(cx - sx * 1.15f, (cy - sy * 3.0f) - CAP_HH, cz - sx * 1.15f);
            col[2 + k].hi = v3(cx + sx * 1.15f, (cy + sy * 3.0f) + CAP_HH, cz + sx * 1.15f);
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

    // ---- physics, agent by agent (boxes do not move during this phase)
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

    // ---- SokobanScenario::step: pushes, in agent order
    auto cell_of = [&](V3 p, int out[3]) { out[0] = (int)floorf(p.x / VOXEL); out[1] = (int)floorf(p.y / VOXEL); out[2] = (int)floorf(p.z / VOXEL); };
    auto box_at = [&](int x, int y, int z) -> int {   // index of the box in that cell, or -1 (lowest index first)
        const unsigned long long m0 = __ballot(ovalid[0] && ox[0] == x && oy[0] == y && oz[0] == z);
        const unsigned long long m1 = __ballot(ovalid[1] && ox[1] == x && oy[1] == y && oz[1] == z);
        if (m0) return __ffsll((long long)m0) - 1;
        if (m1) return 64 + (__ffsll((long long)m1) - 1);
        return -1;
    };
    auto terrain = [&](int x, int y, int z) -> int {
        if (y != 1 || x < 0 || x >= SOKO_DIM || z < 0 || z >= SOKO_DIM) return 0;
        return (int)cells[x * SOKO_DIM + z];
    };
#pragma unroll 1
    for (int i = 0; i < A; ++i) {
        if (!(s_act[i] & ACT_INTERACT)) continue;
        AgentState a;
        phys_load(a, s_ag[i]);
        const Cam cam = camera_of(a);
        const V3 t = cam_to_world(cam, v3(0.0f, -0.44f, -1.0f));   // interactLocation
        int boxPos[3], agentPos[3];
        cell_of(t, boxPos);
        const int bi = box_at(boxPos[0], boxPos[1], boxPos[2]);
        if (bi < 0) continue;
        cell_of(v3(a.pos[0], a.pos[1] + 0.05f, a.pos[2]), agentPos);
        const int d0 = boxPos[0] - agentPos[0], d1 = boxPos[1] - agentPos[1], d2 = boxPos[2] - agentPos[2];
        if (abs(d0) + abs(d1) + abs(d2) != 1) continue;   // only from the adjacent cell
        const int w0 = boxPos[0] + d0, w1 = boxPos[1] + d1, w2 = boxPos[2] + d2;
        bool occupied = false;
        for (int j = 0; j < A; ++j) {
            int c[3];
            cell_of(v3(s_ag[j].pos[0], s_ag[j].pos[1] + 0.05f, s_ag[j].pos[2]), c);
            if (c[0] == w0 && c[1] == w1 && c[2] == w2) occupied = true;
        }
        if (occupied) continue;
        const int fromTerrain = terrain(boxPos[0], boxPos[1], boxPos[2]), toTerrain = terrain(w0, w1, w2);
        if (toTerrain == SOKO_WALL || box_at(w0, w1, w2) >= 0) continue;
#pragma unroll
        for (int k = 0; k < 2; ++k)
            if (oi[k] == bi) { ox[k] = w0; oy[k] = w1; oz[k] = w2; }
        if (fromTerrain != SOKO_GOAL && toTerrain == SOKO_GOAL) {
            ++onGoal;
            reward_team_lds(s_ag, A, 1, i, 1);
            if (onGoal == numObjects && !solved) {
                solved = 1;
                reward_team_lds(s_ag, A, 3, i, 1);
                episodeSec = fmax_sel(episodeSec, episodeLen - 0.3f);   // doneWithTimer()
            }
        } else if (fromTerrain == SOKO_GOAL && toTerrain != SOKO_GOAL) {
            --onGoal;
            reward_team_lds(s_ag, A, 2, i, 1);
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
    for (int k = 0; k < 2; ++k)
        if (ovalid[k]) {
            MovableObject o;
            o.x = (int8_t)ox[k]; o.y = (int8_t)oy[k]; o.z = (int8_t)oz[k]; o.state = 0;
            gobjw[oi[k]] = o;
        }
    if (lane == 0) {
        gh->num_frames = numFrames; gh->done = done; gh->solved = solved; gh->highest_tower = onGoal;
        gh->episode_sec = episodeSec; gh->bar_half_width = bar;
        gv.done[env] = (uint8_t)done;
    }
    agents_store(gv, env, A, s_ag);
    if (done && lane < A) gv.true_objective[(size_t)env * A + lane] = float(solved);

    // ---- the auto-reset of VectorEnv::step: the wave of a finished env swaps the next episode in right here
    if (done) {
        wave_sync();   // one wave per env: orders the stores above before the swap-in's
        swap_in_episode(gv, static_cast<const SokobanBlob *>(gv.blobs), gv.episode_status, env, 0);
    }
}

}  // namespace tick_sokoban
}  // namespace mv
