// megaverse_amd/csrc/mv_tick_rearrange.h -- the rearrange tick as a device function (namespace mv::tick_rearrange): shared by the scenario's own step
// kernel (mv_step_rearrange.hip) and by the union step kernel that steps several gyms with one launch (mv_step_union.hip).
//
//
// Replaces, per env (reference paths relative to src/libs):
//   Env::step                                   env/src/env.cpp:83-152            (shared pieces: mv_physics.h)
//   RearrangeScenario::step / canPlaceObject / placedObject / pickedObject / checkDone / countMatchingObjects
//                                               scenarios/src/scenario_rearrange.cpp:125-180
//   ObjectStackingComponent                     scenarios/include/scenarios/component_object_stacking.hpp:45-168
//   VectorEnv::step done bookkeeping + Env::reset of finished envs (env/src/vector_env.cpp:93-105)
//
// The scene is tiny and almost entirely static: 5 room slabs, 9 static boxes (raised floor + two stepped pedestals,
// addStaticCollidingBox), the target arrangement (<= 7 static items) and its movable copy.  One wavefront per env,
// ONE collider per lane, in the order collision tie-breaks need: lanes 0-4 room, 5-13 static boxes, 14-21 target items,
// 22-29 movable items, 30-37 other agents.  The room is fixed (19 x H x 14), so "is this cell solid?" is arithmetic.
#pragma once
#include <hip/hip_runtime.h>

#include <algorithm>

#include "mv_boxlist.h"
#include "mv_actions.h"
#include "mv_agents.h"
#include "mv_frame.h"
#include "mv_math.h"
#include "mv_physics.h"
#include "mv_rearrange.h"
#include "mv_types.h"

namespace mv {
namespace tick_rearrange {



constexpr int NC = 1;
constexpr int LANE_STATIC = 5, LANE_TARGET = 14, LANE_ITEM = 22, LANE_AGENT = 30;
// VoxelGrid solidity of column (x, z): the room (floor, four walls) plus the undrawn solid layer under both work areas
// (:268-275); cells outside the 32 x 16 x 32 window count as empty like the oracle's dense chunk
__device__ __forceinline__ Bits128 column_solid_room(int x, int z, int H)
{
    Bits128 m{0ull, 0ull};
    if (x < 0 || x >= CX || z < 0 || z >= CZ) return m;
    if (x < ROOM_L && z < ROOM_W) {
        set_range(m, 0, 1);
        if (x == 0 || x == ROOM_L - 1 || z == 0 || z == ROOM_W - 1) set_range(m, 0, min(H, (int)CY));
    }
    if (abs(z - RE_LEFT_Z) <= 3 && (abs(x - RE_LEFT_X) <= 3 || abs(x - RE_RIGHT_X) <= 3)) set_range(m, 1, 2);
    return m;
}


// Episode swap-in: Env::reset of one env from its resident RearrangeBlob (called by the env's whole wavefront: by the stand-alone
// reset kernel for mv_reset, and by the tail of the step kernel for the auto-reset of VectorEnv::step, vector_env.cpp:93-105)
__device__ __forceinline__ void swap_in_episode(const GymView &gv, const RearrangeBlob *blobs, int *status, int env, int force_all)
{
    const int lane = lane_id();
    EnvHeader *gh = gv.hdr + env;
    const int consumed = gh->episodes_consumed;
    const RearrangeBlob *b = blobs + (size_t)env * gv.spares + consumed % gv.spares;   // ring slot of episode number consumed + 1
    if (b->seq != consumed + 1) {   // the host has not delivered the next episode: must never happen (mv_api.hip keeps one ahead)
        if (lane == 0) { gh->starved |= 1; atomicOr(&status[gv.num_envs + 1], (int)ST_STARVED); }
        return;
    }
    const int A = gv.num_agents;
    if (lane < TOWER_BOXES) gv.boxes[(size_t)env * gv.box_stride + lane] = b->boxes[lane];
    if (lane < MAX_ITEMS) {
        gv.items[(size_t)env * MAX_ITEMS + lane] = b->items[lane];
        gv.objects[(size_t)env * MAX_OBJECTS + lane] = b->objects[lane];
    }
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
        gh->layout_color = 0x555555; gh->wall_color = 0x555555; gh->draw_walls = b->draw_walls;
        gh->num_objects = b->num_items; gh->num_boxes = b->num_boxes; gh->num_terrain = b->num_items;
        gh->num_rewards = 0; gh->num_platforms = b->max_matching;
        gh->num_frames = 0; gh->done = 0; gh->highest_tower = 0; gh->solved = 0;
        gh->episode_sec = 0.0f; gh->episode_len = b->episode_len; gh->bz_reward = 0.0f; gh->bar_half_width = 0.24f;
        gh->episodes_consumed = consumed + 1;
        status[env] = consumed + 1;
        atomicAdd(&status[gv.num_envs], 1);
        if (force_all) gv.done[env] = 0;
    }
}

template <int A_MAX>
__device__ __forceinline__ void rearrange_tick(const GymView &gv, const int env)
{
    const int lane = lane_id();
    if (env >= gv.num_envs) return;
    const int A = gv.num_agents;

    EnvHeader *gh = gv.hdr + env;
    const int numItems = gh->num_terrain, numBoxes = gh->num_boxes, H = gh->H;
    int numFrames = gh->num_frames, done = gh->done, solved = gh->solved, maxMatching = gh->num_platforms;
    float episodeSec = gh->episode_sec;
    const float episodeLen = gh->episode_len, lookLimit = gh->p_vertical_look_limit;

    // ---- this lane's static collider / item
    Col col[NC];
    col[0].kind = 0; col[0].lo = col[0].hi = v3(0, 0, 0);
    if (lane < numBoxes && lane < LANE_STATIC) {
        const LayoutBox b = gv.boxes[(size_t)env * gv.box_stride + lane];
        if (b.type & VX_SOLID) {
            col[0].kind = 1;
            col[0].lo = v3(float(b.min[0]), float(b.min[1]) - CAP_HH, float(b.min[2]));
            col[0].hi = v3(float(b.max[0]), float(b.max[1]) + CAP_HH, float(b.max[2]));
        }
    } else if (lane >= LANE_STATIC && lane < LANE_STATIC + NUM_STATIC) {
        V3 lo, hi;
        static_box(lane - LANE_STATIC, lo, hi);
        col[0].kind = 1;
        col[0].lo = v3(lo.x, lo.y - CAP_HH, lo.z);
        col[0].hi = v3(hi.x, hi.y + CAP_HH, hi.z);
    }
    // target item k lives in lane LANE_TARGET + k AND (as the movable copy's description) in lane LANE_ITEM + k
    const bool isTarget = lane >= LANE_TARGET && lane < LANE_TARGET + MAX_ITEMS, isItem = lane >= LANE_ITEM && lane < LANE_ITEM + MAX_ITEMS;
    const int itemIdx = isTarget ? lane - LANE_TARGET : isItem ? lane - LANE_ITEM : -1;
    const bool hasItem = itemIdx >= 0 && itemIdx < numItems;
    int shape = 0, color = 0, offx = 0, offy = 0, offz = 0;
    if (hasItem) {
        const ArrangementItem it = gv.items[(size_t)env * MAX_ITEMS + itemIdx];
        shape = it.shape; color = it.color; offx = it.off[0]; offy = it.off[1]; offz = it.off[2];
    }
    int ox = 0, oy = 0, oz = 0, ostate = 0;   // movable copy (lanes LANE_ITEM..)
    if (isItem && hasItem) {
        const MovableObject o = gv.objects[(size_t)env * MAX_OBJECTS + itemIdx];
        ox = o.x; oy = o.y; oz = o.z; ostate = o.state;
    }
    if (isTarget && hasItem) {
        const V3 h = item_collision_half(shape);
        const float cx = float(offx + RE_LEFT_X) + 0.5f, cy = float(offy + RE_LEFT_Y) + 0.5f, cz = float(offz + RE_LEFT_Z) + 0.5f;
        col[0].kind = 1;
        col[0].lo = v3(cx - h.x, (cy - h.y) - CAP_HH, cz - h.z);
        col[0].hi = v3(cx + h.x, (cy + h.y) + CAP_HH, cz + h.z);
    }
    auto refresh_item_collider = [&]() {
        if (isItem && hasItem) {
            col[0].kind = 0;
            if (ostate <= 0) {
                const V3 h = item_collision_half(shape);
                const float cx = float(ox) + 0.5f, cy = float(oy) + 0.5f, cz = float(oz) + 0.5f;
                col[0].kind = 1;
                col[0].lo = v3(cx - h.x, (cy - h.y) - CAP_HH, cz - h.z);
                col[0].hi = v3(cx + h.x, (cy + h.y) + CAP_HH, cz + h.z);
            }
        }
    };
    refresh_item_collider();

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

    // ---- physics, agent by agent (items do not move during this phase)
#pragma unroll 1
    for (int i = 0; i < A; ++i) {
        if (A_MAX > 1 && lane >= LANE_AGENT && lane < LANE_AGENT + MAX_AGENTS) {
            const int j = lane - LANE_AGENT;
            col[0].kind = 0;
            if (j < A && j != i) {
                col[0].kind = 2;
                col[0].lo = v3(s_ag[j].pos[0], s_ag[j].pos[1], s_ag[j].pos[2]);
                col[0].hi = v3(2 * CAP_HH, 0.0f, 0.0f);
            }
        }
        AgentState a;
        phys_load(a, s_ag[i]);
        player_step<NC>(a, col, dt);
        if (lane == 0) phys_store(s_ag[i], a);
        wave_sync();
    }

    // ---- helpers over the movable items
    auto item_at = [&](int x, int y, int z) -> int {   // index of the standing item in that cell, or -1
        const unsigned long long m = __ballot(isItem && hasItem && ostate == 0 && ox == x && oy == y && oz == z);
        return m ? (__ffsll((long long)m) - 1) - LANE_ITEM : -1;
    };
    auto count_matching = [&]() -> int {   // countMatchingObjects :136-151
        bool match = false;
        const int rx = ox - RE_RIGHT_X, ry = oy - RE_RIGHT_Y, rz = oz - RE_RIGHT_Z;
#pragma unroll
        for (int k = 0; k < MAX_ITEMS; ++k) {
            const int src = LANE_TARGET + k;
            const int ks = __shfl(shape, src, 64), kc = __shfl(color, src, 64), kx = __shfl(offx, src, 64),
                                  ky = __shfl(offy, src, 64), kz = __shfl(offz, src, 64);
            if (k < numItems && ks == shape && kc == color && kx == rx && ky == ry && kz == rz) match = true;
        }
        return __popcll(__ballot(isItem && hasItem && ostate <= 0 && match));
    };
    auto check_done = [&](int i) {   // checkDone :165-180
        const int matches = count_matching();
        if (matches > maxMatching) { reward_team_lds(s_ag, A, 1, i, 1); maxMatching = matches; }
        if (matches >= numItems && !solved) {
            solved = 1;
            reward_team_lds(s_ag, A, 2, i, 1);
            episodeSec = fmax_sel(episodeSec, episodeLen - 0.3f);
        }
    };

    // ---- interact: pick up / put down (items may only be put down on the right work area)
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
                const bool placeable = vx[0] >= 0 && vx[0] < CX && vx[2] >= 0 && vx[2] < CZ && vx[1] < CY;
                Bits128 objs{0ull, 0ull};   // standing items of the column
                if (isItem && hasItem && ostate == 0 && ox == vx[0] && oz == vx[2]) set_range(objs, oy, oy + 1);
                objs.lo = wave_or_u64(objs.lo); objs.hi = wave_or_u64(objs.hi);
                Bits128 occ = column_solid_room(vx[0], vx[2], H);   // solid cells of the column, then | standing items
                const bool empty = !test(occ, vx[1]) && !test(objs, vx[1]);
                occ.lo |= objs.lo; occ.hi |= objs.hi;
                const bool canPlace = abs(vx[0] - RE_RIGHT_X) <= 2 && abs(vx[2] - RE_RIGHT_Z) <= 2;
                if (placeable && empty && !collidesWithAgent && canPlace) {
                    vx[1] = drop_height(occ, vx[1]);
                    if (isItem && itemIdx == carrying) { ox = vx[0]; oy = vx[1]; oz = vx[2]; ostate = 0; }
                    if (lane == 0) s_ag[i].carrying = -1;
                    wave_sync();
                    check_done(i);
                }
            } else {
                const V3 pickup = cam_to_world(cam, v3(0.0f, -0.44f, -1.0f));
                int vx[3];
                voxel_of(pickup, vx);
                const int o0 = item_at(vx[0], vx[1], vx[2]);
                const int o1 = item_at(vx[0], vx[1] + 1, vx[2]);
                const int o2 = item_at(vx[0], vx[1] + 2, vx[2]);
                int oidx = -1;
                if (o0 >= 0 && o1 < 0) oidx = o0;
                else if (o1 >= 0 && o2 < 0) oidx = o1;
                if (oidx >= 0) {
                    if (isItem && itemIdx == oidx) ostate = 1 + i;
                    if (lane == 0) s_ag[i].carrying = oidx;
                    wave_sync();
                    check_done(i);
                }
            }
        }
        wave_sync();
    }

    // ---- timers / done
    episodeSec += dt;
    const float bar = fmax_sel(0.0f, (episodeLen - episodeSec) / episodeLen) * 0.24f;
    if (episodeSec >= episodeLen) done = 1;
    ++numFrames;

    // ---- write back
    if (isItem && hasItem) {
        MovableObject o;
        o.x = (int8_t)ox; o.y = (int8_t)oy; o.z = (int8_t)oz; o.state = (int8_t)ostate;
        gv.objects[(size_t)env * MAX_OBJECTS + itemIdx] = o;
    }
    if (lane == 0) {
        gh->num_frames = numFrames; gh->done = done; gh->solved = solved; gh->num_platforms = maxMatching;
        gh->episode_sec = episodeSec; gh->bar_half_width = bar;
        gv.done[env] = (uint8_t)done;
    }
    agents_store(gv, env, A, s_ag);
    if (done && lane < A) gv.true_objective[(size_t)env * A + lane] = float(solved);   // scenario_rearrange.hpp:94

    // ---- the auto-reset of VectorEnv::step: the wave of a finished env swaps the next episode in right here
    if (done) {
        wave_sync();   // one wave per env: orders the stores above before the swap-in's
        swap_in_episode(gv, static_cast<const RearrangeBlob *>(gv.blobs), gv.episode_status, env, 0);
    }
}

}  // namespace tick_rearrange
}  // namespace mv
