// megaverse_amd/csrc/mv_tick_tower.h -- the tower tick as a device function (namespace mv::tick_tower): shared by the scenario's own step
// kernel (mv_step.hip) and by the union step kernel that steps several gyms with one launch (mv_step_union.hip).
//
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
#pragma once
#include <hip/hip_runtime.h>

#include <algorithm>

#include "mv_actions.h"
#include "mv_agents.h"
#include "mv_frame.h"
#include "mv_math.h"
#include "mv_physics.h"
#include "mv_reset_device.h"
#include "mv_types.h"

namespace mv {
namespace tick_tower {



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
    return wave_or_u64(m);
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

__device__ __forceinline__ void voxel_of(V3 p, int out[3])
{
    out[0] = (int)floorf(p.x); out[1] = (int)floorf(p.y); out[2] = (int)floorf(p.z);
}

__device__ __forceinline__ bool in_chunk(int x, int y, int z) { return x >= 0 && x < CX && y >= 0 && y < CY && z >= 0 && z < CZ; }


// PAR (A_MAX > 1, mv_step.hip): called by ALL waves of the env's workgroup.  The controllers of one env run in action order because agents collide
// with each other (env.cpp:126) -- but two agents that cannot come within a capsule's width of each other this tick do not care about the order, nor
// about each other's position at all.  With PAR the waves share the agents out:
//   * an agent with a neighbour inside (its and the neighbour's speed) x dt + a capsule's width + 0.5 is "near": the near agents run one after the
//     other, in index order, on wave 0, and see each other's positions as the sequential loop would;
//   * every other agent runs on whichever wave has least to do, against the PRE-tick positions of all others (which it will not meet);
//   * every controller records how far it really reached (player_step's reach2); afterwards wave 0 checks every pair that was run apart:
//     |pre_i - pre_j| > reach_i + reach_j + a capsule's width + 0.1 (horizontally) -- within those circles neither capsule ever enters the other's
//     sweeps or depenetration, so the results ARE the sequential loop's.  Should a pair fail (a depenetration push longer than the margin), the
//     env's agents are restored and stepped again in a row.
// A launch lasts as long as its slowest env: that used to be A controllers in a row, now the env with the largest group of near agents.
// pipe_wait (wave-uniform; the software-pipelined multi-tick kernel, mv_step.hip: step_ticks_pipe_kernel): the env's second wave is still reading the
// state of the previous tick for that tick's frame setup -- this tick computes beside it and meets it at a workgroup barrier before it writes anything
// the frame setup reads (objects, header, agents; the chunk is the tick's alone).
template <int A_MAX, bool PAR = false>
__device__ __forceinline__ void tower_tick(const GymView &gv, const int env, const int pipe_wait = 0)
{
    static_assert(!PAR || A_MAX > 1, "one agent: nothing to share out");
    const int lane = lane_id();
    if (env >= gv.num_envs) return;
    const int A = gv.num_agents;
    const int wave = PAR ? __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)) : 0;

    MV_T_BEGIN
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
    // (the records are loaded whether or not the header's counts will keep them -- every env owns MAX_OBJECTS / MAX_BOXES entries --: the loads go out
    // together with the header's instead of one memory round trip behind it)
    static_assert(48 + 32 <= MAX_OBJECTS && TOWER_BOXES <= MAX_BOXES, "lane -> record mapping");
    MovableObject po[2] = {{0, 0, 0, 0}, {0, 0, 0, 0}};
#pragma unroll
    for (int k = 0; k < 2; ++k)
        if (oi[k] >= 0) po[k] = gobj[oi[k]];
    LayoutBox pb{{0, 0, 0}, 0, {0, 0, 0}, 0};
    if (lane < TOWER_BOXES) pb = gv.boxes[(size_t)env * MAX_BOXES + lane];
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        ob.valid[k] = oi[k] >= 0 && oi[k] < h.num_objects;
        ob.x[k] = ob.y[k] = ob.z[k] = 0; ob.state[k] = 0;
        if (ob.valid[k]) { ob.x[k] = po[k].x; ob.y[k] = po[k].y; ob.z[k] = po[k].z; ob.state[k] = po[k].state; }
    }
    auto object_collider = [&](int k) {
        if (ob.valid[k] && ob.state[k] == 0) {
            const float cx = float(ob.x[k]) + 0.5f, cy = float(ob.y[k]) + 0.5f + OBJ_COLL_YOFF, cz = float(ob.z[k]) + 0.5f;
            col[k].kind = 1;
            col[k].lo = v3(cx - OBJ_COLL_HALF, (cy - OBJ_COLL_HALF) - CAP_HH, cz - OBJ_COLL_HALF);
            col[k].hi = v3(cx + OBJ_COLL_HALF, (cy + OBJ_COLL_HALF) + CAP_HH, cz + OBJ_COLL_HALF);
        } else col[k].kind = 0;
    };
    if (lane < TOWER_BOXES) {
        if (lane < h.num_boxes) {
            const LayoutBox b = pb;
            if (b.type & VX_SOLID) {
                col[0].kind = 1;
                col[0].lo = v3(float(b.min[0]), float(b.min[1]) - CAP_HH, float(b.min[2]));
                col[0].hi = v3(float(b.max[0]), float(b.max[1]) + CAP_HH, float(b.max[2]));
            }
        }
    } else object_collider(0);
    if (lane < 32) object_collider(1);

    // ---- agents: records in LDS (mv_agents.h), one agent's physics fields in registers at a time
    __shared__ AgentState s_ag[A_MAX];
    __shared__ int s_act[A_MAX];
    const float dt = DT;
    if (!PAR || wave == 0) {
        agents_load(gv, env, A, s_ag, s_act);
        MV_T(0);   // loads
        // ---- actions -> intents (env.cpp:89-122): agents are independent here, one lane each
        if (lane < A) {
            AgentState a;
            phys_load(a, s_ag[lane]);
            apply_actions(a, s_act[lane], dt, h.p_vertical_look_limit);
            phys_store(s_ag[lane], a);
        }
        wave_sync();
        MV_T(1);   // actions -> intents
#ifdef MV_TICK_TIMING
        if (lane < 8) s_cast_dbg[lane] = 0;
        wave_sync();
#endif
    }
    // the other agents' capsules as colliders (lanes 32 .. 32 + MAX_AGENTS of the second collider slot): `live` = bit set of the agents whose CURRENT
    // record counts, the others are taken where the tick found them
    __shared__ float s_pre[A_MAX][4];   // PAR: x, y, z before the controllers ran, speed x dt
    auto capsules_for = [&](int i, unsigned live) {
        if (A_MAX > 1 && lane >= 32 && lane < 32 + MAX_AGENTS) {
            const int j = lane - 32;
            col[1].kind = 0;
            if (j < A && j != i) {
                col[1].kind = 2;
                col[1].lo = (!PAR || ((live >> j) & 1u)) ? v3(s_ag[j].pos[0], s_ag[j].pos[1], s_ag[j].pos[2]) : v3(s_pre[j][0], s_pre[j][1], s_pre[j][2]);
                col[1].hi = v3(2 * CAP_HH, 0.0f, 0.0f);
            }
        }
    };
    // ---- physics, agent by agent (controllers run in addAction order, env.cpp:126)
    auto controllers_in_a_row = [&]() {
#pragma unroll 1
        for (int i = 0; i < A; ++i) {
            capsules_for(i, ~0u);
            AgentState a;
            phys_load(a, s_ag[i]);
            player_step(a, col, dt);
            if (lane == 0) phys_store(s_ag[i], a);
            wave_sync();
        }
    };
    if (!PAR) controllers_in_a_row();
    else {
        __shared__ AgentState s_ag0[A_MAX];   // the records as the controllers found them (a failed check steps the env again from these)
        __shared__ float s_reach[A_MAX];
        const int nw = (int)(blockDim.x >> 6);
        if (wave == 0) {
            const uint32_t *src = reinterpret_cast<const uint32_t *>(s_ag);
            uint32_t *dst = reinterpret_cast<uint32_t *>(s_ag0);
            for (int q = lane; q < A * AGENT_DWORDS; q += 64) dst[q] = src[q];
            if (lane < A) {
                const AgentState &g = s_ag[lane];
                s_pre[lane][0] = g.pos[0]; s_pre[lane][1] = g.pos[1]; s_pre[lane][2] = g.pos[2];
                s_pre[lane][3] = sqrtf(g.hvx * g.hvx + g.hvz * g.hvz) * dt;
            }
        }
        __syncthreads();
        // who is near whom, and who runs where: the same small scalar computation on every wave
        const float WIDTH = 2.0f * CAP_R;
        unsigned nearMask = 0u;
        for (int i = 0; i < A; ++i)
            for (int j = i + 1; j < A; ++j) {
                const float dx = s_pre[i][0] - s_pre[j][0], dz = s_pre[i][2] - s_pre[j][2], r = (s_pre[i][3] + s_pre[j][3]) + (WIDTH + 0.5f);
                if (dx * dx + dz * dz <= r * r) nearMask |= (1u << i) | (1u << j);
            }
        nearMask = (unsigned)__builtin_amdgcn_readfirstlane((int)nearMask);
        int l0 = __popc(nearMask), l1 = 0, l2 = 0, l3 = 0;   // agents per wave (at most four waves)
        unsigned mine = wave == 0 ? nearMask : 0u;
        for (int i = 0; i < A; ++i) {
            if ((nearMask >> i) & 1u) continue;
            int w = 0, least = l0;   // the least busy wave; ties: the higher one -- wave 0 has the rest of the tick to itself
            if (nw > 1 && l1 <= least) { w = 1; least = l1; }
            if (nw > 2 && l2 <= least) { w = 2; least = l2; }
            if (nw > 3 && l3 <= least) { w = 3; least = l3; }
            l0 += w == 0; l1 += w == 1; l2 += w == 2; l3 += w == 3;
            if (w == wave) mine |= 1u << i;
        }
#pragma unroll 1
        for (int i = 0; i < A; ++i) {
            if (!((mine >> i) & 1u)) continue;   // (wave-uniform)
            const bool near = (nearMask >> i) & 1u;
            capsules_for(i, near ? nearMask : 0u);
            AgentState a;
            phys_load(a, s_ag[i]);
            float r2 = 0.0f;
            player_step(a, col, dt, &r2);
            if (lane == 0) { phys_store(s_ag[i], a); s_reach[i] = sqrtf(r2); }
            wave_sync();
        }
        __syncthreads();
        if (wave != 0) return;   // (the kernel's barrier is next)
        bool apart = true;   // every pair that did not run in sequence stayed out of each other's way
        for (int i = 0; i < A; ++i)
            for (int j = i + 1; j < A; ++j) {
                if (((nearMask >> i) & 1u) && ((nearMask >> j) & 1u)) continue;
                const float dx = s_pre[i][0] - s_pre[j][0], dz = s_pre[i][2] - s_pre[j][2], r = (s_reach[i] + s_reach[j]) + (WIDTH + 0.1f);
                if (!(dx * dx + dz * dz > r * r)) apart = false;
            }
        if (__ballot(!apart) != 0ull || gv.debug_redo) {
            const uint32_t *src = reinterpret_cast<const uint32_t *>(s_ag0);
            uint32_t *dst = reinterpret_cast<uint32_t *>(s_ag);
            for (int q = lane; q < A * AGENT_DWORDS; q += 64) dst[q] = src[q];
            wave_sync();
            controllers_in_a_row();
        }
    }

#ifdef MV_TICK_TIMING
    if (gv.dbg && lane == 0)
        for (int k = 0; k < 5; ++k) { gv.dbg[(size_t)env * 64 + 16 + 8 + k] = s_cast_dbg[k]; gv.dbg[(size_t)env * 64 + 8 + k] += s_cast_dbg[k]; }
#endif
    MV_T(2);   // physics
    // ---- scenario step: interact (component_object_stacking.hpp:45-168)
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
            const bool placeable = vx[0] >= 0 && vx[0] < CX && vx[2] >= 0 && vx[2] < CZ && vx[1] < CY;
            const unsigned long long colObj = column_objects(ob, vx[0], vx[2]);   // wave op, outside the loop
            auto has_obj = [&](int y) { return in_chunk(vx[0], y, vx[2]) && ((colObj >> (y + 32)) & 1ull); };
            const bool empty = !(vox(vx[0], vx[1], vx[2]) & VX_SOLID) && !has_obj(vx[1]);
            // the reference's grid is unbounded; a placement this build's 32 x 16 x 32 chunk cannot hold is refused AND reported
            if (!placeable && !collidesWithAgent && in_zone(h, vx[0], vx[2]) && lane == 0) atomicOr(&gv.episode_status[gv.num_envs + 1], (int)ST_CHUNK);
            if (placeable && empty && !collidesWithAgent && in_zone(h, vx[0], vx[2])) {
                for (;;) {
                    const int by = vx[1] - 1;
                    if (by < -30) break;
                    if ((vox(vx[0], by, vx[2]) & VX_SOLID) || has_obj(by)) break;
                    vx[1] = by;
                }
                const int oidx = carrying;
#pragma unroll
                for (int k = 0; k < 2; ++k)
                    if (oi[k] == oidx) { ob.x[k] = vx[0]; ob.y[k] = vx[1]; ob.z[k] = vx[2]; ob.state[k] = 0; }
                if (lane == 0 && in_chunk(vx[0], vx[1], vx[2])) chunk[(vx[1] * CZ + vx[2]) * CX + vx[0]] |= VX_OBJECT;
                if (lane == 0) s_ag[i].carrying = -1;
                const float newReward = tower_reward(h, ob);
                const float delta = newReward - h.bz_reward;
                h.bz_reward = newReward;
                wave_sync();
                reward_team_lds(s_ag, A, 3, i, delta);
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
                const int pickedBefore = s_ag[i].picked_up;
                wave_sync();
                if (lane == 0) { s_ag[i].carrying = oidx; s_ag[i].picked_up = 1; }
                wave_sync();
                if (!pickedBefore) reward_agent_lds(s_ag, 1, i, 1);
            }
        }
    }
    wave_sync();

    MV_T(3);   // interact
    // ---- fall detection (component_fall_detection.hpp:33-55): one lane per agent
    if (lane < A) {
        AgentState &a = s_ag[lane];
        if (a.pos[1] + 0.05f < -20.0f) {
            int p[3] = {a.spawn[0], a.spawn[1], a.spawn[2]};
            while ((vox(p[0], p[1], p[2]) & VX_SOLID) && p[1] < 1000) ++p[1];
            a.pos[0] = float(p[0]) + 0.5f; a.pos[1] = float(p[1]) + 0.5f; a.pos[2] = float(p[2]) + 0.5f;
            a.m00 = 1.0f; a.m02 = 0.0f; a.m20 = 0.0f; a.m22 = 1.0f;
            a.hvx = 0.0f; a.hvz = 0.0f; a.vvel = 0.0f;
        }
    }
    wave_sync();

    // ---- building-zone visit shaping (scenario_tower_building.cpp:184-198), in agent order
#pragma unroll 1
    for (int i = 0; i < A; ++i) {
        if (s_ag[i].carrying < 0) continue;
        int vx[3];
        voxel_of(v3(s_ag[i].pos[0], s_ag[i].pos[1] + 0.05f, s_ag[i].pos[2]), vx);
        if (in_zone(h, vx[0], vx[2]) && !s_ag[i].visited_zone) {
            wave_sync();
            if (lane == 0) s_ag[i].visited_zone = 1;
            reward_team_lds(s_ag, A, 2, i, 1);
        }
    }

    // ---- timers / done (env.cpp:133-151)
    h.episode_sec += dt;
    h.bar_half_width = fmax_sel(0.0f, (h.episode_len - h.episode_sec) / h.episode_len) * 0.24f;
    if (h.episode_sec >= h.episode_len) h.done = 1;
    ++h.num_frames;

    MV_T(4);   // fall detection, zone shaping, timers
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
    if (lane == 0) {
        EnvHeader *wh = gv.hdr + env;
        wh->num_frames = h.num_frames; wh->done = h.done; wh->highest_tower = h.highest_tower;
        wh->episode_sec = h.episode_sec; wh->bz_reward = h.bz_reward; wh->bar_half_width = h.bar_half_width;
        gv.done[env] = (uint8_t)h.done;
    }
    agents_store(gv, env, A, s_ag);
    if (h.done && lane < A) gv.true_objective[(size_t)env * A + lane] = float(h.highest_tower);   // vector_env.cpp:97-98

    MV_T(5);   // write back
    // ---- VectorEnv::step's auto-reset (vector_env.cpp:93-105): the wave of a finished env regenerates it right here.  About one
    // env in two thousand finishes per tick and its wave is not the slowest of the launch even with the generator on top, so
    // this costs nothing, where a separate "reset whoever is done" launch cost 4-7 us per step.
#ifndef MV_EXP_NO_AUTO_RESET   // (timing experiments only: what the generator's 20 KB of LDS cost the kernels that run beside this one)
    if (h.done) {
        wave_sync();   // one wave per env: orders the stores above before the swap-in's
        (void)tower_swap_in(gv, env, 0);   // the next episode, drawn ahead of time (mv_reset_device.h)
    }
#endif
}

}  // namespace tick_tower
}  // namespace mv
