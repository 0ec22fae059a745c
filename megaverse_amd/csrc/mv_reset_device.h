// megaverse_amd/csrc/mv_reset_device.h -- TowerBuilding episode (re)generation as a device function, so that both the
// stand-alone reset kernel (mv_reset, force_all) and the tail of the step kernel (auto-reset of a finished env) run it.
//
// Replaces, per env:  Env::reset (reference: src/libs/env/src/env.cpp:57-76)
//   -> TowerBuildingScenario::reset + TowerBuildingPlatform::init/generate
//      (src/libs/scenarios/src/scenario_tower_building.cpp:19-89,129-154)
//   -> VoxelGridComponent::addPlatform / toBoundingBoxes (component_voxel_grid.hpp:73-187)
//   -> ObjectStackingComponent::addDrawablesAndCollisions (component_object_stacking.hpp:170-198)
//   -> DefaultScenario::spawnAgents (scenario_default.hpp:80-97)
//
// One wavefront per env.  The RNG-dependent part is inherently serial and runs as uniform scalar code (mv_rng.h); voxel
// fill and write-out are lane-parallel through a 16 KiB LDS image of the chunk so HBM only sees full-width coalesced
// dwordx4 stores.
#pragma once
#include <hip/hip_runtime.h>

#include "mv_math.h"
#include "mv_rng.h"
#include "mv_types.h"

namespace mv {
namespace {

__constant__ unsigned LAYOUT_COLORS[14] = {  // reference: env/include/env/const.hpp:121-136
    0xffffff, 0xffffe6, 0xccffcc, 0xe6ecff, 0xd9d9d9, 0xffebcc, 0xb3b3b3, 0xb3b3b3, 0xb3b3b3, 0xb3b3b3,
    0x555555, 0x555555, 0x555555, 0x555555};

__device__ __forceinline__ unsigned random_layout_color(Mt19937 &g) { return LAYOUT_COLORS[rand_range(g, 0, 14)]; }

__device__ __forceinline__ bool in_zone(const int bz[4], int x, int z) { return x >= bz[0] && x < bz[1] && z >= bz[2] && z < bz[3]; }

__device__ __forceinline__ float building_reward_coeff(int h)
{   // scenario_tower_building.cpp:246-251; 2^h is exact
    float res = float(h) * 0.05f;
    const float p = 0.05f * __uint_as_float((unsigned)(127 + h) << 23);
    res += fmin_sel(p, 20.0f);
    return res;
}

// ---- Env::reset in two halves -----------------------------------------------------------------------------------------------------------
// tower_draw:     everything that consumes the env's random stream -- the serial part: seeding mt19937 is 623 dependent steps, the whole draw ~47 us
//                 of one wavefront, 7 KB of LDS -- into a TowerBlob of the env's ring of resident episodes.  Runs in tower_draw_kernel, launched
//                 behind the step kernels on a stream of its own (mv_api.hip), never inside a tick.
// tower_swap_in:  the blob -> voxel chunk, objects, boxes, header, agents: lane-parallel copies, a few microseconds, no LDS.  Runs at the tail of a
//                 finishing env's tick (VectorEnv::step's auto-reset, vector_env.cpp:93-105) and in the reset kernel (mv_reset).
// Until round 5 both halves ran inside the finishing env's tick: its wave was the launch's straggler by ~47 us, and every step workgroup -- resident
// for a whole batched call beside the observation passes -- carried the generator's LDS.  The episodes are the same, byte for byte: the draws happen
// in the reference's order (the parts of Env::reset that were moved behind them consume nothing).

// Draws episode number `seq` of env `env` (the next one of its generator: tg->generated + 1) into its ring slot.  Called by all 64 lanes of the env's
// wavefront, which must be the whole workgroup (the function orders its LDS traffic with wave_sync()).
__device__ __forceinline__ void tower_draw(const GymView &gv, int env, int seq)
{
    const int lane = lane_id();
    TowerGen *tg = gv.tower_gen + env;
    TowerBlob *blob = const_cast<TowerBlob *>(reinterpret_cast<const TowerBlob *>(gv.blobs)) + (size_t)env * gv.spares + (seq - 1) % gv.spares;
    __shared__ __attribute__((aligned(16))) uint32_t s_mt[624];
    __shared__ __attribute__((aligned(16))) uint16_t s_cand[32 * 32];
    __shared__ __attribute__((aligned(16))) uint16_t s_steps[32 * 32];   // the shuffle's swap partners
    __shared__ MovableObject s_obj[MAX_OBJECTS];

    const int A = gv.num_agents;
    Mt19937 g{s_mt, 624};

    // ---- Env::reset: seed = randRange(0, 1<<30, rng); rng.seed(seed)  (env.cpp:61-62)
    uint32_t seed = tg->seed;
    if (tg->seed_is_env_seed) {
        mt_seed(g, seed);
        seed = (uint32_t)rand_range(g, 0, 1 << 30);
    }
    mt_seed(g, seed);

    // ---- TowerBuildingScenario::reset (:139-141)
    unsigned layoutColor = random_layout_color(g);
    while (layoutColor == 0x555555u) layoutColor = random_layout_color(g);

    // ---- TowerBuildingPlatform::init (:19-39)
    const int height = rand_range(g, 5, 7);
    int length = rand_range(g, 12, 30);
    int width = rand_range(g, 12, 25);
    const int bzL = rand_range(g, 3, 9), bzW = rand_range(g, 3, 9);
    const int matL = rand_range(g, 2, 8), matW = rand_range(g, 2, 8);
    length = max(bzL + matL + 3, length);
    width = max(bzW + matW + 3, width);
    const int bzX = rand_range(g, 1, length - bzL - 1), bzZ = rand_range(g, 1, width - bzW - 1);
    const int matX = rand_range(g, 1, length - matL - 1), matZ = rand_range(g, 1, width - matW - 1);

    // spawn candidates x-major (:41-44), shuffled (:46)
    // (only the first A + 25 entries of the shuffled list are ever used -- agents, then the random objects: they are computed directly,
    // mv_rng.h: shuffle_prefix_u16; the full shuffle remains for the rare draw the direct form cannot take)
    const int nz = width - 2, ncand = (length - 2) * nz;
    auto candidate = [nz](int i) { const int x = 1 + i / nz, z = 1 + i % nz; return (uint16_t)((x << 8) | z); };
    static_assert(MAX_AGENTS + 25 <= 64, "one lane per entry of the shuffled prefix");
    if (!shuffle_prefix_u16(g, ncand, min(ncand, (int)MAX_AGENTS + 25), s_steps, s_cand, candidate)) {
        for (int i = lane; i < ncand; i += 64) s_cand[i] = candidate(i);
        wave_sync();
        shuffle_u16(g, s_cand, ncand);
    }

    const int nSpawn = min(A, ncand);
    const int maxRandomObjects = min(ncand - A, 25);
    const int spawnObjects = rand_range(g, 0, max(1, maxRandomObjects));
    const int numObjects = min(spawnObjects + matL * matW, (int)MAX_OBJECTS);

    // objects: the random ones (:53-66) then the materials rectangle (:68-72)
    for (int i = lane; i < MAX_OBJECTS; i += 64) {
        MovableObject o{0, 0, 0, 0};
        if (i < spawnObjects) {
            const uint16_t c = s_cand[nSpawn + i];
            const int x = c >> 8, z = c & 255;
            const bool inMat = x >= matX && x < matX + matL && z >= matZ && z < matZ + matW;
            o.x = (int8_t)x; o.y = (int8_t)(inMat ? 2 : 1); o.z = (int8_t)z;
        } else if (i < numObjects) {
            const int j = i - spawnObjects;
            o.x = (int8_t)(matX + j / matW); o.y = 1; o.z = (int8_t)(matZ + j % matW);
        }
        s_obj[i] = o;
        blob->objects[i] = o;
    }
    wave_sync();

    // vg.addPlatform(platform, layoutColor, randomLayoutColor(rng), randomBool(rng)) (:145); the
    // reference is built with GCC, which evaluates the arguments right to left.
    const bool drawWalls = random_bool(g);
    const unsigned wallColor = random_layout_color(g);

    const int bz[4] = {bzX, bzX + bzL, bzZ, bzZ + bzW};

    // initial tower reward, summed in object order (scenario_tower_building.cpp:166-173,232-241)
    float bzReward = 0.0f;
    for (int i = 0; i < numObjects; ++i) {
        const MovableObject o = s_obj[i];
        if (in_zone(bz, o.x, o.z)) bzReward += building_reward_coeff(o.y);
    }

    // ---- agents (scenario_default.hpp:80-97, agent.cpp:24-65); one frand per agent, in order
    for (int k = 0; k < A; ++k) {
        const uint16_t c = s_cand[k < nSpawn ? k : 0];
        const float rot = frand(g) * 3.14159274f * 2;
        if (lane == 0) { blob->spawn[k] = (int32_t)c; blob->yaw_rot[k] = rot; }
    }

    // value the next Env::reset() will draw; nothing consumes the env rng during an episode
    const uint32_t nextSeed = (uint32_t)rand_range(g, 0, 1 << 30);

    // `seq` PUBLISHES the episode: every lane's stores to the blob (objects, spawns) are released at agent scope before lane 0 writes it, and tower_swap_in
    // acquires after it has read it -- the host's launch order (the draw before the last one is waited for) already keeps a step kernel off a blob that is
    // being drawn; should a starved env ever race a running draw, it sees seq unset (ST_STARVED) or the whole episode, never a torn one.  Once per
    // episode.
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    wave_sync();
    if (lane == 0) {
        blob->L = length; blob->H = height; blob->W = width;
        blob->bz[0] = bz[0]; blob->bz[1] = bz[1]; blob->bz[2] = bz[2]; blob->bz[3] = bz[3];
        blob->layout_color = (int)layoutColor; blob->wall_color = (int)wallColor; blob->draw_walls = drawWalls ? 1 : 0;
        blob->num_objects = numObjects;
        blob->bz_reward = bzReward;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        __hip_atomic_store(&blob->seq, seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        tg->seed = nextSeed; tg->seed_is_env_seed = 0; tg->generated = seq;
    }
    wave_sync();   // (the LDS arrays are reused by the next draw of this wave)
}

// Env::reset of env `env` from the next episode of its ring.  Called by all 64 lanes of ONE wavefront (stores are ordered with wave_sync()).
// -> false: the episode is not resident -- its draw kernel has not run yet, which the host's launch order rules out (mv_api.hip) -- the env keeps its
// done state, repeats its done step, and ST_STARVED is raised (reported by mv_step like the host-generated scenarios' starvation).
__device__ __forceinline__ bool tower_swap_in(const GymView &gv, int env, int force_all)
{
    const int lane = lane_id();
    EnvHeader *hdr = gv.hdr + env;
    const int consumed = hdr->episodes_consumed;
    // ring slot of episode number consumed + 1
    const TowerBlob *b = reinterpret_cast<const TowerBlob *>(gv.blobs) + (size_t)env * gv.spares + consumed % gv.spares;
    if (__hip_atomic_load(&b->seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != consumed + 1) {
        if (lane == 0) { hdr->starved |= 1; atomicOr(&gv.episode_status[gv.num_envs + 1], (int)ST_STARVED); }
        return false;
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");   // (pairs with tower_draw's release before it published seq)
    const int A = gv.num_agents;
    const int length = b->L, height = b->H, width = b->W, numObjects = b->num_objects;
    const bool drawWalls = b->draw_walls != 0;

    // ---- voxel chunk: floor, then the four walls override (platforms.hpp:167-190, component_voxel_grid.hpp:73-90).  One 16-cell run per lane
    // per round, formed in registers and stored straight to the env's chunk (coalesced 16 B / lane); the object bits are then or-ed into
    // their bytes in global memory.
    uint8_t *gbytes = gv.chunk + (size_t)env * CHUNK_BYTES;
    uint4 *gchunk = reinterpret_cast<uint4 *>(gbytes);
    const uint32_t vFloor = VX_SOLID | VX_OPAQUE;
    const uint32_t vWall = VX_SOLID | (drawWalls ? VX_OPAQUE : 0) | (1u << VX_COLOR_SHIFT);
    static_assert(CHUNK_BYTES / 16 == 64 * CY && CX == 32, "one layer of the chunk per round of the wave");
    for (int y = 0; y < CY; ++y) {   // (a round of the wave is one layer: y is uniform)
        const int grp = y * 64 + lane;
        const int x0 = (grp & 1) * 16, z = (grp >> 1) & (CZ - 1);
        uint32_t w[4] = {0u, 0u, 0u, 0u};
        // the layers above the walls are empty: ten rounds of sixteen store zeros without forming them cell by cell (~1000 vector instructions less in a
        // finishing env's tick; r12v: the closed loop's step launch stays at 18.2 us -- it is the diversified states of a long run that make it longer than
        // the 15.7 us of the first ticks after a reset, not the swap-ins)
        if (y < height || y == 0) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            uint32_t word = 0;
#pragma unroll
            for (int bb = 0; bb < 4; ++bb) {
                const int x = x0 + q * 4 + bb;
                uint32_t v = 0;
                if (x < length && z < width) {
                    if (y == 0) v = vFloor;
                    if (y < height && (x == 0 || x == length - 1 || z == 0 || z == width - 1)) v = vWall;
                }
                word |= v << (8 * bb);
            }
            w[q] = word;
        }
        }
        gchunk[grp] = make_uint4(w[0], w[1], w[2], w[3]);
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");   // the fill before the bytes below are read back (other lanes' stores)
    wave_sync();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    MovableObject *gobj = gv.objects + (size_t)env * MAX_OBJECTS;
    for (int i = lane; i < MAX_OBJECTS; i += 64) {
        const MovableObject o = b->objects[i];   // (zero beyond num_objects)
        gobj[i] = o;
        if (i < numObjects) {
            volatile uint8_t *cell = gbytes + (o.y * CZ + o.z) * CX + o.x;
            *cell = (uint8_t)(*cell | VX_OBJECT);   // distinct cells, byte-wide read-modify-write
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    wave_sync();

    // canonical layout parallelepipeds (== the generic greedy merge the oracle runs on the voxels:
    // keys sorted by (type, slot), scan y,z,x, grow x then z then y).  For this room that is the
    // interior floor slab and four wall slabs.
    if (lane < TOWER_BOXES) {
        LayoutBox bx{{0, 0, 0}, 0, {0, 0, 0}, 0};
        const int wallType = VX_SOLID | (drawWalls ? VX_OPAQUE : 0);
        const int floorIdx = drawWalls ? 0 : 4;       // key 12 vs wall key 13 (drawn) / 5 (invisible)
        const int wallIdx = drawWalls ? lane - 1 : lane;
        if (lane == floorIdx) {
            bx.min[0] = 1; bx.min[1] = 0; bx.min[2] = 1; bx.max[0] = length - 1; bx.max[1] = 1; bx.max[2] = width - 1;
            bx.type = VX_SOLID | VX_OPAQUE; bx.slot = 0;
        } else if (lane < 5) {
            bx.type = wallType; bx.slot = 1; bx.min[1] = 0; bx.max[1] = height;
            if (wallIdx == 0) { bx.min[0] = 0; bx.min[2] = 0; bx.max[0] = length; bx.max[2] = 1; }
            if (wallIdx == 1) { bx.min[0] = 0; bx.min[2] = 1; bx.max[0] = 1; bx.max[2] = width; }
            if (wallIdx == 2) { bx.min[0] = length - 1; bx.min[2] = 1; bx.max[0] = length; bx.max[2] = width; }
            if (wallIdx == 3) { bx.min[0] = 1; bx.min[2] = width - 1; bx.max[0] = length - 1; bx.max[2] = width; }
        }
        gv.boxes[(size_t)env * MAX_BOXES + lane] = bx;
    }

    // ---- agents (scenario_default.hpp:80-97, agent.cpp:24-65)
    for (int k = 0; k < A; ++k) {
        const int c = b->spawn[k];
        const int sx = c >> 8, sz = c & 255;
        float cs, sn;
        yaw_matrix(b->yaw_rot[k], cs, sn);
        if (lane == 0) {
            AgentState *ga = gv.agents + (size_t)env * A + k;
            AgentState a = *ga;   // keeps the per-agent reward shaping
            a.pos[0] = float(sx) + 0.5f; a.pos[1] = 2.0f + 0.0f + 1.75f; a.pos[2] = float(sz) + 0.5f;
            a.m00 = cs; a.m02 = sn; a.m20 = -sn; a.m22 = cs;
            a.pitch = 0.0f; a.hvx = 0.0f; a.hvz = 0.0f; a.vvel = 0.0f; a.voffset = 0.0f; a.step_offset = 0.0f;
            a.jump_speed = 10.0f; a.was_jumping = 0; a.carrying = -1; a.picked_up = 0; a.visited_zone = 0;
            a.spawn[0] = sx; a.spawn[1] = 2; a.spawn[2] = sz;
            a.last_reward = 0.0f; a.total_reward = 0.0f;
            *ga = a;
            gv.rewards[(size_t)env * A + k] = 0.0f;   // EnvState::reset zero-fills lastReward (env.hpp:141-143)
            gv.actions[(size_t)env * A + k] = 0;
        }
    }

    if (lane == 0) {
        EnvHeader h = *hdr;
        h.L = length; h.H = height; h.W = width;
        h.bz[0] = b->bz[0]; h.bz[1] = b->bz[1]; h.bz[2] = b->bz[2]; h.bz[3] = b->bz[3];
        h.layout_color = b->layout_color; h.wall_color = b->wall_color; h.draw_walls = drawWalls ? 1 : 0;
        h.num_objects = numObjects; h.num_boxes = 5;
        h.num_frames = 0; h.done = 0; h.highest_tower = 0;
        h.episode_sec = 0.0f;
        h.episode_len = h.p_episode_len_sec + 4.0f * float(numObjects);   // :263-266
        h.bz_reward = b->bz_reward;
        h.bar_half_width = 0.24f;
        h.episodes_consumed = consumed + 1;
        *hdr = h;
        if (force_all) gv.done[env] = 0;
    }
    return true;
}

}  // namespace
}  // namespace mv
