// megaverse_amd/csrc/mv_api_debug.hip -- test hooks of the C ABI (include/megaverse_hip.h: mv_debug_*): state snapshots in the oracle's layout, pose setters,
// the host-side episode generators and the feeder without a device, the device's RNG helpers and single fp32 operations.  Used by tests/ only.
#include "mv_api_internal.h"

__global__ void set_agent_pos_kernel(AgentState *agents, int idx, float x, float y,
                                     float z) { agents[idx].pos[0] = x; agents[idx].pos[1] = y; agents[idx].pos[2] = z; }
__global__ void set_agent_yaw_kernel(AgentState *agents, int idx, float c,
                                     float s) { agents[idx].m00 = c; agents[idx].m02 = s; agents[idx].m20 = -s; agents[idx].m22 = c; }
__global__ void set_agent_velocity_kernel(AgentState *agents, int idx, float hvx, float hvz,
                                          float vvel) { agents[idx].hvx = hvx; agents[idx].hvz = hvz; agents[idx].vvel = vvel; }

__global__ void debug_rng_kernel(uint32_t seed, int what, const int32_t *lo, const int32_t *hi, int n, void *out)
{
    __shared__ uint32_t s_mt[624];
    __shared__ uint16_t s_items[4096];
    Mt19937 g{s_mt, 624};
    mt_seed(g, seed);
    const int lane = threadIdx.x & 63;
    if (what == 0) {
        for (int i = 0; i < n; ++i) { const uint32_t v = mt_next(g); if (lane == 0) ((uint32_t *)out)[i] = v; }
    } else if (what == 1) {
        for (int i = 0; i < n; ++i) { const int v = rand_range(g, lo[i], hi[i]); if (lane == 0) ((int32_t *)out)[i] = v; }
    } else if (what == 2) {
        for (int i = 0; i < n; ++i) { const float v = frand(g); if (lane == 0) ((float *)out)[i] = v; }
    } else if (what == 3) {
        for (int i = lane; i < n; i += 64) s_items[i] = (uint16_t)i;
        __syncthreads();
        shuffle_u16(g, s_items, n);
        for (int i = lane; i < n; i += 64) ((int32_t *)out)[i] = s_items[i];
    }
}

__global__ void debug_math_kernel(int what, const float *a, const float *b, int n, float *out)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    if (what == 0) out[i] = a[i] / b[i];
    else if (what == 1) out[i] = sqrtf(a[i]);
    else if (what == 2) { float s, c; sincos_poly(a[i], s, c); out[2 * i] = s; out[2 * i + 1] = c; }
    else if (what == 3) out[i] = a[i] * b[i] + a[i];   // must NOT be contracted into an fma
    else if (what == 4) out[i] = floorf(a[i]);
}

extern "C" {

// ---- test hooks ---------------------------------------------------------------------------------
#pragma pack(push, 4)
struct SnapAgent {
    float pos[3], basis[4], pitch, hv[2], vvel, voffset, step_offset, jump_speed;
    int32_t was_jumping, carrying, picked_up, visited_zone, spawn[3];
    float last_reward, total_reward, shaping[NUM_SHAPING];
};
struct Snap {
    int32_t scenario, L, H, W, bz[4], layout_color, wall_color, draw_walls, num_objects, num_boxes, num_frames, done, highest_tower,
        num_agents, num_terrain, num_rewards, num_platforms, solved;
    float episode_sec, episode_len, bz_reward, bar_half_width;
    int32_t boxes[COLLECT_MAX_BOXES][8];
    int32_t terrain[MAX_TERRAIN][8];
    int8_t objects[MAX_OBJECTS][4];
    int8_t rewards[COLLECT_MAX_REWARDS][4];
    SnapAgent agents[MAX_AGENTS];
    uint8_t chunk[CHUNK_BYTES];
    int8_t heightmap[HM_DIM * HM_DIM];
    int32_t num_items, items[MAX_ITEMS][5];
    uint8_t soko[32 * 32];   // Sokoban level cells
    int32_t hex_num_boxes, hex_num_objs;
    float hex_target[3];
    HexRec hex_boxes[HEX_MAX_BOXES], hex_objs[HEX_MAX_OBJS];
};
#pragma pack(pop)

int mv_debug_set_agent_pos(mv_gym *g, int32_t env, int32_t agent, float x, float y, float z)
{
    if (check(g)) return -1;
    if (env < 0 || env >= g->N || agent < 0 || agent >= g->A) return fail("mv_debug_set_agent_pos: index out of range");
    if (sim_join(g)) return -1;
    hipLaunchKernelGGL(set_agent_pos_kernel, dim3(1), dim3(1), 0, g->stream, g->gv.agents, env * g->A + agent, x, y, z);
    HIP_TRY(hipGetLastError());
    return 0;
}

int mv_debug_set_agent_yaw(mv_gym *g, int32_t env, int32_t agent, float c, float s)
{
    if (check(g)) return -1;
    if (env < 0 || env >= g->N || agent < 0 || agent >= g->A) return fail("mv_debug_set_agent_yaw: index out of range");
    if (sim_join(g)) return -1;
    hipLaunchKernelGGL(set_agent_yaw_kernel, dim3(1), dim3(1), 0, g->stream, g->gv.agents, env * g->A + agent, c, s);
    HIP_TRY(hipGetLastError());
    return 0;
}

int mv_debug_set_agent_velocity(mv_gym *g, int32_t env, int32_t agent, float hvx, float hvz, float vvel)
{
    if (check(g)) return -1;
    if (env < 0 || env >= g->N || agent < 0 || agent >= g->A) return fail("mv_debug_set_agent_velocity: index out of range");
    if (sim_join(g)) return -1;
    hipLaunchKernelGGL(set_agent_velocity_kernel, dim3(1), dim3(1), 0, g->stream, g->gv.agents, env * g->A + agent, hvx, hvz, vvel);
    HIP_TRY(hipGetLastError());
    return 0;
}

int mv_debug_snapshot_size(const mv_gym *) { return (int)sizeof(Snap); }

int mv_debug_snapshot(mv_gym *g, int32_t env, void *out)
{
    if (check(g)) return -1;
    if (env < 0 || env >= g->N) return fail("mv_debug_snapshot: index out of range");
    HIP_TRY(hipStreamSynchronize(g->simStream));
    HIP_TRY(hipStreamSynchronize(g->stream));
    EnvHeader h;
    std::vector<LayoutBox> boxes(g->gv.box_stride);
    std::vector<MovableObject> objs(MAX_OBJECTS);
    std::vector<AgentState> ag(g->A);
    Snap *s = new Snap();
    std::memset(s, 0, sizeof *s);
    hipError_t e = hipMemcpy(&h, g->gv.hdr + env, sizeof h, hipMemcpyDeviceToHost);
    if (e == hipSuccess) e = hipMemcpy(boxes.data(), g->gv.boxes + (size_t)env * g->gv.box_stride, g->gv.box_stride * sizeof(LayoutBox), hipMemcpyDeviceToHost);
    if (e == hipSuccess) e = hipMemcpy(objs.data(), g->gv.objects + (size_t)env * MAX_OBJECTS, MAX_OBJECTS * sizeof(MovableObject), hipMemcpyDeviceToHost);
    if (e == hipSuccess) e = hipMemcpy(ag.data(), g->gv.agents + (size_t)env * g->A, g->A * sizeof(AgentState), hipMemcpyDeviceToHost);
    if (e == hipSuccess && g->gv.chunk) e = hipMemcpy(s->chunk, g->gv.chunk + (size_t)env * CHUNK_BYTES, CHUNK_BYTES, hipMemcpyDeviceToHost);
    std::vector<TerrainBox> terr(MAX_TERRAIN);
    std::vector<MovableObject> rew(g->gv.reward_stride);
    if (e == hipSuccess && g->gv.terrain) e = hipMemcpy(terr.data(), g->gv.terrain + (size_t)env * MAX_TERRAIN,
        MAX_TERRAIN * sizeof(TerrainBox), hipMemcpyDeviceToHost);
    if (e == hipSuccess && g->gv.rewards_obj) e = hipMemcpy(rew.data(), g->gv.rewards_obj + (size_t)env * g->gv.reward_stride,
        g->gv.reward_stride * sizeof(MovableObject), hipMemcpyDeviceToHost);
    if (e == hipSuccess && g->gv.items) {
        std::vector<ArrangementItem> its(MAX_ITEMS);
        e = hipMemcpy(its.data(), g->gv.items + (size_t)env * MAX_ITEMS, MAX_ITEMS * sizeof(ArrangementItem), hipMemcpyDeviceToHost);
        s->num_items = h.num_terrain;
        for (int i = 0; i < h.num_terrain && i < MAX_ITEMS; ++i) {
            s->items[i][0] = its[i].shape; s->items[i][1] = its[i].color;
            s->items[i][2] = its[i].off[0]; s->items[i][3] = its[i].off[1]; s->items[i][4] = its[i].off[2];
        }
    }
    if (e == hipSuccess && g->gv.soko_cells) e = hipMemcpy(s->soko, g->gv.soko_cells + (size_t)env * (SOKO_DIM * SOKO_DIM),
        SOKO_DIM * SOKO_DIM, hipMemcpyDeviceToHost);
    std::memset(s->heightmap, 0xff, sizeof s->heightmap);
    if (e == hipSuccess && g->gv.heightmap) e = hipMemcpy(s->heightmap, g->gv.heightmap + (size_t)env * HM_BYTES, sizeof s->heightmap, hipMemcpyDeviceToHost);
    if (e != hipSuccess) { delete s; return fail(std::string("mv_debug_snapshot: ") + hipGetErrorString(e)); }
    if (h.scenario == SCN_REARRANGE) h.num_terrain = 0;   // (the header reuses it for the item count, reported as num_items)
    if (e == hipSuccess && g->gv.hex_boxes) {   // Hex*: the header's box / collider / reward counts describe the hex lists
        s->hex_num_boxes = h.num_boxes; s->hex_num_objs = h.num_rewards;
        s->hex_target[0] = h.hex_target[0]; s->hex_target[1] = 0.0f; s->hex_target[2] = h.hex_target[1];
        e = hipMemcpy(s->hex_boxes, g->gv.hex_boxes + (size_t)env * HEX_MAX_BOXES,
                      (size_t)std::min(h.num_boxes, (int)HEX_MAX_BOXES) * sizeof(HexRec), hipMemcpyDeviceToHost);
        if (e == hipSuccess) e = hipMemcpy(s->hex_objs, g->gv.hex_objs + (size_t)env * HEX_MAX_OBJS,
            (size_t)std::min(h.num_rewards, (int)HEX_MAX_OBJS) * sizeof(HexRec), hipMemcpyDeviceToHost);
        if (e != hipSuccess) { delete s; return fail(std::string("mv_debug_snapshot: ") + hipGetErrorString(e)); }
        h.num_boxes = 0; h.num_rewards = 0; h.num_terrain = 0;
    }
    s->scenario = h.scenario; s->num_terrain = h.num_terrain; s->num_rewards = h.num_rewards; s->num_platforms = h.num_platforms; s->solved = h.solved;
    for (int i = 0; i < h.num_terrain && i < MAX_TERRAIN; ++i) {
        const TerrainBox &t = terr[i];
        int32_t *o = s->terrain[i];
        o[0] = t.min[0]; o[1] = t.min[1]; o[2] = t.min[2]; o[3] = t.max[0]; o[4] = t.max[1]; o[5] = t.max[2]; o[6] = t.type; o[7] = 0;
    }
    for (int i = 0; i < h.num_rewards && i < g->gv.reward_stride; ++i) {
        s->rewards[i][0] = rew[i].x; s->rewards[i][1] = rew[i].y; s->rewards[i][2] = rew[i].z; s->rewards[i][3] = rew[i].state;
    }
    s->L = h.L; s->H = h.H; s->W = h.W;
    for (int i = 0; i < 4; ++i) s->bz[i] = h.bz[i];
    s->layout_color = h.layout_color; s->wall_color = h.wall_color; s->draw_walls = h.draw_walls;
    s->num_objects = h.num_objects; s->num_boxes = h.num_boxes; s->num_frames = h.num_frames; s->done = h.done;
    s->highest_tower = h.highest_tower; s->num_agents = g->A;
    s->episode_sec = h.episode_sec; s->episode_len = h.episode_len; s->bz_reward = h.bz_reward; s->bar_half_width = h.bar_half_width;
    for (int i = 0; i < h.num_boxes && i < g->gv.box_stride; ++i) {
        const LayoutBox &b = boxes[i];
        int32_t *o = s->boxes[i];
        o[0] = b.min[0]; o[1] = b.min[1]; o[2] = b.min[2]; o[3] = b.max[0]; o[4] = b.max[1]; o[5] = b.max[2]; o[6] = b.type; o[7] = b.slot;
    }
    for (int i = 0; i < h.num_objects && i < MAX_OBJECTS; ++i) {
        s->objects[i][0] = objs[i].x; s->objects[i][1] = objs[i].y; s->objects[i][2] = objs[i].z; s->objects[i][3] = objs[i].state;
    }
    for (int i = 0; i < g->A; ++i) {
        const AgentState &a = ag[i];
        SnapAgent &o = s->agents[i];
        o.pos[0] = a.pos[0]; o.pos[1] = a.pos[1]; o.pos[2] = a.pos[2];
        o.basis[0] = a.m00; o.basis[1] = a.m02; o.basis[2] = a.m20; o.basis[3] = a.m22;
        o.pitch = a.pitch; o.hv[0] = a.hvx; o.hv[1] = a.hvz; o.vvel = a.vvel; o.voffset = a.voffset;
        o.step_offset = a.step_offset; o.jump_speed = a.jump_speed; o.was_jumping = a.was_jumping; o.carrying = a.carrying;
        o.picked_up = a.picked_up; o.visited_zone = a.visited_zone;
        for (int k = 0; k < 3; ++k) o.spawn[k] = a.spawn[k];
        o.last_reward = a.last_reward; o.total_reward = a.total_reward;
        for (int k = 0; k < NUM_SHAPING; ++k) o.shaping[k] = a.shaping[k];
    }
    std::memcpy(out, s, sizeof *s);
    delete s;
    return 0;
}

// Host-only test hook (no device needed): the n-th episode an env seeded with `env_seed` generates, as the raw
// blob the reset kernel consumes (EpisodeBlob for the Obstacles family, CollectBlob for Collect).
int mv_debug_generate_episode(const char *scenario_name, int32_t num_agents, int32_t env_seed, int32_t n, float base_episode_len,
                              void *out, int32_t out_bytes)
{
    int scenario = SCN_TOWER;
    ObstacleConfig oc;
    if (!scenario_name || !scenario_from_name(lower(scenario_name), scenario, oc) || scenario == SCN_TOWER || scenario == SCN_SOKOBAN || scenario == SCN_EMPTY)
        return fail("mv_debug_generate_episode: the Obstacles family, Collect, Rearrange, HexMemory and HexExplore (Sokoban: mv_debug_generate_sokoban)");
    if (num_agents < 1 || num_agents > MAX_AGENTS || n < 1) return fail("mv_debug_generate_episode: bad arguments");
    const bool hex = scenario == SCN_HEX_MEMORY || scenario == SCN_HEX_EXPLORE;
    const size_t bytes = scenario == SCN_COLLECT ? sizeof(CollectBlob) : scenario == SCN_REARRANGE
                                                          ? sizeof(RearrangeBlob) : hex ? sizeof(HexBlob) : sizeof(EpisodeBlob);
    if (!out) return (int)bytes;
    if ((size_t)out_bytes < bytes) return fail("mv_debug_generate_episode: buffer too small");
    std::mt19937 rng;
    rng.seed((unsigned long)env_seed);
    std::vector<uint8_t> buf(bytes, 0);
    for (int i = 0; i < n; ++i) {
        std::memset(buf.data(), 0, bytes);
        if (scenario == SCN_COLLECT) generate_collect_episode(rng, num_agents, base_episode_len, *reinterpret_cast<CollectBlob *>(buf.data()));
        else if (scenario == SCN_HEX_MEMORY) generate_hex_memory_episode(rng, num_agents, base_episode_len, *reinterpret_cast<HexBlob *>(buf.data()));
        else if (scenario == SCN_HEX_EXPLORE) generate_hex_explore_episode(rng, num_agents, base_episode_len, *reinterpret_cast<HexBlob *>(buf.data()));
        else if (scenario == SCN_REARRANGE) generate_rearrange_episode(rng, num_agents, base_episode_len, *reinterpret_cast<RearrangeBlob *>(buf.data()));
        else generate_obstacles_episode(rng, oc, num_agents, base_episode_len, *reinterpret_cast<EpisodeBlob *>(buf.data()));
    }
    std::memcpy(out, buf.data(), bytes);
    return (int)bytes;
}

// Host-only test hook: drives an EpisodeFeeder (worker pool, per-env ordering, recycle) without a device and checks
// every episode it delivers against a straight sequential generation from the same seeds.  Returns 0 when equal.
int mv_debug_feeder_selftest(const char *scenario_name, int32_t num_envs, int32_t num_agents, int32_t threads, int32_t rounds)
{
    int scenario = SCN_TOWER;
    ObstacleConfig oc;
    if (!scenario_name || !scenario_from_name(lower(scenario_name), scenario, oc) || scenario == SCN_TOWER || scenario == SCN_SOKOBAN || scenario == SCN_EMPTY)
        return fail("mv_debug_feeder_selftest: the Obstacles family, Collect, Rearrange, HexMemory and HexExplore");
    const bool hex = scenario == SCN_HEX_MEMORY || scenario == SCN_HEX_EXPLORE;
    const size_t bytes = scenario == SCN_COLLECT ? sizeof(CollectBlob) : scenario == SCN_REARRANGE
                                                          ? sizeof(RearrangeBlob) : hex ? sizeof(HexBlob) : sizeof(EpisodeBlob);
    std::vector<uint8_t> slots((size_t)num_envs * bytes, 0), want(bytes);
    std::vector<uint32_t> seeds(num_envs);
    for (int i = 0; i < num_envs; ++i) seeds[i] = 1000u + 7u * (uint32_t)i;
    std::vector<std::mt19937> rng(num_envs);
    for (int i = 0; i < num_envs; ++i) rng[i].seed((unsigned long)seeds[i]);
    EpisodeFeeder feeder(scenario, oc, num_envs, num_agents, 60.0f, slots.data(), bytes, 0, threads);
    feeder.reseed(seeds, std::vector<int>(num_envs, 1));
    for (int r = 1; r <= rounds; ++r)
        for (int k = 0; k < num_envs; ++k) {
            const int i = (r & 1) ? k : num_envs - 1 - k;   // consume in varying order
            size_t used = 0;
            const uint8_t *got = feeder.wait_ready(i, r, &used);
            if (!got) return fail("feeder selftest: episode not delivered");
            std::memset(want.data(), 0, bytes);
            if (hex) {   // the box list comes last and only its used prefix is meaningful
                HexBlob &b = *reinterpret_cast<HexBlob *>(want.data());
                if (scenario == SCN_HEX_MEMORY) generate_hex_memory_episode(rng[i], num_agents, 60.0f, b);
                else generate_hex_explore_episode(rng[i], num_agents, 60.0f, b);
                b.seq = r;
                const HexBlob &a = *reinterpret_cast<const HexBlob *>(got);
                if (used != offsetof(HexBlob, boxes) + sizeof(HexRec) * (size_t)b.num_boxes || std::memcmp(&a, &b, offsetof(HexBlob, objs)) ||
                    std::memcmp(a.objs, b.objs, sizeof(HexRec) * (size_t)b.num_objs) || std::memcmp(a.boxes, b.boxes, sizeof(HexRec) * (size_t)b.num_boxes))
                    return fail("feeder selftest: Hex episode differs from sequential generation");
                feeder.recycle(i, nullptr);
                continue;
            }
            if (scenario == SCN_REARRANGE) {
                RearrangeBlob &b = *reinterpret_cast<RearrangeBlob *>(want.data());
                generate_rearrange_episode(rng[i], num_agents, 60.0f, b);
                b.seq = r;
                if (std::memcmp(got, &b, sizeof b)) return fail("feeder selftest: Rearrange episode differs from sequential generation");
                feeder.recycle(i, nullptr);
                continue;
            }
            if (scenario == SCN_COLLECT) {
                CollectBlob &b = *reinterpret_cast<CollectBlob *>(want.data());
                generate_collect_episode(rng[i], num_agents, 60.0f, b);
                b.seq = r;
            } else {
                EpisodeBlob &b = *reinterpret_cast<EpisodeBlob *>(want.data());
                generate_obstacles_episode(rng[i], oc, num_agents, 60.0f, b);
                b.seq = r;
            }
            if (used > bytes) return fail("feeder selftest: used bytes out of range");
            // compare the meaningful fields: counts first, then the used prefix of each array via the generators' own layout
            if (scenario == SCN_COLLECT) {
                const CollectBlob &a = *reinterpret_cast<const CollectBlob *>(got), &b = *reinterpret_cast<const CollectBlob *>(want.data());
                if (a.seq != b.seq || a.num_boxes != b.num_boxes || a.num_objects != b.num_objects || a.num_rewards != b.num_rewards ||
                    std::memcmp(a.boxes, b.boxes, sizeof(LayoutBox) * (size_t)b.num_boxes) || std::memcmp(a.heightmap, b.heightmap, HM_DIM * HM_DIM) ||
                    std::memcmp(a.spawn, b.spawn, sizeof a.spawn) || std::memcmp(a.yaw_frand, b.yaw_frand, sizeof(float) * (size_t)num_agents))
                    return fail("feeder selftest: Collect episode differs from sequential generation");
            } else {
                const EpisodeBlob &a = *reinterpret_cast<const EpisodeBlob *>(got), &b = *reinterpret_cast<const EpisodeBlob *>(want.data());
                if (a.seq != b.seq || a.num_boxes != b.num_boxes || a.num_objects != b.num_objects || a.num_rewards != b.num_rewards ||
                    std::memcmp(a.boxes, b.boxes, sizeof(LayoutBox) * (size_t)b.num_boxes) || std::memcmp(a.spawn, b.spawn, sizeof a.spawn) ||
                    std::memcmp(a.yaw_frand, b.yaw_frand, sizeof(float) * (size_t)num_agents))
                    return fail("feeder selftest: Obstacles episode differs from sequential generation");
            }
            feeder.recycle(i, nullptr);
        }
    return 0;
}

// Host-only test hook for the Sokoban generator: the first `n` episodes an env
// seeded with env_seed generates from the level files under $BOXOBAN_LEVELS, as n consecutive SokobanBlob records.
int mv_debug_generate_sokoban(int32_t num_agents, int32_t env_seed, int32_t n, float base_episode_len, void *out, int32_t out_bytes)
{
    if (!out) return (int)sizeof(SokobanBlob);
    if (num_agents < 1 || num_agents > MAX_AGENTS || n < 1 || (size_t)out_bytes < (size_t)n * sizeof(SokobanBlob))
        return fail("mv_debug_generate_sokoban: bad arguments");
    const std::vector<std::string> files = find_boxoban_level_files();
    if (files.empty()) return fail("mv_debug_generate_sokoban: no Boxoban levels found (BOXOBAN_LEVELS)");
    std::mt19937 rng;
    rng.seed((unsigned long)env_seed);
    SokobanLevels levels;
    for (int i = 0; i < n; ++i)
        if (!generate_sokoban_episode(rng, levels, files, num_agents, base_episode_len, reinterpret_cast<SokobanBlob *>(out)[i]))
            return fail("mv_debug_generate_sokoban: unreadable level file");
    return n;
}

int mv_debug_rng(int32_t device, uint32_t seed, int32_t what, const int32_t *lo, const int32_t *hi, int32_t n, void *out)
{
    HIP_TRY(hipSetDevice(device));
    if (what == 3 && n > 4096) return fail("mv_debug_rng: shuffle n <= 4096");
    int32_t *dlo = nullptr, *dhi = nullptr;
    void *dout = nullptr;
    HIP_TRY(hipMalloc(&dout, (size_t)n * 4));
    if (what == 1) {
        HIP_TRY(hipMalloc((void **)&dlo, (size_t)n * 4));
        HIP_TRY(hipMalloc((void **)&dhi, (size_t)n * 4));
        HIP_TRY(hipMemcpy(dlo, lo, (size_t)n * 4, hipMemcpyHostToDevice));
        HIP_TRY(hipMemcpy(dhi, hi, (size_t)n * 4, hipMemcpyHostToDevice));
    }
    hipLaunchKernelGGL(debug_rng_kernel, dim3(1), dim3(64), 0, nullptr, seed, what, dlo, dhi, n, dout);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpy(out, dout, (size_t)n * 4, hipMemcpyDeviceToHost));
    (void)hipFree(dout); (void)hipFree(dlo); (void)hipFree(dhi);
    return 0;
}

int mv_debug_math(int32_t device, int32_t what, const float *a, const float *b, int32_t n, float *out)
{
    HIP_TRY(hipSetDevice(device));
    float *da = nullptr, *db = nullptr, *dout = nullptr;
    const size_t outN = (what == 2) ? 2 * (size_t)n : (size_t)n;
    HIP_TRY(hipMalloc((void **)&da, (size_t)n * 4));
    HIP_TRY(hipMalloc((void **)&db, (size_t)n * 4));
    HIP_TRY(hipMalloc((void **)&dout, outN * 4));
    HIP_TRY(hipMemcpy(da, a, (size_t)n * 4, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(db, b ? b : a, (size_t)n * 4, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(debug_math_kernel, dim3((n + 255) / 256), dim3(256), 0, nullptr, what, da, db, n, dout);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpy(out, dout, outN * 4, hipMemcpyDeviceToHost));
    (void)hipFree(da); (void)hipFree(db); (void)hipFree(dout);
    return 0;
}

}  // extern "C"
