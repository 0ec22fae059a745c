// megaverse_amd/csrc/mv_api.hip -- host side of libmegaverse_hip.so: the C ABI declared in
// include/megaverse_hip.h -- this file: create / close (HBM allocation), seeding, reset, actions, output rings, getters, reward shaping,
// the episode refill protocol; mv_api_step.hip: stepping (pipelining, batched calls, groups, profiling); mv_api_debug.hip: test hooks;
// mv_api_internal.h: what they share (struct mv_gym).
//
// Mirrors class MegaverseGym of the reference (src/libs/bindings/megaverse.cpp:34-262) method by
// method; the per-step control flow mirrors VectorEnv::step (src/libs/env/src/vector_env.cpp:89-108):
//   step all envs  ->  for done envs: record trueObjective, reset  ->  draw.
// There is NO CPU fallback: if no HIP device can be opened mv_create fails.
#include "mv_api_internal.h"

thread_local std::string mvapi::g_err;

static const char *SHAPING_KEYS_TOWER[4] = {"teamSpirit", "towerPickedUpObject", "towerVisitedBuildingZoneWithObject",
                                           "towerBuildingReward"};
static const float SHAPING_DEFAULT_TOWER[4] = {0.1f, 0.1f, 0.1f, 1.0f};   // scenario_tower_building.hpp:44-52
// scenario_obstacles.hpp:36-44, teamSpirit 0 from Scenario::init (scenario.hpp:94-103)
static const char *SHAPING_KEYS_OBST[5] = {"teamSpirit", "obstaclesAgentAtExit", "obstaclesAllAgentsAtExit", "obstaclesExtraReward",
                                          "obstaclesAgentCarriedObjectToExit"};
static const float SHAPING_DEFAULT_OBST[5] = {0.0f, 1.0f, 5.0f, 0.5f, 0.0f};
// scenario_collect.hpp:44-52
// scenario_rearrange.hpp:96-102
static const char *SHAPING_KEYS_REARRANGE[3] = {"teamSpirit", "rearrangeOneMoreObjectCorrectPosition", "rearrangeAllObjectsCorrectPosition"};
static const float SHAPING_DEFAULT_REARRANGE[3] = {0.0f, 1.0f, 10.0f};
// scenario_sokoban.hpp:40-47, teamSpirit 0
static const char *SHAPING_KEYS_SOKOBAN[4] = {"teamSpirit", "sokobanBoxOnTarget", "sokobanBoxLeavesTarget", "sokobanAllBoxesOnTarget"};
static const float SHAPING_DEFAULT_SOKOBAN[4] = {0.0f, 1.0f, -1.0f, 10.0f};
// EmptyScenario::defaultRewardShaping() is {} (scenario_empty.hpp:28) + Scenario::init's teamSpirit
static const char *SHAPING_KEYS_EMPTY[1] = {"teamSpirit"};
// scenario_hex_memory.hpp:37-43, scenario_hex_explore.hpp:28-31 (+ teamSpirit 0)
static const char *SHAPING_KEYS_HEX_MEMORY[3] = {"teamSpirit", "memoryCollectGood", "memoryCollectBad"};
static const float SHAPING_DEFAULT_HEX_MEMORY[3] = {0.0f, 1.0f, -1.0f};
static const char *SHAPING_KEYS_HEX_EXPLORE[2] = {"teamSpirit", "exploreSolved"};
static const float SHAPING_DEFAULT_HEX_EXPLORE[2] = {0.0f, 5.0f};
static const char *SHAPING_KEYS_COLLECT[5] = {"teamSpirit", "collectSingleGood", "collectSingleBad", "collectAll", "collectAbyss"};
static const float SHAPING_DEFAULT_COLLECT[5] = {0.0f, 1.0f, -1.0f, 5.0f, -0.5f};
static const int ACTION_SPACE[6] = {3, 3, 3, 2, 2, 3};                          // env.cpp:33

// ------------------------------------------------------------------------------------------------
// small utility kernels
// ------------------------------------------------------------------------------------------------
__global__ void masks_from_multidiscrete_kernel(const int32_t *md, int32_t *masks, int n)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) masks[i] = action_mask_of(md + (size_t)i * 6);
}

// step outputs of this parity -> the public arrays (on the caller's stream: ordered with its consumers).  true_objective is only ever
// recorded by a finishing env (vector_env.cpp:96-101): the others keep the value of their last episode.
__global__ void publish_kernel(const float *s_rew, const uint8_t *s_done, const float *s_true, float *rew, uint8_t *done, float *true_obj, int n_envs, int A)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_envs * A) return;
    rew[i] = s_rew[i];
    const int e = i / A;
    if (s_done[e]) true_obj[i] = s_true[i];
    if (i < n_envs) done[i] = s_done[i];
}

__global__ void clear_flags_kernel(int *word, int reported) { atomicAnd(word, ~reported); }   // only the bits that were reported: a bit raised since stays
__global__ void set_shaping_kernel(AgentState *agents, int idx, int key, float v) { agents[idx].shaping[key] = v; }

namespace mvapi {
// ------------------------------------------------------------------------------------------------
std::string lower(const char *s)
{
    std::string r(s ? s : "");
    for (auto &c : r) c = (char)std::tolower((unsigned char)c);
    return r;
}

int check(mv_gym *g)
{
    if (!g) return fail("null gym handle");
    if (g->closed) return fail("gym is closed");
    return 0;
}

// Where a tick's public outputs go: the observation slab and the reward / done arrays -- or, with mv_set_output_ring, entry (tick % count)
// of the caller's rings.  true_objective is state (only a finishing env records it, vector_env.cpp:96-101): never ringed.
OutPtrs outputs_of(const mv_gym *g, unsigned long long tick)
{
    OutPtrs o{g->obs, g->gv.rewards, g->gv.done};
    if (g->ringCount > 0) {
        const size_t r = (size_t)(tick % (unsigned long long)g->ringCount), NA = (size_t)g->N * g->A;
        if (g->ringObs) o.obs = reinterpret_cast<uint32_t *>(g->ringObs + r * NA * (size_t)g->w * g->h * 4);
        if (g->ringRewards) o.rewards = g->ringRewards + r * NA;
        if (g->ringDone) o.done = g->ringDone + r * (size_t)g->N;
    }
    return o;
}
OutPtrs last_outputs(const mv_gym *g) { return outputs_of(g, g->ringTick ? g->ringTick - 1 : 0); }   // of the last tick (reset / render / getters)

int refresh_mirrors(mv_gym *g)
{
    if (g->mirrorsFresh) return 0;
    const size_t NA = (size_t)g->N * g->A;
    const OutPtrs o = last_outputs(g);
    HIP_TRY(hipMemcpyAsync(g->hRewards.data(), o.rewards, NA * sizeof(float), hipMemcpyDeviceToHost, g->stream));
    HIP_TRY(hipMemcpyAsync(g->hTrueObj.data(), g->gv.true_objective, NA * sizeof(float), hipMemcpyDeviceToHost, g->stream));
    HIP_TRY(hipMemcpyAsync(g->hDone.data(), o.done, (size_t)g->N, hipMemcpyDeviceToHost, g->stream));
    HIP_TRY(hipStreamSynchronize(g->stream));
    g->mirrorsFresh = true;
    return 0;
}

// The next observation pass's cost histogram: advances hist3 and makes sure the histogram is zero before the pass's frame setup counts into it
// (stream s: where that setup runs).  setupClearsNext: the frame setup of this pass clears the histogram after it (the one-launch-per-tick path).
int take_hist(mv_gym *g, hipStream_t s, bool setupClearsNext)
{
    g->hist3 = (g->hist3 + 1) % g->hists;
    if (!g->histClean[(size_t)g->hist3])
        HIP_TRY(hipMemsetAsync(g->gv.lpt_hist + (size_t)g->hist3 * LPT_BUCKETS * LPT_SUBS, 0, (size_t)LPT_BUCKETS * LPT_SUBS * sizeof(int32_t), s));
    g->histClean[(size_t)g->hist3] = 0;
    if (setupClearsNext) g->histClean[(size_t)((g->hist3 + 1) % g->hists)] = 1;
    return 0;
}

// the view a kernel launch gets: the buffers of slot q, this pass's cost histogram, the action-sampling request
GymView view(const mv_gym *g, int q, const OutPtrs *direct)   // direct: the step writes the public output arrays itself (not pipelined)
{
    GymView v = g->gvp[q];
    if (direct) { v.rewards = direct->rewards; v.done = direct->done; v.true_objective = g->gv.true_objective; }
    v.sample_on = g->gv.sample_on; v.sample_seed = g->gv.sample_seed; v.sample_step = g->gv.sample_step;
    v.md_actions = nullptr;
    v.lpt_parity = g->hist3;
    v.depth_sort = v.sort_scratch != nullptr && g->fastPixels != 0;   // (the exact kernel resolves depth ties by list position: its lists stay as found)
    return v;
}

// Before anything on the caller's stream reads or writes simulator state (reset, render, hires, seeds, test hooks): it waits for the
// simulation stream, and the next step will wait for it.
int sim_join(mv_gym *g)
{
    if (g->simDoneValid && g->simOnOwnStream) HIP_TRY(hipStreamWaitEvent(g->stream, g->simDone, 0));
    g->simMustWaitUser = true;
    return 0;
}

// TowerBuilding: before anything on the caller's stream touches the generators or the ring of drawn episodes (mv_reset, mv_seed): the last draw launch
int tower_join(mv_gym *g)
{
    if (g->genStream && g->drawCount > 0) HIP_TRY(hipStreamWaitEvent(g->stream, g->drawDone[(size_t)((g->drawCount - 1) & 1ull)], 0));
    return 0;
}

// TowerBuilding, a stepping call: before its step launches the simulation stream waits for the draw launch BEFORE the last one (two episodes are resident per
// env; what the last launch may still be drawing replaces an episode consumed a call ago: not needed yet) ...
int tower_draw_before(mv_gym *g, hipStream_t sim)
{
    if (!g->genStream || g->drawCount < 2) return 0;
    // (draw launches come every drawPeriod ticks: a tick-by-tick caller waits once per launch, not per tick)
    if (g->drawWaitedCount == g->drawCount && g->drawWaitedOn == sim) return 0;
    HIP_TRY(hipStreamWaitEvent(sim, g->drawDone[(size_t)(g->drawCount & 1ull)], 0));
    g->drawWaitedCount = g->drawCount; g->drawWaitedOn = sim;
    return 0;
}
// ... and behind them, every drawPeriod ticks, the draw kernel goes to its own stream: it tops up the rings of the envs that finished
int tower_draw_after(mv_gym *g, hipEvent_t after, int ticks)
{
    if (!g->genStream) return 0;
    g->ticksSinceDraw += ticks;
    if (g->ticksSinceDraw < g->drawPeriod) return 0;
    g->ticksSinceDraw = 0;
    HIP_TRY(hipStreamWaitEvent(g->genStream, after, 0));
    launch_tower_draw(g->gv, g->genStream);
    HIP_TRY(hipEventRecord(g->drawDone[(size_t)(g->drawCount & 1ull)], g->genStream));
    ++g->drawCount;
    return 0;
}

int publish_outputs(mv_gym *g, int q, const OutPtrs &o)   // on the caller's stream
{
    const GymView &v = g->gvp[q];
    const int n = g->N * g->A;
    hipLaunchKernelGGL(publish_kernel, dim3((n + 255) / 256), dim3(256), 0, g->stream, v.rewards, v.done, v.true_objective, o.rewards, o.done,
                       g->gv.true_objective, g->N, g->A);
    HIP_TRY(hipGetLastError());
    return 0;
}
}  // namespace mvapi

extern "C" {

const char *mv_last_error(void) { return g_err.c_str(); }
int mv_abi_version(void) { return 2; }

int mv_device_count(void)
{
    int n = 0;
    if (hipInit(0) != hipSuccess || hipGetDeviceCount(&n) != hipSuccess) { (void)hipGetLastError(); return 0; }
    return n;
}

int mv_action_space_sizes(int32_t *out6)
{
    for (int i = 0; i < 6; ++i) out6[i] = ACTION_SPACE[i];
    return 0;
}

}  // extern "C" (helper below has C++ linkage)

// scenario name -> kernel family + generator parameters (scenarios/init.hpp:30-52)
namespace mvapi {
bool scenario_from_name(const std::string &scen, int &scenario, ObstacleConfig &oc)
{
    if (scen == "towerbuilding") scenario = SCN_TOWER;
    else if (scen == "obstacleseasy") scenario = SCN_OBSTACLES;                       // scenario_obstacles.hpp:112-138
    else if (scen == "obstaclesmedium") { scenario = SCN_OBSTACLES; oc.min_platforms = 2; oc.max_platforms = 4; oc.min_lava = 2; oc.max_lava = 5; }
    else if (scen == "obstacleshard") {                                               // :164-189
        scenario = SCN_OBSTACLES; oc.min_platforms = 2; oc.max_platforms = 7; oc.min_gap = 2; oc.max_gap = 3; oc.min_lava = 3; oc.max_lava = 10;
        oc.min_height = 2; oc.max_height = 4;
    } else if (scen == "obstacleswalls" || scen == "obstaclessteps" || scen == "obstacleslava") {   // :190-268
        scenario = SCN_OBSTACLES; oc.min_platforms = 1; oc.max_platforms = 4; oc.min_gap = 1; oc.max_gap = 3; oc.min_lava = 2; oc.max_lava = 10;
        oc.min_height = 1; oc.max_height = 3; oc.carried_object_to_exit = 1.0f;
        oc.platform_types[0] = scen == "obstacleswalls" ? 1 : scen == "obstaclessteps" ? 3 : 2;
        oc.num_platform_types = 1;
    } else if (scen == "collect") scenario = SCN_COLLECT;                              // scenarios/init.hpp:45
    else if (scen == "rearrange") scenario = SCN_REARRANGE;                            // scenarios/init.hpp:49
    else if (scen == "sokoban") scenario = SCN_SOKOBAN;                                // scenarios/init.hpp:46
    else if (scen == "empty") scenario = SCN_EMPTY;                                    // scenarios/init.hpp:34
    else if (scen == "hexmemory") scenario = SCN_HEX_MEMORY;                           // scenarios/init.hpp:47
    else if (scen == "hexexplore") scenario = SCN_HEX_EXPLORE;                         // scenarios/init.hpp:48
    else return false;
    return true;
}
}  // namespace mvapi

// cores this process may use: the affinity mask, capped by the cgroup's CPU quota (a container with 16 of the host's 192 cores sees all of them in the mask)
static int usable_host_cores()
{
    int n = (int)std::thread::hardware_concurrency();
    cpu_set_t set;
    if (sched_getaffinity(0, sizeof set, &set) == 0) n = CPU_COUNT(&set);
    for (const char *path : {"/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"}) {
        FILE *f = std::fopen(path, "r");
        if (!f) continue;
        char a[64] = {0}, b[64] = {0};
        const int got = std::fscanf(f, "%63s %63s", a, b);
        std::fclose(f);
        if (got >= 1 && std::strcmp(a, "max") != 0 && std::atol(a) > 0) {
            long period = got >= 2 ? std::atol(b) : 0;
            if (period <= 0) {   // cgroup v1: the period is in a file of its own
                if (FILE *p = std::fopen("/sys/fs/cgroup/cpu/cpu.cfs_period_us", "r")) { if (std::fscanf(p, "%ld", &period) != 1) period = 0; std::fclose(p); }
            }
            if (period > 0) n = std::min(n, std::max(1, (int)((std::atol(a) + period - 1) / period)));
        }
        break;
    }
    return std::max(1, n);
}

// the simulation stream (another queue priority than the caller's was measured, low and high: no gain, r06k)
// The simulation stream has the device's highest stream priority: when a call's observation launch and the next call's step launch become ready together --
// the end of a step launch releases both -- the step launch's few fat workgroups (one or four waves of 128-168 VGPRs) should reach the chip first; behind an
// observation launch that has filled it with 72-VGPR waves they wait for holes that never get large enough until that launch has drained.  Measured (r08s, two
// runs each, normal / high): ObstaclesHard 512 envs 18.6 / 20.8 M obs/s, 1024 envs 24.1 / 26.3, Sokoban 26.2 / 28.2, Mixed 64 x 64 16.8 / 18.3, Mixed4 20.6 /
// 18.7, TowerBuilding (1024, 4096, 512 x 4), Collect, Empty: unchanged.  MV_SIM_PRIORITY=normal: a stream of default priority.
static hipError_t create_sim_stream(hipStream_t *s)
{
    static const char *prio = getenv("MV_SIM_PRIORITY");
    if (!prio || std::strcmp(prio, "normal")) {
        int lo = 0, hi = 0;
        if (hipDeviceGetStreamPriorityRange(&lo, &hi) == hipSuccess
            && hipStreamCreateWithPriority(s, hipStreamNonBlocking, hi) == hipSuccess) return hipSuccess;
        (void)hipGetLastError();   // (no priorities on this device / runtime: a stream of default priority does the same work)
    }
    return hipStreamCreateWithFlags(s, hipStreamNonBlocking);
}

extern "C" {

int mv_create(const mv_config *cfg, mv_gym **out)
{
    if (!cfg || !out) return fail("mv_create: null argument");
    *out = nullptr;
    // Scenario::create (scenario.hpp:61-77) is fatal on unknown names; we return an error instead
    const std::string scen = lower(cfg->scenario);
    int scenario = SCN_TOWER;
    ObstacleConfig oc;
    if (!scenario_from_name(scen, scenario, oc))
        return fail("Unknown scenario " + scen
                    + " (this build accelerates: TowerBuilding, ObstaclesEasy/Medium/Hard/Walls/Steps/Lava, Collect, Rearrange, Sokoban, HexMemory, "
                         "HexExplore, Empty)");
    std::vector<std::string> levelFiles;
    if (scenario == SCN_SOKOBAN) {   // SokobanScenario's constructor looks the level files up (scenario_sokoban.cpp:40-78); none is fatal there too
        levelFiles = find_boxoban_level_files();
        if (levelFiles.empty()) return fail("Sokoban: no Boxoban level files (set BOXOBAN_LEVELS to the directory that holds unfiltered/train/000.txt ...)");
    }
    if (cfg->num_envs < 1 || cfg->num_agents_per_env < 1 || cfg->num_agents_per_env > MAX_AGENTS)
        return fail("mv_create: num_envs >= 1 and 1 <= num_agents_per_env <= 8 required");
    if (cfg->obs_width < 1 || cfg->obs_height < 1 || cfg->obs_width > 1024
        || cfg->obs_height > 1024) return fail("mv_create: observation size must be within 1..1024");

    int ndev = 0;
    hipError_t derr = hipInit(0);
    if (derr == hipSuccess) derr = hipGetDeviceCount(&ndev);
    if (derr != hipSuccess || ndev <= 0)
        return fail(std::string("mv_create: no HIP device available (this library has no CPU fallback): ") + hipGetErrorString(derr));
    if (cfg->device < 0 || cfg->device >= ndev) return fail("mv_create: bad device ordinal");
    HIP_TRY(hipSetDevice(cfg->device));

    mv_gym *g = new mv_gym();
    g->device = cfg->device;
    g->w = cfg->obs_width; g->h = cfg->obs_height;
    g->N = cfg->num_envs; g->A = cfg->num_agents_per_env;
    g->scenario = scenario;
    g->numShaping = scenario == SCN_TOWER || scenario == SCN_SOKOBAN ? 4 : scenario == SCN_REARRANGE || scenario == SCN_HEX_MEMORY ? 3
                  : scenario == SCN_HEX_EXPLORE ? 2 : scenario == SCN_EMPTY ? 1 : 5;
    g->shapingKeys = scenario == SCN_TOWER ? SHAPING_KEYS_TOWER : scenario == SCN_OBSTACLES ? SHAPING_KEYS_OBST
                   : scenario == SCN_COLLECT ? SHAPING_KEYS_COLLECT : scenario == SCN_SOKOBAN ? SHAPING_KEYS_SOKOBAN
                   : scenario == SCN_HEX_MEMORY ? SHAPING_KEYS_HEX_MEMORY : scenario == SCN_HEX_EXPLORE ? SHAPING_KEYS_HEX_EXPLORE
                   : scenario == SCN_EMPTY ? SHAPING_KEYS_EMPTY : SHAPING_KEYS_REARRANGE;
    // Resident episodes per env: two where an episode ends at its time limit only (TowerBuilding, Empty); THREE where a goal can end it early (the exit pad,
    // every diamond collected, the level solved, the arrangement matched, the maze's target found): the host's run-ahead is bounded in TICKS (refill_episodes),
    // the status words are read back every 16th, and a lucky env can finish twice inside that window -- a third resident episode covers it where two starved
    // (ADVICE r05).
    g->spares = scenario == SCN_TOWER || scenario == SCN_EMPTY ? 2 : 3;
    g->envOffset = cfg->total_envs > 0 ? cfg->env_offset : 0;
    g->envStride = cfg->total_envs > 0 && cfg->env_stride > 1 ? cfg->env_stride : 1;
    g->gv.sample_on = 0; g->gv.sample_seed = g->gv.sample_step = 0;
    g->gv.env_offset = g->envOffset; g->gv.env_stride = g->envStride;
    g->totalEnvs = cfg->total_envs > 0 ? cfg->total_envs : cfg->num_envs;
    const size_t N = g->N, NA = (size_t)g->N * g->A;

    GymView &gv = g->gv;
    gv.num_envs = g->N; gv.num_agents = g->A;
    const bool obstacles = scenario == SCN_OBSTACLES || scenario == SCN_EMPTY, collect = scenario == SCN_COLLECT,
            rearrange = scenario == SCN_REARRANGE, sokoban = scenario == SCN_SOKOBAN;
    const bool hex = scenario == SCN_HEX_MEMORY || scenario == SCN_HEX_EXPLORE;
    const bool hostEpisodes = obstacles || collect || rearrange || sokoban || hex;
    gv.scenario = scenario;
    gv.box_stride = collect ? COLLECT_MAX_BOXES : MAX_BOXES;
    gv.reward_stride = collect ? COLLECT_MAX_REWARDS : MAX_REWARDS;
    g->blobBytes = collect ? sizeof(CollectBlob) : obstacles ? sizeof(EpisodeBlob) : rearrange ? sizeof(RearrangeBlob)
                                    : sokoban ? sizeof(SokobanBlob) : hex ? sizeof(HexBlob) : sizeof(TowerBlob);
    // ONE arena for all simulator state: a step touches ~8 arrays per env, separate small allocations
    // cost a TLB miss each per wave (measured: 83 % of the physics kernel's time was spent waiting on
    // ~30 memory operations); one large allocation is backed by large pages.
    auto up = [](size_t v) { return (v + 4095) & ~size_t(4095); };
    const size_t szHdr = up(N * sizeof(EnvHeader)), szBoxes = up(N * (size_t)gv.box_stride * sizeof(LayoutBox)),
                 szObj = up(N * MAX_OBJECTS * sizeof(MovableObject)), szAg = up(NA * sizeof(AgentState)),
                 szChunk = up(N * (size_t)CHUNK_BYTES), szAct = up(NA * sizeof(int32_t)), szRew = up(NA * sizeof(float)),
                 szDone = up(N), szObjv = up(NA * sizeof(float)), szMd = up(NA * 6 * sizeof(int32_t)),
                 szObs = up(NA * (size_t)g->w * g->h * 4);
    const size_t szTerrain = obstacles ? up(N * MAX_TERRAIN * sizeof(TerrainBox)) : 0,
                 szRewObj = hostEpisodes ? up(N * (size_t)gv.reward_stride * sizeof(MovableObject)) : 0,
                 szHeight = collect ? up(N * (size_t)HM_BYTES) : 0, szItems = rearrange ? up(N * MAX_ITEMS * sizeof(ArrangementItem))
                                         : 0, szCells = sokoban ? up(N * (size_t)(SOKO_DIM * SOKO_DIM)) : 0,
                 szHexB = hex ? up(N * (size_t)HEX_MAX_BOXES * sizeof(HexRec)) : 0, szHexO = hex ? up(N * (size_t)HEX_MAX_OBJS * sizeof(HexRec)) : 0,
                                   szBlobs = up(N * g->blobBytes * (size_t)g->spares), szCnt = up((N + 2) * sizeof(int32_t)),
                                                szGen = hostEpisodes ? 0 : up(N * sizeof(TowerGen));
    gv.vis_stride = hex ? 2048 : collect ? 1024 : 256;
    if (const char *e = getenv("MV_DEBUG_VIS_STRIDE")) gv.vis_stride = std::min(gv.vis_stride, std::max(8, atoi(e)));   // (tests: provoke ST_VISIBLE)
    gv.debug_redo = getenv("MV_DEBUG_FORCE_REDO") && atoi(getenv("MV_DEBUG_FORCE_REDO")) ? 1 : 0;   // (tests: mv_tick_tower.h's sequential redo)
    gv.spares = g->spares;
    const size_t szVisP = up(NA * (size_t)gv.vis_stride * 32), szVisR = up(NA * (size_t)gv.vis_stride * 8), szVisC = up(NA * sizeof(int32_t)),
                 szLpt = up(NA * sizeof(int32_t)) + up((NA + 1) * sizeof(int32_t)) + up(NA * (size_t)FRAME_HDR_BYTES)
                            + up((size_t)LPT_BUCKETS * LPT_SUBS * lpt_sub_capacity(NA) * sizeof(int32_t));
    // per slot: frame lists, headers, cost lists, and the staging copies of rewards / dones / true objectives
    const size_t szParity = szVisP + szVisR + szVisC + szLpt + szRew + szDone + szObjv;
    // Ticks per call of mv_step_n (`batch`; a gym holds PIPE_GROUPS x batch hand-over slots of szParity bytes each): 16 -- one tail of the one-launch
    // observation pass per 16 ticks, measured against 8: TowerBuilding 1024 envs 26.6 -> 28.3 M obs/s -- where the 48 slots that takes stay under 2.25 GiB,
    // else 8 (a Hex frame's slot is 80 KB: 3.8 GB per 1024 frames at 16, 1.9 GB at 8 -- and 16 buys it nothing: 9.32 / 9.40 M obs/s, r09k; Collect's 42 KB: 2.1
    // GB at 16, 13.9 -> 14.8 M; TowerBuilding's 12 KB: 0.6 GB).  MV_PIPE_BATCH=1..16 overrides; mv_recommended_ticks_per_call says what to ask for.
    g->batch = (size_t)PIPE_GROUPS * 16 * szParity <= (size_t(9) << 28) ? 16 : 8;   // (2.25 GiB)
    if (const char *e = getenv("MV_PIPE_BATCH")) g->batch = std::min((int)PIPE_BATCH_MAX, std::max(1, atoi(e)));
    g->slots = PIPE_GROUPS * g->batch;
    g->hists = g->slots + 1;
    g->histClean.assign((size_t)g->hists, 1);   // (the arena is zeroed below)
    g->gvp.resize((size_t)g->slots);
    g->parity = g->slots - 1;
    g->group = PIPE_GROUPS - 1;
    // (+ one "workgroups that have looked their frame up" counter per histogram, behind them)
    const size_t szHist = up((size_t)g->hists * (LPT_BUCKETS * LPT_SUBS + 1) * sizeof(int32_t));
    gv.lpt_hists = g->hists;
    // long lists: the list as found, before the frame setup deals it into depth classes (mv_frame.h: DepthSortScratch); MV_DEPTH_SORT=0: lists stay as found
    const bool depthSortOn = !(getenv("MV_DEPTH_SORT") && atoi(getenv("MV_DEPTH_SORT")) == 0);
    const size_t szSort = gv.vis_stride > 256 && depthSortOn ? up(NA * (size_t)gv.vis_stride * 40) : 0;
    const size_t total = szSort + szHdr + szBoxes + szObj + szAg + szAct + szRew + szDone + szObjv + szMd + (hostEpisodes ? 0 : szChunk) + szObs + szTerrain +
                         szRewObj + szHeight + szItems + szCells + szHexB + szHexO + szBlobs + szCnt + szGen + (size_t)g->slots * szParity + szHist;
    {
        hipError_t e_ = hipMalloc((void **)&g->arena, total);
        if (e_ != hipSuccess) {
            const int b = g->batch;
            mv_destroy(g);
            return fail("hipMalloc arena (" + std::to_string(total >> 20) + " MiB, of which "
                        + std::to_string(((size_t)PIPE_GROUPS * b * szParity) >> 20) + " MiB are the " +
                        std::to_string(PIPE_GROUPS * b) + " hand-over slots of " + std::to_string(b)
                                       + " ticks per call: a smaller MV_PIPE_BATCH shrinks them): " + hipGetErrorString(e_));
        }
        g->arenaBytes = total;
        (void)hipMemset(g->arena, 0, total);
    }
    {
        uint8_t *p = g->arena;
        gv.hdr = (EnvHeader *)p; p += szHdr;
        gv.boxes = (LayoutBox *)p; p += szBoxes;
        gv.objects = (MovableObject *)p; p += szObj;
        gv.agents = (AgentState *)p; p += szAg;
        gv.actions = (int32_t *)p; p += szAct;
        gv.rewards = (float *)p; p += szRew;
        gv.done = p; p += szDone;
        gv.true_objective = (float *)p; p += szObjv;
        g->dMultiDiscrete = (int32_t *)p; p += szMd;
        if (!hostEpisodes) { gv.chunk = p; p += szChunk; }
        g->ownedObs = (uint32_t *)p; p += szObs;
        g->dStatus = (int *)p; p += szCnt;
        gv.episode_status = g->dStatus;
        if (obstacles) { gv.terrain = (TerrainBox *)p; p += szTerrain; }
        if (hostEpisodes) { gv.rewards_obj = (MovableObject *)p; p += szRewObj; }
        g->dBlobs = p; p += szBlobs;   // the ring of resident next episodes: uploaded by the host's feeder, or (TowerBuilding) drawn on the device
        gv.blobs = g->dBlobs;
        if (!hostEpisodes) { gv.tower_gen = (TowerGen *)p; p += szGen; }
        if (collect) { gv.heightmap = (int8_t *)p; p += szHeight; }
        if (rearrange) { gv.items = (ArrangementItem *)p; p += szItems; }
        if (sokoban) { gv.soko_cells = p; p += szCells; }
        if (hex) { gv.hex_boxes = (HexRec *)p; p += szHexB; gv.hex_objs = (HexRec *)p; p += szHexO; }
        gv.lpt_hist = (int32_t *)p; p += szHist;
        gv.sort_scratch = szSort ? p : nullptr; p += szSort;
        gv.depth_sort = 0;
        for (int q = 0; q < g->slots; ++q) {   // gv.rewards / done / true_objective stay the public arrays; the slot views write their own
            GymView &v = g->gvp[q];
            v = gv;
            v.vis_prims = p; p += szVisP;
            v.vis_rects = p; p += szVisR;
            v.vis_count = (int32_t *)p; p += szVisC;
            v.lpt_bucket = (int32_t *)p; p += up(NA * sizeof(int32_t));
            v.lpt_order = (int32_t *)p; p += up((NA + 1) * sizeof(int32_t));
            v.vis_hdr = p; p += up(NA * (size_t)FRAME_HDR_BYTES);
            v.lpt_list = (int32_t *)p; p += up((size_t)LPT_BUCKETS * LPT_SUBS * lpt_sub_capacity(NA) * sizeof(int32_t));
            v.rewards = (float *)p; p += szRew;
            v.done = p; p += szDone;
            v.true_objective = (float *)p; p += szObjv;
        }
    }
    if (getenv("MV_TICK_TIMING") && atoi(getenv("MV_TICK_TIMING"))) {   // (an instrumented build, -DMV_TICK_TIMING: phase cycle sums, printed by mv_close)
        if (hipMalloc((void **)&g->gv.dbg, N * 64 * sizeof(unsigned long long)) == hipSuccess)
            (void)hipMemset(g->gv.dbg, 0, N * 64 * sizeof(unsigned long long));
        else g->gv.dbg = nullptr;
        for (int q = 0; q < g->slots; ++q) g->gvp[q].dbg = g->gv.dbg;
    }
    if (const char *e = getenv("MV_PIXEL_MODE")) g->fastPixels = lower(e) == "exact" ? 0 : 1;
    if (const char *e = getenv("MV_PIPELINE")) g->pipelined = atoi(e) != 0;
    g->obs = g->ownedObs;
    for (int b = 0; b < 2; ++b) {
        if (hipHostMalloc((void **)&g->hActions[b], NA * sizeof(int32_t), hipHostMallocDefault) != hipSuccess ||
            hipEventCreateWithFlags(&g->actionsCopied[b], hipEventDisableTiming) != hipSuccess) {
            mv_destroy(g);
            return fail("mv_create: pinned staging allocation failed");
        }
        std::memset(g->hActions[b], 0, NA * sizeof(int32_t));
    }
    g->hRewards.assign(NA, 0.0f); g->hTrueObj.assign(NA, 0.0f); g->hDone.assign(N, 0);

    // headers: float params + unseeded envs take their seed from random_device (env.hpp:169)
    float episodeLen = sokoban ? 80.0f : 60.0f, lookLimit = 0.2f;   // scenario.hpp:225-232; Sokoban: scenario_sokoban.hpp:49-53
    for (int k = 0; k < cfg->num_params; ++k) {
        const char *key = cfg->param_keys[k];
        const float v = cfg->param_vals[k];
        if (!std::strcmp(key, "episodeLengthSec")) episodeLen = v;
        if (!std::strcmp(key, "verticalLookLimitRad")) lookLimit = v;
        auto ip = [&](const char *name, int &dst) { if (!std::strcmp(key, name)) dst = int(std::lround(v)); };   // Platform::param()
        ip("obstaclesMinNumPlatforms", oc.min_platforms); ip("obstaclesMaxNumPlatforms", oc.max_platforms);
        ip("obstaclesMinGap", oc.min_gap); ip("obstaclesMaxGap", oc.max_gap); ip("obstaclesMinLava", oc.min_lava);
        ip("obstaclesMaxLava", oc.max_lava); ip("obstaclesMinHeight", oc.min_height); ip("obstaclesMaxHeight", oc.max_height);
        if (!std::strcmp(key, "obstaclesNumAllowedMaxDifficulty")) oc.num_allowed_max_difficulty = int(v);
    }
    g->obst = oc;
    g->baseEpisodeLen = episodeLen;
    {   // status words (episodes consumed, error flags) travel back on a side stream for every scenario
        bool ok = hipHostMalloc((void **)&g->hStatus, (N + 2) * sizeof(int), hipHostMallocDefault) == hipSuccess &&
                  hipStreamCreateWithFlags(&g->copyStream, hipStreamNonBlocking) == hipSuccess &&
                  create_sim_stream(&g->simStream) == hipSuccess &&
                  hipEventCreateWithFlags(&g->userMark[0], hipEventDisableTiming) == hipSuccess &&
                  hipEventCreateWithFlags(&g->userMark[1], hipEventDisableTiming) == hipSuccess &&
                  hipEventCreateWithFlags(&g->userMark[2], hipEventDisableTiming) == hipSuccess &&
                  hipEventCreateWithFlags(&g->userNow, hipEventDisableTiming) == hipSuccess &&
                  hipEventCreateWithFlags(&g->simDone, hipEventDisableTiming) == hipSuccess &&
                  hipEventCreateWithFlags(&g->stepDone, hipEventDisableTiming) == hipSuccess &&
                  hipEventCreateWithFlags(&g->statusCopied, hipEventDisableTiming) == hipSuccess;
        if (!ok) {
            mv_destroy(g);
            return fail("mv_create: status staging / copy stream allocation failed");
        }
        std::memset(g->hStatus, 0, (N + 2) * sizeof(int));
    }
    {
        // An env needs a fresh resident episode at every reset.  Two are kept resident, and the consumed counts are read back
        // every 16th step -- unless episodes can time out within a few ticks (a small or negative episodeLengthSec: the Obstacles
        // family never goes below 35 s per platform, the others take the parameter as is -- TowerBuilding adds 4 s per object), then after every step.
        const float minLenSec = scenario == SCN_OBSTACLES ? std::max(episodeLen, 35.0f) : episodeLen;
        g->statusPeriod = minLenSec * 15.0f >= 64.0f ? 16 : 1;
        if (const char *e = getenv("MV_STATUS_PERIOD")) g->statusPeriod = std::max(1, atoi(e));   // (tests: provoke starvation)
    }
    if (!hostEpisodes) {   // TowerBuilding: the generators' state (unseeded envs take their seed from random_device, env.hpp:169), the draw stream
        std::vector<TowerGen> tg(N);
        std::random_device rdev;
        for (auto &t : tg) { t.seed = (uint32_t)rdev(); t.seed_is_env_seed = 1; t.generated = 0; t.pad = 0; }
        bool ok = hipMemcpy(gv.tower_gen, tg.data(), N * sizeof(TowerGen), hipMemcpyHostToDevice) == hipSuccess &&
                  hipEventCreateWithFlags(&g->drawDone[0], hipEventDisableTiming) == hipSuccess &&
                  hipEventCreateWithFlags(&g->drawDone[1], hipEventDisableTiming) == hipSuccess;
        if (!ok) {
            mv_destroy(g);
            return fail("mv_create: generator state / draw stream allocation failed");
        }
        // (the draws share the side stream of the status read-backs: a stream more per gym and HIP's few hardware queues are oversubscribed -- measured with
        // two gyms stepped in turn, each with its own draw stream: 13.7 -> 8.5 M obs/s, the two gyms' caller streams had come to share a queue)
        g->genStream = g->copyStream;
        g->drawPeriod = g->statusPeriod > 1 ? 8 : 1;
        if (const char *e = getenv("MV_DRAW_PERIOD")) g->drawPeriod = g->statusPeriod > 1 ? std::max(1, atoi(e)) : 1;   // (measurements: r12m)
    }
    if (hostEpisodes) {
        g->uploaded.assign(N, 0);
        // Worker threads of the episode feeder: what the caller asks for (MegaverseGym's num_simulation_threads), or -- 0 / negative -- this process's
        // share of the host: the cores it may run on (affinity mask, cgroup quota) divided by the ranks of the job (total_envs / num_envs shards, one
        // process per GPU), at most 16.  What a thread sustains (episodes per second, measured: DESIGN.md 3.2): ObstaclesHard 15 k, ObstaclesEasy 44 k,
        // Collect 4.5 k, HexMemory 22 k, Rearrange 330 k; what one GPU consumes at its benchmark rate: ObstaclesHard ~7 k, Collect ~12 k.
        // ONE core of the share stays with the caller's thread (the one that enqueues the launches): with two cores and the two feeder threads this rule
        // used to start, ObstaclesHard 1024 envs ran at 21.5 - 25.6 M obs/s, with one thread at 29.4 M -- its rate with sixteen cores --, Collect 13.6 / 15.9
        // (r12p, r12q; what a thread sustains on that box: ObstaclesHard 40 k, Collect 28 k, HexMemory 46 k episodes per second).
        const int share = usable_host_cores() / std::max(1, g->totalEnvs / std::max(1, g->N));
        g->feederThreads = cfg->num_simulation_threads > 0 ? std::min(32, (int)cfg->num_simulation_threads) : std::max(1, std::min(16, share - 1));
        if (const char *e = getenv("MV_FEEDER_THREADS")) g->feederThreads = std::min(64, std::max(1, atoi(e)));
        g->uploadEvents.assign(64, nullptr);
        // Collect: the episodes may be drawn on the DEVICE (mv_collect_draw.h: the same episodes, byte for byte) instead of by the host's worker threads --
        // MV_COLLECT_DEVICE_GEN=1 / 0, otherwise where this process's share of the host is under three cores (eight ranks under a 16-CPU quota: two host
        // thread holds Collect at 15.9 M obs/s against 16.6 M) and the gym has 256 envs and more (r12n, two cores, host / device: a quarter of a
        // Mixed4 batch 18.0 / 19.8 M obs/s, an eighth of a Mixed batch 16.9 / 16.2: too few episodes to pay for the draw launches).  The staging slots are device memory then.
        g->blobsOnDevice = collect && (getenv("MV_COLLECT_DEVICE_GEN") ? atoi(getenv("MV_COLLECT_DEVICE_GEN")) != 0
                                                                        : cfg->num_simulation_threads <= 0 && !getenv("MV_FEEDER_THREADS") && share < 3 && N >= 256);
        bool ok = g->blobsOnDevice ? hipMalloc((void **)&g->hBlobs, N * g->blobBytes) == hipSuccess && hipMemset(g->hBlobs, 0, N * g->blobBytes) == hipSuccess
                                   : hipHostMalloc((void **)&g->hBlobs, N * g->blobBytes, hipHostMallocDefault) == hipSuccess;
        for (auto &e : g->uploadEvents) ok = ok && hipEventCreateWithFlags(&e, hipEventDisableTiming) == hipSuccess;
        if (!ok) {
            mv_destroy(g);
            return fail("mv_create: pinned episode staging allocation failed");
        }
        g->feeder = std::make_unique<EpisodeFeeder>(scenario, oc, g->N, g->A, episodeLen, g->hBlobs, g->blobBytes, g->device, g->feederThreads, levelFiles,
                                                    g->blobsOnDevice);
        if (g->feeder->failed()) {
            mv_destroy(g);
            return fail("mv_create: the device-side episode generator's allocations failed");
        }
        std::vector<uint32_t> seeds(N);
        std::random_device rdev;   // unseeded envs take their seed from random_device (env.hpp:169)
        for (auto &v : seeds) v = (uint32_t)rdev();
        g->feeder->reseed(seeds, std::vector<int>(N, 1));
    }
    std::vector<EnvHeader> hh(N);
    std::random_device rd;
    for (size_t i = 0; i < N; ++i) {
        std::memset(&hh[i], 0, sizeof(EnvHeader));
        hh[i].p_episode_len_sec = episodeLen; hh[i].p_vertical_look_limit = lookLimit;
        hh[i].next_seed = (uint32_t)rd(); hh[i].seed_is_env_seed = 1;
        hh[i].bar_half_width = 0.24f;
        hh[i].scenario = scenario;
    }
    std::vector<AgentState> ha(NA);
    for (size_t i = 0; i < NA; ++i) {
        std::memset(&ha[i], 0, sizeof(AgentState));
        for (int k = 0; k < g->numShaping; ++k)
            ha[i].shaping[k] = scenario == SCN_TOWER ? SHAPING_DEFAULT_TOWER[k] : scenario == SCN_COLLECT ? SHAPING_DEFAULT_COLLECT[k]
                                                     : scenario == SCN_REARRANGE ? SHAPING_DEFAULT_REARRANGE[k]
                                                     : scenario == SCN_SOKOBAN ? SHAPING_DEFAULT_SOKOBAN[k]
                                                     : scenario == SCN_HEX_MEMORY ? SHAPING_DEFAULT_HEX_MEMORY[k]
                                                     : scenario == SCN_HEX_EXPLORE ? SHAPING_DEFAULT_HEX_EXPLORE[k]
                                                     : scenario == SCN_EMPTY ? 0.0f
                                                     : (k == 4 ? oc.carried_object_to_exit : SHAPING_DEFAULT_OBST[k]);
        ha[i].carrying = -1; ha[i].jump_speed = 10.0f; ha[i].m00 = 1.0f; ha[i].m22 = 1.0f;
    }
    {
        std::vector<int32_t> iota(NA);
        for (size_t i = 0; i < NA; ++i) iota[i] = (int32_t)i;
        // identity until the first frame sort
        for (int q = 0; q < g->slots; ++q) (void)hipMemcpy(g->gvp[q].lpt_order, iota.data(), NA * sizeof(int32_t), hipMemcpyHostToDevice);
    }
    if (hipMemcpy(gv.hdr, hh.data(), N * sizeof(EnvHeader), hipMemcpyHostToDevice) != hipSuccess ||
        hipMemcpy(gv.agents, ha.data(), NA * sizeof(AgentState), hipMemcpyHostToDevice) != hipSuccess) {
        mv_destroy(g);
        return fail("mv_create: initial upload failed");
    }
    *out = g;
    return 0;
}

int mv_close(mv_gym *g)
{
    if (!g || g->closed) return 0;
    (void)hipSetDevice(g->device);
    if (g->inGroup) {   // a member leaves: everything the group has in flight first, then every member gets its own stream and events back
        (void)hipStreamSynchronize(g->simStream);
        (void)hipStreamSynchronize(g->stream);
        group_detach(g);
    }
    // order: nothing may still target the arena (episode uploads / status read-backs on the copy stream, kernels on the step
    // stream) or the pinned slots (feeder workers) when they are freed.  The step stream may be caller-owned and already gone:
    // its errors are ignored, the device-wide synchronise below covers whatever was enqueued on it.
    if (g->simStream) (void)hipStreamSynchronize(g->simStream);
    (void)hipStreamSynchronize(g->stream);
    (void)hipGetLastError();
    if (g->genStream) (void)hipStreamSynchronize(g->genStream);
    if (g->copyStream) (void)hipStreamSynchronize(g->copyStream);
    (void)hipDeviceSynchronize();
    g->feeder.reset();   // joins the workers before their slots go away
    GymView &gv = g->gv;
    if (gv.dbg) {   // tick phase timing of an instrumented build.  Per env 64 counters: 0..15 sums over all ticks (0..7 phase cycles, 8..12 cast statistics),
                    // 16..31 the same of the last tick, 32..47 of the env's longest tick, 48..55 launch lifetimes on the 100 MHz clock (mv_step.hip)
        const int N = g->N;
        std::vector<unsigned long long> h((size_t)N * 64);
        if (hipMemcpy(h.data(), gv.dbg, h.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost) == hipSuccess) {
            auto at = [&](int e, int k) { return h[(size_t)e * 64 + k]; };
            static const char *names[8] = {"loads", "actions", "physics", "interact", "fall/zone/timers", "write-back",
                    "tick (both waves, to the barrier)", "frame setup (both waves)"};
            for (int k = 0; k < 8; ++k) {
                double sum = 0.0; unsigned long long mx = 0;
                for (int e = 0; e < N; ++e) { sum += (double)at(e, k); mx = std::max(mx, at(e, k)); }
                std::fprintf(stderr, "[mv tick timing] %-36s mean %.0f cycles per env (all steps summed), max env %llu\n", names[k], sum / N, mx);
            }
            {   // frame setup phases (thread 0 of the workgroup), slots 56..61
                static const char *fn[6] = {"cameras", "slot records", "screen rectangles", "list positions", "records written", "header + cost bin"};
                for (int k = 0; k < 6; ++k) {
                    double sum = 0.0;
                    for (int e = 0; e < N; ++e) sum += (double)at(e, 56 + k);
                    std::fprintf(stderr, "[mv tick timing] frame setup: %-20s mean %.0f cycles per env (all frames summed)\n", fn[k], sum / N);
                }
            }
            double cs[5] = {0, 0, 0, 0, 0};
            for (int e = 0; e < N; ++e) for (int k = 0; k < 5; ++k) cs[k] += (double)at(e, 8 + k);
            std::fprintf(stderr, "[mv tick timing] casts, all ticks of all envs: %.0f sweeps, %.0f casts started, %.0f wave iterations slot by slot, %.0f if "
                         "a lane's casts were queued\n", cs[0], cs[4], cs[1], cs[2]);
            const bool tickOnly = at(0, 49) != 0;
            if (tickOnly) {
                int worst = 0; double mxsum = 0.0, wi[5] = {0, 0, 0, 0, 0};
                for (int e = 0; e < N; ++e) {
                    mxsum += (double)at(e, 54);
                    if (at(e, 54) > at(worst, 54)) worst = e;
                    for (int k = 0; k < 5; ++k) wi[k] += (double)at(e, 32 + 8 + k);
                }
                std::fprintf(stderr, "[mv tick timing] longest tick of an env: %.2f us on average over envs, with on average %.1f sweeps, %.1f casts, %.1f "
                             "wave iterations (%.1f queued), longest cast %.1f\n",
                             mxsum / N * 0.01, wi[0] / N, wi[4] / N, wi[1] / N, wi[2] / N, wi[3] / N);
                std::fprintf(stderr, "[mv tick timing] the longest of all (env %d, %.2f us), cycles per phase: loads %llu actions %llu physics %llu interact "
                             "%llu fall %llu write-back %llu; %llu sweeps, %llu casts, %llu wave iterations (%llu queued), longest cast %llu\n",
                             worst, (double)at(worst, 54) * 0.01, at(worst, 32), at(worst, 33), at(worst, 34), at(worst, 35), at(worst, 36), at(worst, 37),
                             at(worst, 40), at(worst, 44), at(worst, 41), at(worst, 42), at(worst, 43));
                double rl = 0.0, rc = 0.0;
                for (int e = 0; e < N; ++e) { rl += (double)at(e, 52); rc += (double)at(e, 53); }
                std::fprintf(stderr, "[mv tick timing] tick-only launches: %.0f ticks regenerated their env and lived %.2f us on average\n",
                             rc, rc > 0 ? rl / rc * 0.01 : 0.0);
            }
            {   // {sum of wave-0 lifetimes, launches, start and end of the last one}
                const int b = tickOnly ? 48 : 52;
                double life = 0.0, cnt = 0.0; unsigned long long s0 = ~0ull, s1 = 0, e1 = 0;
                for (int e = 0; e < N; ++e) {
                    life += (double)at(e, b); cnt += (double)at(e, b + 1);
                    if (at(e, b + 1)) { s0 = std::min(s0, at(e, b + 2)); s1 = std::max(s1, at(e, b + 2)); e1 = std::max(e1, at(e, b + 3)); }
                }
                if (cnt > 0)
                    std::fprintf(stderr, "[mv tick timing] %s launches: wave 0 lives %.2f us on average; last launch: starts spread over %.2f us, first "
                                 "start to last end %.2f us\n",
                                 tickOnly ? "tick-only" : "fused", life / cnt * 0.01, (double)(s1 - s0) * 0.01, (double)(e1 - s0) * 0.01);
            }
        }
        (void)hipFree(gv.dbg);
    }
    if (g->arena) (void)hipFree(g->arena);
    if (g->hiresObs) (void)hipFree(g->hiresObs);
    if (g->hBlobs) (void)(g->blobsOnDevice ? hipFree(g->hBlobs) : hipHostFree(g->hBlobs));
    if (g->hStatus) (void)hipHostFree(g->hStatus);
    if (g->stepDone) (void)hipEventDestroy(g->stepDone);
    if (g->statusCopied) (void)hipEventDestroy(g->statusCopied);
    for (hipEvent_t e : g->uploadEvents) if (e) (void)hipEventDestroy(e);
    g->uploadEvents.clear();
    for (int i = 0; i < 2; ++i) {
        if (g->passStream[i]) (void)hipStreamDestroy(g->passStream[i]);
        if (g->callStart[i]) (void)hipEventDestroy(g->callStart[i]);
        g->passStream[i] = nullptr; g->callStart[i] = nullptr;
    }
    for (hipEvent_t &e : g->drawDone) { if (e) (void)hipEventDestroy(e); e = nullptr; }
    g->genStream = nullptr;
    if (g->copyStream) (void)hipStreamDestroy(g->copyStream);
    if (g->simStream) (void)hipStreamDestroy(g->simStream);
    for (hipEvent_t &e : g->userMark) { if (e) (void)hipEventDestroy(e); e = nullptr; }
    if (g->userNow) (void)hipEventDestroy(g->userNow);
    g->userNow = nullptr;
    if (g->simDone) (void)hipEventDestroy(g->simDone);
    g->simStream = nullptr; g->simDone = nullptr;
    g->hBlobs = nullptr; g->hStatus = nullptr; g->stepDone = g->statusCopied = nullptr; g->lastStep = nullptr;
            g->copyStream = nullptr; g->dBlobs = nullptr; g->dStatus = nullptr;
    g->arena = nullptr;
    for (int b = 0; b < 2; ++b) {
        if (g->hActions[b]) (void)hipHostFree(g->hActions[b]);
        if (g->actionsCopied[b]) (void)hipEventDestroy(g->actionsCopied[b]);
        g->hActions[b] = nullptr; g->actionsCopied[b] = nullptr;
    }
    for (hipEvent_t e : g->profEvents) (void)hipEventDestroy(e);
    g->profEvents.clear();
    gv = GymView{};
    g->ownedObs = g->hiresObs = g->obs = nullptr; g->dMultiDiscrete = nullptr;
    g->closed = true;
    return 0;
}

int mv_destroy(mv_gym *g)
{
    if (!g) return 0;
    mv_close(g);
    delete g;
    return 0;
}

int mv_num_agents(const mv_gym *g) { return g ? g->A : -1; }

int mv_set_stream(mv_gym *g, void *s)
{
    if (check(g)) return -1;
    if (g->inGroup) return fail("mv_set_stream: the gym belongs to a group (set the stream before mv_group_create)");
    HIP_TRY(hipStreamSynchronize(g->simStream));
    HIP_TRY(hipStreamSynchronize(g->stream));
    g->stream = (hipStream_t)s;
    g->simMustWaitUser = true;
    g->markCount = 0;   // (the marks were recorded on the old stream)
    return 0;
}

int mv_set_obs_buffer(mv_gym *g, void *p)
{
    if (check(g)) return -1;
    g->obs = p ? (uint32_t *)p : g->ownedObs;
    return 0;
}

int mv_set_pixel_mode(mv_gym *g, int32_t mode)
{
    if (check(g)) return -1;
    if (mode != MV_PIXELS_EXACT && mode != MV_PIXELS_FAST) return fail("mv_set_pixel_mode: mode must be MV_PIXELS_EXACT (0) or MV_PIXELS_FAST (1)");
    g->fastPixels = mode;
    return 0;
}

int mv_get_pixel_mode(const mv_gym *g) { return g ? g->fastPixels : -1; }

int mv_set_pipelining(mv_gym *g, int32_t on)
{
    if (check(g)) return -1;
    if (g->inGroup) return fail("mv_set_pipelining: the gym belongs to a group (set it before mv_group_create)");
    if (sim_join(g)) return -1;
    g->pipelined = on != 0;
    return 0;
}

int mv_get_pipelining(const mv_gym *g) { return g ? g->pipelined : -1; }

int mv_seed(mv_gym *g, int32_t seed)
{   // MegaverseGym::seed, megaverse.cpp:60-69: master rng -> one randRange(0, 1<<30) per env
    if (check(g)) return -1;
    HIP_TRY(hipSetDevice(g->device));
    g->rng.seed((unsigned long)seed);
    std::vector<uint32_t> seeds(g->N);
    for (int i = 0; i < g->totalEnvs; ++i) {
        const int noise = std::uniform_int_distribution<>{0, (1 << 30) - 1}(g->rng);
        const int rel = i - g->envOffset;
        if (rel >= 0 && rel % g->envStride == 0 && rel / g->envStride < g->N) seeds[rel / g->envStride] = (uint32_t)noise;
    }
    if (g->hostEpisodes()) {   // Env::seed (env.cpp:52-55) on the host-side episode generators
        HIP_TRY(hipStreamSynchronize(g->simStream));
        HIP_TRY(hipStreamSynchronize(g->stream));
        HIP_TRY(hipStreamSynchronize(g->copyStream));
        // the current counts, not the last periodic read-back
        HIP_TRY(hipMemcpy(g->hStatus, g->dStatus, (size_t)(g->N + 2) * sizeof(int), hipMemcpyDeviceToHost));
        g->statusPending = false;
        g->pendingAge = 0;
        g->stepsSinceStatus = 0;
        std::vector<int> first(g->N);
        for (int i = 0; i < g->N; ++i) {
            // the episodes resident on the device were drawn from the old stream: the ring is wiped (sequence number 0 matches
            // nothing) and every env counts as "consumed everything uploaded"
            g->uploaded[i] = g->hStatus[i];
            first[i] = g->uploaded[i] + 1;
        }
        HIP_TRY(hipMemset(g->dBlobs, 0, (size_t)g->N * g->spares * g->blobBytes));
        g->deficit = 0;
        g->refillForce = true;
        g->feeder->reseed(seeds, first);
        return 0;
    }
    // TowerBuilding: Env::seed on the device-side generators -- what they drew ahead from the old stream is dropped, the rings are drawn again from the new one
    if (sim_join(g) || tower_join(g)) return -1;
    uint32_t *d = nullptr;
    HIP_TRY(hipMalloc((void **)&d, g->N * sizeof(uint32_t)));
    HIP_TRY(hipMemcpyAsync(d, seeds.data(), g->N * sizeof(uint32_t), hipMemcpyHostToDevice, g->stream));
    launch_tower_seed(g->gv, d, g->stream);
    launch_tower_draw(g->gv, g->stream);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipStreamSynchronize(g->stream));
    HIP_TRY(hipFree(d));
    return 0;
}

int mv_render(mv_gym *g)
{
    if (check(g)) return -1;
    HIP_TRY(hipSetDevice(g->device));
    if (sim_join(g)) return -1;
    if (take_hist(g, g->stream, true)) return -1;
    if (launch_raster(view(g, g->parity), last_outputs(g).obs, g->w, g->h, g->stream, nullptr, g->fastPixels))
        return fail("mv_render: observation size above 1024x1024");
    HIP_TRY(hipGetLastError());
    return 0;
}

}  // extern "C"

namespace mvapi {
// ---- status flags ----------------------------------------------------------------------------------------------------
// Kernels raise ST_* bits in status[N + 1], the host generators GEN_* bits (mv_gen.h); both are limits the reference does not
// have.  They are reported ONCE, by the mv_step / mv_reset that sees them, and cleared: the gym stays usable.  They are WARNINGS: the call
// that reports one does all of its work and returns 1 instead of 0 (mv_last_error() has the text); -1 stays what it was, a real failure.
int check_status_flags(mv_gym *g)
{
    const int N = g->N;
    const int flags = g->hStatus[N + 1], gen = g->feeder ? g->feeder->take_overflow() : 0;
    if (!flags && !gen) return 0;
    std::string msg;
    if (flags & ST_STARVED) msg += "an env finished again before its next episode was resident (it repeated its done step; the next episodes are being "
        "uploaded now); ";
    if (flags & ST_CANDIDATES) msg += "collision candidate list overflow (more than 128 bodies around one agent); ";
    if (flags & ST_VISIBLE) msg += "a frame had more visible primitives than the raster keeps (256; Collect 1024; Hex* 2048): the excess was not drawn; ";
    if (flags & ST_CHUNK) msg += "an object placement outside the 32 x 16 x 32 voxel chunk was refused (the reference's grid is unbounded); ";
    if (gen & GEN_SLABS) msg += "a generated layout merged into more slabs than an episode record holds (128, Collect 1024): the excess was dropped; ";
    if (gen & GEN_TERRAIN) msg += "more than 16 terrain boxes in a generated episode; ";
    if (gen & GEN_OBJECTS) msg += "more than 80 movable boxes in a generated episode; ";
    if (gen & GEN_REWARDS) msg += "more reward objects than an episode record holds (16, Collect 96); ";
    if (gen & GEN_COORDS) msg += "a generated level extends beyond +-127 voxels (int8 object coordinates); ";
    if (flags) {   // clear the reported bits in the device word (and the mirror): a bit a kernel raised after this read-back is reported next time
        g->hStatus[N + 1] = 0;
        hipLaunchKernelGGL(clear_flags_kernel, dim3(1), dim3(1), 0, g->simOnOwnStream ? g->simStream : g->stream, g->dStatus + N + 1, flags);
        HIP_TRY(hipGetLastError());
    }
    if (!g->warning.empty()) g->warning += " | ";
    g->warning += "capacity limit hit: " + msg + "reported once, this call did all of its work";
    return 1;
}
// what a call that did its work returns: 0, or 1 with the warning text where mv_last_error() finds it
int finish_with_warning(mv_gym *g)
{
    if (g->warning.empty()) return 0;
    g_err = g->warning;
    g->warning.clear();
    return 1;
}

// ---- episode refill protocol (host-generated scenarios) --------------------------------------------------------------
// Each env keeps up to `spares` (2) generated episodes resident in HBM, a ring indexed by the episode's sequence number
// (dBlobs[env][(seq - 1) % spares]); a finished env swaps the next one in at the tail of the step kernel and bumps status[env] /
// status[N].  A copy stream reads the status words back after the step kernel every `statusPeriod` steps; a later mv_step looks at
// them and -- still on the copy stream, ordered against the step stream by events only -- tops the ring up from the feeder's pinned
// slots, where the episodes were generated ahead of time by the worker pool.  The step path itself only ever enqueues; it waits
// for the host only if an env has NO resident episode left and its next one is still being generated.  An upload never overlaps a
// step kernel (stepDone): a finished env must not read a half-written slot.
// One pass over the envs: whoever has a free ring entry (by the consumed counts last read back: g->consumedSeen) and a generated episode waiting in its pinned
// slot gets it uploaded -- one episode per env and pass: the feeder keeps ONE episode per env ahead -- and what is still missing afterwards is g->deficit.
static int upload_pass(mv_gym *g)
{
    const int N = g->N, K = g->spares;
    hipEvent_t ev = g->uploadEvents[g->uploadRing++ % g->uploadEvents.size()];
    HIP_TRY(hipEventSynchronize(ev));   // 64 batches ago
    std::vector<int> &batch = g->uploadBatch;
    batch.clear();
    int deficit = 0;
    bool waited = false;
    // Consecutive envs whose next episode goes to the same ring slot travel as ONE strided copy (rows: the pinned slots, blobBytes apart, to the ring slots,
    // spares x blobBytes apart).  Scenarios whose episodes all last the same -- Empty, Rearrange, Sokoban, Hex*: every env of the batch finishes on the very
    // same tick -- used to pay a thousand hipMemcpyAsync calls, ~5 ms of host time, at every such tick.
    int runFirst = -1, runLen = 0, runSlot = 0;
    size_t runBytes = 0;
    // Host-generated episodes travel on the copy stream (a DMA engine's work, ordered against the step launches by events).  Device-drawn ones are copied
    // device to device by ONE small kernel per pass (collect_blob_copy_kernel), and that goes to the SIMULATION stream itself, in order between the step
    // launches: on the copy stream it would queue behind whatever shares its hardware queue (an observation launch lasts a millisecond) with the step
    // launches waiting for it (r12a: 13.7 M obs/s; on the simulation stream as one hipMemcpyAsync per episode 15.7 M, r12b; the host feeder: 16.8 M).
    hipStream_t up = g->blobsOnDevice ? g->simStream : g->copyStream;
    std::vector<int32_t> devEnvs, devSlots;
    auto flush_run = [&]() -> int {
        if (runLen <= 0) return 0;
        uint8_t *dst = g->dBlobs + ((size_t)runFirst * K + (size_t)runSlot) * g->blobBytes;
        const uint8_t *src = g->hBlobs + (size_t)runFirst * g->blobBytes;
        const hipMemcpyKind kind = g->blobsOnDevice ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice;   // (device-drawn Collect episodes: staging is device memory)
        if (runLen == 1) HIP_TRY(hipMemcpyAsync(dst, src, runBytes, kind, up));
        else HIP_TRY(hipMemcpy2DAsync(dst, (size_t)K * g->blobBytes, src, g->blobBytes, runBytes, (size_t)runLen, kind, up));
        runLen = 0;
        return 0;
    };
    for (int i = 0; i < N; ++i) {
        const int consumed = g->consumedSeen[(size_t)i];
        if (g->uploaded[i] >= consumed + K) continue;            // ring full
        const int need = g->uploaded[i] + 1;
        const bool must = g->uploaded[i] == consumed;            // nothing resident: the next reset would starve
        if (!must && !g->feeder->is_ready(i, need)) { deficit += consumed + K - g->uploaded[i]; continue; }   // later
        size_t bytes = 0;
        const uint8_t *src = g->feeder->wait_ready(i, need, &bytes);
        if (!src) return fail(g->feeder->failed() ? std::string(g->feeder->device_gen() ? "episode feeder: a HIP call of the device-side generator failed (Collect)"
                                                                                       : "episode feeder: a level file could not be read (Sokoban)")
                                                  : "episode feeder: episode " + std::to_string(need)
                                                          + " of env " + std::to_string(i) + " was never generated");
        if (!waited && g->lastStep) { HIP_TRY(hipStreamWaitEvent(up, g->lastStep, 0)); waited = true; }
        const int slot = (need - 1) % K;
        if (g->blobsOnDevice) {   // (device-drawn: gathered, one copy launch for the pass below)
            devEnvs.push_back(i); devSlots.push_back(slot);
            ++g->uploaded[i];
            deficit += consumed + K - g->uploaded[i];
            batch.push_back(i);
            continue;
        }
        if (runLen > 0 && (i != runFirst + runLen || slot != runSlot) && flush_run()) return -1;
        if (runLen == 0) { runFirst = i; runSlot = slot; runBytes = 0; }
        ++runLen;
        runBytes = std::max(runBytes, bytes);   // (the used prefix of the longest record of the run: what lies behind a shorter one's is never read)
        (void)src;
        ++g->uploaded[i];
        deficit += consumed + K - g->uploaded[i];
        batch.push_back(i);
    }
    if (flush_run()) return -1;
    if (!devEnvs.empty()) {
        launch_collect_blob_copy(devEnvs.data(), devSlots.data(), (int)devEnvs.size(), g->hBlobs, g->dBlobs, g->blobBytes, K, up);
        HIP_TRY(hipGetLastError());
    }
    if (!batch.empty()) {
        HIP_TRY(hipEventRecord(ev, up));
        for (int i : batch) g->feeder->recycle(i, ev);   // regenerate a slot only once its upload has left it
        HIP_TRY(hipStreamWaitEvent(g->simStream, ev, 0));   // (a step that runs on the caller's stream, and mv_reset, wait for lastUpload themselves)
        g->lastUpload = ev;
        g->uploadNotOnUser = true;
    }
    g->deficit = deficit;
    return 0;
}

int refill_episodes(mv_gym *g, int k)
{
    if (g->statusPending) {
        // The read-back was enqueued behind a step launch the host is normally ahead of: waiting for it at once would drain that run-ahead every statusPeriod
        // ticks.  With long episodes (period 16, two resident episodes per env) the words may arrive later: look again at the next call -- every period a fresh
        // read-back takes the pending one's place, so a host that runs ahead never finds it ready -- and wait for the latest one once max(32, 4 k) TICKS have
        // been enqueued since the first.  That wait is what bounds the host's run-ahead, and with it how late a refill can land: an upload is ordered behind
        // the LAST step launch enqueued.  It costs nothing: the host catches up with the STEP launches, which run up to three calls ahead of the observation
        // passes the device is busy with.  (Until round 5 the bound was 32 CALLS -- 256 ticks at 8 per call, 512 at 16: a HexExplore env that found its goal
        // twice within ~300 ticks starved, scripts/soak.py in r08z.  Measured, r08x2, bound 3 k / 6 k / 32 k ticks at k = 16: TowerBuilding 28.8 / 28.7 / 28.1
        // M obs/s, Empty 38.3 / 39.4 / 39.9; with ONE read-back in flight instead (polled until ready, no forced wait) the host's run-ahead was bounded by
        // nothing: 26.3-28.9
        // / 36-38.
        // r08x4, three runs each, this scheme / the 32-call bound: ObstaclesHard 512 envs 20.9 / 21.5, Empty 38.9 / 39.6: what the bound costs.)
        const int bound = std::max(32, 4 * k);
        if (g->statusPeriod > 1 && g->pendingAge < bound && hipEventQuery(g->statusCopied) == hipErrorNotReady) {
            (void)hipGetLastError();   // ("not ready" is an answer, not an error to report at the end of the step)
            g->pendingAge += k;
            // (no fresh counts: but envs known to be short of an episode whose successor was not generated yet -- or had just sent one: one episode per env and
            // pass -- are served now, not at the next read-back: with every env finishing every 70 ticks and a pass every 80 the ring fell behind until it
            // starved)
            if (g->hostEpisodes() && g->deficit > 0 && !g->consumedSeen.empty() && upload_pass(g)) return -1;
            return 0;
        }
        (void)hipGetLastError();
        HIP_TRY(hipEventSynchronize(g->statusCopied));
        g->statusPending = false;
        g->pendingAge = 0;
    }
    const int N = g->N;
    const bool starved = (g->hStatus[N + 1] & ST_STARVED) != 0;
    if (starved && g->hostEpisodes()) {   // recover: take the current counts and upload synchronously below
        HIP_TRY(hipStreamSynchronize(g->simStream));
        // (closed-loop / unpipelined steps run on the caller's stream, which may be a non-blocking one)
        if (!g->simOnOwnStream) HIP_TRY(hipStreamSynchronize(g->stream));
        HIP_TRY(hipStreamSynchronize(g->copyStream));
        const int keep = g->hStatus[N + 1];
        HIP_TRY(hipMemcpy(g->hStatus, g->dStatus, (size_t)(N + 2) * sizeof(int), hipMemcpyDeviceToHost));
        g->hStatus[N + 1] |= keep;
        g->refillForce = true;
    }
    if (g->hostEpisodes() && (g->refillForce || g->deficit > 0 || g->hStatus[N] != g->lastTotalSeen)) {
        // (the pinned words are the target of the next read-back: the passes between two of them work from this copy)
        g->consumedSeen.assign(g->hStatus, g->hStatus + N);
        if (upload_pass(g)) return -1;
        g->lastTotalSeen = g->hStatus[N];
        g->refillForce = false;
    }
    return check_status_flags(g) < 0 ? -1 : 0;
}

// after a step / reset kernel: read the status words back without touching the step path
int read_back_status(mv_gym *g, hipEvent_t after)
{
    HIP_TRY(hipStreamWaitEvent(g->copyStream, after, 0));
    HIP_TRY(hipMemcpyAsync(g->hStatus, g->dStatus, (size_t)(g->N + 2) * sizeof(int), hipMemcpyDeviceToHost, g->copyStream));
    HIP_TRY(hipEventRecord(g->statusCopied, g->copyStream));
    if (!g->statusPending) g->pendingAge = 0;   // (a read-back issued while one is pending takes its place -- the event is re-recorded -- and keeps its age)
    g->statusPending = true;
    return 0;
}

// mv_set_actions_device keeps the caller's multi-discrete buffer until the next step kernel reads it.  Whatever else touches the actions or
// comes between the two -- a reset, a host-side setter -- first turns the pending buffer into bitmasks (on the caller's stream, where the
// buffer's producer ran), so that the buffer is read NOW, while it is certainly alive, and the last writer wins as it did when
// mv_set_actions_device converted at once.
int flush_device_actions(mv_gym *g)
{
    if (!g->mdActions) return 0;
    const int n = g->N * g->A;
    hipLaunchKernelGGL(masks_from_multidiscrete_kernel, dim3((n + 255) / 256), dim3(256), 0, g->stream, g->mdActions, g->gv.actions, n);
    HIP_TRY(hipGetLastError());
    g->mdActions = nullptr;
    g->simMustWaitUser = true;
    return 0;
}

}  // namespace mvapi

extern "C" {

int mv_reset(mv_gym *g)
{   // MegaverseGym::reset (megaverse.cpp:76-93) -> VectorEnv::reset (vector_env.cpp:110-120)
    if (check(g)) return -1;
    HIP_TRY(hipSetDevice(g->device));
    if (sim_join(g)) return -1;
    if (flush_device_actions(g)) return -1;   // (a buffer handed over before the reset is read now, not by whatever step comes after it)
    if (g->hostEpisodes()) {
        // the periodic status read-back may be up to 15 ticks old: an env that auto-reset since then has consumed its
        // resident episode without the host knowing -- take the current counts before deciding what to upload
        HIP_TRY(hipStreamSynchronize(g->simStream));
        HIP_TRY(hipStreamSynchronize(g->stream));
        HIP_TRY(hipStreamSynchronize(g->copyStream));
        HIP_TRY(hipMemcpy(g->hStatus, g->dStatus, (size_t)(g->N + 2) * sizeof(int), hipMemcpyDeviceToHost));
        g->statusPending = false;
        g->pendingAge = 0;
        g->stepsSinceStatus = 0;
        g->refillForce = true;
        if (refill_episodes(g, 1) < 0) return -1;       // every env has an unconsumed episode resident
        if (g->lastUpload) HIP_TRY(hipStreamWaitEvent(g->stream, g->lastUpload, 0));
        const OutPtrs outs = last_outputs(g);
        const GymView v = view(g, g->parity, &outs);
        if (g->scenario == SCN_OBSTACLES || g->scenario == SCN_EMPTY) launch_reset_obstacles(v, (const EpisodeBlob *)g->dBlobs, g->dStatus, 1, g->stream);
        else if (g->scenario == SCN_REARRANGE) launch_reset_rearrange(v, (const RearrangeBlob *)g->dBlobs, g->dStatus, 1, g->stream);
        else if (g->scenario == SCN_SOKOBAN) launch_reset_sokoban(v, (const SokobanBlob *)g->dBlobs, g->dStatus, 1, g->stream);
        else if (g->scenario == SCN_HEX_MEMORY || g->scenario == SCN_HEX_EXPLORE) launch_reset_hex(v, (const HexBlob *)g->dBlobs, g->dStatus, 1, g->stream);
        else launch_reset_collect(v, (const CollectBlob *)g->dBlobs, g->dStatus, 1, g->stream);
        HIP_TRY(hipEventRecord(g->stepDone, g->stream));   // (the reset kernel reads the ring too)
        g->lastStep = g->stepDone;
        if (read_back_status(g, g->stepDone)) return -1;  // the second resident episodes go up with the next steps
    } else {   // TowerBuilding: every ring topped up, every env takes its next episode, the rings topped up again
        if (tower_join(g)) return -1;
        const OutPtrs outs = last_outputs(g);
        const GymView v = view(g, g->parity, &outs);
        launch_tower_draw(v, g->stream);
        launch_reset(v, 1, g->stream);
        launch_tower_draw(v, g->stream);
    }
    HIP_TRY(hipGetLastError());
    g->wasReset = true;
    g->mirrorsFresh = false;
    if (mv_render(g)) return -1;
    return finish_with_warning(g);
}

int mv_set_actions(mv_gym *g, int32_t env, int32_t agent, const int32_t *actions, int32_t n)
{
    if (check(g)) return -1;
    if (env < 0 || env >= g->N || agent < 0 || agent >= g->A) return fail("mv_set_actions: index out of range");
    if (g->mdActions) {   // device actions pending: they are converted now, and the host's (uploaded by the next step) then win, as the last writer
        HIP_TRY(hipSetDevice(g->device));
        if (flush_device_actions(g)) return -1;
    }
    int idx = 0, mask = 0;
    for (int i = 0; i < n && i < 6; ++i) {
        if (actions[i] > 0) mask |= 1 << (idx + actions[i]);
        idx += ACTION_SPACE[i] - 1;
    }
    g->hActions[g->stage][(size_t)env * g->A + agent] = mask;
    g->actionsDirty = true;
    return 0;
}

int mv_set_actions_batched(mv_gym *g, const int32_t *host_actions)
{
    if (check(g)) return -1;
    HIP_TRY(hipSetDevice(g->device));
    const int n = g->N * g->A;
    g->mdActions = nullptr;   // (this call sets every agent's action: a pending device buffer is superseded, and no longer referenced)
    HIP_TRY(hipMemcpyAsync(g->dMultiDiscrete, host_actions, (size_t)n * 6 * sizeof(int32_t), hipMemcpyHostToDevice, g->stream));
    hipLaunchKernelGGL(masks_from_multidiscrete_kernel, dim3((n + 255) / 256), dim3(256), 0, g->stream, g->dMultiDiscrete, g->gv.actions, n);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipStreamSynchronize(g->stream));   // host_actions may be pageable and reused by the caller
    g->simMustWaitUser = true;
    return 0;
}

int mv_set_actions_device(mv_gym *g, const int32_t *device_actions)
{
    if (check(g)) return -1;
    HIP_TRY(hipSetDevice(g->device));
    if (!device_actions) return fail("mv_set_actions_device: null pointer");
    // No launch here: the next step kernel converts multi-discrete -> bitmask itself (mv_actions.h: action_of), one kernel boundary less per
    // tick of a policy in the loop.  The buffer is read when that step kernel runs, in the order of the caller's stream.
    g->mdActions = device_actions;
    g->simMustWaitUser = true;   // the actions come from the caller's stream (a policy that read the last observations): a true dependency
    return 0;
}

int mv_set_sample_policy(mv_gym *g, int32_t policy)
{
    if (check(g)) return -1;
    if (policy != MV_POLICY_MULTIDISCRETE
        && policy != MV_POLICY_SINGLE_BIT) return fail("mv_set_sample_policy: MV_POLICY_MULTIDISCRETE (1) or MV_POLICY_SINGLE_BIT (2)");
    g->samplePolicy = policy;
    return 0;
}

int mv_set_output_ring(mv_gym *g, int32_t count, void *obs, float *rewards, uint8_t *dones)
{
    if (check(g)) return -1;
    if (count < 0 || (count > 0 && !obs && !rewards && !dones)) return fail("mv_set_output_ring: count >= 0 and at least one ring required");
    if (sim_join(g)) return -1;   // (ticks in flight keep the pointers they were given)
    g->ringCount = count; g->ringTick = 0;
    g->ringObs = count ? (uint8_t *)obs : nullptr; g->ringRewards = count ? rewards : nullptr; g->ringDone = count ? dones : nullptr;
    g->mirrorsFresh = false;
    return 0;
}

int mv_set_pass_overlap(mv_gym *g, int32_t on)
{
    if (check(g)) return -1;
    HIP_TRY(hipSetDevice(g->device));
    if (on && !g->passStream[0]) {
        for (int i = 0; i < 2; ++i) {
            HIP_TRY(hipStreamCreateWithFlags(&g->passStream[i], hipStreamNonBlocking));
            HIP_TRY(hipEventCreateWithFlags(&g->callStart[i], hipEventDisableTiming));
        }
    }
    if (!on && g->passStream[0]) {   // switched off: the two pass streams go (a process holds few hardware queues: idle streams are not free for the gyms made later)
        if (sim_join(g)) return -1;
        HIP_TRY(hipStreamSynchronize(g->stream));
        for (int i = 0; i < 2; ++i) {
            HIP_TRY(hipStreamSynchronize(g->passStream[i]));
            (void)hipStreamDestroy(g->passStream[i]); (void)hipEventDestroy(g->callStart[i]);
            g->passStream[i] = nullptr; g->callStart[i] = nullptr;
        }
    }
    g->passOverlap = on ? 1 : 0;
    g->overlapCalls = 0;
    return 0;
}

int mv_sample_random_actions(mv_gym *g, uint32_t seed, uint32_t step)
{   // the draw itself happens inside the next step kernel (mv_actions.h): no separate launch, no action buffer traffic
    if (check(g)) return -1;
    g->samplePending = true;
    g->gv.sample_seed = seed;
    g->gv.sample_step = step;
    return 0;
}

int mv_synchronize(mv_gym *g)
{
    if (check(g)) return -1;
    HIP_TRY(hipStreamSynchronize(g->simStream));
    HIP_TRY(hipStreamSynchronize(g->stream));
    return 0;
}

int mv_is_done(mv_gym *g, int32_t env)
{
    if (check(g)) return -1;
    if (env < 0 || env >= g->N) return fail("mv_is_done: index out of range");
    if (refresh_mirrors(g)) return -1;
    return g->hDone[env] ? 1 : 0;
}

int mv_get_dones(mv_gym *g, uint8_t *out)
{
    if (check(g) || refresh_mirrors(g)) return -1;
    std::memcpy(out, g->hDone.data(), g->N);
    return 0;
}

int mv_get_last_rewards(mv_gym *g, float *out)
{
    if (check(g) || refresh_mirrors(g)) return -1;
    std::memcpy(out, g->hRewards.data(), (size_t)g->N * g->A * sizeof(float));
    return 0;
}

int mv_true_objective(mv_gym *g, int32_t env, int32_t agent, float *out)
{
    if (check(g)) return -1;
    if (env < 0 || env >= g->N || agent < 0 || agent >= g->A) return fail("mv_true_objective: index out of range");
    if (refresh_mirrors(g)) return -1;
    *out = g->hTrueObj[(size_t)env * g->A + agent];
    return 0;
}

int mv_get_true_objectives(mv_gym *g, float *out)
{
    if (check(g) || refresh_mirrors(g)) return -1;
    std::memcpy(out, g->hTrueObj.data(), (size_t)g->N * g->A * sizeof(float));
    return 0;
}

int mv_get_observation(mv_gym *g, int32_t env, int32_t agent, uint8_t *out)
{
    if (check(g)) return -1;
    if (env < 0 || env >= g->N || agent < 0 || agent >= g->A) return fail("mv_get_observation: index out of range");
    const size_t frameBytes = (size_t)g->w * g->h * 4;
    HIP_TRY(hipMemcpyAsync(out, (const uint8_t *)last_outputs(g).obs + ((size_t)env * g->A + agent) * frameBytes,
            frameBytes, hipMemcpyDeviceToHost, g->stream));
    HIP_TRY(hipStreamSynchronize(g->stream));
    return 0;
}

void *mv_obs_device_ptr(mv_gym *g) { return (g && !g->closed) ? g->obs : nullptr; }
void *mv_rewards_device_ptr(mv_gym *g) { return (g && !g->closed) ? g->gv.rewards : nullptr; }
void *mv_dones_device_ptr(mv_gym *g) { return (g && !g->closed) ? g->gv.done : nullptr; }
void *mv_true_objectives_device_ptr(mv_gym *g) { return (g && !g->closed) ? g->gv.true_objective : nullptr; }

int mv_set_render_resolution(mv_gym *g, int32_t w, int32_t h)
{
    if (check(g)) return -1;
    if (w < 1 || h < 1) return fail("mv_set_render_resolution: bad size");
    g->renderW = w; g->renderH = h;
    return 0;
}

int mv_draw_hires(mv_gym *g)
{   // MegaverseGym::drawHires, megaverse.cpp:154-178: a second renderer at renderW x renderH
    if (check(g)) return -1;
    HIP_TRY(hipSetDevice(g->device));
    if (!g->hiresObs || g->hiresW != g->renderW || g->hiresH != g->renderH) {
        if (g->hiresObs) { HIP_TRY(hipStreamSynchronize(g->stream)); HIP_TRY(hipFree(g->hiresObs)); g->hiresObs = nullptr; }
        HIP_TRY(hipMalloc((void **)&g->hiresObs, (size_t)g->N * g->A * g->renderW * g->renderH * 4));
        g->hiresW = g->renderW; g->hiresH = g->renderH;
    }
    if (sim_join(g)) return -1;
    if (take_hist(g, g->stream, true)) return -1;
    if (launch_raster(view(g, g->parity), g->hiresObs, g->hiresW, g->hiresH, g->stream, nullptr,
        g->fastPixels)) return fail("mv_draw_hires: render size above 1024x1024");
    HIP_TRY(hipGetLastError());
    return 0;
}

int mv_get_hires_observation(mv_gym *g, int32_t env, int32_t agent, uint8_t *out)
{
    if (check(g)) return -1;
    if (!g->hiresObs) return fail("mv_get_hires_observation: call mv_draw_hires first");
    if (env < 0 || env >= g->N || agent < 0 || agent >= g->A) return fail("mv_get_hires_observation: index out of range");
    const size_t frameBytes = (size_t)g->hiresW * g->hiresH * 4;
    HIP_TRY(hipMemcpyAsync(out, (const uint8_t *)g->hiresObs + ((size_t)env * g->A + agent) * frameBytes, frameBytes, hipMemcpyDeviceToHost, g->stream));
    HIP_TRY(hipStreamSynchronize(g->stream));
    return 0;
}

int mv_draw_overview(mv_gym *g) { return check(g) ? -1 : 0; }

int mv_num_reward_shaping_keys(const mv_gym *g) { return g ? g->numShaping : 0; }
const char *mv_reward_shaping_key(const mv_gym *g, int32_t i) { return (g && i >= 0 && i < g->numShaping) ? g->shapingKeys[i] : nullptr; }

static int shaping_index(const mv_gym *g, const char *key)
{
    for (int k = 0; k < g->numShaping; ++k)
        if (!std::strcmp(key, g->shapingKeys[k])) return k;
    return -1;
}

int mv_get_reward_shaping(mv_gym *g, int32_t env, int32_t agent, const char *key, float *out)
{
    if (check(g)) return -1;
    const int k = shaping_index(g, key);
    if (k < 0) return fail(std::string("unknown reward shaping key ") + key);
    if (env < 0 || env >= g->N || agent < 0 || agent >= g->A) return fail("mv_get_reward_shaping: index out of range");
    if (sim_join(g)) return -1;
    HIP_TRY(hipMemcpyAsync(out, &g->gv.agents[(size_t)env * g->A + agent].shaping[k], sizeof(float), hipMemcpyDeviceToHost, g->stream));
    HIP_TRY(hipStreamSynchronize(g->stream));
    return 0;
}

int mv_set_reward_shaping(mv_gym *g, int32_t env, int32_t agent, const char *key, float v)
{
    if (check(g)) return -1;
    const int k = shaping_index(g, key);
    if (k < 0) return fail(std::string("unknown reward shaping key ") + key);
    if (env < 0 || env >= g->N || agent < 0 || agent >= g->A) return fail("mv_set_reward_shaping: index out of range");
    if (sim_join(g)) return -1;
    hipLaunchKernelGGL(set_shaping_kernel, dim3(1), dim3(1), 0, g->stream, g->gv.agents, env * g->A + agent, k, v);
    HIP_TRY(hipGetLastError());
    return 0;
}

}  // extern "C"
