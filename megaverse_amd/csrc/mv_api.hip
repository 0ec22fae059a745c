// megaverse_amd/csrc/mv_api.hip -- host side of libmegaverse_hip.so: the C ABI declared in
// include/megaverse_hip.h, HBM allocation, kernel sequencing on one HIP stream.
//
// Mirrors class MegaverseGym of the reference (src/libs/bindings/megaverse.cpp:34-262) method by
// method; the per-step control flow mirrors VectorEnv::step (src/libs/env/src/vector_env.cpp:89-108):
//   step all envs  ->  for done envs: record trueObjective, reset  ->  draw.
// There is NO CPU fallback: if no HIP device can be opened mv_create fails.
#include <hip/hip_runtime.h>

#include <sched.h>

#include <algorithm>
#include <cctype>
#include <cstdio>
#include <cstring>
#include <memory>
#include <random>
#include <string>
#include <thread>
#include <vector>

#include "../../include/megaverse_hip.h"
#include "mv_feeder.h"
#include "mv_actions.h"
#include "mv_gen.h"
#include "mv_raster.h"
#include "mv_math.h"
#include "mv_rng.h"
#include "mv_types.h"
#include "mv_union.h"

namespace mv {
void launch_reset(const GymView &gv, int force_all, hipStream_t stream);
void launch_tower_draw(const GymView &gv, hipStream_t stream);                          // TowerBuilding: tops every env's ring of drawn episodes up (mv_reset.hip)
void launch_tower_seed(const GymView &gv, const uint32_t *seeds, hipStream_t stream);   // Env::seed for every env's generator
// step kernels: one 256-thread workgroup per env = the tick (wave 0) + the frame setup of the env's frames for a W x H observation
// (render = 0: tick only)
void launch_step(const GymView &gv, hipStream_t stream, int W, int H, int render, hipEvent_t done = nullptr);
void launch_step_ticks(const GymView *views, int k, hipStream_t stream, int W, int H, hipEvent_t done = nullptr);   // k ticks + frame setups of every env, one launch (mv_step.hip)
void launch_step_obstacles(const GymView &gv, hipStream_t stream, int W, int H, int render);
void launch_step_obstacles_ticks(const GymView *views, int k, hipStream_t stream, int W, int H);   // k ticks + frame setups of every env (one agent), one launch
void launch_step_rearrange_ticks(const GymView *views, int k, hipStream_t stream, int W, int H);
void launch_step_sokoban_ticks(const GymView *views, int k, hipStream_t stream, int W, int H);
void launch_step_collect_ticks(const GymView *views, int k, hipStream_t stream, int W, int H);
void launch_step_hex_ticks(const GymView *views, int k, hipStream_t stream, int W, int H);
void launch_reset_obstacles(const GymView &gv, const EpisodeBlob *blobs, int *status, int force_all, hipStream_t stream);
void launch_step_rearrange(const GymView &gv, hipStream_t stream, int W, int H, int render);
void launch_reset_rearrange(const GymView &gv, const RearrangeBlob *blobs, int *status, int force_all, hipStream_t stream);
void launch_step_collect(const GymView &gv, hipStream_t stream, int W, int H, int render);
void launch_step_sokoban(const GymView &gv, hipStream_t stream, int W, int H, int render);
void launch_reset_sokoban(const GymView &gv, const SokobanBlob *blobs, int *status, int force_all, hipStream_t stream);
void launch_reset_collect(const GymView &gv, const CollectBlob *blobs, int *status, int force_all, hipStream_t stream);
void launch_step_hex(const GymView &gv, hipStream_t stream, int W, int H, int render);
void launch_reset_hex(const GymView &gv, const HexBlob *blobs, int *status, int force_all, hipStream_t stream);
}  // namespace mv

using namespace mv;

static thread_local std::string g_err;
static int fail(const std::string &msg)
{
    g_err = msg;
    return -1;
}
#define HIP_TRY(expr)                                                                                  \
    do {                                                                                               \
        hipError_t e_ = (expr);                                                                        \
        if (e_ != hipSuccess) return fail(std::string(#expr) + ": " + hipGetErrorString(e_));          \
    } while (0)

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
static const char *SHAPING_KEYS_EMPTY[1] = {"teamSpirit"};   // EmptyScenario::defaultRewardShaping() is {} (scenario_empty.hpp:28) + Scenario::init's teamSpirit
// scenario_hex_memory.hpp:37-43, scenario_hex_explore.hpp:28-31 (+ teamSpirit 0)
static const char *SHAPING_KEYS_HEX_MEMORY[3] = {"teamSpirit", "memoryCollectGood", "memoryCollectBad"};
static const float SHAPING_DEFAULT_HEX_MEMORY[3] = {0.0f, 1.0f, -1.0f};
static const char *SHAPING_KEYS_HEX_EXPLORE[2] = {"teamSpirit", "exploreSolved"};
static const float SHAPING_DEFAULT_HEX_EXPLORE[2] = {0.0f, 5.0f};
static const char *SHAPING_KEYS_COLLECT[5] = {"teamSpirit", "collectSingleGood", "collectSingleBad", "collectAll", "collectAbyss"};
static const float SHAPING_DEFAULT_COLLECT[5] = {0.0f, 1.0f, -1.0f, 5.0f, -0.5f};
static const int ACTION_SPACE[6] = {3, 3, 3, 2, 2, 3};                          // env.cpp:33

static_assert(PIPE_GROUPS == 3, "userMark events are created one by one in mv_create");
struct mv_gym;
struct mv_group {
    std::vector<mv_gym *> gyms;   // gyms[0] is the leader; empty once a member was closed
};
struct mv_gym {
    int device = 0;
    int w = 0, h = 0, renderW = 768, renderH = 432;   // megaverse.cpp:261
    int N = 0, A = 0, envOffset = 0, envStride = 1, totalEnvs = 0;
    bool samplePending = false;                  // mv_sample_random_actions: the next step draws its own actions
    int samplePolicy = POLICY_MULTIDISCRETE;     // mv_set_sample_policy: which generator mv_sample_random_actions requests
    bool closed = false, wasReset = false;
    hipStream_t stream = nullptr;                // the caller's stream: observation passes, published outputs, everything it may consume
    // One-step-ahead pipelining: the step kernels run on an internal stream.  A step only waits for what the caller had
    // enqueued on its stream when the PREVIOUS mv_step began (consumers of outputs two steps old), so step t + 1 overlaps the observation
    // pass of step t whenever nothing on the caller's stream feeds it (device-sampled or host-provided actions).  Everything a step
    // hands to the observation pass or to the caller exists PIPE_BUFS times: frame lists / headers / cost lists, and the rewards /
    // dones / true objectives, which the observation pass (on the caller's stream) publishes into the stable public arrays.
    // Slots: PIPE_GROUPS groups of `batch` hand-over buffers.  One call -- mv_step (one tick) or mv_step_n (up to `batch` ticks) -- takes the
    // next group; its step kernels wait for the mark recorded PIPE_GROUPS - 1 calls ago.  With k ticks per call the two cross-queue
    // hand-overs (mark -> simulation stream, simDone -> caller's stream, ~10 us of command-processor time each) are paid once per k ticks.
    hipStream_t simStream = nullptr;
    int pipelined = 1;                           // mv_set_pipelining / MV_PIPELINE: 0 = everything on the caller's stream, in order
    bool simOnOwnStream = false;                 // where the last step ran
    hipEvent_t userMark[PIPE_GROUPS] = {};       // completed when the last observation pass of a stepping call is, round-robin over the calls
    hipEvent_t userNow = nullptr;                // recorded on `stream` when the simulation must wait for all of it
    long dbgCalls = 0;                           // (instrumented builds)
    unsigned long long markCount = 0;             // (64 bits: a training run takes 2^31 steps in a day and a half)
    bool simMustWaitUser = true;                 // the caller's stream holds work the next step depends on (reset, render, device actions, ...)
    hipEvent_t simDone = nullptr;                // after the last kernel on simStream
    bool simDoneValid = false;
    int batch = 8, slots = PIPE_GROUPS * 8, hists = PIPE_GROUPS * 8 + 1;   // slots per group (MV_PIPE_BATCH), slots, cost histograms
    int group = 0;                               // slot group of the last stepping call
    int parity = 0, hist3 = 0;                   // hand-over slot of the last tick; cost histogram of the last pass
    // histClean[h]: cost histogram h is (or, in stream order, will be) all zero when the next frame setup counts into it.  A pass drawn by the
    // one-launch observation kernel clears its histogram itself once its last workgroup has looked its frame up (mv_raster.hip: hist_done); a
    // tick of the one-launch-per-tick path clears the NEXT pass's in its frame setup (mv_frame.h); what neither covers -- the hand-over between
    // the two paths -- is cleared by take_hist with a memset.
    std::vector<uint8_t> histClean;
    std::vector<GymView> gvp;                    // [slots] gv with the buffers of each slot swapped in
    GymView gv{};
    const int32_t *mdActions = nullptr;          // mv_set_actions_device: the caller's multi-discrete buffer, read by the next step kernel
    // mv_set_pass_overlap(1), ring at least two calls deep: the one-launch observation passes of consecutive batched calls go to two internal streams
    // in turn, so that the passes of call c + 1 start -- their step launch permitting -- while those of call c drain (a launch ends with its last
    // workgroups finishing alone, and the next one could not begin before: ~7 % of a 1024-env call).  The caller's stream waits for every call's
    // passes as before; what the passes of call c wait for on the caller's side is what was enqueued before call c - 1 began (callStart).
    int passOverlap = 0;
    hipStream_t passStream[2] = {nullptr, nullptr};
    hipEvent_t callStart[2] = {nullptr, nullptr};
    unsigned long long overlapCalls = 0;   // consecutive calls that took the overlapped path (0: the last call's passes ran on the caller's stream)
    // mv_set_output_ring: tick number t (since the ring was set) leaves its observations / rewards / dones in entry t % ringCount
    int ringCount = 0;
    unsigned long long ringTick = 0;
    uint8_t *ringObs = nullptr, *ringDone = nullptr;
    float *ringRewards = nullptr;
    std::string warning;                         // soft conditions (capacity flags) of the last call: returned as 1, not as an error
    // mv_group: the gyms of a group share the leader's simulation stream and events; a member keeps its own handles here until it leaves
    mv_group *inGroup = nullptr;
    hipStream_t ownSimStream = nullptr;
    hipEvent_t ownUserMark[PIPE_GROUPS] = {}, ownSimDone = nullptr, ownStepDone = nullptr;
    uint8_t *arena = nullptr;
    uint32_t *obs = nullptr, *ownedObs = nullptr, *hiresObs = nullptr;
    int hiresW = 0, hiresH = 0;
    int fastPixels = 1;                          // mv_set_pixel_mode: 1 = raster_fast_kernel (default), 0 = bit-exact raster_kernel
    // host mirrors
    int32_t *hActions[2] = {nullptr, nullptr};   // pinned staging, double buffered
    hipEvent_t actionsCopied[2] = {nullptr, nullptr};
    int stage = 0;
    bool actionsDirty = false;
    int32_t *dMultiDiscrete = nullptr;           // [N*A*6] scratch for batched host actions
    std::vector<float> hRewards, hTrueObj;
    std::vector<uint8_t> hDone;
    bool mirrorsFresh = false;
    std::mt19937 rng{std::random_device{}()};    // megaverse.cpp:253
    // scenario
    int scenario = SCN_TOWER;
    int numShaping = 4;
    const char *const *shapingKeys = SHAPING_KEYS_TOWER;
    ObstacleConfig obst;
    float baseEpisodeLen = 60.0f;
    // Obstacles / Collect: background episode feeder + one resident episode per env (refill protocol below)
    std::unique_ptr<EpisodeFeeder> feeder;
    int feederThreads = 1;
    std::vector<int> uploaded, uploadBatch;         // episodes uploaded per env; envs of the current upload batch
    uint8_t *dBlobs = nullptr, *hBlobs = nullptr;   // device [N][blobBytes], pinned feeder slots [N][blobBytes]
    size_t blobBytes = 0;                           // sizeof(EpisodeBlob) or sizeof(CollectBlob)
    bool hostEpisodes() const { return scenario != SCN_TOWER; }
    // TowerBuilding: the episode generator's serial half (tower_draw_kernel, ~47 us of one wavefront per finished env) runs on a stream of its own,
    // behind the step launch whose finished envs it refills and beside everything else; a stepping call waits for the draw launch BEFORE the last one
    // (two episodes are resident per env: what the last launch is still drawing is not needed yet).  drawPeriod: ticks between draw launches -- 8 where
    // episodes last at least 64 ticks, every call where they can be a few ticks long (those calls are one tick each: mv_step_n).
    hipStream_t genStream = nullptr;
    hipEvent_t stepForDraw = nullptr, drawDone[2] = {nullptr, nullptr};
    unsigned long long drawCount = 0;
    int ticksSinceDraw = 0, drawPeriod = 1;
    int *dStatus = nullptr, *hStatus = nullptr;     // [N + 2]: consumed per env, total, error flags (device, pinned mirror)
    int lastTotalSeen = 0;
    bool statusPending = false, refillForce = true;
    int pendingAge = 0;                             // steps since the pending read-back was first looked for
    int stepsSinceStatus = 0;
    int spares = 2;                                 // resident episodes per env (ring); the host keeps uploaded <= consumed + spares
    int statusPeriod = 16;                          // steps between status read-backs (1 when episodes can be only a few ticks long)
    int deficit = 0;                                // spares still to be uploaded (their episodes were not generated yet at the last look)
    hipEvent_t stepDone = nullptr;                  // after the last step kernel: uploads never overlap a kernel that may read the ring
    hipStream_t copyStream = nullptr;               // status read-back + episode uploads, off the step path
    hipEvent_t resetDone = nullptr, statusCopied = nullptr;
    bool stepDoneValid = false;
    std::vector<hipEvent_t> uploadEvents;           // ring, one per upload batch
    hipEvent_t lastUpload = nullptr;                // the most recent batch (mv_reset: the caller's stream waits for it too)
    bool uploadNotOnUser = false;                   // ... and a step that runs on the caller's stream has not waited for it yet
    size_t uploadRing = 0;
    // in-stream profiling
    std::vector<hipEvent_t> profEvents;          // 5 per profiled tick: [0] [1] around the step kernel (its stream), [2] [3] [4] before the
                                                 // observation pass, between frame sort and raster, after the raster (the caller's stream)
    int profMax = 0, profCount = 0;
    std::vector<int> profTicks;                  // ticks an entry covers: 1, or the k ticks of a batched call whose launches are timed as a whole
};

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
__global__ void set_agent_pos_kernel(AgentState *agents, int idx, float x, float y, float z) { agents[idx].pos[0] = x; agents[idx].pos[1] = y; agents[idx].pos[2] = z; }
__global__ void set_agent_yaw_kernel(AgentState *agents, int idx, float c, float s) { agents[idx].m00 = c; agents[idx].m02 = s; agents[idx].m20 = -s; agents[idx].m22 = c; }
__global__ void set_agent_velocity_kernel(AgentState *agents, int idx, float hvx, float hvz, float vvel) { agents[idx].hvx = hvx; agents[idx].hvz = hvz; agents[idx].vvel = vvel; }

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

// ------------------------------------------------------------------------------------------------
static std::string lower(const char *s)
{
    std::string r(s ? s : "");
    for (auto &c : r) c = (char)std::tolower((unsigned char)c);
    return r;
}

static int check(mv_gym *g)
{
    if (!g) return fail("null gym handle");
    if (g->closed) return fail("gym is closed");
    return 0;
}

// Where a tick's public outputs go: the observation slab and the reward / done arrays -- or, with mv_set_output_ring, entry (tick % count)
// of the caller's rings.  true_objective is state (only a finishing env records it, vector_env.cpp:96-101): never ringed.
struct OutPtrs { uint32_t *obs; float *rewards; uint8_t *done; };
static OutPtrs outputs_of(const mv_gym *g, unsigned long long tick)
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
static OutPtrs last_outputs(const mv_gym *g) { return outputs_of(g, g->ringTick ? g->ringTick - 1 : 0); }   // of the last tick (reset / render / getters)

static int refresh_mirrors(mv_gym *g)
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
static int take_hist(mv_gym *g, hipStream_t s, bool setupClearsNext)
{
    g->hist3 = (g->hist3 + 1) % g->hists;
    if (!g->histClean[(size_t)g->hist3])
        HIP_TRY(hipMemsetAsync(g->gv.lpt_hist + (size_t)g->hist3 * LPT_BUCKETS * LPT_SUBS, 0, (size_t)LPT_BUCKETS * LPT_SUBS * sizeof(int32_t), s));
    g->histClean[(size_t)g->hist3] = 0;
    if (setupClearsNext) g->histClean[(size_t)((g->hist3 + 1) % g->hists)] = 1;
    return 0;
}

// the view a kernel launch gets: the buffers of slot q, this pass's cost histogram, the action-sampling request
static GymView view(const mv_gym *g, int q, const OutPtrs *direct = nullptr)   // direct: the step writes the public output arrays itself (not pipelined)
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
static int sim_join(mv_gym *g)
{
    if (g->simDoneValid && g->simOnOwnStream) HIP_TRY(hipStreamWaitEvent(g->stream, g->simDone, 0));
    g->simMustWaitUser = true;
    return 0;
}

// TowerBuilding: before anything on the caller's stream touches the generators or the ring of drawn episodes (mv_reset, mv_seed): the last draw launch
static int tower_join(mv_gym *g)
{
    if (g->genStream && g->drawCount > 0) HIP_TRY(hipStreamWaitEvent(g->stream, g->drawDone[(size_t)((g->drawCount - 1) & 1ull)], 0));
    return 0;
}

// TowerBuilding, a stepping call: before its step launches the simulation stream waits for the draw launch BEFORE the last one (two episodes are resident per
// env; what the last launch may still be drawing replaces an episode consumed a call ago: not needed yet) ...
static int tower_draw_before(mv_gym *g, hipStream_t sim)
{
    if (g->genStream && g->drawCount >= 2) HIP_TRY(hipStreamWaitEvent(sim, g->drawDone[(size_t)(g->drawCount & 1ull)], 0));
    return 0;
}
// ... and behind them, every drawPeriod ticks, the draw kernel goes to its own stream: it tops up the rings of the envs that finished
static int tower_draw_after(mv_gym *g, hipStream_t sim, int ticks)
{
    if (!g->genStream) return 0;
    g->ticksSinceDraw += ticks;
    if (g->ticksSinceDraw < g->drawPeriod) return 0;
    g->ticksSinceDraw = 0;
    HIP_TRY(hipEventRecord(g->stepForDraw, sim));
    HIP_TRY(hipStreamWaitEvent(g->genStream, g->stepForDraw, 0));
    launch_tower_draw(g->gv, g->genStream);
    HIP_TRY(hipEventRecord(g->drawDone[(size_t)(g->drawCount & 1ull)], g->genStream));
    ++g->drawCount;
    return 0;
}

static int publish_outputs(mv_gym *g, int q, const OutPtrs &o)   // on the caller's stream
{
    const GymView &v = g->gvp[q];
    const int n = g->N * g->A;
    hipLaunchKernelGGL(publish_kernel, dim3((n + 255) / 256), dim3(256), 0, g->stream, v.rewards, v.done, v.true_objective, o.rewards, o.done,
                       g->gv.true_objective, g->N, g->A);
    HIP_TRY(hipGetLastError());
    return 0;
}

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
static bool scenario_from_name(const std::string &scen, int &scenario, ObstacleConfig &oc)
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
static hipError_t create_sim_stream(hipStream_t *s) { return hipStreamCreateWithFlags(s, hipStreamNonBlocking); }

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
        return fail("Unknown scenario " + scen + " (this build accelerates: TowerBuilding, ObstaclesEasy/Medium/Hard/Walls/Steps/Lava, Collect, Rearrange, Sokoban, HexMemory, HexExplore, Empty)");
    std::vector<std::string> levelFiles;
    if (scenario == SCN_SOKOBAN) {   // SokobanScenario's constructor looks the level files up (scenario_sokoban.cpp:40-78); none is fatal there too
        levelFiles = find_boxoban_level_files();
        if (levelFiles.empty()) return fail("Sokoban: no Boxoban level files (set BOXOBAN_LEVELS to the directory that holds unfiltered/train/000.txt ...)");
    }
    if (cfg->num_envs < 1 || cfg->num_agents_per_env < 1 || cfg->num_agents_per_env > MAX_AGENTS)
        return fail("mv_create: num_envs >= 1 and 1 <= num_agents_per_env <= 8 required");
    if (cfg->obs_width < 1 || cfg->obs_height < 1 || cfg->obs_width > 1024 || cfg->obs_height > 1024) return fail("mv_create: observation size must be within 1..1024");

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
    g->envOffset = cfg->total_envs > 0 ? cfg->env_offset : 0;
    g->envStride = cfg->total_envs > 0 && cfg->env_stride > 1 ? cfg->env_stride : 1;
    g->gv.sample_on = 0; g->gv.sample_seed = g->gv.sample_step = 0;
    g->gv.env_offset = g->envOffset; g->gv.env_stride = g->envStride;
    g->totalEnvs = cfg->total_envs > 0 ? cfg->total_envs : cfg->num_envs;
    if (const char *e = getenv("MV_PIPE_BATCH")) g->batch = std::min((int)PIPE_BATCH_MAX, std::max(1, atoi(e)));   // ticks per call of mv_step_n
    g->slots = PIPE_GROUPS * g->batch;
    g->hists = g->slots + 1;
    g->histClean.assign((size_t)g->hists, 1);   // (the arena is zeroed below)
    g->gvp.resize((size_t)g->slots);
    g->parity = g->slots - 1;
    g->group = PIPE_GROUPS - 1;
    const size_t N = g->N, NA = (size_t)g->N * g->A;

    GymView &gv = g->gv;
    gv.num_envs = g->N; gv.num_agents = g->A;
    const bool obstacles = scenario == SCN_OBSTACLES || scenario == SCN_EMPTY, collect = scenario == SCN_COLLECT, rearrange = scenario == SCN_REARRANGE, sokoban = scenario == SCN_SOKOBAN;
    const bool hex = scenario == SCN_HEX_MEMORY || scenario == SCN_HEX_EXPLORE;
    const bool hostEpisodes = obstacles || collect || rearrange || sokoban || hex;
    gv.scenario = scenario;
    gv.box_stride = collect ? COLLECT_MAX_BOXES : MAX_BOXES;
    gv.reward_stride = collect ? COLLECT_MAX_REWARDS : MAX_REWARDS;
    g->blobBytes = collect ? sizeof(CollectBlob) : obstacles ? sizeof(EpisodeBlob) : rearrange ? sizeof(RearrangeBlob) : sokoban ? sizeof(SokobanBlob) : hex ? sizeof(HexBlob) : sizeof(TowerBlob);
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
                 szHeight = collect ? up(N * (size_t)HM_BYTES) : 0, szItems = rearrange ? up(N * MAX_ITEMS * sizeof(ArrangementItem)) : 0, szCells = sokoban ? up(N * (size_t)(SOKO_DIM * SOKO_DIM)) : 0,
                 szHexB = hex ? up(N * (size_t)HEX_MAX_BOXES * sizeof(HexRec)) : 0, szHexO = hex ? up(N * (size_t)HEX_MAX_OBJS * sizeof(HexRec)) : 0, szBlobs = up(N * g->blobBytes * (size_t)g->spares), szCnt = up((N + 2) * sizeof(int32_t)), szGen = hostEpisodes ? 0 : up(N * sizeof(TowerGen));
    gv.vis_stride = hex ? 2048 : collect ? 1024 : 256;
    if (const char *e = getenv("MV_DEBUG_VIS_STRIDE")) gv.vis_stride = std::min(gv.vis_stride, std::max(8, atoi(e)));   // (tests: provoke ST_VISIBLE)
    gv.debug_redo = getenv("MV_DEBUG_FORCE_REDO") && atoi(getenv("MV_DEBUG_FORCE_REDO")) ? 1 : 0;   // (tests: mv_tick_tower.h's sequential redo)
    gv.spares = g->spares;
    const size_t szVisP = up(NA * (size_t)gv.vis_stride * 32), szVisR = up(NA * (size_t)gv.vis_stride * 8), szVisC = up(NA * sizeof(int32_t)),
                 szLpt = up(NA * sizeof(int32_t)) + up((NA + 1) * sizeof(int32_t)) + up(NA * (size_t)FRAME_HDR_BYTES) + up((size_t)LPT_BUCKETS * LPT_SUBS * lpt_sub_capacity(NA) * sizeof(int32_t));
    // per slot: frame lists, headers, cost lists, and the staging copies of rewards / dones / true objectives
    const size_t szParity = szVisP + szVisR + szVisC + szLpt + szRew + szDone + szObjv, szHist = up((size_t)g->hists * (LPT_BUCKETS * LPT_SUBS + 1) * sizeof(int32_t));   // (+ one "workgroups that have looked their frame up" counter per histogram, behind them)
    gv.lpt_hists = g->hists;
    // long lists: the list as found, before the frame setup deals it into depth classes (mv_frame.h: DepthSortScratch); MV_DEPTH_SORT=0: lists stay as found
    const bool depthSortOn = !(getenv("MV_DEPTH_SORT") && atoi(getenv("MV_DEPTH_SORT")) == 0);
    const size_t szSort = gv.vis_stride > 256 && depthSortOn ? up(NA * (size_t)gv.vis_stride * 40) : 0;
    const size_t total = szSort + szHdr + szBoxes + szObj + szAg + szAct + szRew + szDone + szObjv + szMd + (hostEpisodes ? 0 : szChunk) + szObs + szTerrain +
                         szRewObj + szHeight + szItems + szCells + szHexB + szHexO + szBlobs + szCnt + szGen + (size_t)g->slots * szParity + szHist;
    {
        hipError_t e_ = hipMalloc((void **)&g->arena, total);
        if (e_ != hipSuccess) { mv_destroy(g); return fail(std::string("hipMalloc arena: ") + hipGetErrorString(e_)); }
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
        if (hipMalloc((void **)&g->gv.dbg, N * 64 * sizeof(unsigned long long)) == hipSuccess) (void)hipMemset(g->gv.dbg, 0, N * 64 * sizeof(unsigned long long));
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
                  hipEventCreateWithFlags(&g->resetDone, hipEventDisableTiming) == hipSuccess &&
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
                  hipEventCreateWithFlags(&g->stepForDraw, hipEventDisableTiming) == hipSuccess &&
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
    }
    if (hostEpisodes) {
        g->uploaded.assign(N, 0);
        // Worker threads of the episode feeder: what the caller asks for (MegaverseGym's num_simulation_threads), or -- 0 / negative -- this process's
        // share of the host: the cores it may run on (affinity mask, cgroup quota) divided by the ranks of the job (total_envs / num_envs shards, one
        // process per GPU), at most 16.  What a thread sustains (episodes per second, measured: DESIGN.md 3.2): ObstaclesHard 15 k, ObstaclesEasy 44 k,
        // Collect 4.5 k, HexMemory 22 k, Rearrange 330 k; what one GPU consumes at its benchmark rate: ObstaclesHard ~7 k, Collect ~12 k.
        g->feederThreads = cfg->num_simulation_threads > 0 ? std::min(32, (int)cfg->num_simulation_threads)
                                                           : std::max(1, std::min(16, usable_host_cores() / std::max(1, g->totalEnvs / std::max(1, g->N))));
        if (const char *e = getenv("MV_FEEDER_THREADS")) g->feederThreads = std::min(64, std::max(1, atoi(e)));
        g->uploadEvents.assign(64, nullptr);
        bool ok = hipHostMalloc((void **)&g->hBlobs, N * g->blobBytes, hipHostMallocDefault) == hipSuccess;
        for (auto &e : g->uploadEvents) ok = ok && hipEventCreateWithFlags(&e, hipEventDisableTiming) == hipSuccess;
        if (!ok) {
            mv_destroy(g);
            return fail("mv_create: pinned episode staging allocation failed");
        }
        g->feeder = std::make_unique<EpisodeFeeder>(scenario, oc, g->N, g->A, episodeLen, g->hBlobs, g->blobBytes, g->device, g->feederThreads, levelFiles);
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
        for (int q = 0; q < g->slots; ++q) (void)hipMemcpy(g->gvp[q].lpt_order, iota.data(), NA * sizeof(int32_t), hipMemcpyHostToDevice);   // identity until the first frame sort
    }
    if (hipMemcpy(gv.hdr, hh.data(), N * sizeof(EnvHeader), hipMemcpyHostToDevice) != hipSuccess ||
        hipMemcpy(gv.agents, ha.data(), NA * sizeof(AgentState), hipMemcpyHostToDevice) != hipSuccess) {
        mv_destroy(g);
        return fail("mv_create: initial upload failed");
    }
    *out = g;
    return 0;
}

static void group_detach(mv_gym *g);

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
            static const char *names[8] = {"loads", "actions", "physics", "interact", "fall/zone/timers", "write-back", "tick (both waves, to the barrier)", "frame setup (both waves)"};
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
            std::fprintf(stderr, "[mv tick timing] casts, all ticks of all envs: %.0f sweeps, %.0f casts started, %.0f wave iterations slot by slot, %.0f if a lane's casts were queued\n", cs[0], cs[4], cs[1], cs[2]);
            const bool tickOnly = at(0, 49) != 0;
            if (tickOnly) {
                int worst = 0; double mxsum = 0.0, wi[5] = {0, 0, 0, 0, 0};
                for (int e = 0; e < N; ++e) {
                    mxsum += (double)at(e, 54);
                    if (at(e, 54) > at(worst, 54)) worst = e;
                    for (int k = 0; k < 5; ++k) wi[k] += (double)at(e, 32 + 8 + k);
                }
                std::fprintf(stderr, "[mv tick timing] longest tick of an env: %.2f us on average over envs, with on average %.1f sweeps, %.1f casts, %.1f wave iterations (%.1f queued), longest cast %.1f\n",
                             mxsum / N * 0.01, wi[0] / N, wi[4] / N, wi[1] / N, wi[2] / N, wi[3] / N);
                std::fprintf(stderr, "[mv tick timing] the longest of all (env %d, %.2f us), cycles per phase: loads %llu actions %llu physics %llu interact %llu fall %llu write-back %llu; %llu sweeps, %llu casts, %llu wave iterations (%llu queued), longest cast %llu\n",
                             worst, (double)at(worst, 54) * 0.01, at(worst, 32), at(worst, 33), at(worst, 34), at(worst, 35), at(worst, 36), at(worst, 37),
                             at(worst, 40), at(worst, 44), at(worst, 41), at(worst, 42), at(worst, 43));
                double rl = 0.0, rc = 0.0;
                for (int e = 0; e < N; ++e) { rl += (double)at(e, 52); rc += (double)at(e, 53); }
                std::fprintf(stderr, "[mv tick timing] tick-only launches: %.0f ticks regenerated their env and lived %.2f us on average\n", rc, rc > 0 ? rl / rc * 0.01 : 0.0);
            }
            {   // {sum of wave-0 lifetimes, launches, start and end of the last one}
                const int b = tickOnly ? 48 : 52;
                double life = 0.0, cnt = 0.0; unsigned long long s0 = ~0ull, s1 = 0, e1 = 0;
                for (int e = 0; e < N; ++e) {
                    life += (double)at(e, b); cnt += (double)at(e, b + 1);
                    if (at(e, b + 1)) { s0 = std::min(s0, at(e, b + 2)); s1 = std::max(s1, at(e, b + 2)); e1 = std::max(e1, at(e, b + 3)); }
                }
                if (cnt > 0)
                    std::fprintf(stderr, "[mv tick timing] %s launches: wave 0 lives %.2f us on average; last launch: starts spread over %.2f us, first start to last end %.2f us\n",
                                 tickOnly ? "tick-only" : "fused", life / cnt * 0.01, (double)(s1 - s0) * 0.01, (double)(e1 - s0) * 0.01);
            }
        }
        (void)hipFree(gv.dbg);
    }
    if (g->arena) (void)hipFree(g->arena);
    if (g->hiresObs) (void)hipFree(g->hiresObs);
    if (g->hBlobs) (void)hipHostFree(g->hBlobs);
    if (g->hStatus) (void)hipHostFree(g->hStatus);
    if (g->resetDone) (void)hipEventDestroy(g->resetDone);
    if (g->stepDone) (void)hipEventDestroy(g->stepDone);
    if (g->statusCopied) (void)hipEventDestroy(g->statusCopied);
    for (hipEvent_t e : g->uploadEvents) if (e) (void)hipEventDestroy(e);
    g->uploadEvents.clear();
    for (int i = 0; i < 2; ++i) {
        if (g->passStream[i]) (void)hipStreamDestroy(g->passStream[i]);
        if (g->callStart[i]) (void)hipEventDestroy(g->callStart[i]);
        g->passStream[i] = nullptr; g->callStart[i] = nullptr;
    }
    if (g->stepForDraw) (void)hipEventDestroy(g->stepForDraw);
    for (hipEvent_t &e : g->drawDone) { if (e) (void)hipEventDestroy(e); e = nullptr; }
    g->genStream = nullptr; g->stepForDraw = nullptr;
    if (g->copyStream) (void)hipStreamDestroy(g->copyStream);
    if (g->simStream) (void)hipStreamDestroy(g->simStream);
    for (hipEvent_t &e : g->userMark) { if (e) (void)hipEventDestroy(e); e = nullptr; }
    if (g->userNow) (void)hipEventDestroy(g->userNow);
    g->userNow = nullptr;
    if (g->simDone) (void)hipEventDestroy(g->simDone);
    g->simStream = nullptr; g->simDone = nullptr;
    g->hBlobs = nullptr; g->hStatus = nullptr; g->resetDone = g->stepDone = g->statusCopied = nullptr; g->copyStream = nullptr; g->dBlobs = nullptr; g->dStatus = nullptr;
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
        HIP_TRY(hipMemcpy(g->hStatus, g->dStatus, (size_t)(g->N + 2) * sizeof(int), hipMemcpyDeviceToHost));   // the current counts, not the last periodic read-back
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
    if (launch_raster(view(g, g->parity), last_outputs(g).obs, g->w, g->h, g->stream, nullptr, g->fastPixels)) return fail("mv_render: observation size above 1024x1024");
    HIP_TRY(hipGetLastError());
    return 0;
}

// ---- status flags ----------------------------------------------------------------------------------------------------
// Kernels raise ST_* bits in status[N + 1], the host generators GEN_* bits (mv_gen.h); both are limits the reference does not
// have.  They are reported ONCE, by the mv_step / mv_reset that sees them, and cleared: the gym stays usable.  They are WARNINGS: the call
// that reports one does all of its work and returns 1 instead of 0 (mv_last_error() has the text); -1 stays what it was, a real failure.
static int check_status_flags(mv_gym *g)
{
    const int N = g->N;
    const int flags = g->hStatus[N + 1], gen = g->feeder ? g->feeder->take_overflow() : 0;
    if (!flags && !gen) return 0;
    std::string msg;
    if (flags & ST_STARVED) msg += "an env finished again before its next episode was resident (it repeated its done step; the next episodes are being uploaded now); ";
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
static int finish_with_warning(mv_gym *g)
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
static int refill_episodes(mv_gym *g)
{
    if (g->statusPending) {
        // The read-back was enqueued behind a step kernel the host is normally several ticks ahead of: waiting for it here would drain
        // that run-ahead every statusPeriod steps.  With long episodes (period 16, two resident episodes per env) the words may arrive a few
        // steps later: look again at the next step, but never let them age beyond 32 steps.
        if (g->statusPeriod > 1 && g->pendingAge < 32 && hipEventQuery(g->statusCopied) == hipErrorNotReady) {
            (void)hipGetLastError();   // ("not ready" is an answer, not an error to report at the end of the step)
            ++g->pendingAge;
            return 0;
        }
        (void)hipGetLastError();
        HIP_TRY(hipEventSynchronize(g->statusCopied));
        g->statusPending = false;
        g->pendingAge = 0;
    }
    const int N = g->N, K = g->spares;
    const bool starved = (g->hStatus[N + 1] & ST_STARVED) != 0;
    if (starved && g->hostEpisodes()) {   // recover: take the current counts and upload synchronously below
        HIP_TRY(hipStreamSynchronize(g->simStream));
        if (!g->simOnOwnStream) HIP_TRY(hipStreamSynchronize(g->stream));   // (closed-loop / unpipelined steps run on the caller's stream, which may be a non-blocking one)
        HIP_TRY(hipStreamSynchronize(g->copyStream));
        const int keep = g->hStatus[N + 1];
        HIP_TRY(hipMemcpy(g->hStatus, g->dStatus, (size_t)(N + 2) * sizeof(int), hipMemcpyDeviceToHost));
        g->hStatus[N + 1] |= keep;
        g->refillForce = true;
    }
    if (g->hostEpisodes() && (g->refillForce || g->deficit > 0 || g->hStatus[N] != g->lastTotalSeen)) {
        hipEvent_t ev = g->uploadEvents[g->uploadRing++ % g->uploadEvents.size()];
        HIP_TRY(hipEventSynchronize(ev));   // 64 batches ago
        std::vector<int> &batch = g->uploadBatch;
        batch.clear();
        int deficit = 0;
        bool waited = false;
        for (int i = 0; i < N; ++i) {
            const int consumed = g->hStatus[i];
            if (g->uploaded[i] >= consumed + K) continue;            // ring full
            const int need = g->uploaded[i] + 1;
            const bool must = g->uploaded[i] == consumed;            // nothing resident: the next reset would starve
            if (!must && !g->feeder->is_ready(i, need)) { deficit += consumed + K - g->uploaded[i]; continue; }   // later
            size_t bytes = 0;
            const uint8_t *src = g->feeder->wait_ready(i, need, &bytes);
            if (!src) return fail(g->feeder->failed() ? std::string("episode feeder: a level file could not be read (Sokoban)")
                                                      : "episode feeder: episode " + std::to_string(need) + " of env " + std::to_string(i) + " was never generated");
            if (!waited && g->stepDoneValid) { HIP_TRY(hipStreamWaitEvent(g->copyStream, g->stepDone, 0)); waited = true; }
            HIP_TRY(hipMemcpyAsync(g->dBlobs + ((size_t)i * K + (size_t)((need - 1) % K)) * g->blobBytes, src, bytes, hipMemcpyHostToDevice, g->copyStream));
            ++g->uploaded[i];
            deficit += consumed + K - g->uploaded[i];
            batch.push_back(i);
        }
        if (!batch.empty()) {
            HIP_TRY(hipEventRecord(ev, g->copyStream));
            for (int i : batch) g->feeder->recycle(i, ev);   // regenerate a slot only once its upload has left it
            HIP_TRY(hipStreamWaitEvent(g->simStream, ev, 0));   // (a step that runs on the caller's stream, and mv_reset, wait for lastUpload themselves)
            g->lastUpload = ev;
            g->uploadNotOnUser = true;
        }
        g->deficit = deficit;
        g->lastTotalSeen = g->hStatus[N];
        g->refillForce = false;
    }
    return check_status_flags(g) < 0 ? -1 : 0;
}

// after a step / reset kernel: read the status words back without touching the step path
static int read_back_status(mv_gym *g, hipStream_t after)
{
    HIP_TRY(hipEventRecord(g->resetDone, after));
    HIP_TRY(hipStreamWaitEvent(g->copyStream, g->resetDone, 0));
    HIP_TRY(hipMemcpyAsync(g->hStatus, g->dStatus, (size_t)(g->N + 2) * sizeof(int), hipMemcpyDeviceToHost, g->copyStream));
    HIP_TRY(hipEventRecord(g->statusCopied, g->copyStream));
    g->statusPending = true;
    return 0;
}

// mv_set_actions_device keeps the caller's multi-discrete buffer until the next step kernel reads it.  Whatever else touches the actions or
// comes between the two -- a reset, a host-side setter -- first turns the pending buffer into bitmasks (on the caller's stream, where the
// buffer's producer ran), so that the buffer is read NOW, while it is certainly alive, and the last writer wins as it did when
// mv_set_actions_device converted at once.
static int flush_device_actions(mv_gym *g)
{
    if (!g->mdActions) return 0;
    const int n = g->N * g->A;
    hipLaunchKernelGGL(masks_from_multidiscrete_kernel, dim3((n + 255) / 256), dim3(256), 0, g->stream, g->mdActions, g->gv.actions, n);
    HIP_TRY(hipGetLastError());
    g->mdActions = nullptr;
    g->simMustWaitUser = true;
    return 0;
}

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
        if (refill_episodes(g) < 0) return -1;          // every env has an unconsumed episode resident
        if (g->lastUpload) HIP_TRY(hipStreamWaitEvent(g->stream, g->lastUpload, 0));
        const OutPtrs outs = last_outputs(g);
        const GymView v = view(g, g->parity, &outs);
        if (g->scenario == SCN_OBSTACLES || g->scenario == SCN_EMPTY) launch_reset_obstacles(v, (const EpisodeBlob *)g->dBlobs, g->dStatus, 1, g->stream);
        else if (g->scenario == SCN_REARRANGE) launch_reset_rearrange(v, (const RearrangeBlob *)g->dBlobs, g->dStatus, 1, g->stream);
        else if (g->scenario == SCN_SOKOBAN) launch_reset_sokoban(v, (const SokobanBlob *)g->dBlobs, g->dStatus, 1, g->stream);
        else if (g->scenario == SCN_HEX_MEMORY || g->scenario == SCN_HEX_EXPLORE) launch_reset_hex(v, (const HexBlob *)g->dBlobs, g->dStatus, 1, g->stream);
        else launch_reset_collect(v, (const CollectBlob *)g->dBlobs, g->dStatus, 1, g->stream);
        HIP_TRY(hipEventRecord(g->stepDone, g->stream));   // (the reset kernel reads the ring too)
        g->stepDoneValid = true;
        if (read_back_status(g, g->stream)) return -1;  // the second resident episodes go up with the next steps
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
    if (policy != MV_POLICY_MULTIDISCRETE && policy != MV_POLICY_SINGLE_BIT) return fail("mv_set_sample_policy: MV_POLICY_MULTIDISCRETE (1) or MV_POLICY_SINGLE_BIT (2)");
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

// -> whether `done` rides on the launch (TowerBuilding's launcher); otherwise the caller records it
static bool launch_step_of(const mv_gym *g, const GymView &v, hipStream_t sim, int fused, hipEvent_t done = nullptr)
{
    if (g->scenario == SCN_OBSTACLES || g->scenario == SCN_EMPTY) launch_step_obstacles(v, sim, g->w, g->h, fused);
    else if (g->scenario == SCN_COLLECT) launch_step_collect(v, sim, g->w, g->h, fused);
    else if (g->scenario == SCN_REARRANGE) launch_step_rearrange(v, sim, g->w, g->h, fused);
    else if (g->scenario == SCN_SOKOBAN) launch_step_sokoban(v, sim, g->w, g->h, fused);
    else if (g->scenario == SCN_HEX_MEMORY || g->scenario == SCN_HEX_EXPLORE) launch_step_hex(v, sim, g->w, g->h, fused);
    else { launch_step(v, sim, g->w, g->h, fused, done); return done != nullptr; }
    return false;
}

// One stepping call = k ticks (mv_step: 1; mv_step_n: up to `batch`) of n gyms that share one pair of streams (n = 1: a gym on its own;
// n > 1: an mv_group, stepped by union launches).  gs[0] is the leader: the stream state that changes with every call -- marks, which stream
// the last step ran on -- is kept on it and mirrored to the others.  policy != POLICY_NONE: tick j draws its actions inside the step kernel
// from (seed, first_index + j); POLICY_NONE: the first tick acts on what mv_set_actions* left, the following ones on cleared actions
// (env.cpp:141-142 clears them after every tick).
// kCall: the ticks of the CALLER's call this chunk belongs to (mv_step_n splits a call of more than `batch` ticks): what the ring contract of
// the overlapped passes is stated in (include/megaverse_hip.h).
static int step_gyms(mv_gym *const *gs, int n, bool render, int k, int policy, uint32_t seed, uint32_t first_index, int kCall)
{
    mv_gym *const L = gs[0];
    int batch = L->batch;
    bool mustWait = false, allFast = true, anyHostEpisodes = false;
    for (int i = 0; i < n; ++i) {
        mv_gym *g = gs[i];
        if (check(g)) return -1;
        if (!g->wasReset) return fail("mv_step: call mv_reset first");
        batch = std::min(batch, g->batch);
        mustWait = mustWait || g->simMustWaitUser;
        allFast = allFast && g->fastPixels != 0;
        anyHostEpisodes = anyHostEpisodes || g->hostEpisodes();
    }
    if (k < 1 || k > batch) return fail("mv_step_n: 1 <= k <= " + std::to_string(batch) + " (MV_PIPE_BATCH) required");
    HIP_TRY(hipSetDevice(L->device));
    for (int i = 0; i < n; ++i)
        if (refill_episodes(gs[i]) < 0) return -1;
    // ---- what this call must wait for on the caller's stream.  Always: whatever was there when the call PIPE_GROUPS - 1 calls ago began --
    // the observation passes and the consumers of the call that used this slot group last.  Everything, when the caller's stream
    // feeds the simulation (reset / render / device actions / test hooks since the last step).  (Both raster kernels read nothing but the
    // frame lists and headers of their tick: the simulator state may move on underneath them.)
    // Not pipelined (mv_set_pipelining(0)), or ONE tick whose inputs come from the caller's stream (a policy in the loop: nothing can overlap,
    // the two queue hand-overs, ~10 us each, would be pure cost): the step runs on the caller's stream like everything else.
    const bool own = L->pipelined != 0 && !(mustWait && k == 1);
    hipStream_t sim = own ? L->simStream : L->stream;
    if (own) {
        // The simulation stream may reuse a slot group once the observation passes that read it are done: the END of the call PIPE_GROUPS calls
        // ago (userMark, completed by that call's last pass, see below).  When the caller's stream feeds the simulation (reset / render / device
        // actions / test hooks since the last step), or the last step ran there: everything enqueued on it so far.
        if (mustWait || !L->simOnOwnStream) {
            HIP_TRY(hipEventRecord(L->userNow, L->stream));
            HIP_TRY(hipStreamWaitEvent(sim, L->userNow, 0));
        } else if (L->markCount >= PIPE_GROUPS) HIP_TRY(hipStreamWaitEvent(sim, L->userMark[L->markCount % PIPE_GROUPS], 0));
    } else {
        if (L->simOnOwnStream && L->simDoneValid) HIP_TRY(hipStreamWaitEvent(sim, L->simDone, 0));   // the last step ran on the other stream
        for (int i = 0; i < n; ++i) {   // (episode uploads make the simulation stream wait)
            if (gs[i]->uploadNotOnUser && gs[i]->lastUpload) HIP_TRY(hipStreamWaitEvent(sim, gs[i]->lastUpload, 0));
            gs[i]->uploadNotOnUser = false;
        }
        L->markCount = 0;
    }
    for (int i = 0; i < n; ++i) {
        mv_gym *g = gs[i];
        if (tower_draw_before(g, sim)) return -1;
        g->simMustWaitUser = false;
        g->simOnOwnStream = own;
        g->markCount = L->markCount;
        if (g->actionsDirty) {
            const int s = g->stage;
            HIP_TRY(hipMemcpyAsync(g->gv.actions, g->hActions[s], (size_t)g->N * g->A * sizeof(int32_t), hipMemcpyHostToDevice, sim));
            HIP_TRY(hipEventRecord(g->actionsCopied[s], sim));
            g->stage = 1 - s;
            HIP_TRY(hipEventSynchronize(g->actionsCopied[g->stage]));   // long done: recorded one step ago
            std::memset(g->hActions[g->stage], 0, (size_t)g->N * g->A * sizeof(int32_t));   // actions are cleared every tick (env.cpp:141-142)
            g->actionsDirty = false;
        }
        g->group = (g->group + 1) % PIPE_GROUPS;
    }
    const int fused = render ? 1 : 0;   // the step kernel also builds the frame lists when an observation pass follows
    if (L->gv.dbg) {   // (instrumented builds: MV_TICK_TIMING_SKIP=n starts the statistics after n stepping calls -- the steady state, not the first ticks of fresh episodes)
        static const long skip = getenv("MV_TICK_TIMING_SKIP") ? atol(getenv("MV_TICK_TIMING_SKIP")) : 0;
        if (skip > 0 && ++L->dbgCalls == skip) HIP_TRY(hipMemsetAsync(L->gv.dbg, 0, (size_t)L->N * 64 * sizeof(unsigned long long), sim));
    }
    // A batched call hands over ONCE: all k step kernels, then all k observation passes.  (Handing over tick by tick when the caller's stream is
    // found idle -- the first call after a synchronisation -- was built and measured on 20-step runs: 15.0-15.6 M obs/s against 16.2 M without;
    // short runs use short calls instead, bench.py's --batch.)
    std::vector<GymView> views((size_t)n * k);
    std::vector<OutPtrs> outs((size_t)n * k);
    hipEvent_t *evs[PIPE_BATCH_MAX];
    // ---- the k step kernels, back to back on the simulation stream
    bool simDoneRodeAlong = false;   // (the last step kernel's dispatch packet completes simDone itself)
    // One TowerBuilding gym, several rendered ticks with device-drawn actions, nothing timed per tick: ONE step launch runs the k ticks of every
    // env (launch_step_ticks; MV_STEP_TICKS=0: k launches).  Its views are collected in the loop below.
    static const bool ticksOff = getenv("MV_STEP_TICKS") && atoi(getenv("MV_STEP_TICKS")) == 0;
    const bool obstFamily = L->scenario == SCN_OBSTACLES || L->scenario == SCN_EMPTY;
    // (every scenario with one agent per env; several agents: TowerBuilding only -- two waves per env, launch_step_ticks)
    const bool canMultiTick = !ticksOff && n == 1 && (L->A == 1 || L->scenario == SCN_TOWER) && k >= 2 && k <= MAX_STEP_TICKS && render && policy != POLICY_NONE && !L->gv.dbg;
    // (the one-launch observation passes only beside the one-launch step: k separate step kernels starve beside a pass that long -- 70-190 us each, r04l)
    const bool canBatchRaster = canMultiTick && render && allFast && n == 1 && k >= 2 && L->ringObs && L->ringCount >= k;
    // timing (mv_profile_begin): a batched call that takes both one-launch paths is timed as a whole -- one entry, events around the step
    // launch and around the raster launch, k ticks -- so that the figures are those of the launches the product runs; otherwise tick by tick
    const bool profiling = render && L->profCount < L->profMax;
    hipEvent_t *callEv = nullptr;
    if (profiling && canMultiTick && canBatchRaster && k <= MAX_UNION) {
        callEv = &L->profEvents[(size_t)L->profCount * 5];
        L->profTicks[(size_t)L->profCount] = k;
        ++L->profCount;
    }
    const bool multiTick = canMultiTick && (!profiling || callEv);
    // A group (n > 1), several rendered ticks with device-drawn actions, every member with an observation ring at least k deep and one agent per env: ONE
    // union step launch runs the k ticks of every env of every gym (step_union_ticks_kernel) and ONE launch draws their k x n observation passes
    // (raster_union_batch_kernel) -- two launches per call where the tick-by-tick path takes 2 k (BASELINE configs[4]: the scenarios of a multi-task batch).
    bool groupBatch = n > 1 && !ticksOff && own && render && allFast && k >= 2 && k <= MAX_STEP_TICKS && policy != POLICY_NONE && !profiling && raster_union_batch_applicable(k, n, L->w, L->h);
    for (int i = 0; i < n && groupBatch; ++i) groupBatch = gs[i]->A == 1 && gs[i]->ringObs && gs[i]->ringCount >= k && !gs[i]->gv.dbg;
    for (int j = 0; j < k; ++j) {
        const bool prof = !callEv && render && L->profCount < L->profMax;
        evs[j] = prof ? &L->profEvents[(size_t)L->profCount * 5] : nullptr;
        if (prof) { L->profTicks[(size_t)L->profCount] = 1; ++L->profCount; }
        UnionStepArgs ua;
        ua.n = n;
        int envs = 0;
        for (int i = 0; i < n; ++i) {
            mv_gym *g = gs[i];
            if (policy != POLICY_NONE) { g->gv.sample_on = policy; g->gv.sample_seed = seed; g->gv.sample_step = first_index + (uint32_t)j; }
            else { g->gv.sample_on = (j == 0 && g->samplePending) ? g->samplePolicy : (int)POLICY_NONE; }
            g->parity = g->group * g->batch + j;
            if (render && take_hist(g, sim, !(multiTick || groupBatch))) return -1;   // (this pass's frame setup fills the next cost histogram; one launch per tick: and clears the one after)
            OutPtrs &o = outs[(size_t)j * n + i];
            o = outputs_of(g, g->ringTick++);
            GymView &v = views[(size_t)j * n + i];
            v = view(g, g->parity, own ? nullptr : &o);
            if (j == 0 && g->gv.sample_on == POLICY_NONE) v.md_actions = g->mdActions;
            if (groupBatch) v.lpt_no_clear = 1;   // (the passes clear their histograms themselves: mv_raster.hip, hist_done)
            if (n > 1) { ua.first[i] = envs; ua.gv[i] = v; envs += g->N; }
        }
        if (evs[j]) HIP_TRY(hipEventRecord(evs[j][0], sim));
        bool simDoneRides = false;
        if (multiTick) {
            views[(size_t)j].lpt_no_clear = 1;
            if (j == k - 1) {
                // (the cost histograms of the call's passes are clean: take_hist.  In the steady state of batched calls nothing is cleared here at
                // all -- every pass of the one-launch observation kernel leaves its histogram zero -- where r06l's kernel traces showed two fill
                // kernels in front of every step launch, the second one waiting 30 us for a wave slot beside the observation passes: the chain
                // of step launches is what bounds a batched call's rate at 1024 envs, 344 + 39 us per call against 288 us of passes.)
                if (callEv) HIP_TRY(hipEventRecord(callEv[0], sim));
                if (obstFamily) launch_step_obstacles_ticks(views.data(), k, sim, L->w, L->h);
                else if (L->scenario == SCN_REARRANGE) launch_step_rearrange_ticks(views.data(), k, sim, L->w, L->h);
                else if (L->scenario == SCN_SOKOBAN) launch_step_sokoban_ticks(views.data(), k, sim, L->w, L->h);
                else if (L->scenario == SCN_COLLECT) launch_step_collect_ticks(views.data(), k, sim, L->w, L->h);
                else if (L->scenario == SCN_HEX_MEMORY || L->scenario == SCN_HEX_EXPLORE) launch_step_hex_ticks(views.data(), k, sim, L->w, L->h);
                else launch_step_ticks(views.data(), k, sim, L->w, L->h, own && !callEv ? L->simDone : nullptr);
                if (callEv) HIP_TRY(hipEventRecord(callEv[1], sim));
                simDoneRides = own && !callEv && L->scenario == SCN_TOWER;
            }
        } else if (n == 1) simDoneRides = launch_step_of(L, views[(size_t)j * n], sim, fused, own && j == k - 1 && !evs[j] ? L->simDone : nullptr);
        else if (groupBatch) {
            if (j == k - 1) {   // every tick's views are collected: one launch for the k ticks of all n gyms
                UnionTicksArgs ta;
                ta.n = n; ta.k = k;
                for (int i = 0; i < n; ++i) {
                    ta.first[i] = ua.first[i];
                    ta.gv[i] = views[(size_t)i];   // tick 0's
                    ta.slot_stride[i] = (int64_t)((const uint8_t *)views[(size_t)n + i].vis_prims - (const uint8_t *)views[(size_t)i].vis_prims);
                }
                for (int i = n; i <= MAX_UNION; ++i) ta.first[i] = envs;
                for (int i = n; i < MAX_UNION; ++i) { ta.gv[i] = views[0]; ta.slot_stride[i] = 0; }
                launch_step_union_ticks(ta, sim, L->w, L->h);
            }
        } else {
            for (int i = n; i <= MAX_UNION; ++i) ua.first[i] = envs;
            launch_step_union(ua, sim, L->w, L->h, fused);
        }
        if (evs[j]) HIP_TRY(hipEventRecord(evs[j][1], sim));
        simDoneRodeAlong = simDoneRodeAlong || simDoneRides;
    }
    if (own && !simDoneRodeAlong) HIP_TRY(hipEventRecord(L->simDone, sim));   // (not pipelined: stream order does it)
    for (int i = 0; i < n; ++i)
        if (tower_draw_after(gs[i], sim, k)) return -1;
    // (every step kernel regenerates / swaps the next episode into the envs it finishes)
    // An env needs a fresh resident episode only at its NEXT reset, normally hundreds of steps away, and two are resident: the status
    // words are read back -- and the refill considered -- every statusPeriod-th step (16; 1 when episodes can be a few ticks long).
    if (anyHostEpisodes) HIP_TRY(hipEventRecord(L->stepDone, sim));   // (the gyms of a group share the leader's event)
    for (int i = 0; i < n; ++i) {
        mv_gym *g = gs[i];
        g->samplePending = false;
        g->mdActions = nullptr;
        if (own) g->simDoneValid = true;
        if (anyHostEpisodes) g->stepDoneValid = true;
        g->stepsSinceStatus += k;
        if (g->stepsSinceStatus >= g->statusPeriod) {   // (TowerBuilding regenerates finished envs in the kernel: only the error flags matter)
            if (read_back_status(g, sim)) return -1;
            g->stepsSinceStatus = 0;
        }
        g->mirrorsFresh = false;
    }
    // ---- the caller's stream: per tick the step's outputs, then the observation pass
    if (L->passOverlap && L->callStart[0]) HIP_TRY(hipEventRecord(L->callStart[(int)(L->overlapCalls & 1ull)], L->stream));   // (before this call enqueues anything there)
    if (own) HIP_TRY(hipStreamWaitEvent(L->stream, L->simDone, 0));
    std::vector<PublishTo> pubs((size_t)n);
    std::vector<uint32_t *> obsPtrs((size_t)n);
    // One gym, several ticks, every tick's observations in a slab of its own (an output ring at least k deep), nothing timed per tick: the
    // observation passes of up to MAX_UNION ticks go out as ONE launch (launch_raster_batch: the next tick's expensive frames fill the tail of
    // the previous tick's pass).  The ticks are collected below and launched at the end of their chunk.
    bool batchRaster = canBatchRaster;
    for (int j = 0; j < k; ++j) batchRaster = batchRaster && !evs[j];
    // overlapped passes (mv_set_pass_overlap): this call's one launch goes to an internal stream
    // (an env must not finish in two consecutive calls: their passes may publish its true objective in either order -- episodes of at least
    // baseEpisodeLen seconds, 15 ticks each)
    // The ring is two CALLS deep, in the caller's ticks per call (a call of 16 ticks runs as two chunks of 8: the second-next chunk's passes would overwrite
    // what the consumer of the previous CALL -- enqueued after both of its chunks -- may still be reading), and rewards / dones have rings of
    // their own (two passes in flight would both publish the single arrays, in either order).
    const bool overlap = batchRaster && own && !callEv && L->passOverlap && L->passStream[0] && L->ringCount >= 2 * std::max(k, kCall) && L->ringRewards && L->ringDone &&
                         k <= MAX_UNION && L->baseEpisodeLen * 15.0f > float(2 * k + 2);
    hipStream_t passOn = L->stream;
    if (overlap) {
        const int me = (int)(L->overlapCalls & 1ull);
        passOn = L->passStream[me];
        HIP_TRY(hipStreamWaitEvent(passOn, L->simDone, 0));                                      // this call's ticks
        if (L->overlapCalls >= 1) HIP_TRY(hipStreamWaitEvent(passOn, L->callStart[1 - me], 0));   // what the caller had enqueued when the previous call began
        else { HIP_TRY(hipEventRecord(L->userNow, L->stream)); HIP_TRY(hipStreamWaitEvent(passOn, L->userNow, 0)); }   // (first overlapped call: everything so far)
    }
    std::vector<PublishTo> chunkPubs;
    std::vector<uint32_t *> chunkObs;
    int chunkFirst = 0;
    for (int j = 0; j < k; ++j) {
        if (evs[j]) HIP_TRY(hipEventRecord(evs[j][2], L->stream));
        for (int i = 0; i < n; ++i) {
            const OutPtrs &o = outs[(size_t)j * n + i];
            pubs[i] = PublishTo{o.rewards, o.done, gs[i]->gv.true_objective};
            obsPtrs[i] = o.obs;
            if (own && (!render || !allFast) && publish_outputs(gs[i], gs[i]->group * gs[i]->batch + j, o)) return -1;   // (the fast observation pass publishes with its first workgroups)
        }
        // the call's last pass completes this call's mark (what the simulation stream waits for before it reuses the slot group)
        hipEvent_t mark = own && j == k - 1 ? L->userMark[L->markCount % PIPE_GROUPS] : nullptr;
        if (render && batchRaster) {
            const bool pubInRaster = own;
            chunkPubs.push_back(pubs[0]);
            chunkObs.push_back(obsPtrs[0]);
            static const int chunkMax = getenv("MV_RASTER_BATCH") ? std::min((int)MAX_UNION, std::max(1, atoi(getenv("MV_RASTER_BATCH")))) : (int)MAX_UNION;   // (0: off, launch_raster_batch declines)
            if (j == k - 1 || (int)chunkObs.size() >= chunkMax) {
                const int cn = (int)chunkObs.size();
                if (callEv) { HIP_TRY(hipEventRecord(callEv[2], L->stream)); HIP_TRY(hipEventRecord(callEv[3], L->stream)); }
                int r = cn >= 2 ? launch_raster_batch(&views[(size_t)chunkFirst], chunkObs.data(), pubInRaster ? chunkPubs.data() : nullptr, cn, L->w, L->h, overlap && cn == k ? passOn : L->stream, mark) : 1;
                if (r == 0 && overlap && cn == k) HIP_TRY(hipStreamWaitEvent(L->stream, mark, 0));   // the caller's stream sees the call's outputs as always
                if (r < 0) return fail("mv_step: observation size above 1024x1024");
                if (r == 0)   // (every pass of the one-launch kernel leaves its cost histogram zero)
                    for (int q = 0; q < cn; ++q) L->histClean[(size_t)views[(size_t)chunkFirst + q].lpt_parity] = 1;
                if (r == 1)   // (not applicable to this gym -- long lists -- or a chunk of one tick: tick by tick)
                    for (int q = 0; q < cn; ++q)
                    {
                        if (launch_raster(views[(size_t)chunkFirst + q], chunkObs[q], L->w, L->h, L->stream, nullptr, 1, /*setup_done=*/1, pubInRaster ? &chunkPubs[q] : nullptr,
                                          q == cn - 1 ? mark : nullptr))
                            return fail("mv_step: observation size above 1024x1024");
                        if (views[(size_t)chunkFirst + q].lpt_no_clear) L->histClean[(size_t)views[(size_t)chunkFirst + q].lpt_parity] = 1;   // (self_clear, mv_raster.hip)
                    }
                if (callEv) HIP_TRY(hipEventRecord(callEv[4], L->stream));
                chunkFirst = j + 1;
                chunkPubs.clear(); chunkObs.clear();
            }
        } else if (render && groupBatch) {
            if (j == k - 1) {   // the k x n observation passes of the call with one launch
                std::vector<PublishTo> allPubs((size_t)n * k);
                std::vector<uint32_t *> allObs((size_t)n * k);
                for (size_t q = 0; q < (size_t)n * k; ++q) {
                    allPubs[q] = PublishTo{outs[q].rewards, outs[q].done, gs[q % (size_t)n]->gv.true_objective};
                    allObs[q] = outs[q].obs;
                }
                const int r = launch_raster_union_batch(views.data(), allObs.data(), allPubs.data(), k, n, L->w, L->h, L->stream, mark);
                if (r != 0) return fail(r == -2 ? "mv_group_step: the hand-over slots of a batched call are not one slot apart (internal)" : "mv_group_step: observation size above 1024x1024");
                for (size_t q = 0; q < (size_t)n * k; ++q) gs[q % (size_t)n]->histClean[(size_t)views[q].lpt_parity] = 1;   // (every pass leaves its cost histogram zero)
            }
        } else if (render) {
            const bool pubInRaster = own && allFast;
            if (n > 1 && allFast) {
                if (launch_raster_union(&views[(size_t)j * n], obsPtrs.data(), pubInRaster ? pubs.data() : nullptr, n, L->w, L->h, L->stream, evs[j] ? evs[j][3] : nullptr, mark))
                    return fail("mv_step: observation size above 1024x1024");
            } else {
                for (int i = 0; i < n; ++i) {
                    const GymView &v = views[(size_t)j * n + i];
                    if (launch_raster(v, obsPtrs[i], L->w, L->h, L->stream, evs[j] && i == 0 ? evs[j][3] : nullptr, gs[i]->fastPixels, /*setup_done=*/1,
                                      pubInRaster ? &pubs[i] : nullptr, i == n - 1 ? mark : nullptr))
                        return fail("mv_step: observation size above 1024x1024");
                    if (v.lpt_no_clear && gs[i]->fastPixels) gs[i]->histClean[(size_t)v.lpt_parity] = 1;   // (self_clear, mv_raster.hip)
                }
            }
        } else if (mark) HIP_TRY(hipEventRecord(mark, L->stream));
        if (evs[j]) HIP_TRY(hipEventRecord(evs[j][4], L->stream));
    }
    if (own) {
        ++L->markCount;
        for (int i = 0; i < n; ++i) gs[i]->markCount = L->markCount;
    }
    L->overlapCalls = overlap ? L->overlapCalls + 1 : 0;
    HIP_TRY(hipGetLastError());
    int rc = 0;
    std::string text;
    for (int i = 0; i < n; ++i)
        if (!gs[i]->warning.empty()) {
            text += (text.empty() ? "" : " | ") + (n > 1 ? "gym " + std::to_string(i) + ": " : std::string()) + gs[i]->warning;
            gs[i]->warning.clear();
            rc = 1;
        }
    if (rc) g_err = text;
    return rc;
}

static int step_impl(mv_gym *g, bool render, int k, int policy, uint32_t seed, uint32_t first_index, int kCall = 0)
{
    if (g && g->inGroup) return fail("this gym belongs to an mv_group: step the group (mv_group_step)");
    return step_gyms(&g, 1, render, k, policy, seed, first_index, kCall > 0 ? kCall : k);
}

int mv_step(mv_gym *g) { return step_impl(g, true, 1, POLICY_NONE, 0, 0); }
int mv_step_no_render(mv_gym *g) { return step_impl(g, false, 1, POLICY_NONE, 0, 0); }

int mv_step_n(mv_gym *g, int32_t k, int32_t policy, uint32_t seed, uint32_t first_step_index)
{
    if (check(g)) return -1;
    if (policy != MV_POLICY_NONE && policy != MV_POLICY_MULTIDISCRETE && policy != MV_POLICY_SINGLE_BIT) return fail("mv_step_n: unknown policy");
    if (k < 1) return fail("mv_step_n: k >= 1 required");
    int rc = 0;
    // Episodes that can end within a few ticks (statusPeriod 1: the refill protocol looks at the consumed counts after every tick) are
    // stepped one tick per call; otherwise `batch` ticks at a time.
    const int chunk = g->statusPeriod <= 1 ? 1 : g->batch;
    for (int done = 0; done < k; done += chunk) {
        const int n = std::min(chunk, k - done);
        const int r = step_impl(g, true, n, policy, seed, first_step_index + (uint32_t)done, k);
        if (r < 0) return -1;
        if (r > 0) { g->warning += (g->warning.empty() ? "" : " | ") + g_err; rc = 1; }   // (every chunk's warning text is kept)
    }
    if (rc) { g_err = g->warning; g->warning.clear(); }
    return rc;
}

// ---- groups: several gyms of one job stepped with union launches (mv_step_union.hip, mv_raster.hip: launch_raster_union)
static void group_detach(mv_gym *g)
{   // back to the gym's own simulation stream and events (they were kept aside while it was a member)
    if (!g->inGroup) return;
    mv_group *grp = g->inGroup;
    for (mv_gym *m : grp->gyms) {
        if (m != grp->gyms[0]) {
            m->simStream = m->ownSimStream; m->simDone = m->ownSimDone; m->stepDone = m->ownStepDone;
            for (int q = 0; q < PIPE_GROUPS; ++q) m->userMark[q] = m->ownUserMark[q];
        }
        m->inGroup = nullptr;
        m->simMustWaitUser = true; m->simOnOwnStream = false; m->simDoneValid = false; m->stepDoneValid = false; m->markCount = 0;
    }
    grp->gyms.clear();   // (the handle stays valid until mv_group_destroy; stepping it is an error from now on)
}

int mv_group_create(mv_gym *const *gyms, int32_t n, mv_group **out)
{
    if (!gyms || !out || n < 1 || n > MAX_UNION) return fail("mv_group_create: 1 <= n <= 8 gyms required");
    *out = nullptr;
    mv_gym *L = gyms[0];
    for (int i = 0; i < n; ++i) {
        mv_gym *g = gyms[i];
        if (check(g)) return -1;
        if (g->inGroup) return fail("mv_group_create: a gym already belongs to a group");
        for (int j = 0; j < i; ++j) if (gyms[j] == g) return fail("mv_group_create: the same gym twice");
        if (g->device != L->device || g->w != L->w || g->h != L->h || g->A != L->A || g->stream != L->stream || g->batch != L->batch || g->pipelined != L->pipelined)
            return fail("mv_group_create: the gyms of a group share device, observation size, agents per env, stream (mv_set_stream first), batch and pipelining");
    }
    HIP_TRY(hipSetDevice(L->device));
    for (int i = 0; i < n; ++i) {   // nothing in flight on the streams a member is about to leave
        HIP_TRY(hipStreamSynchronize(gyms[i]->simStream));
        HIP_TRY(hipStreamSynchronize(gyms[i]->stream));
    }
    mv_group *grp = new mv_group();
    grp->gyms.assign(gyms, gyms + n);
    for (int i = 0; i < n; ++i) {
        mv_gym *g = gyms[i];
        g->inGroup = grp;
        if (i > 0) {
            g->ownSimStream = g->simStream; g->ownSimDone = g->simDone; g->ownStepDone = g->stepDone;
            g->simStream = L->simStream; g->simDone = L->simDone; g->stepDone = L->stepDone;
            for (int q = 0; q < PIPE_GROUPS; ++q) { g->ownUserMark[q] = g->userMark[q]; g->userMark[q] = L->userMark[q]; }
        }
        g->simMustWaitUser = true; g->simOnOwnStream = false; g->simDoneValid = false; g->stepDoneValid = false; g->markCount = 0;
    }
    *out = grp;
    return 0;
}

int mv_group_destroy(mv_group *grp)
{
    if (!grp) return 0;
    if (!grp->gyms.empty()) {
        mv_gym *L = grp->gyms[0];
        (void)hipSetDevice(L->device);
        (void)hipStreamSynchronize(L->simStream);
        (void)hipStreamSynchronize(L->stream);
        group_detach(L);
    }
    delete grp;
    return 0;
}

int mv_group_step(mv_group *grp, int32_t k, int32_t render, int32_t policy, uint32_t seed, uint32_t first_step_index)
{
    if (!grp || grp->gyms.empty()) return fail("mv_group_step: the group is gone (a member was closed)");
    if (policy != MV_POLICY_NONE && policy != MV_POLICY_MULTIDISCRETE && policy != MV_POLICY_SINGLE_BIT) return fail("mv_group_step: unknown policy");
    if (k < 1) return fail("mv_group_step: k >= 1 required");
    int chunk = grp->gyms[0]->batch;
    for (mv_gym *g : grp->gyms)
        if (!g->closed && g->statusPeriod <= 1) chunk = 1;   // (episodes of a few ticks: the refill protocol looks at the consumed counts after every tick)
    int rc = 0;
    std::string text;
    for (int done = 0; done < k; done += chunk) {
        const int r = step_gyms(grp->gyms.data(), (int)grp->gyms.size(), render != 0, std::min(chunk, k - done), policy, seed, first_step_index + (uint32_t)done, k);
        if (r < 0) return -1;
        if (r > 0) { text += (text.empty() ? "" : " | ") + g_err; rc = 1; }
    }
    if (rc) g_err = text;
    return rc;
}

int mv_step_many(mv_gym *const *gyms, int32_t n, int32_t render, int32_t sample, uint32_t seed, uint32_t step_index)
{   // several gyms of one job (MultiTaskGym: one per scenario, one stream each) stepped by one call: at eight sub-gyms the per-call cost of
    // the language binding is a third of the step.  EVERY gym is stepped, whatever another one reports: a failure (or a warning) is
    // collected and returned after the loop, so the sub-gyms never get out of step with each other.
    if (!gyms || n < 0) return fail("mv_step_many: bad arguments");
    int rc = 0;
    std::string msgs;
    for (int i = 0; i < n; ++i) {
        int r = sample ? mv_sample_random_actions(gyms[i], seed, step_index) : 0;
        if (r == 0) r = step_impl(gyms[i], render != 0, 1, POLICY_NONE, 0, 0);
        if (r != 0) {
            msgs += (msgs.empty() ? "gym " : " | gym ") + std::to_string(i) + ": " + g_err;
            if (r < 0 || rc == 0) rc = r < 0 ? -1 : 1;
        }
    }
    if (rc) g_err = msgs;
    return rc;
}

int mv_synchronize(mv_gym *g)
{
    if (check(g)) return -1;
    HIP_TRY(hipStreamSynchronize(g->simStream));
    HIP_TRY(hipStreamSynchronize(g->stream));
    return 0;
}

int mv_profile_begin(mv_gym *g, int32_t max_steps)
{
    if (check(g)) return -1;
    if (max_steps < 0) return fail("mv_profile_begin: max_steps < 0");
    HIP_TRY(hipSetDevice(g->device));
    while ((int)g->profEvents.size() < max_steps * 5) {
        hipEvent_t e;
        HIP_TRY(hipEventCreate(&e));
        g->profEvents.push_back(e);
    }
    g->profTicks.assign((size_t)max_steps, 1);
    g->profMax = max_steps;
    g->profCount = 0;
    return 0;
}

int mv_profile_end(mv_gym *g, float *avg_ms4, int32_t *counts4)
{
    if (check(g)) return -1;
    HIP_TRY(hipStreamSynchronize(g->simStream));
    HIP_TRY(hipStreamSynchronize(g->stream));
    // every interval lies on ONE stream: [0] step kernel = events 0 -> 1 (the stream the step ran on); [2] publish / frame sort = 2 -> 3 and
    // [3] raster = 3 -> 4 (the caller's stream).  [1] (the old status read-back gap) is gone: it spanned two streams when pipelined.
    double sum[4] = {0, 0, 0, 0};
    static const int FROM[4] = {0, -1, 2, 3};
    for (int i = 0; i < g->profCount; ++i)
        for (int k = 0; k < 4; ++k) {
            if (FROM[k] < 0) continue;
            float ms = 0.0f;
            HIP_TRY(hipEventElapsedTime(&ms, g->profEvents[(size_t)i * 5 + FROM[k]], g->profEvents[(size_t)i * 5 + FROM[k] + 1]));
            sum[k] += ms;
        }
    int ticks = 0;   // (an entry of a batched call covers its k ticks: the averages are per tick)
    for (int i = 0; i < g->profCount; ++i) ticks += g->profTicks[(size_t)i];
    for (int k = 0; k < 4; ++k) {
        avg_ms4[k] = ticks ? (float)(sum[k] / ticks) : 0.0f;
        counts4[k] = ticks;
    }
    g->profMax = 0;
    g->profCount = 0;
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
    HIP_TRY(hipMemcpyAsync(out, (const uint8_t *)last_outputs(g).obs + ((size_t)env * g->A + agent) * frameBytes, frameBytes, hipMemcpyDeviceToHost, g->stream));
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
    if (launch_raster(view(g, g->parity), g->hiresObs, g->hiresW, g->hiresH, g->stream, nullptr, g->fastPixels)) return fail("mv_draw_hires: render size above 1024x1024");
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
    if (e == hipSuccess && g->gv.terrain) e = hipMemcpy(terr.data(), g->gv.terrain + (size_t)env * MAX_TERRAIN, MAX_TERRAIN * sizeof(TerrainBox), hipMemcpyDeviceToHost);
    if (e == hipSuccess && g->gv.rewards_obj) e = hipMemcpy(rew.data(), g->gv.rewards_obj + (size_t)env * g->gv.reward_stride, g->gv.reward_stride * sizeof(MovableObject), hipMemcpyDeviceToHost);
    if (e == hipSuccess && g->gv.items) {
        std::vector<ArrangementItem> its(MAX_ITEMS);
        e = hipMemcpy(its.data(), g->gv.items + (size_t)env * MAX_ITEMS, MAX_ITEMS * sizeof(ArrangementItem), hipMemcpyDeviceToHost);
        s->num_items = h.num_terrain;
        for (int i = 0; i < h.num_terrain && i < MAX_ITEMS; ++i) {
            s->items[i][0] = its[i].shape; s->items[i][1] = its[i].color;
            s->items[i][2] = its[i].off[0]; s->items[i][3] = its[i].off[1]; s->items[i][4] = its[i].off[2];
        }
    }
    if (e == hipSuccess && g->gv.soko_cells) e = hipMemcpy(s->soko, g->gv.soko_cells + (size_t)env * (SOKO_DIM * SOKO_DIM), SOKO_DIM * SOKO_DIM, hipMemcpyDeviceToHost);
    std::memset(s->heightmap, 0xff, sizeof s->heightmap);
    if (e == hipSuccess && g->gv.heightmap) e = hipMemcpy(s->heightmap, g->gv.heightmap + (size_t)env * HM_BYTES, sizeof s->heightmap, hipMemcpyDeviceToHost);
    if (e != hipSuccess) { delete s; return fail(std::string("mv_debug_snapshot: ") + hipGetErrorString(e)); }
    if (h.scenario == SCN_REARRANGE) h.num_terrain = 0;   // (the header reuses it for the item count, reported as num_items)
    if (e == hipSuccess && g->gv.hex_boxes) {   // Hex*: the header's box / collider / reward counts describe the hex lists
        s->hex_num_boxes = h.num_boxes; s->hex_num_objs = h.num_rewards;
        s->hex_target[0] = h.hex_target[0]; s->hex_target[1] = 0.0f; s->hex_target[2] = h.hex_target[1];
        e = hipMemcpy(s->hex_boxes, g->gv.hex_boxes + (size_t)env * HEX_MAX_BOXES, (size_t)std::min(h.num_boxes, (int)HEX_MAX_BOXES) * sizeof(HexRec), hipMemcpyDeviceToHost);
        if (e == hipSuccess) e = hipMemcpy(s->hex_objs, g->gv.hex_objs + (size_t)env * HEX_MAX_OBJS, (size_t)std::min(h.num_rewards, (int)HEX_MAX_OBJS) * sizeof(HexRec), hipMemcpyDeviceToHost);
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
    const size_t bytes = scenario == SCN_COLLECT ? sizeof(CollectBlob) : scenario == SCN_REARRANGE ? sizeof(RearrangeBlob) : hex ? sizeof(HexBlob) : sizeof(EpisodeBlob);
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
    const size_t bytes = scenario == SCN_COLLECT ? sizeof(CollectBlob) : scenario == SCN_REARRANGE ? sizeof(RearrangeBlob) : hex ? sizeof(HexBlob) : sizeof(EpisodeBlob);
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
