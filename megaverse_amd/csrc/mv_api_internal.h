// megaverse_amd/csrc/mv_api_internal.h -- what the host side of libmegaverse_hip.so shares between its translation units:
//   mv_api.hip        create / close, seeding, reset, actions, output rings, getters, reward shaping (MegaverseGym's methods but step)
//   mv_api_step.hip   stepping: pipelining, batched calls, overlapped passes, groups (union launches), in-stream profiling
//   mv_api_debug.hip  test hooks: snapshots, pose setters, host-side generators, RNG / arithmetic probes
// The C ABI itself is include/megaverse_hip.h; nothing here is exported under a C name.
#pragma once
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
// TowerBuilding: tops every env's ring of drawn episodes up (mv_reset.hip)
void launch_tower_draw(const GymView &gv, hipStream_t stream);
void launch_tower_seed(const GymView &gv, const uint32_t *seeds, hipStream_t stream);   // Env::seed for every env's generator
// Collect, device-drawn episodes (mv_collect_draw.hip): staging slots of envs[i] -> their ring slots slots[i], one launch per 64 episodes
void launch_collect_blob_copy(const int32_t *envs, const int32_t *slots, int count, const uint8_t *staging, uint8_t *ring, size_t blob_bytes, int spares,
                              hipStream_t stream);
// step kernels: one 256-thread workgroup per env = the tick (wave 0) + the frame setup of the env's frames for a W x H observation
// (render = 0: tick only)
void launch_step(const GymView &gv, hipStream_t stream, int W, int H, int render, hipEvent_t done = nullptr);
// k ticks + frame setups of every env, one launch (mv_step.hip)
void launch_step_ticks(const GymView *views, int k, hipStream_t stream, int W, int H, hipEvent_t done = nullptr);
void launch_step_obstacles(const GymView &gv, hipStream_t stream, int W, int H, int render);
// k ticks + frame setups of every env (one agent), one launch
void launch_step_obstacles_ticks(const GymView *views, int k, hipStream_t stream, int W, int H, hipEvent_t done = nullptr);
void launch_step_rearrange_ticks(const GymView *views, int k, hipStream_t stream, int W, int H, hipEvent_t done = nullptr);
void launch_step_sokoban_ticks(const GymView *views, int k, hipStream_t stream, int W, int H, hipEvent_t done = nullptr);
void launch_step_collect_ticks(const GymView *views, int k, hipStream_t stream, int W, int H, hipEvent_t done = nullptr);
void launch_step_hex_ticks(const GymView *views, int k, hipStream_t stream, int W, int H, hipEvent_t done = nullptr);
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


namespace mvapi {
extern thread_local std::string g_err;   // mv_last_error()
inline int fail(const std::string &msg)
{
    g_err = msg;
    return -1;
}
}  // namespace mvapi
using namespace mvapi;
#define HIP_TRY(expr)                                                                                  \
    do {                                                                                               \
        hipError_t e_ = (expr);                                                                        \
        if (e_ != hipSuccess) return fail(std::string(#expr) + ": " + hipGetErrorString(e_));          \
    } while (0)

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
    // slots per group (ticks per call: set by mv_create from the slots' footprint, or MV_PIPE_BATCH), slots, cost histograms
    int batch = 8, slots = PIPE_GROUPS * 8, hists = PIPE_GROUPS * 8 + 1;
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
    hipStream_t ownSimStream = nullptr, ownCopyStream = nullptr;
    hipEvent_t ownUserMark[PIPE_GROUPS] = {}, ownSimDone = nullptr, ownStepDone = nullptr;
    uint8_t *arena = nullptr;
    size_t arenaBytes = 0;
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
    const char *const *shapingKeys = nullptr;
    ObstacleConfig obst;
    float baseEpisodeLen = 60.0f;
    // Obstacles / Collect: background episode feeder + one resident episode per env (refill protocol below)
    std::unique_ptr<EpisodeFeeder> feeder;
    int feederThreads = 1;
    std::vector<int> consumedSeen;                  // the consumed counts of the last status read-back that was looked at (host copy)
    std::vector<int> uploaded, uploadBatch;         // episodes uploaded per env; envs of the current upload batch
    uint8_t *dBlobs = nullptr, *hBlobs = nullptr;   // device [N][blobBytes], pinned feeder slots [N][blobBytes]
    bool blobsOnDevice = false;                     // Collect with the device-side generator: hBlobs is device memory, filled by collect_draw_kernel (mv_feeder.cpp)
    size_t blobBytes = 0;                           // sizeof(EpisodeBlob) or sizeof(CollectBlob)
    bool hostEpisodes() const { return scenario != SCN_TOWER; }
    // TowerBuilding: the episode generator's serial half (tower_draw_kernel, ~47 us of one wavefront per finished env) runs on a stream of its own,
    // behind the step launch whose finished envs it refills and beside everything else; a stepping call waits for the draw launch BEFORE the last one
    // (two episodes are resident per env: what the last launch is still drawing is not needed yet).  drawPeriod: ticks between draw launches -- 8 where
    // episodes last at least 64 ticks, every call where they can be a few ticks long (those calls are one tick each: mv_step_n).
    hipStream_t genStream = nullptr;
    hipEvent_t drawDone[2] = {nullptr, nullptr};
    unsigned long long drawCount = 0, drawWaitedCount = 0;   // draw launches so far; the count at the last tower_draw_before wait ...
    hipStream_t drawWaitedOn = nullptr;                       // ... and the stream that waited
    int ticksSinceDraw = 0, drawPeriod = 1;
    int *dStatus = nullptr, *hStatus = nullptr;     // [N + 2]: consumed per env, total, error flags (device, pinned mirror)
    int lastTotalSeen = 0;
    bool statusPending = false, refillForce = true;
    int pendingAge = 0;                             // ticks enqueued since the pending read-back was issued
    int stepsSinceStatus = 0;
    int spares = 2;                                 // resident episodes per env (ring); the host keeps uploaded <= consumed + spares
    int statusPeriod = 16;                          // steps between status read-backs (1 when episodes can be only a few ticks long)
    int deficit = 0;                                // spares still to be uploaded (their episodes were not generated yet at the last look)
    hipEvent_t stepDone = nullptr;                  // after the last step kernel: uploads never overlap a kernel that may read the ring
    hipStream_t copyStream = nullptr;               // status read-back + episode uploads, off the step path
    hipEvent_t statusCopied = nullptr;
    // the event behind the last kernel that may read the ring (a step launch's simDone / stepDone, mv_reset's stepDone)
    hipEvent_t lastStep = nullptr;
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

namespace mvapi {
// Where a tick's public outputs go: the observation slab and the reward / done arrays -- or, with mv_set_output_ring, entry (tick % count)
// of the caller's rings.  true_objective is state (only a finishing env records it, vector_env.cpp:96-101): never ringed.
struct OutPtrs { uint32_t *obs; float *rewards; uint8_t *done; };
std::string lower(const char *s);
int check(mv_gym *g);
OutPtrs outputs_of(const mv_gym *g, unsigned long long tick);
OutPtrs last_outputs(const mv_gym *g);   // of the last tick (reset / render / getters)
int refresh_mirrors(mv_gym *g);
int take_hist(mv_gym *g, hipStream_t s, bool setupClearsNext);
GymView view(const mv_gym *g, int q, const OutPtrs *direct = nullptr);   // direct: the step writes the public output arrays itself (not pipelined)
int sim_join(mv_gym *g);
int tower_join(mv_gym *g);
int tower_draw_before(mv_gym *g, hipStream_t sim);
int tower_draw_after(mv_gym *g, hipEvent_t after, int ticks);
int publish_outputs(mv_gym *g, int q, const OutPtrs &o);   // on the caller's stream
bool scenario_from_name(const std::string &scen, int &scenario, ObstacleConfig &oc);
int check_status_flags(mv_gym *g);
int finish_with_warning(mv_gym *g);
int refill_episodes(mv_gym *g, int k);   // k: the ticks of the stepping call that is about to be enqueued
int read_back_status(mv_gym *g, hipEvent_t after);   // after: an event recorded behind the kernel whose status words are wanted
int flush_device_actions(mv_gym *g);
void group_detach(mv_gym *g);   // (mv_api_step.hip)
}  // namespace mvapi
