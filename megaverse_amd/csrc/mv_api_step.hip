// megaverse_amd/csrc/mv_api_step.hip -- stepping: MegaverseGym::step (megaverse.cpp:118-121 -> VectorEnv::step, vector_env.cpp:89-108) as kernel sequencing on
// two streams -- one-step-ahead pipelining, batched calls (mv_step_n: one step launch + one observation launch per call), overlapped passes, groups of gyms
// stepped with union launches (mv_group_*), several gyms per call (mv_step_many) -- and the in-stream kernel timing (mv_profile_*).  DESIGN.md 3.4, 3.6.
#include "mv_api_internal.h"

extern "C" {

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
        if (refill_episodes(gs[i], k) < 0) return -1;
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
    // (instrumented builds: MV_TICK_TIMING_SKIP=n starts the statistics after n stepping calls -- the steady state, not the first ticks of fresh episodes)
    if (L->gv.dbg) {
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
    const bool canMultiTick = !ticksOff && n == 1 && (L->A == 1 || L->scenario == SCN_TOWER) && k >= 2
                                                      && k <= MAX_STEP_TICKS && render && policy != POLICY_NONE && !L->gv.dbg;
    // (the one-launch observation passes only beside the one-launch step: k separate step kernels starve beside a pass that long -- 70-190 us each, r04l)
    const bool canBatchRaster = canMultiTick && render && allFast && n == 1 && k >= 2 && L->ringObs && L->ringCount >= k;
    // timing (mv_profile_begin): a batched call that takes both one-launch paths is timed as a whole -- one entry, events around the step
    // launch and around the raster launch, k ticks -- so that the figures are those of the launches the product runs; otherwise tick by tick
    const bool profiling = render && L->profCount < L->profMax;
    hipEvent_t *callEv = nullptr;
    if (profiling && canMultiTick && canBatchRaster && k <= MAX_STEP_TICKS) {
        callEv = &L->profEvents[(size_t)L->profCount * 5];
        L->profTicks[(size_t)L->profCount] = k;
        ++L->profCount;
    }
    const bool multiTick = canMultiTick && (!profiling || callEv);
    // A group (n > 1), several rendered ticks with device-drawn actions, every member with an observation ring at least k deep and one agent per env: ONE
    // union step launch runs the k ticks of every env of every gym (step_union_ticks_kernel) and ONE launch draws their k x n observation passes
    // (raster_union_batch_kernel) -- two launches per call where the tick-by-tick path takes 2 k (BASELINE configs[4]: the scenarios of a multi-task batch).
    bool groupBatch = n > 1 && !ticksOff && own && render && allFast && k >= 2 && k <= MAX_STEP_TICKS && policy != POLICY_NONE
            && !profiling && raster_union_batch_applicable(k, n, L->w, L->h);
    for (int i = 0; i < n && groupBatch; ++i) groupBatch = gs[i]->A == 1 && gs[i]->ringObs && gs[i]->ringCount >= k && !gs[i]->gv.dbg;
    {   // ... and all of the group's envs resident at once: the union step launch keeps a workgroup per env alive for the whole call (four waves of 168 VGPRs for the long-list
        // gyms); beyond one round of the chip the tick-by-tick launches win (Mixed 64 x 64, batched / tick by tick: 512 envs 12.2 / 8.6 M obs/s, 1024: 17.6 /
        // 16.2, 2048: 16.1 / 20.5; r08z_rules)
        int envs = 0;
        for (int i = 0; i < n; ++i) envs += gs[i]->N;
        groupBatch = groupBatch && envs <= 1024;
    }
    // not pipelined: the caller's stream needs no event behind the step launches; the side streams do when a draw launch, a status read-back or an upload will
    // wait for this call
    bool sideWaits = anyHostEpisodes;
    for (int i = 0; i < n; ++i)
        sideWaits = sideWaits || (gs[i]->genStream && gs[i]->ticksSinceDraw + k >= gs[i]->drawPeriod) || gs[i]->stepsSinceStatus + k >= gs[i]->statusPeriod;
    bool stepDoneRodeAlong = false;
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
            // (this pass's frame setup fills the next cost histogram; one launch per tick: and clears the one after)
            if (render && take_hist(g, sim, !(multiTick || groupBatch))) return -1;
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
                // (completed by the last launch's own dispatch packet: no marker behind it on the simulation stream)
                hipEvent_t rides = own && !callEv ? L->simDone : nullptr;
                const int chunkTicks = 8;   // (a launch holds the views of up to 8 ticks as its arguments, mv_types.h: StepTicksArgs8)
                for (int j0 = 0; j0 < k; j0 += chunkTicks) {
                    const int kk = std::min(chunkTicks, k - j0);
                    const GymView *vw = views.data() + j0;
                    hipEvent_t r = j0 + kk == k ? rides : nullptr;
                    if (obstFamily) launch_step_obstacles_ticks(vw, kk, sim, L->w, L->h, r);
                    else if (L->scenario == SCN_REARRANGE) launch_step_rearrange_ticks(vw, kk, sim, L->w, L->h, r);
                    else if (L->scenario == SCN_SOKOBAN) launch_step_sokoban_ticks(vw, kk, sim, L->w, L->h, r);
                    else if (L->scenario == SCN_COLLECT) launch_step_collect_ticks(vw, kk, sim, L->w, L->h, r);
                    else if (L->scenario == SCN_HEX_MEMORY || L->scenario == SCN_HEX_EXPLORE) launch_step_hex_ticks(vw, kk, sim, L->w, L->h, r);
                    else launch_step_ticks(vw, kk, sim, L->w, L->h, r);
                }
                if (callEv) HIP_TRY(hipEventRecord(callEv[1], sim));
                simDoneRides = rides != nullptr;
            }
        } else if (n == 1) {
            hipEvent_t rides = j == k - 1 && !evs[j] ? (own ? L->simDone : sideWaits ? L->stepDone : nullptr) : nullptr;
            const bool rode = launch_step_of(L, views[(size_t)j * n], sim, fused, rides);
            simDoneRides = rode && own;
            stepDoneRodeAlong = rode && !own;
        }
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
                launch_step_union_ticks(ta, sim, L->w, L->h, own ? L->simDone : nullptr);
                simDoneRides = own;
            }
        } else {
            for (int i = n; i <= MAX_UNION; ++i) ua.first[i] = envs;
            launch_step_union(ua, sim, L->w, L->h, fused);
        }
        if (evs[j]) HIP_TRY(hipEventRecord(evs[j][1], sim));
        simDoneRodeAlong = simDoneRodeAlong || simDoneRides;
    }
    // ONE event behind the call's step launches -- simDone (pipelined: often completed by the last launch's own dispatch packet), else stepDone -- is what
    // the caller's stream, the episode draws / uploads and the status read-backs wait for.  (There used to be up to 2 + n event records here -- simDone,
    // stepDone, TowerBuilding's stepForDraw, a resetDone per member whose status was due; every one is a marker packet the simulation queue works off before
    // it reaches the NEXT call's step launch, while that call's observation launch -- one wait on the caller's queue -- was already taking the chip: a step
    // launch that starts behind the observation launch it runs beside waits for that launch's workgroups to drain, r08q timeline.)
    hipEvent_t after = L->simDone;
    if (own) { if (!simDoneRodeAlong) HIP_TRY(hipEventRecord(L->simDone, sim)); }
    else {
        after = sideWaits ? L->stepDone : nullptr;
        if (sideWaits && !stepDoneRodeAlong) HIP_TRY(hipEventRecord(L->stepDone, sim));
    }
    for (int i = 0; i < n; ++i)
        if (tower_draw_after(gs[i], after, k)) return -1;
    // (every step kernel regenerates / swaps the next episode into the envs it finishes)
    // An env needs a fresh resident episode only at its NEXT reset, normally hundreds of steps away, and two are resident: the status
    // words are read back -- and the refill considered -- every statusPeriod-th step (16; 1 when episodes can be a few ticks long).
    for (int i = 0; i < n; ++i) {
        mv_gym *g = gs[i];
        g->samplePending = false;
        g->mdActions = nullptr;
        if (own) g->simDoneValid = true;
        if (after) g->lastStep = after;   // (uploads never overlap a kernel that may read the ring: refill_episodes; host-generated scenarios always have one)
        g->stepsSinceStatus += k;
        if (g->stepsSinceStatus >= g->statusPeriod) {   // (TowerBuilding: the error flags)
            if (read_back_status(g, after)) return -1;
            g->stepsSinceStatus = 0;
        }
        g->mirrorsFresh = false;
    }
    // ---- the caller's stream: per tick the step's outputs, then the observation pass
    // (before this call enqueues anything there)
    if (L->passOverlap && L->callStart[0]) HIP_TRY(hipEventRecord(L->callStart[(int)(L->overlapCalls & 1ull)], L->stream));
    if (own) HIP_TRY(hipStreamWaitEvent(L->stream, L->simDone, 0));
    std::vector<PublishTo> pubs((size_t)n);
    std::vector<uint32_t *> obsPtrs((size_t)n);
    // One gym, several ticks, every tick's observations in a slab of its own (an output ring at least k deep), nothing timed per tick: the
    // observation passes of up to MAX_STEP_TICKS ticks go out as ONE launch (launch_raster_batch: the next tick's expensive frames fill the tail of
    // the previous tick's pass).  The ticks are collected below and launched at the end of their chunk.
    bool batchRaster = canBatchRaster;
    for (int j = 0; j < k; ++j) batchRaster = batchRaster && !evs[j];
    // overlapped passes (mv_set_pass_overlap): this call's one launch goes to an internal stream
    // (an env must not finish in two consecutive calls: their passes may publish its true objective in either order -- episodes of at least
    // baseEpisodeLen seconds, 15 ticks each)
    // The ring is two CALLS deep, in the caller's ticks per call (a call of 16 ticks runs as two chunks of 8: the second-next chunk's passes would overwrite
    // what the consumer of the previous CALL -- enqueued after both of its chunks -- may still be reading), and rewards / dones have rings of
    // their own (two passes in flight would both publish the single arrays, in either order).
    const bool overlap = batchRaster && own && !callEv && L->passOverlap && L->passStream[0]
            && L->ringCount >= 2 * std::max(k, kCall) && L->ringRewards && L->ringDone &&
                         k <= MAX_STEP_TICKS && L->baseEpisodeLen * 15.0f > float(2 * k + 2);
    hipStream_t passOn = L->stream;
    if (overlap) {
        const int me = (int)(L->overlapCalls & 1ull);
        passOn = L->passStream[me];
        HIP_TRY(hipStreamWaitEvent(passOn, L->simDone, 0));                                      // this call's ticks
        if (L->overlapCalls >= 1) HIP_TRY(hipStreamWaitEvent(passOn, L->callStart[1 - me], 0));   // what the caller had enqueued when the previous call began
        // (first overlapped call: everything so far)
        else { HIP_TRY(hipEventRecord(L->userNow, L->stream)); HIP_TRY(hipStreamWaitEvent(passOn, L->userNow, 0)); }
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
            // (the fast observation pass publishes with its first workgroups)
            if (own && (!render || !allFast) && publish_outputs(gs[i], gs[i]->group * gs[i]->batch + j, o)) return -1;
        }
        // the call's last pass completes this call's mark (what the simulation stream waits for before it reuses the slot group)
        hipEvent_t mark = own && j == k - 1 ? L->userMark[L->markCount % PIPE_GROUPS] : nullptr;
        if (render && batchRaster) {
            const bool pubInRaster = own;
            chunkPubs.push_back(pubs[0]);
            chunkObs.push_back(obsPtrs[0]);
            // (0: off, launch_raster_batch declines)
            static const int chunkMax = getenv("MV_RASTER_BATCH") ? std::min((int)MAX_STEP_TICKS, std::max(1, atoi(getenv("MV_RASTER_BATCH"))))
                                               : (int)MAX_STEP_TICKS;
            if (j == k - 1 || (int)chunkObs.size() >= chunkMax) {
                const int cn = (int)chunkObs.size();
                if (callEv) { HIP_TRY(hipEventRecord(callEv[2], L->stream)); HIP_TRY(hipEventRecord(callEv[3], L->stream)); }
                int r = cn >= 2 ? launch_raster_batch(&views[(size_t)chunkFirst], chunkObs.data(), pubInRaster ? chunkPubs.data()
                                                      : nullptr, cn, L->w, L->h, overlap && cn == k ? passOn : L->stream, mark) : 1;
                if (r == 0 && overlap && cn == k) HIP_TRY(hipStreamWaitEvent(L->stream, mark, 0));   // the caller's stream sees the call's outputs as always
                if (r < 0) return fail("mv_step: observation size above 1024x1024");
                if (r == 0)   // (every pass of the one-launch kernel leaves its cost histogram zero)
                    for (int q = 0; q < cn; ++q) L->histClean[(size_t)views[(size_t)chunkFirst + q].lpt_parity] = 1;
                if (r == 1)   // (not applicable to this gym -- long lists -- or a chunk of one tick: tick by tick)
                    for (int q = 0; q < cn; ++q)
                    {
                        if (launch_raster(views[(size_t)chunkFirst + q], chunkObs[q], L->w, L->h, L->stream,
                            nullptr, 1, /*setup_done=*/1, pubInRaster ? &chunkPubs[q] : nullptr,
                                          q == cn - 1 ? mark : nullptr))
                            return fail("mv_step: observation size above 1024x1024");
                        // (self_clear, mv_raster.hip)
                        if (views[(size_t)chunkFirst + q].lpt_no_clear) L->histClean[(size_t)views[(size_t)chunkFirst + q].lpt_parity] = 1;
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
                if (r != 0) return fail(r == -2 ? "mv_group_step: the hand-over slots of a batched call are not one slot apart (internal)"
                    : "mv_group_step: observation size above 1024x1024");
                // (every pass leaves its cost histogram zero)
                for (size_t q = 0; q < (size_t)n * k; ++q) gs[q % (size_t)n]->histClean[(size_t)views[q].lpt_parity] = 1;
            }
        } else if (render) {
            const bool pubInRaster = own && allFast;
            if (n > 1 && allFast) {
                if (launch_raster_union(&views[(size_t)j * n], obsPtrs.data(), pubInRaster ? pubs.data() : nullptr,
                    n, L->w, L->h, L->stream, evs[j] ? evs[j][3] : nullptr, mark))
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
}  // extern "C"
void mvapi::group_detach(mv_gym *g)
{   // back to the gym's own simulation stream and events (they were kept aside while it was a member)
    if (!g->inGroup) return;
    mv_group *grp = g->inGroup;
    for (mv_gym *m : grp->gyms) {
        if (m != grp->gyms[0]) {
            m->simStream = m->ownSimStream; m->simDone = m->ownSimDone; m->stepDone = m->ownStepDone;
            for (int q = 0; q < PIPE_GROUPS; ++q) m->userMark[q] = m->ownUserMark[q];
            if (m->ownCopyStream) {
                (void)hipStreamSynchronize(m->copyStream);   // (the leader's: this member's uploads and read-backs are on it)
                if (m->genStream) m->genStream = m->ownCopyStream;
                m->copyStream = m->ownCopyStream; m->ownCopyStream = nullptr;
            }
        }
        m->inGroup = nullptr;
        m->simMustWaitUser = true; m->simOnOwnStream = false; m->simDoneValid = false; m->lastStep = nullptr; m->markCount = 0;
    }
    grp->gyms.clear();   // (the handle stays valid until mv_group_destroy; stepping it is an error from now on)
}
extern "C" {

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
        if (g->device != L->device || g->w != L->w || g->h != L->h || g->A != L->A || g->stream != L->stream || g->pipelined != L->pipelined)
            return fail("mv_group_create: the gyms of a group share device, observation size, agents per env, stream (mv_set_stream first) and pipelining");
    }
    HIP_TRY(hipSetDevice(L->device));
    for (int i = 0; i < n; ++i) {   // nothing in flight on the streams a member is about to leave
        HIP_TRY(hipStreamSynchronize(gyms[i]->simStream));
        HIP_TRY(hipStreamSynchronize(gyms[i]->stream));
        if (gyms[i]->copyStream) HIP_TRY(hipStreamSynchronize(gyms[i]->copyStream));
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
            // ONE copy stream for the group's status read-backs and episode uploads.  A device has four hardware queues and HIP deals its streams over
            // them: with a copy stream per member some of them shared a queue with the caller's stream or the simulation stream, and a read-back that
            // waits there for the step launch holds up the observation launch queued behind it (60-180 us gaps every other call in the traces of r08k;
            // Mixed 64 x 64, a copy stream per member / one for the group: 14.1 / 17.3 M obs/s, r08r).
            if (L->copyStream && g->copyStream) {
                g->ownCopyStream = g->copyStream; g->copyStream = L->copyStream;
                if (g->genStream) g->genStream = g->copyStream;
            }
        }
        g->simMustWaitUser = true; g->simOnOwnStream = false; g->simDoneValid = false; g->lastStep = nullptr; g->markCount = 0;
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
    // A call's chunks: what every member's slot groups hold, and at most MAX_GROUP_TICKS -- the two-launch batched path's limit
    // (raster_union_batch_applicable): a chunk
    // of 9..16 ticks would fall back to two launches per TICK.
    int chunk = (int)MAX_GROUP_TICKS;
    for (mv_gym *g : grp->gyms) chunk = std::min(chunk, g->batch);
    for (mv_gym *g : grp->gyms)
        if (!g->closed && g->statusPeriod <= 1) chunk = 1;   // (episodes of a few ticks: the refill protocol looks at the consumed counts after every tick)
    int rc = 0;
    std::string text;
    for (int done = 0; done < k; done += chunk) {
        const int r = step_gyms(grp->gyms.data(), (int)grp->gyms.size(), render != 0, std::min(chunk, k - done),
                                policy, seed, first_step_index + (uint32_t)done, k);
        if (r < 0) return -1;
        if (r > 0) { text += (text.empty() ? "" : " | ") + g_err; rc = 1; }
    }
    if (rc) g_err = text;
    return rc;
}

// the measured rules bench.py used to carry (r08p / r08y: 16 against 8 ticks per call, M obs/s: TowerBuilding 1024 envs 28.5 / 26.8, ObstaclesHard 1024 24.5 /
// 22.7, Rearrange 23.7 / 21.5,
// Collect 15.0 / 14.5; 512 envs 16.2 / 19.3, ObstaclesHard 512 13.2 / 18.8, Sokoban 19.9 / 25.9, 512 envs x 4 agents 19.5 / 26.0, 4096 envs 28.4 / 30.5;
// overlap r07a/b/j)
int mv_recommended_ticks_per_call(const mv_gym *g)
{
    if (!g || g->closed) return 1;
    if (g->statusPeriod <= 1) return 1;
    const int frames = g->N * g->A;
    int k = frames >= 1024 && frames < 2048 && g->scenario != SCN_SOKOBAN ? 16 : 8;
    if (g->inGroup) k = std::min(k, (int)MAX_GROUP_TICKS);
    return std::max(1, std::min(k, g->batch));
}

int mv_host_generator_threads(const mv_gym *g)
{
    if (!g || g->closed) return -1;
    return g->feeder && !g->feeder->device_gen() ? g->feeder->num_threads() : 0;
}

int mv_recommended_pass_overlap(const mv_gym *g)
{
    if (!g || g->closed || g->inGroup) return 0;
    if (g->statusPeriod <= 1) return 0;   // (episodes of a few ticks: stepped tick by tick, and two passes in flight may not both publish an env's true objective)
    // Measured on / off, M obs/s (r10za, r10zb, r10zc; the Obstacles family and Sokoban: r07j): wherever the passes are what a call waits for, the next call's
    // begin in the tail of this one's -- Rearrange 25.6 -> 28.6, HexMemory 9.5 -> 9.8, HexExplore 10.8 -> 11.0, Collect 16.6 -> 16.9; TowerBuilding 512 envs 25.4 ->
    // 26.9, 1024: 32.5 -> 34.4, 2048: 33.7 -> 35.0, 4096: 34.9 -> 35.6, 512 x 2 agents 22.1 -> 23.7 -- but 256 envs 17.1 -> 16.6 and 512 x 4 agents 28.8 -> 24.5
    // (their step launches, not their passes, bound them), Empty 46.9 -> 46.3.
    if (g->scenario == SCN_EMPTY) return 0;
    if (g->scenario == SCN_TOWER) return g->A <= 2 && g->N * g->A >= 512 ? 1 : 0;
    return 1;
}

int64_t mv_arena_bytes(const mv_gym *g) { return g ? (int64_t)g->arenaBytes : 0; }

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

}  // extern "C"
