// megaverse_amd/csrc/mv_feeder.cpp -- see mv_feeder.h
#include "mv_feeder.h"

#include <chrono>
#include <cstddef>

#include "mv_collect_draw.h"

namespace mv {

void launch_collect_draw(void *states, const int32_t *envs, int count, uint8_t *slots, size_t slot_bytes, int num_agents, float base_episode_len,
                         hipStream_t stream, int32_t *flag_word);   // mv_collect_draw.hip

EpisodeFeeder::EpisodeFeeder(int scenario, const ObstacleConfig &cfg, int num_envs, int num_agents, float base_episode_len, uint8_t *slots,
                             size_t slot_bytes, int device, int num_threads, std::vector<std::string> level_files, bool device_gen)
    : scenario_{scenario}, num_envs_{num_envs}, num_agents_{num_agents}, device_{device}, cfg_{cfg}, base_len_{base_episode_len},
      slots_{slots}, slot_bytes_{slot_bytes}, rng_(num_envs), next_seq_(num_envs, 1), ready_seq_(num_envs), used_bytes_(num_envs, 0),
      soko_files_{std::move(level_files)}, soko_levels_(scenario == SCN_SOKOBAN ? num_envs : 0), soko_undo_(scenario == SCN_SOKOBAN ? num_envs : 0),
      device_gen_{device_gen && scenario == SCN_COLLECT}, seeds_(num_envs, 0)
{
    std::random_device rd;   // unseeded envs: Env::EnvState::rng{std::random_device{}()} (env.hpp:169)
    for (int i = 0; i < num_envs; ++i) { rng_[i].seed(rd()); ready_seq_[i].store(0, std::memory_order_relaxed); }
    if (device_gen_) {
        (void)hipSetDevice(device_);
        const bool ok = hipMalloc(&d_states_, size_t(num_envs) * sizeof(cdraw::GenState)) == hipSuccess &&
                        hipMalloc((void **)&d_list_, size_t(num_envs) * sizeof(int32_t)) == hipSuccess &&
                        hipMalloc((void **)&d_flags_, sizeof(int32_t)) == hipSuccess && hipMemset(d_flags_, 0, sizeof(int32_t)) == hipSuccess &&
                        hipHostMalloc((void **)&h_list_, size_t(num_envs) * sizeof(int32_t), hipHostMallocDefault) == hipSuccess &&
                        hipHostMalloc((void **)&h_flags_, sizeof(int32_t), hipHostMallocDefault) == hipSuccess &&
                        hipStreamCreateWithFlags(&gen_stream_, hipStreamNonBlocking) == hipSuccess;
        if (!ok) failed_.store(true, std::memory_order_release);
        for (int i = 0; i < num_envs; ++i) seeds_[i] = (uint32_t)rd();
        if (const char *e = getenv("MV_DRAW_LINGER_MS")) linger_ms_ = std::max(0, atoi(e));
        if (ok && !upload_states()) failed_.store(true, std::memory_order_release);
        workers_.emplace_back([this] { device_worker_main(); });
        return;
    }
    for (int t = 0; t < std::max(1, num_threads); ++t) workers_.emplace_back([this] { worker_main(); });
}

EpisodeFeeder::~EpisodeFeeder()
{
    {
        std::lock_guard<std::mutex> lk(mu_);
        stop_ = true;
    }
    cv_task_.notify_all();
    for (auto &w : workers_) w.join();
    if (device_gen_) {
        if (gen_stream_) { (void)hipStreamSynchronize(gen_stream_); (void)hipStreamDestroy(gen_stream_); }
        if (d_states_) (void)hipFree(d_states_);
        if (d_list_) (void)hipFree(d_list_);
        if (d_flags_) (void)hipFree(d_flags_);
        if (h_list_) (void)hipHostFree(h_list_);
        if (h_flags_) (void)hipHostFree(h_flags_);
    }
}

bool EpisodeFeeder::upload_states()
{
    std::vector<cdraw::GenState> st((size_t)num_envs_);
    for (int i = 0; i < num_envs_; ++i) st[(size_t)i] = cdraw::GenState{seeds_[i], 1, next_seq_[i], 0};
    return hipMemcpy(d_states_, st.data(), st.size() * sizeof(cdraw::GenState), hipMemcpyHostToDevice) == hipSuccess;
}

void EpisodeFeeder::reseed(const std::vector<uint32_t> &values, const std::vector<int> &first_seq)
{
    std::unique_lock<std::mutex> lk(mu_);
    tasks_.clear();                                           // not started yet: their episodes would come from the old streams
    cv_done_.wait(lk, [this] { return in_flight_ == 0; });    // started ones finish first (they own rng_[env])
    for (int i = 0; i < num_envs_; ++i) {
        if (scenario_ == SCN_SOKOBAN) {   // episodes first_seq .. next_seq - 1 were generated ahead and are dropped: undo what they took
            std::deque<SokoUndo> &undo = soko_undo_[i];
            for (int q = next_seq_[i] - 1; q >= first_seq[i] && !undo.empty(); --q) {
                if (undo.back().reloaded) soko_levels_[i].pending.clear();
                else soko_levels_[i].pending.push_back(std::move(undo.back().level));
                undo.pop_back();
            }
            undo.clear();
        }
        rng_[i].seed((unsigned long)values[i]);               // Env::seed, env.cpp:52-55
        seeds_[i] = values[i];
        next_seq_[i] = first_seq[i];
        ready_seq_[i].store(0, std::memory_order_release);
        tasks_.push_back(Task{i, nullptr});
    }
    if (device_gen_ && !failed() && !upload_states()) failed_.store(true, std::memory_order_release);   // (no batch is in flight: the states are the host's to write)
    lk.unlock();
    cv_task_.notify_all();
}

const uint8_t *EpisodeFeeder::wait_ready(int env, int seq, size_t *used_bytes)
{
    if (failed()) return nullptr;
    if (ready_seq_[env].load(std::memory_order_acquire) != seq) {
        std::unique_lock<std::mutex> lk(mu_);
        if (device_gen_) { urgent_ = true; cv_task_.notify_all(); }   // (the device worker's batch stops gathering)
        // a healthy pool needs well under a millisecond per episode; a minute means the protocol was violated
        // (asking for an episode that was never scheduled): fail loudly instead of hanging the caller
        if (!cv_done_.wait_for(lk, std::chrono::seconds(60), [&] { return failed() || ready_seq_[env].load(std::memory_order_acquire) == seq; }) || failed())
            return nullptr;
    }
    if (used_bytes) *used_bytes = used_bytes_[env];
    return slots_ + size_t(env) * slot_bytes_;
}

void EpisodeFeeder::recycle(int env, hipEvent_t copied)
{
    {
        std::lock_guard<std::mutex> lk(mu_);
        tasks_.push_back(Task{env, copied});
    }
    cv_task_.notify_one();
}

void EpisodeFeeder::generate(int env)
{
    uint8_t *slot = slots_ + size_t(env) * slot_bytes_;
    const int seq = next_seq_[env]++;
    size_t used = slot_bytes_;
    if (scenario_ == SCN_OBSTACLES || scenario_ == SCN_EMPTY) {
        EpisodeBlob &b = *reinterpret_cast<EpisodeBlob *>(slot);
        if (scenario_ == SCN_EMPTY) generate_empty_episode(rng_[env], num_agents_, base_len_, b);
        else generate_obstacles_episode(rng_[env], cfg_, num_agents_, base_len_, b);
        b.seq = seq;
    } else if (scenario_ == SCN_REARRANGE) {
        RearrangeBlob &b = *reinterpret_cast<RearrangeBlob *>(slot);
        generate_rearrange_episode(rng_[env], num_agents_, base_len_, b);
        b.seq = seq;
    } else if (scenario_ == SCN_SOKOBAN) {
        SokobanBlob &b = *reinterpret_cast<SokobanBlob *>(slot);
        SokobanLevels &levels = soko_levels_[env];
        SokoUndo undo;
        undo.reloaded = levels.pending.empty();
        if (!undo.reloaded) undo.level = levels.pending.back();
        if (!generate_sokoban_episode(rng_[env], levels, soko_files_, num_agents_, base_len_, b)) {
            failed_.store(true, std::memory_order_release);
            return;
        }
        b.seq = seq;
        soko_undo_[env].push_back(std::move(undo));
        while (soko_undo_[env].size() > 8) soko_undo_[env].pop_front();   // never more than spares + 1 episodes ahead
    } else if (scenario_ == SCN_HEX_MEMORY || scenario_ == SCN_HEX_EXPLORE) {
        HexBlob &b = *reinterpret_cast<HexBlob *>(slot);
        if (scenario_ == SCN_HEX_MEMORY) generate_hex_memory_episode(rng_[env], num_agents_, base_len_, b);
        else generate_hex_explore_episode(rng_[env], num_agents_, base_len_, b);
        b.seq = seq;
        used = offsetof(HexBlob, boxes) + size_t(b.num_boxes) * sizeof(HexRec);   // the box list is last: used prefix only
    } else {
        CollectBlob &b = *reinterpret_cast<CollectBlob *>(slot);
        generate_collect_episode(rng_[env], num_agents_, base_len_, b);
        b.seq = seq;
        used = offsetof(CollectBlob, boxes) + size_t(b.num_boxes) * sizeof(LayoutBox);   // the slab list is last: used prefix only
    }
    if (const int f = generator_overflow_take()) overflow_.fetch_or(f, std::memory_order_relaxed);   // (raised on this worker thread)
    used_bytes_[env] = used;
    ready_seq_[env].store(seq, std::memory_order_release);
}

// Device mode's only worker: every env whose slot is free, in ONE launch of collect_draw_kernel on this feeder's stream -- while it runs (an episode is 0.7 ms
// of one wavefront on average, 2.5 ms the slowest) the next batch gathers.  The launch is ordered behind the uploads that still read the batch's slots.
void EpisodeFeeder::device_worker_main()
{
    (void)hipSetDevice(device_);
    std::vector<Task> batch;
    for (;;) {
        batch.clear();
        {
            std::unique_lock<std::mutex> lk(mu_);
            cv_task_.wait(lk, [this] { return stop_ || !tasks_.empty(); });
            // A launch lasts as long as its slowest episode -- 2 to 3 ms (5 to 7 before the slab merge went to all lanes), however few it draws -- and while ANY kernel of another queue is in flight the
            // observation launches run 9 % slower (r12h / r12j: with the draws back to back, in flight 84 % of the time, 1014 -> 1105 us; the same with a launch
            // that only sleeps).  So the batch gathers for a while -- an env's ring holds two more episodes -- unless somebody is waiting for one (wait_ready:
            // a forced reset, a re-seed, episodes of a few ticks).
            if (!stop_ && !urgent_) cv_task_.wait_for(lk, std::chrono::milliseconds(linger_ms_), [this] { return stop_ || urgent_; });
            if (stop_) return;
            urgent_ = false;
            batch.assign(tasks_.begin(), tasks_.end());
            tasks_.clear();
            in_flight_ += (int)batch.size();
        }
        bool ok = !failed();
        hipEvent_t last = nullptr;
        for (size_t k = 0; k < batch.size() && ok; ++k) {
            h_list_[k] = batch[k].env;
            if (batch[k].after && batch[k].after != last) { ok = hipStreamWaitEvent(gen_stream_, batch[k].after, 0) == hipSuccess; last = batch[k].after; }
        }
        const int count = (int)batch.size();
        ok = ok && hipMemcpyAsync(d_list_, h_list_, size_t(count) * sizeof(int32_t), hipMemcpyHostToDevice, gen_stream_) == hipSuccess;
        if (ok) {
            launch_collect_draw(d_states_, d_list_, count, slots_, slot_bytes_, num_agents_, base_len_, gen_stream_, d_flags_);
            ok = hipGetLastError() == hipSuccess && hipMemcpyAsync(h_flags_, d_flags_, sizeof(int32_t), hipMemcpyDeviceToHost, gen_stream_) == hipSuccess &&
                 hipStreamSynchronize(gen_stream_) == hipSuccess;
        }
        if (ok && *h_flags_) {
            overflow_.fetch_or(*h_flags_, std::memory_order_relaxed);
            ok = hipMemsetAsync(d_flags_, 0, sizeof(int32_t), gen_stream_) == hipSuccess;
        }
        if (!ok) failed_.store(true, std::memory_order_release);
        for (const Task &t : batch) {
            const int seq = next_seq_[t.env]++;
            used_bytes_[t.env] = slot_bytes_;   // (how many slabs the device drew is not known here: the whole record travels, device to device)
            if (ok) ready_seq_[t.env].store(seq, std::memory_order_release);
        }
        {
            std::lock_guard<std::mutex> lk(mu_);
            in_flight_ -= (int)batch.size();
        }
        cv_done_.notify_all();
    }
}

void EpisodeFeeder::worker_main()
{
    (void)hipSetDevice(device_);
    for (;;) {
        Task t;
        {
            std::unique_lock<std::mutex> lk(mu_);
            cv_task_.wait(lk, [this] { return stop_ || !tasks_.empty(); });
            if (stop_) return;
            t = tasks_.front();
            tasks_.pop_front();
            ++in_flight_;
        }
        if (t.after) (void)hipEventSynchronize(t.after);   // the slot is still the source of an in-flight upload
        generate(t.env);
        {
            std::lock_guard<std::mutex> lk(mu_);
            --in_flight_;
        }
        cv_done_.notify_all();
    }
}

}  // namespace mv
