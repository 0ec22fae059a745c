// megaverse_amd/csrc/mv_feeder.cpp -- see mv_feeder.h
#include "mv_feeder.h"

#include <chrono>
#include <cstddef>

namespace mv {

EpisodeFeeder::EpisodeFeeder(int scenario, const ObstacleConfig &cfg, int num_envs, int num_agents, float base_episode_len, uint8_t *slots,
                             size_t slot_bytes, int device, int num_threads, std::vector<std::string> level_files)
    : scenario_{scenario}, num_envs_{num_envs}, num_agents_{num_agents}, device_{device}, cfg_{cfg}, base_len_{base_episode_len},
      slots_{slots}, slot_bytes_{slot_bytes}, rng_(num_envs), next_seq_(num_envs, 1), ready_seq_(num_envs), used_bytes_(num_envs, 0),
      soko_files_{std::move(level_files)}, soko_levels_(scenario == SCN_SOKOBAN ? num_envs : 0), soko_undo_(scenario == SCN_SOKOBAN ? num_envs : 0)
{
    std::random_device rd;   // unseeded envs: Env::EnvState::rng{std::random_device{}()} (env.hpp:169)
    for (int i = 0; i < num_envs; ++i) { rng_[i].seed(rd()); ready_seq_[i].store(0, std::memory_order_relaxed); }
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
        next_seq_[i] = first_seq[i];
        ready_seq_[i].store(0, std::memory_order_release);
        tasks_.push_back(Task{i, nullptr});
    }
    lk.unlock();
    cv_task_.notify_all();
}

const uint8_t *EpisodeFeeder::wait_ready(int env, int seq, size_t *used_bytes)
{
    if (failed()) return nullptr;
    if (ready_seq_[env].load(std::memory_order_acquire) != seq) {
        std::unique_lock<std::mutex> lk(mu_);
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
