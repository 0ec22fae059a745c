// megaverse_amd/csrc/mv_feeder.h -- background episode generation for the host-generated scenarios
// (Obstacles family, Collect).
//
// The reference generates an env's next episode inside VectorEnv::step, serially, on the thread that is
// stepping (vector_env.cpp:93-105 -> Env::reset): "reset is the serial straggler" (SURVEY.md 8a row E).
// Here generation is off the step path entirely: every env always has its NEXT episode generated ahead of
// time in a pinned host slot by a small worker pool, so consuming an episode costs the step path one
// host-to-device copy enqueue.  An env's episodes form one RNG stream (Env::seed, then one
// randRange + re-seed per Env::reset), so per-env generation order is all that has to be preserved.
#pragma once
#include <hip/hip_runtime.h>

#include <atomic>
#include <condition_variable>
#include <deque>
#include <mutex>
#include <random>
#include <string>
#include <thread>
#include <vector>

#include "mv_gen.h"

namespace mv {

class EpisodeFeeder {
public:
    // slots: pinned host memory, num_envs * slot_bytes, owned by the caller
    // level_files: Sokoban only (the Boxoban level files found at construction, scenario_sokoban.cpp:40-78)
    // device_gen (Collect only): `slots` is DEVICE memory and the episodes are drawn there by collect_draw_kernel (mv_collect_draw.h) -- one worker thread
    // gathers the envs whose slots are free into a batch, launches the kernel for them on a stream of its own and waits for it; everything else --
    // is_ready / wait_ready / recycle / reseed, the refill protocol around them -- is what it is for the host generators.
    EpisodeFeeder(int scenario, const ObstacleConfig &cfg, int num_envs, int num_agents, float base_episode_len, uint8_t *slots,
                  size_t slot_bytes, int device, int num_threads, std::vector<std::string> level_files = {}, bool device_gen = false);
    ~EpisodeFeeder();
    EpisodeFeeder(const EpisodeFeeder &) = delete;
    EpisodeFeeder &operator=(const EpisodeFeeder &) = delete;

    // Env::seed for every env (values.size() == num_envs).  Drops episodes generated from the old streams and starts
    // generating `first_seq[i]` (the sequence number the device expects next) for every env.
    void reseed(const std::vector<uint32_t> &values, const std::vector<int> &first_seq);

    // Blocks until episode `seq` of `env` is complete in its slot; returns the slot and how many bytes of it are used.
    const uint8_t *wait_ready(int env, int seq, size_t *used_bytes);

    // non-blocking: is episode `seq` of `env` complete in its slot?
    bool is_ready(int env, int seq) const { return ready_seq_[env].load(std::memory_order_acquire) == seq; }

    // The slot of `env` was handed to an asynchronous copy that `copied` completes: once it has, generate seq + 1 into it.
    void recycle(int env, hipEvent_t copied);

    int num_threads() const { return int(workers_.size()); }
    bool device_gen() const { return device_gen_; }
    bool failed() const { return failed_.load(std::memory_order_acquire); }   // a generator gave up (Sokoban: unreadable level file)
    int take_overflow() { return overflow_.exchange(0, std::memory_order_relaxed); }   // GEN_* flags this feeder's generators raised since the last call

private:
    struct Task { int env; hipEvent_t after; };
    void worker_main();
    void device_worker_main();
    void generate(int env);

    const int scenario_, num_envs_, num_agents_, device_;
    const ObstacleConfig cfg_;
    const float base_len_;
    uint8_t *const slots_;
    const size_t slot_bytes_;
    std::vector<std::mt19937> rng_;
    std::vector<int> next_seq_;                  // sequence number the next generated episode of env i gets (worker-owned once queued)
    std::vector<std::atomic<int>> ready_seq_;    // sequence number of the complete episode in slot i (0 = none)
    std::vector<size_t> used_bytes_;
    std::mutex mu_;
    std::condition_variable cv_task_, cv_done_;
    std::deque<Task> tasks_;
    int in_flight_ = 0;
    bool stop_ = false;
    std::vector<std::thread> workers_;
    std::atomic<bool> failed_{false};
    std::atomic<int> overflow_{0};
    // Sokoban keeps state across episodes and across Env::seed: the shuffled rest of the level file picked last
    // (SokobanScenario::levels).  Episodes generated ahead of a re-seed are dropped, so what they took from that list is put back.
    struct SokoUndo { bool reloaded; std::vector<std::string> level; };
    std::vector<std::string> soko_files_;
    std::vector<SokobanLevels> soko_levels_;
    std::vector<std::deque<SokoUndo>> soko_undo_;
    // device mode: the generators' states on the device (cdraw::GenState per env), a batch's env list (pinned + device), the flag word its kernels raise
    const bool device_gen_;
    bool urgent_ = false;       // under mu_: somebody waits for an episode, launch what is there
    int linger_ms_ = 15;        // how long a batch gathers otherwise (MV_DRAW_LINGER_MS; r12k: 0 / 10 / 25 / 50 / 100 ms: 16.4 / 16.6 / 16.5 / 16.0 / 14.7 M obs/s)
    void *d_states_ = nullptr;
    int32_t *h_list_ = nullptr, *d_list_ = nullptr, *h_flags_ = nullptr, *d_flags_ = nullptr;
    hipStream_t gen_stream_ = nullptr;
    bool upload_states();   // next_seq_ / seeds -> d_states_ (reseed, under mu_ with no batch in flight)
    std::vector<uint32_t> seeds_;
};

}  // namespace mv
