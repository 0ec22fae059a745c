/*
 * include/megaverse_hip.h -- C ABI of libmegaverse_hip.so, the MI355X-native drop-in for the
 * reference's MegaverseGym hot path.
 *
 * Every entry point replaces one method of class MegaverseGym in the reference's pybind11 module
 * (reference: src/libs/bindings/megaverse.cpp; the Python-visible table is at :267-292).  A
 * maintainer binds these from the existing pybind shim or via ctypes
 * (megaverse_amd/extension.py does exactly that).  Plain pointers and sizes only; no torch, no
 * STL, no exceptions.  Return value: 0 = ok, negative = error (mv_last_error() has the text);
 * the reference instead logs and calls exit(-1) (src/libs/util/src/tiny_logger.cpp:109-113).
 * mv_step* / mv_reset may also return 1 = done, with a WARNING in mv_last_error(): a fixed capacity this build has and the
 * reference has not (visible primitives per frame, collision candidates, the voxel chunk, an episode record, a starved episode
 * ring) was hit since the last report.  The call did all of its work; the condition is reported once.
 *
 * Threading: like the reference every call is made from one host thread per gym.
 * All device work is enqueued on one HIP stream (mv_set_stream; default: the null stream);
 * mv_step() returns without synchronising, host getters synchronise that stream.
 */
#ifndef MEGAVERSE_HIP_H
#define MEGAVERSE_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct mv_gym mv_gym;

typedef struct mv_config {
    const char *scenario;          /* case-insensitive registered name (scenarios/init.hpp:26-57); "TowerBuilding" */
    int32_t obs_width, obs_height; /* megaverse.cpp:38 w, h */
    int32_t num_envs;              /* envs simulated by THIS process (one process per GPU) */
    int32_t num_agents_per_env;
    int32_t num_simulation_threads;/* worker threads of the background episode generator (every scenario but TowerBuilding, whose generator runs on the device);
                                      <= 0: this process's share of the host's cores (cores it may use / ranks of the job, at most 16); MV_FEEDER_THREADS overrides;
                                      stepping itself has no threads */
    int32_t use_vulkan;            /* accepted for signature parity; ignored */
    int32_t device;                /* HIP device ordinal */
    const char *const *param_keys; /* FloatParams (megaverse.cpp:45, scenario.hpp:225-242) */
    const float *param_vals;
    int32_t num_params;
    /* env sharding across GPUs: this process owns global envs [env_offset, env_offset+num_envs)
     * of total_envs; mv_seed() draws the per-env seeds for the whole job so that a sharded run is
     * bit-identical to the single-process run.  0 / 0 = not sharded. */
    int32_t env_offset, total_envs;
    /* 0 or 1: this process owns a contiguous block.  k > 1: it owns every k-th env starting at env_offset
     * (global index of local env j = env_offset + j * env_stride): the layout of a multi-task job that deals
     * scenarios round-robin by env index (megaverse_env.py:27-39, one gym per scenario). */
    int32_t env_stride;
} mv_config;

const char *mv_last_error(void);
int mv_device_count(void);   /* HIP devices this process can see (0 when there is none or the runtime cannot start) */
/* Version of this ABI.  2: mv_step* / mv_reset / mv_step_many / mv_group_step return 1 for "done, with a warning" (version 1 returned -1 for the same
 * conditions BEFORE doing the work): callers written as `if (mv_step(g)) fail();` must test `< 0` instead -- ask here which contract the library has. */
int mv_abi_version(void);

/* MegaverseGym::MegaverseGym (megaverse.cpp:38-58) / close (:227-243).  mv_close is idempotent
 * and valid before the first reset (megaverse/tests/test_env.py:28-30). */
int mv_create(const mv_config *cfg, mv_gym **out);
int mv_close(mv_gym *g);
int mv_destroy(mv_gym *g); /* mv_close + free the handle */

int mv_num_agents(const mv_gym *g);                 /* numAgents(), megaverse.cpp:71-74 */
int mv_action_space_sizes(int32_t *out6);           /* actionSpaceSizes(), :95-98 -> {3,3,3,2,2,3} */
int mv_seed(mv_gym *g, int32_t seed);               /* seed(), :60-69 */
int mv_reset(mv_gym *g);                            /* reset(), :76-93 (+ first render) */

/* setActions(), :100-116: multi-discrete -> Action bitmask for one agent (host staging) */
int mv_set_actions(mv_gym *g, int32_t env_idx, int32_t agent_idx, const int32_t *actions, int32_t n);
/* batched forms the reference lacks (SURVEY.md 3.2 hot loop iii): [N*A][6] multi-discrete.  The device form launches nothing: the buffer
 * is read by the NEXT step kernel, in the order of the caller's stream (keep it unchanged until that mv_step has been enqueued). */
int mv_set_actions_batched(mv_gym *g, const int32_t *host_actions);
int mv_set_actions_device(mv_gym *g, const int32_t *device_actions);
/* benchmark policy: i.i.d. uniform per head, counter-based (seed, step, agent, head) -> action;
 * same stream as megaverse_amd.rollout.sample_actions() on the host */
int mv_sample_random_actions(mv_gym *g, uint32_t seed, uint32_t step_index);
/* which generator mv_sample_random_actions / mv_step_n draw from: MV_POLICY_MULTIDISCRETE (default; = action_space.sample(),
 * megaverse_env.py:110-112) or MV_POLICY_SINGLE_BIT = Action(1 << randRange(0, NumActions)), the reference's own benchmark policy
 * (src/apps/megaverse_test_app.cpp:140-147); host twins: megaverse_amd/rollout.py */
enum { MV_POLICY_NONE = 0, MV_POLICY_MULTIDISCRETE = 1, MV_POLICY_SINGLE_BIT = 2 };
int mv_set_sample_policy(mv_gym *g, int32_t policy);
/* step several gyms of one job with one call (no reference counterpart: its multi-task runs are separate processes,
 * the scripts under megaverse_rl/runs): for each gym, optionally mv_sample_random_actions(seed, step_index), then mv_step / mv_step_no_render */
int mv_step_many(mv_gym *const *gyms, int32_t n, int32_t render, int32_t sample, uint32_t seed, uint32_t step_index);

/* Groups: up to 8 gyms of one job -- one per scenario of a multi-task batch, the reference's layout (megaverse/megaverse_env.py:27-39: one
 * MegaverseGym per task) -- stepped TOGETHER: one step launch and one observation launch per tick for all of them -- per CALL of 2..8 ticks
 * when every member has output rings (mv_set_output_ring) at least that deep and the group holds at most 1024 envs (all resident at once; a longer call is split into chunks of 8) -- on one shared pair of streams (BASELINE.json configs[4]: scenarios dealt round-robin over the envs of one batch; every gym keeps its env_offset /
 * env_stride, so seeds and sampled actions are the job-wide ones).  The members must share device, observation size, agents per env and
 * stream.  While grouped a gym is stepped through the group only; everything else (reset, seed, getters, shaping) stays per gym.
 * mv_group_step: k ticks like mv_step_n (render = 0: no observation pass).  Closing a member dissolves the group. */
typedef struct mv_group mv_group;
int mv_group_create(mv_gym *const *gyms, int32_t n, mv_group **out);
int mv_group_step(mv_group *grp, int32_t k, int32_t render, int32_t policy, uint32_t seed, uint32_t first_step_index);
int mv_group_destroy(mv_group *grp);

int mv_step(mv_gym *g);                             /* step(), :118-121: VectorEnv::step incl. auto-reset + render */
int mv_step_no_render(mv_gym *g);                   /* physics/logic/auto-reset only */
/* k open-loop ticks with one call = k iterations of the reference's benchmark loop body "for every agent setAction(random); venv.step()"
 * (megaverse_test_app.cpp:140-147 + vector_env.cpp:89-108): tick j draws its actions from (policy, seed, first_step_index + j) inside the
 * step kernel and renders every agent's observation.  Exactly the ticks k calls of mv_sample_random_actions + mv_step make -- but the
 * simulation stream and the caller's stream hand over to each other once per call instead of once per tick.  policy
 * MV_POLICY_NONE: the first tick acts on what mv_set_actions* left, the others on cleared actions.  The public arrays hold the LAST tick's
 * outputs -- or, with mv_set_output_ring, every tick's.  k may exceed the internal batch (mv_recommended_ticks_per_call; MV_PIPE_BATCH overrides its sizing): the call splits it. */
int mv_step_n(mv_gym *g, int32_t k, int32_t policy, uint32_t seed, uint32_t first_step_index);
/* Rollout rings (no reference counterpart: its learner copies each step's observation out of the gym, megaverse_env.py:121-130): tick
 * number t since this call leaves its observations in obs[t % count] ([count][N*A][h][w][4]), its rewards in rewards[t % count] ([count][N*A])
 * and its dones in dones[t % count] ([count][N]); a NULL ring keeps that output where it was.  count = 0 switches back to the single
 * slab / arrays.  Host getters (mv_get_observation, mv_get_last_rewards, ...) read the entry of the last tick. */
int mv_set_output_ring(mv_gym *g, int32_t count, void *obs, float *rewards, uint8_t *dones);
/* Overlapped observation passes (opt-in).  With rings for all three outputs at least TWO calls deep (count >= 2 k, k = the ticks of one mv_step_n call,
 * also where k exceeds the internal batch; otherwise the call runs as without the option) the one-launch observation passes of consecutive
 * mv_step_n calls run on two internal streams in turn: the passes of call c + 1 begin while the last workgroups of call c's drain.  The caller's
 * stream still waits for every call's passes before anything enqueued after the call runs.  The price is the ring's contract: an entry must be
 * consumed -- the consumer enqueued on the caller's stream -- before the NEXT stepping call after the one that produced it is issued (without
 * overlap: before the call that overwrites it).  on = 0 also releases the two internal streams (a HIP process shares few hardware queues among its streams:
 * GPU_MAX_HW_QUEUES, 4 by default).  No reference counterpart (the reference renders synchronously, vector_env.cpp:112-118). */
int mv_set_pass_overlap(mv_gym *g, int32_t on);
/* What a caller who just wants throughput should ask mv_step_n for -- the measured rules that used to live in bench.py (no reference counterpart):
 * mv_recommended_ticks_per_call: 16 (one tail of the one-launch observation pass per 16 ticks) for 1024 .. 2047 agent frames per tick where the gym's slot groups hold
 * 16 (mv_create sizes them by footprint: 16 where the 48 hand-over slots that takes stay under 2.25 GiB, else 8; MV_PIPE_BATCH overrides) and the scenario is not Sokoban; 8 otherwise; 1 where episodes can end within a few ticks
 * (such gyms are stepped tick by tick whatever k says).  A gym in a group: at most 8 (the two-launch group call's limit, and only while the group's envs are
 * all resident at once: 1024).  mv_recommended_pass_overlap: 1 wherever the
 * observation passes are what a call waits for (the next call's begin in the tail of this one's) -- every scenario but Empty; TowerBuilding with one or two agents per
 * env from 512 frames per tick on (with fewer frames, or four agents per env, the step launch bounds the call and the second pass stream only costs) --, 0 for
 * gyms whose episodes last a few ticks (the library declines to overlap there) and in groups.
 * mv_arena_bytes: the device memory this gym holds (state + the hand-over slots of PIPE_GROUPS x ticks-per-call ticks + its own observation slab). */
int mv_recommended_ticks_per_call(const mv_gym *g);
int mv_recommended_pass_overlap(const mv_gym *g);
int64_t mv_arena_bytes(const mv_gym *g);
/* Who draws this gym's episodes (Env::reset, env.cpp:57-76, off the step path in every case): the number of host threads of its episode feeder, or 0 when the
 * episodes are drawn on the device -- TowerBuilding always (tower_draw_kernel); Collect where the process's share of the host is under three cores and the gym has 256 envs and more, or
 * MV_COLLECT_DEVICE_GEN=1 (collect_draw_kernel: the host generator's episodes, byte for byte).  -1: no such gym. */
int mv_host_generator_threads(const mv_gym *g);
int mv_render(mv_gym *g);                           /* observation pass only */

int mv_is_done(mv_gym *g, int32_t env_idx);         /* isDone(), :123-126 -> 0/1, <0 on error */
int mv_get_dones(mv_gym *g, uint8_t *out);          /* [N] */
int mv_get_last_rewards(mv_gym *g, float *out);     /* getLastRewards(), :128-137 -> [N*A] env-major */
int mv_true_objective(mv_gym *g, int32_t env_idx, int32_t agent_idx, float *out); /* :203-206 */
int mv_get_true_objectives(mv_gym *g, float *out);  /* [N*A] */

/* getObservation(), :139-143: (h, w, 4) uint8, rows bottom-up like glReadPixels.  The reference
 * returns a view of host memory; here the frame lives in HBM: copy one frame out ... */
int mv_get_observation(mv_gym *g, int32_t env_idx, int32_t agent_idx, uint8_t *out_host);
/* ... or take the device slab [N*A][h][w][4] (valid until mv_close; rewritten by every step) */
void *mv_obs_device_ptr(mv_gym *g);
void *mv_rewards_device_ptr(mv_gym *g);             /* float [N*A] */
void *mv_dones_device_ptr(mv_gym *g);               /* uint8 [N] */
void *mv_true_objectives_device_ptr(mv_gym *g);     /* float [N*A] */
/* let the caller own the observation slab (e.g. a torch tensor / an RCCL gather buffer) */
int mv_set_obs_buffer(mv_gym *g, void *device_ptr);
int mv_set_stream(mv_gym *g, void *hip_stream);

/* Pixel arithmetic of the observation pass (MagnumEnvRenderer::draw, magnum_env_renderer.cpp:288-330, is a GPU
 * rasteriser: its pixels are defined up to fp32 rounding).  MV_PIXELS_FAST (default): hardware reciprocal / rsqrt,
 * within the tolerance DESIGN.md "pixel tolerance" states (<= 1 of 255 per channel except for a <= 1e-4 fraction of
 * silhouette / depth-tie pixels).  MV_PIXELS_EXACT: every fp32 operation correctly rounded, RGBA8 bit-identical to
 * the CPU oracle (parity tests).  Env var MV_PIXEL_MODE=exact|fast sets the default of new gyms. */
enum { MV_PIXELS_EXACT = 0, MV_PIXELS_FAST = 1 };
int mv_set_pixel_mode(mv_gym *g, int32_t mode);
int mv_get_pixel_mode(const mv_gym *g);

/* One-step-ahead pipelining (no reference counterpart).  On (default): the step kernels run on an internal stream,
 * and the step of tick t + 1 may overlap the observation pass of tick t whenever nothing the caller enqueued on its stream feeds
 * it (device-sampled or host-provided actions).  Nothing observable changes: rewards / dones / true objectives are published into
 * the arrays above ON THE CALLER'S STREAM, ordered with the observations, and a step never overwrites what a consumer enqueued
 * before the previous mv_step may still be reading.  mv_set_actions_device makes the next step wait (a policy in the loop is a true
 * dependency).  Off: everything runs on the caller's stream in order -- cheaper when several gyms already overlap each other
 * (MultiTaskGym).  Env var MV_PIPELINE=0|1 sets the default of new gyms. */
int mv_set_pipelining(mv_gym *g, int32_t on);
int mv_get_pipelining(const mv_gym *g);

/* setRenderResolution/drawHires/getHiresObservation (:145-178,203-207); drawOverview is a no-op
 * exactly like a reference build without WITH_GUI (:180-201) */
int mv_set_render_resolution(mv_gym *g, int32_t w, int32_t h);
int mv_draw_hires(mv_gym *g);
int mv_get_hires_observation(mv_gym *g, int32_t env_idx, int32_t agent_idx, uint8_t *out_host);
int mv_draw_overview(mv_gym *g);

/* getRewardShaping/setRewardShaping (:208-217); one key at a time across the C boundary */
int mv_num_reward_shaping_keys(const mv_gym *g);
const char *mv_reward_shaping_key(const mv_gym *g, int32_t i);
int mv_get_reward_shaping(mv_gym *g, int32_t env_idx, int32_t agent_idx, const char *key, float *out);
int mv_set_reward_shaping(mv_gym *g, int32_t env_idx, int32_t agent_idx, const char *key, float value);

int mv_synchronize(mv_gym *g);

/* In-stream kernel timing with HIP events on the gym's own stream (bench.py roofline leg, in a loop of its own: never inside
 * the timed region).  After mv_profile_begin the next max_steps calls of mv_step record events around the launches;
 * mv_profile_end synchronises and returns the mean milliseconds and sample count per interval: [0] step kernel (physics + logic +
 * auto-reset of finished envs + frame setup of the env's frames), [1] always 0 (retired), [2] output publish + frame sort (exact
 * pixel mode; ~0 in the fast mode), [3] raster.  Every interval is taken between two events of ONE stream.  The counterpart in the reference is the TinyProfiler timers around venv.step()
 * (src/apps/megaverse_test_app.cpp:68-74). */
int mv_profile_begin(mv_gym *g, int32_t max_steps);
int mv_profile_end(mv_gym *g, float *avg_ms4, int32_t *counts4);

/* test hooks: packed state snapshot of one env (layout in DESIGN.md, same bytes as the oracle's
 * mvo_snapshot) and the raw device RNG streams */
int mv_debug_set_agent_pos(mv_gym *g, int32_t env_idx, int32_t agent_idx, float x, float y, float z); /* teleport (fall-detection tests) */
/* scripted single-step physics cases (tests/test_canonical_poses*.py): the yaw basis from (cos, sin) as DefaultKinematicAgent's spawn builds it
 * (agent.cpp:42-46), and the controller's velocities */
int mv_debug_set_agent_yaw(mv_gym *g, int32_t env_idx, int32_t agent_idx, float c, float s);
int mv_debug_set_agent_velocity(mv_gym *g, int32_t env_idx, int32_t agent_idx, float hvx, float hvz, float vvel);
int mv_debug_snapshot_size(const mv_gym *g);
int mv_debug_snapshot(mv_gym *g, int32_t env_idx, void *out_host);
int mv_debug_rng(int32_t device, uint32_t seed, int32_t what, const int32_t *lo, const int32_t *hi, int32_t n, void *out_host);
int mv_debug_math(int32_t device, int32_t what, const float *a, const float *b, int32_t n, float *out_host);
/* Host-only (no device): the n-th (1-based) episode an env seeded with env_seed generates for a host-generated
 * scenario (Obstacles family, Collect), as the raw blob the reset kernel swaps in.  Returns the blob size in
 * bytes (out == NULL: size query only), -1 on error.  Replaces Env::reset's scenario->reset() + spawnAgents
 * draws (env.cpp:57-76). */
int mv_debug_generate_episode(const char *scenario, int32_t num_agents, int32_t env_seed, int32_t n, float base_episode_len,
                              void *out, int32_t out_bytes);
/* Host-only: runs the background episode feeder (worker pool that keeps every env's next episode generated ahead of
 * time; replaces the serial Env::reset inside VectorEnv::step, vector_env.cpp:93-105) for `rounds` episodes per env
 * and compares each delivered episode with sequential generation.  0 = identical. */
int mv_debug_feeder_selftest(const char *scenario, int32_t num_envs, int32_t num_agents, int32_t threads, int32_t rounds);
/* Host-only: the first n episodes of the Sokoban level generator (scenario_sokoban.cpp:80-170; its kernels come next) as n
 * packed records; out == NULL: record size.  Returns n, -1 on error. */
int mv_debug_generate_sokoban(int32_t num_agents, int32_t env_seed, int32_t n, float base_episode_len, void *out, int32_t out_bytes);

/* Collect's episode generator as it runs on the DEVICE (megaverse_amd/csrc/mv_collect_draw.h; the product's feeder uses it where mv_host_generator_threads
 * says 0 for a Collect gym).  _host: the same code compiled for the CPU, no device needed -- the n-th episode of an env seeded
 * with env_seed, the record mv_debug_generate_episode("Collect", ...) returns (seq = n).  _device: `count` envs draw their first n episodes on the GPU, out
 * receives the n-th of each (count x the record size), *ms_per_launch the mean duration of a launch of `count` wavefronts.  Replaces CollectScenario::reset +
 * createLandscape + addEpisodeDrawables (scenario_collect.cpp:20-161,190-214), siv::PerlinNoise (perlin_noise.hpp:118-126,315-318). */
int mv_debug_collect_draw_host(int32_t num_agents, int32_t env_seed, int32_t n, float base_episode_len, void *out, int32_t out_bytes);
int mv_debug_collect_draw_device(int32_t device, int32_t num_agents, const int32_t *env_seeds, int32_t count, int32_t n, float base_episode_len, void *out,
                                 int64_t out_bytes, float *ms_per_launch);

#ifdef __cplusplus
}
#endif
#endif
