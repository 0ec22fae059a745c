/*
 * oracle/mv_oracle.h -- C API of the CPU oracle.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under megaverse_amd/ may include, link or
 * call this.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline
 * leg load liboracle (as the checker / the timed CPU baseline).
 *
 * What this is: a dependency-free CPU restatement of the reference's
 * VectorEnv::step() path for the scenarios TowerBuilding, Obstacles{Easy,Medium,Hard,Walls,Steps,Lava},
 * Collect, Rearrange, Sokoban, HexMemory, HexExplore and Empty
 *   reference: src/libs/env/src/vector_env.cpp:89-120 (step/reset order)
 *              src/libs/env/src/env.cpp:57-152          (Env::reset/step)
 *              src/libs/env/src/kinematic_character_controller.cpp (controller)
 *              src/libs/scenarios/src/scenario_*.cpp, component_*.{hpp,cpp}, platforms.hpp, layout_utils.cpp (scenarios)
 *              src/libs/mazes/src (honeycomb maze + Kruskal, Hex scenarios)
 *              src/libs/magnum_rendering/src/magnum_env_renderer.cpp:158-340 (pixels)
 *
 * PARITY STATUS
 *   pinned   : RNG helpers randRange/frand/randomBool, triangularNumber, splitString, the
 *              Perlin noise of Collect's landscape and the honeycomb maze generator are
 *              checked against the reference's own util.hpp / math_utils.hpp /
 *              string_utils.cpp / perlin_noise.hpp / mazes library compiled in place
 *              (oracle/_ref, see oracle/Makefile), the RNG also against the C++ standard's mt19937
 *              known answer (10000th output == 4123659995); action-mask table,
 *              getCoords example (voxel_grid_tests.cpp:25), reward/episode
 *              formulas are spec-derived known answers (tests/test_oracle_spec.py).
 *              Round 3: canonical-pose controller cases whose outcome is derived BY HAND from
 *              the cited lines of kinematic_character_controller.cpp / agent.cpp (wall slide at
 *              30 / 45 / 60 degrees, a box that is no step but can be jumped on, ledges below the
 *              step height walked up, two agents head-on in controller order) and asserted on
 *              this oracle with a stated tolerance: tests/test_canonical_poses.py.
 *   UNPINNED : physics trajectories and pixels.  The reference delegates them to
 *              Bullet 2.89 and Magnum/OpenGL, neither of which is vendored in
 *              /root/reference nor installed here, and the reference's tests hold
 *              no golden vector for them.  "parity unpinned".
 *              (The restated convex cast forms Bullet's advancement dist / (-(d . n)) as
 *              (dist |v|) / (-(d . v)), one divide per iteration -- round 3, together with
 *              the device code, operation for operation.)
 */
#ifndef MV_ORACLE_H
#define MV_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct mvo_gym mvo_gym;

/* Same argument meaning as MegaverseGym's constructor (bindings/megaverse.cpp:38-58).
 * param_keys/param_vals: float params (scenario.hpp:225-242), n_params entries. */
mvo_gym *mvo_create(const char *scenario, int w, int h, int num_envs, int num_agents_per_env,
                    int num_simulation_threads, const char *const *param_keys,
                    const float *param_vals, int n_params);
void mvo_close(mvo_gym *g);

void mvo_seed(mvo_gym *g, int seed);
void mvo_reset(mvo_gym *g);
void mvo_set_actions(mvo_gym *g, int env_idx, int agent_idx, const int *actions, int n);
void mvo_set_action_mask(mvo_gym *g, int env_idx, int agent_idx, int mask);
void mvo_set_action_masks(mvo_gym *g, const int *masks); /* [N*A] env-major */
void mvo_step(mvo_gym *g);
/* physics+logic+auto-reset only, no rendering (for long rollouts in tests) */
void mvo_step_norender(mvo_gym *g);
void mvo_render(mvo_gym *g);
void mvo_set_raster(mvo_gym *g, int tiled);      /* 1: tile-culled software raster (byte-identical image; bench.py's timing leg), 0 (default): brute force */
void mvo_render_env(mvo_gym *g, int env_idx);   /* only this env's agents' frames */
void mvo_get_dones(mvo_gym *g, uint8_t *out);    /* [N] */

int mvo_is_done(mvo_gym *g, int env_idx);
void mvo_get_last_rewards(mvo_gym *g, float *out); /* [N*A] env-major */
float mvo_true_objective(mvo_gym *g, int env_idx, int agent_idx);
const uint8_t *mvo_get_observation(mvo_gym *g, int env_idx, int agent_idx); /* h*w*4, rows bottom-up */
float mvo_get_reward_shaping(mvo_gym *g, int env_idx, int agent_idx, const char *key, int *found);
void mvo_set_reward_shaping(mvo_gym *g, int env_idx, int agent_idx, const char *key, float v);

/* Packed state snapshot, layout documented in DESIGN.md ("snapshot format");
 * identical to what mv_debug_snapshot() of the HIP library writes. */
void mvo_debug_set_agent_pos(mvo_gym *g, int env_idx, int agent_idx, float x, float y, float z); /* test hook: teleport */
void mvo_debug_set_agent_yaw(mvo_gym *g, int env_idx, int agent_idx, float c, float s);           /* test hook: yaw basis from (cos, sin) */
void mvo_debug_set_agent_velocity(mvo_gym *g, int env_idx, int agent_idx, float hvx, float hvz, float vvel);
int mvo_snapshot_size(mvo_gym *g);
void mvo_snapshot(mvo_gym *g, int env_idx, void *out);

/* ---- spec-level helpers exposed for known-answer tests ---- */
/* Collect landscape noise: siv::PerlinNoise(seed).accumulatedOctaveNoise2D_0_1 (util/perlin_noise.hpp:315-318) */
void mvo_perlin_octave2_01(uint32_t seed, const double *xs, const double *ys, int n, int octaves, double *out);
uint32_t mvo_mt19937_nth(uint32_t seed, int n);            /* n-th output (1-based) */
int mvo_rand_range_seq(uint32_t seed, const int *lo, const int *hi, int n, int *out);
void mvo_frand_seq(uint32_t seed, int n, float *out);
void mvo_shuffle_iota(uint32_t seed, int n, int *out);      /* std::shuffle of 0..n-1 */
int mvo_action_mask(const int *actions, int n);             /* megaverse.cpp:100-116 */
void mvo_get_coords(const float *v, int *out);              /* voxel_grid.hpp:144-149 */
int mvo_hex_maze(int size, uint32_t seed, int *cells_out, int *border_counts, int *border_to, double *border_xy, double *centers,
                 double *bounds);                          /* src/libs/mazes: HoneyCombMaze + Kruskal + RemoveBorders */
int mvo_triangular_number(int n);                          /* util/math_utils.hpp:7-10 */
float mvo_building_reward_coeff(float height);              /* scenario_tower_building.cpp:246-251 */
void mvo_sincos(float x, float *s, float *c);               /* the fp32 polynomial both sides use */

#ifdef __cplusplus
}
#endif
#endif
