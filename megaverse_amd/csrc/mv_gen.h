// megaverse_amd/csrc/mv_gen.h -- host-side episode generators (see mv_gen_obstacles.cpp)
#pragma once
#include <random>
#include <string>
#include <vector>

#include "mv_types.h"

namespace mv {

// ObstaclesScenario float params as integers (scenario_obstacles.hpp:46-68; per registered name :94-268)
struct ObstacleConfig {
    int min_platforms = 1, max_platforms = 2, min_gap = 1, max_gap = 2, min_lava = 1, max_lava = 4, min_height = 1, max_height = 3;
    int num_allowed_max_difficulty = 1;
    int platform_types[4] = {1, 2, 3, 4};   // WALL, LAVA, STEP, GAP
    int num_platform_types = 4;
    float carried_object_to_exit = 0.0f;    // default of the obstaclesAgentCarriedObjectToExit shaping key
};

// Fixed capacities of the episode records (the reference has none): a generator that would exceed one drops the excess AND raises
// a flag here; mv_step / mv_reset report it (mv_api.hip: check_status_flags).  Process-wide, cleared when reported.
enum : int { GEN_SLABS = 1, GEN_TERRAIN = 2, GEN_OBJECTS = 4, GEN_REWARDS = 8, GEN_COORDS = 16 };
void generator_overflow_raise(int flags);
int generator_overflow_take();   // returns the flags raised ON THIS THREAD since the last call and clears them

// Advances `rng` exactly like Env::reset + ObstaclesScenario::reset + spawnAgents and fills `out`.
void generate_obstacles_episode(std::mt19937 &rng, const ObstacleConfig &cfg, int num_agents, float base_episode_len, EpisodeBlob &out);

// Empty (scenario_empty.{hpp,cpp}, the reference's performance-test scenario): Env::reset's seed draw + the agents' spawn rotations;
// one static 20 x 2 x 20 box as the only layout slab, every agent at (1, 1, 1).  Same record as the Obstacles family.
void generate_empty_episode(std::mt19937 &rng, int num_agents, float base_episode_len, EpisodeBlob &out);

// Advances `rng` exactly like Env::reset + CollectScenario::reset + spawnAgents + addEpisodeDrawables and fills `out`.
void generate_collect_episode(std::mt19937 &rng, int num_agents, float base_episode_len, CollectBlob &out);

// Advances `rng` exactly like Env::reset + RearrangeScenario::reset + spawnAgents + addEpisodeDrawables and fills `out`.
void generate_rearrange_episode(std::mt19937 &rng, int num_agents, float base_episode_len, RearrangeBlob &out);

// HexExplore / HexMemory: advance `rng` exactly like Env::reset + the scenario's reset + spawnAgents + addEpisodeDrawables and fill `out`
// (the maze's Kruskal is seeded with the episode seed: mv_gen_hex.cpp)
void generate_hex_explore_episode(std::mt19937 &rng, int num_agents, float base_episode_len, HexBlob &out);
void generate_hex_memory_episode(std::mt19937 &rng, int num_agents, float base_episode_len, HexBlob &out);

// Sokoban keeps state across episodes: the shuffled levels of the file picked last (SokobanScenario::levels)
struct SokobanLevels {
    std::vector<std::vector<std::string>> pending;
};
// $BOXOBAN_LEVELS (or ~/datasets/boxoban) / unfiltered / train / 000.txt .. 999.txt that exist (scenario_sokoban.cpp:40-78)
std::vector<std::string> find_boxoban_level_files();
// Advances `rng` exactly like Env::reset + SokobanScenario::reset + spawnAgents and fills `out`; false if a level file is unreadable.
bool generate_sokoban_episode(std::mt19937 &rng, SokobanLevels &levels, const std::vector<std::string> &files, int num_agents,
                              float base_episode_len, SokobanBlob &out);

}  // namespace mv
