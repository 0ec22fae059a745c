/*
 * oracle/mv_oracle.cpp -- CPU restatement ("oracle") of the reference hot path.
 *
 * TEST INFRASTRUCTURE.  See mv_oracle.h for who may use it and for the parity
 * status (RNG helpers pinned; physics + pixels "parity unpinned").
 *
 * Plain scalar C++17, libstdc++ only.  Build: oracle/Makefile (g++ -O2
 * -ffp-contract=off, no -ffast-math) so that every fp32 operation is a single
 * IEEE-754 rounding in source order -- the HIP kernels are built the same way and
 * tests compare the two bit for bit.
 *
 * Each block cites the reference file:line it follows (paths relative to
 * /root/reference/src/libs).  Where the reference hands the arithmetic to Bullet
 * 2.89 / Magnum (not vendored) the published algorithm is restated and marked [3P].
 */
#include "mv_oracle.h"

#include <algorithm>
#include <functional>
#include <numeric>
#include <atomic>
#include <cfloat>
#include <cmath>
#include <condition_variable>
#include <cstdio>
#include <cstring>
#include <map>
#include <memory>
#include <mutex>
#include <random>
#include <set>
#include <string>
#include <thread>
#include <vector>
#ifdef __linux__
#include <pthread.h>
#include <sched.h>
#endif

namespace mvo {

// ------------------------------------------------------------------------------------------------
// RNG helpers -- util/include/util/util.hpp:25-56.  Same libstdc++ templates as the reference.
// ------------------------------------------------------------------------------------------------
using Rng = std::mt19937;

static inline int randRange(int low, int high, Rng &rng) { return std::uniform_int_distribution<>{low, high - 1}(rng); }
static inline bool randomBool(Rng &rng) { return bool(randRange(0, 2, rng)); }
static inline float frand(Rng &rng) { return std::uniform_real_distribution<float>{0, 1}(rng); }

// ------------------------------------------------------------------------------------------------
// Constants (every number read from the cited reference line)
// ------------------------------------------------------------------------------------------------
static const float DT = 1.0f / 15.0f;                 // env/include/env/env.hpp:160-161
static const float CAP_R = 0.33f;                     // env/src/agent.cpp:53
static const float CAP_HH = 1.05f * 0.5f;             // env/src/agent.cpp:52 (btCapsuleShape height/2)
static const float AGENT_HEIGHT = 1.75f;              // env/include/env/agent.hpp:110
static const float STEP_HEIGHT = 0.2f;                // env/src/agent.cpp:59
static const float GRAVITY = 1.4f * 9.8f;             // kinematic_character_controller.hpp:169
static const float FALL_SPEED = 55.0f;                // kinematic_character_controller.cpp:135
static const float MAX_H_SPEED = 4.5f;                // .hpp:173
static const float MAX_AIR_SPEED = 1.0f;              // .hpp:174
static const float NORMAL_DECEL = 15.0f;              // .hpp:175
static const float MAX_ACCEL = 35.0f + 15.0f;         // .hpp:176
static const float MAX_AIR_ACCEL = 3.0f;              // .hpp:176
static const float EXCEED_DECEL = (35.0f + 15.0f) * 2;// .hpp:177
static const float MAX_PEN_DEPTH = 0.041f;            // .hpp:155
static const float MAX_SLOPE_COS = 0.70710678f;       // .cpp:146 cos(45 deg)
static const float ALLOWED_CCD_PEN = 0.04f;           // [3P] btDispatcherInfo::m_allowedCcdPenetration default
static const float CAST_RADIUS = 0.001f;              // [3P] btContinuousConvexCollision "radius"
static const int CAST_MAX_ITER = 64;                  // [3P] MAX_ITERATIONS
static const float SIMD_EPS = FLT_EPSILON;            // [3P] SIMD_EPSILON
static const float ROTATE_RAD = 3.5f, ROTATE_X_RAD = 1.5f; // env/include/env/agent.hpp:109
static const float PI_F = 3.14159274f;                // Magnum::Constants::pi() as float
static const float OBJ_HALF = 0.39f;                  // component_object_stacking.hpp:172
static const float OBJ_COLL_HALF = 0.39f * 1.15f;     // :181 collision scale
static const float OBJ_COLL_YOFF = -0.05f;            // :182 collision offset
static const float CARRY_SCALE = 0.78f;               // :63

enum { CX = 32, CY = 16, CZ = 32, CHUNK = CX * CY * CZ };
enum { MAX_BOXES = 1024, MAX_OBJECTS = 80, MAX_AGENTS = 8, MAX_TERRAIN = 16, MAX_REWARDS = 96, MAX_SHAPING = 8 };
enum { HM_DIM = 42 };   // Collect heightfield: maxWidth == maxLength == 42 (scenario_collect.cpp:63)
enum { SCN_TOWER = 0, SCN_OBSTACLES = 1, SCN_COLLECT = 2, SCN_REARRANGE = 3, SCN_SOKOBAN = 4, SCN_EMPTY = 5, SCN_HEX_MEMORY = 6, SCN_HEX_EXPLORE = 7 };
enum { HEX_MAX_BOXES = 2048, HEX_MAX_OBJS = 128, HEX_FRAMES = 3 };   // Hex*: float boxes (floor, walls, edgings, landmarks), collectables
enum { HEX_PILLAR = 0, HEX_DIAMOND = 1, HEX_SPHERE = 2 };            // scenario_hex_memory.cpp:163-168 ShapeType
enum { SOKO_DIM = 32, SOKO_WALL = 1, SOKO_GOAL = 2 };   // Sokoban level cells (scenario_sokoban.cpp:28-33), levels up to 32 x 32
enum { MAX_STATIC = 16, MAX_ITEMS = 8 };   // Rearrange: static colliding boxes, arrangement items (arrangementSize < 8)
enum { SHAPE_BOX = 0, SHAPE_CAPSULE = 1, SHAPE_SPHERE = 2, SHAPE_CYLINDER = 4 };   // DrawableType, env.hpp:58-69
enum { TERRAIN_EXIT = 1, TERRAIN_LAVA = 2, TERRAIN_BUILDING_ZONE = 4 };   // scenarios/platforms.hpp:28-34
static const unsigned COLOR_EXIT_PAD = 0x50c878, COLOR_RED = 0xff0000, COLOR_GREEN = 0x3bb372;   // const.hpp:25-56

// voxel byte: voxel_state.hpp:10-37 + platforms.hpp:28-34 folded into one byte per cell
enum { VX_SOLID = 1, VX_OPAQUE = 2, VX_OBJECT = 4, VX_TERRAIN_SHIFT = 3, VX_COLOR_SHIFT = 6 };

static const unsigned LAYOUT_COLORS[14] = {  // env/include/env/const.hpp:121-136
    0xffffff, 0xffffe6, 0xccffcc, 0xe6ecff, 0xd9d9d9, 0xffebcc, 0xb3b3b3, 0xb3b3b3, 0xb3b3b3, 0xb3b3b3,
    0x555555, 0x555555, 0x555555, 0x555555};
static const unsigned COLOR_BUILDING_ZONE = 0x555555, COLOR_MOVABLE_BOX = 0xadd8e6, COLOR_AGENT_EYES = 0x2c3e50,
                      COLOR_UI_BAR = 0x2eb5d0;  // const.hpp:25-56
static const unsigned ALL_COLORS[22] = {  // const.hpp:58-83 allColors
    0xffdd3c, 0x3bb372, 0x50c878, 0x2eb5d0, 0xadd8e6, 0x3a7fa6, 0x2c3e50, 0xffb400, 0xb3b3b3, 0x555555, 0x222222,
    0xffffff, 0xff0000, 0xffa770, 0xd468ee, 0xffe6e6, 0xffffe6, 0xccffcc, 0xe6ecff, 0xd9d9d9, 0xf2e6ff, 0xffebcc};
static const unsigned AGENT_COLORS[7] = {0xffdd3c, 0x3bb372, 0x2eb5d0, 0xffb400, 0xd468ee, 0x222222, 0xff0000};  // const.hpp:85

struct V3 {
    float x, y, z;
};
static inline V3 v3(float x, float y, float z) { return V3{x, y, z}; }
static inline V3 operator+(V3 a, V3 b) { return v3(a.x + b.x, a.y + b.y, a.z + b.z); }
static inline V3 operator-(V3 a, V3 b) { return v3(a.x - b.x, a.y - b.y, a.z - b.z); }
static inline V3 operator*(V3 a, float s) { return v3(a.x * s, a.y * s, a.z * s); }
static inline float dot(V3 a, V3 b) { return (a.x * b.x + a.y * b.y) + a.z * b.z; }
static inline float len2(V3 a) { return dot(a, a); }

// ------------------------------------------------------------------------------------------------
// fp32 sin/cos used wherever a result feeds state or pixels (cephes-style minimax polynomials,
// one rounding per op, identical sequence in the HIP kernels).  |x| <= ~8.
// ------------------------------------------------------------------------------------------------
static void mv_sincos(float x, float *s_out, float *c_out)
{
    const float TWO_OVER_PI = 0.636619772f;
    const float PIO2_HI = 1.57079625f;        // 0x3FC90FDA
    const float PIO2_LO = 7.54978942e-08f;    // pi/2 - PIO2_HI
    float kf = floorf(x * TWO_OVER_PI + 0.5f);
    int k = (int)kf;
    float r = x - kf * PIO2_HI;
    r = r - kf * PIO2_LO;
    float r2 = r * r;
    float sp = ((-1.9515295891e-4f * r2 + 8.3321608736e-3f) * r2 - 1.6666654611e-1f) * r2 * r + r;
    float cp = ((2.443315711809948e-5f * r2 - 1.388731625493765e-3f) * r2 + 4.166664568298827e-2f) * r2 * r2 -
               0.5f * r2 + 1.0f;
    float s, c;
    switch (k & 3) {
        case 0: s = sp; c = cp; break;
        case 1: s = cp; c = -sp; break;
        case 2: s = -sp; c = -cp; break;
        default: s = -cp; c = sp; break;
    }
    *s_out = s;
    *c_out = c;
}

// y-axis rotation matrix entries (c, s) the way Bullet builds them from an axis-angle quaternion:
// [3P] btQuaternion::setRotation + btMatrix3x3::setRotation, restated for axis (0,1,0).
static void yaw_matrix(float angle, float *c_out, float *s_out)
{
    float sh, ch;
    mv_sincos(angle * 0.5f, &sh, &ch);
    float qy = sh, w = ch;
    float d = qy * qy + w * w;
    float s = 2.0f / d;
    float ys = qy * s;
    float wy = w * ys;
    float yy = qy * ys;
    *c_out = 1.0f - yy;
    *s_out = wy;
}

// ------------------------------------------------------------------------------------------------
// State
// ------------------------------------------------------------------------------------------------
struct Box {  // merged layout parallelepiped, voxel units, max exclusive
    int min[3], max[3];
    int type, slot;
};

struct TerrainBox {  // platforms.hpp terrainBoxes, voxel units, max exclusive
    int min[3], max[3];
    int type;
};

struct RewardObj {  // diamond: scenario_obstacles.cpp:251-258 (green), scenario_collect.cpp:190-214 (green +1 / red -1)
    int x, y, z;
    int active;     // 0 collected, 1 there (Collect: 1 = +1 reward, 2 = -1 reward)
};

struct Object {  // movable box: component_object_stacking.hpp:170-198
    int x, y, z;
    int state;  // 0 = placed at voxel, 1+k = carried by agent k, -1 = placed but its grid cell was erased (Collect, see collect_step)
};

struct Agent {
    V3 pos;                      // ghost origin == capsule centre (agent.cpp:45)
    float m00, m02, m20, m22;    // yaw basis (agent.cpp:128-133 accumulates matrix products)
    float pitch;                 // currXRotation (agent.cpp:110-126)
    float hvx, hvz;              // horizontalVelocity (y is always 0)
    float vvel, voffset;         // m_verticalVelocity, m_verticalOffset
    float step_offset;           // m_currentStepOffset (persists across steps)
    float jump_speed;            // m_jumpSpeed (10 until first jump)
    int was_jumping;
    int carrying;                // object index or -1
    int picked_up, visited_zone; // scenario_tower_building.hpp:24-28 ; Obstacles: visited_zone == agentReachedExit
    int spawn[3];                // fallDetection.agentInitialPositions
    float last_reward, total_reward;
    float shaping[MAX_SHAPING];  // per scenario, see SHAPING_KEYS_*
    int action;                  // bitmask env.hpp:22-42
};

static const char *SHAPING_KEYS_TOWER[4] = {"teamSpirit", "towerPickedUpObject", "towerVisitedBuildingZoneWithObject",
                                           "towerBuildingReward"};
static const float SHAPING_DEFAULT_TOWER[4] = {0.1f, 0.1f, 0.1f, 1.0f};  // scenario_tower_building.hpp:44-52
// scenario_obstacles.hpp:36-44 (+ Scenario::init teamSpirit 0, scenario.hpp:94-103)
static const char *SHAPING_KEYS_OBST[5] = {"teamSpirit", "obstaclesAgentAtExit", "obstaclesAllAgentsAtExit", "obstaclesExtraReward",
                                          "obstaclesAgentCarriedObjectToExit"};
static const float SHAPING_DEFAULT_OBST[5] = {0.0f, 1.0f, 5.0f, 0.5f, 0.0f};
// scenario_collect.hpp:44-52 (+ teamSpirit 0)
// scenario_rearrange.hpp:96-102 (+ teamSpirit 0)
static const char *SHAPING_KEYS_REARRANGE[3] = {"teamSpirit", "rearrangeOneMoreObjectCorrectPosition", "rearrangeAllObjectsCorrectPosition"};
static const float SHAPING_DEFAULT_REARRANGE[3] = {0.0f, 1.0f, 10.0f};
// scenario_sokoban.hpp:40-47 (+ teamSpirit 0)
static const char *SHAPING_KEYS_SOKOBAN[4] = {"teamSpirit", "sokobanBoxOnTarget", "sokobanBoxLeavesTarget", "sokobanAllBoxesOnTarget"};
static const float SHAPING_DEFAULT_SOKOBAN[4] = {0.0f, 1.0f, -1.0f, 10.0f};
// scenario_hex_memory.hpp:37-43, scenario_hex_explore.hpp:28-31 (+ teamSpirit 0)
static const char *SHAPING_KEYS_HEX_MEMORY[3] = {"teamSpirit", "memoryCollectGood", "memoryCollectBad"};
static const float SHAPING_DEFAULT_HEX_MEMORY[3] = {0.0f, 1.0f, -1.0f};
static const char *SHAPING_KEYS_HEX_EXPLORE[2] = {"teamSpirit", "exploreSolved"};
static const float SHAPING_DEFAULT_HEX_EXPLORE[2] = {0.0f, 5.0f};
static const char *SHAPING_KEYS_COLLECT[5] = {"teamSpirit", "collectSingleGood", "collectSingleBad", "collectAll", "collectAbyss"};
static const float SHAPING_DEFAULT_COLLECT[5] = {0.0f, 1.0f, -1.0f, 5.0f, -0.5f};

enum { PT_EMPTY = 0, PT_WALL, PT_LAVA, PT_STEP, PT_GAP, PT_START, PT_EXIT, PT_TRANSITION };
struct ObstacleParams {  // scenario_obstacles.hpp:46-68 defaults, overridden per registered name :94-268
    int minPlatforms = 1, maxPlatforms = 2, minGap = 1, maxGap = 2, minLava = 1, maxLava = 4, minHeight = 1, maxHeight = 3;
    int numAllowedMaxDifficulty = 1;
    std::vector<int> platformTypes = {PT_WALL, PT_LAVA, PT_STEP, PT_GAP};
};

struct Env {
    int scenario = SCN_TOWER;
    ObstacleParams op;
    int numShaping = 4;
    const char *const *shapingKeys = SHAPING_KEYS_TOWER;
    int numTerrain = 0, numRewards = 0, numPlatforms = 0, solved = 0;
    // Rearrange (scenario_rearrange.hpp): the target arrangement (drawn, static, on the left pedestal), the movable copy on the
    // right pedestal == objects[0..numItems) in item order, the pedestals / raised floor as float boxes; numPlatforms holds
    // maxMatchingObjects
    struct ArrItem { int shape; unsigned color; int off[3]; };
    struct StaticBox { V3 lo, hi; unsigned color; };
    int numItems = 0, numStatic = 0;
    ArrItem items[MAX_ITEMS];
    StaticBox statics[MAX_STATIC];
    // Sokoban (scenario_sokoban.{hpp,cpp}): voxels are 2 units wide; the level's wall / goal cells (drawn as caps and pads, the
    // walls themselves are solid but invisible); objects[] are the pushable boxes; highestTower holds numBoxesOnGoal.  The
    // shuffled levels of the file picked last live on across episodes (SokobanScenario::levels).
    float voxelSize = 1.0f;
    std::vector<uint8_t> soko = std::vector<uint8_t>(SOKO_DIM * SOKO_DIM, 0);   // [x * SOKO_DIM + z]: SOKO_WALL | SOKO_GOAL
    std::vector<std::vector<std::string>> sokoLevels;
    const std::vector<std::string> *sokoFiles = nullptr;                         // shared, owned by the gym
    // HexMemory / HexExplore (component_hexagonal_maze.cpp, scenario_hex_{memory,explore}.cpp): the floor, walls, edgings and
    // landmarks as float boxes, each in the world frame or in one of the three wall orientations (a rotation about Y: the
    // vertical capsule stays vertical, so physics and rays treat such a box as an axis-aligned box in ITS frame); the collectables
    // (HexExplore: hexObjs[0] is the reward diamond).  numPlatforms holds goodObjects.size(), highestTower goodObjectsCollected.
    struct HexBox { int frame; V3 lo, hi; unsigned color; int collide; };           // frame: -1 world, 0..2 = HEX_ROT[k]
    struct HexObj { V3 pos, scale; int shape, good, alive; unsigned color; int vox[3]; };   // pos/scale: what addObject got
    std::vector<HexBox> hexBoxes;
    std::vector<HexObj> hexObjs;
    V3 hexTarget = V3{0, 0, 0};                                                     // HexExplore rewardObjectCoords
    // Collect: numPlatforms holds numPositiveRewards, highestTower holds positiveRewardsCollected (scenario_collect.hpp:76)
    std::vector<int8_t> heightmap = std::vector<int8_t>(HM_DIM * HM_DIM, -1);   // [x * HM_DIM + z]: top solid y of the column, -1 = no voxels
    TerrainBox terrain[MAX_TERRAIN];
    RewardObj rewards[MAX_REWARDS];
    int numAgents = 1;
    Rng rng{std::random_device{}()};  // env.hpp:169
    // float params (scenario.hpp:225-232)
    float p_episodeLengthSec = 60.0f, p_verticalLookLimitRad = 0.2f;

    int L = 0, H = 0, W = 0;
    int bz[4] = {0, 0, 0, 0};  // minx, maxx, minz, maxz  (min.y == max.y == 1)
    unsigned layoutColor = 0, wallColor = 0;
    int drawWalls = 1;
    int numObjects = 0, numBoxes = 0;
    Box boxes[MAX_BOXES];
    Object objects[MAX_OBJECTS];
    Agent agents[MAX_AGENTS];
    std::vector<uint8_t> chunk = std::vector<uint8_t>(CHUNK, 0);

    int done = 0, numFrames = 0, highestTower = 0;
    float episodeSec = 0, episodeLen = 0, bzReward = 0;
    float barHalfWidth = 0.24f;  // scenario_default.hpp:69,164-169

    static int cell(int x, int y, int z) { return (y * CZ + z) * CX + x; }
    static bool inChunk(int x, int y, int z) { return x >= 0 && x < CX && y >= 0 && y < CY && z >= 0 && z < CZ; }
    uint8_t vox(int x, int y, int z) const { return inChunk(x, y, z) ? chunk[cell(x, y, z)] : 0; }
};

// ------------------------------------------------------------------------------------------------
// Procedural generation
// ------------------------------------------------------------------------------------------------
static unsigned randomLayoutColor(Rng &rng) { return LAYOUT_COLORS[randRange(0, 14, rng)]; }  // const.hpp:139-143

static void fill_box(Env &e, int x0, int y0, int z0, int x1, int y1, int z1, uint8_t v)
{   // component_voxel_grid.hpp:83-90 addBoundingBox
    for (int x = x0; x < x1; ++x)
        for (int y = y0; y < y1; ++y)
            for (int z = z0; z < z1; ++z)
                if (Env::inChunk(x, y, z)) e.chunk[Env::cell(x, y, z)] = v;
}

// Canonical greedy merge of layout voxels into parallelepipeds.  The reference
// (component_voxel_grid.hpp:108-187) visits an unordered_map in hash order, which makes its
// decomposition implementation-defined; the union is the same.  Ours:
// keys sorted by (type, colour slot); per key scan y, then z, then lowest x; grow the x-run, then
// +z, then +y.
static void merge_boxes(Env &e)
{
    std::vector<uint8_t> visited(CHUNK, 0);
    e.numBoxes = 0;
    for (int key = 0; key < 16; ++key) {
        const int type = key >> 2, slot = key & 3;
        if (type == 0) continue;
        for (int y = 0; y < CY; ++y)
            for (int z = 0; z < CZ; ++z)
                for (int x = 0; x < CX; ++x) {
                    auto match = [&](int xx, int yy, int zz) {
                        if (!Env::inChunk(xx, yy, zz)) return false;
                        const int c = Env::cell(xx, yy, zz);
                        const uint8_t v = e.chunk[c];
                        return !visited[c] && (v & 3) == type && (v >> VX_COLOR_SHIFT) == slot;
                    };
                    if (!match(x, y, z)) continue;
                    int x1 = x;
                    while (match(x1 + 1, y, z)) ++x1;
                    int z1 = z;
                    for (;;) {
                        bool ok = true;
                        for (int xx = x; xx <= x1 && ok; ++xx) ok = match(xx, y, z1 + 1);
                        if (!ok) break;
                        ++z1;
                    }
                    int y1 = y;
                    for (;;) {
                        bool ok = true;
                        for (int zz = z; zz <= z1 && ok; ++zz)
                            for (int xx = x; xx <= x1 && ok; ++xx) ok = match(xx, y1 + 1, zz);
                        if (!ok) break;
                        ++y1;
                    }
                    for (int yy = y; yy <= y1; ++yy)
                        for (int zz = z; zz <= z1; ++zz)
                            for (int xx = x; xx <= x1; ++xx) visited[Env::cell(xx, yy, zz)] = 1;
                    if (e.numBoxes < MAX_BOXES) {
                        Box &b = e.boxes[e.numBoxes++];
                        b.min[0] = x; b.min[1] = y; b.min[2] = z;
                        b.max[0] = x1 + 1; b.max[1] = y1 + 1; b.max[2] = z1 + 1;
                        b.type = type; b.slot = slot;
                    }
                }
    }
}

static bool in_building_zone(const Env &e, int x, int z)
{   // scenario_tower_building.cpp:227-230 (x/z interval only, any height)
    return x >= e.bz[0] && x < e.bz[1] && z >= e.bz[2] && z < e.bz[3];
}

static int triangular_number(int n) { return n * (n + 1) / 2; }   // util/math_utils.hpp:7-10 (pinned: tests/test_oracle_spec.py)

static float building_reward_coeff(float height)
{   // scenario_tower_building.cpp:246-251 ; powf(2,h) is exact for integral h
    float res = height * 0.05f;
    res += std::min(0.05f * ldexpf(1.0f, (int)height), 20.0f);
    return res;
}

static float tower_reward(const Env &e)
{   // scenario_tower_building.cpp:232-241.  The reference sums over an unordered_set; we fix the
    // order to object index.
    float r = 0.0f;
    for (int i = 0; i < e.numObjects; ++i) {
        const Object &o = e.objects[i];
        if (o.state == 0 && in_building_zone(e, o.x, o.z)) r += building_reward_coeff(float(o.y));
    }
    return r;
}

struct C3 { int x, y, z; };

// DefaultScenario::spawnAgents, scenario_default.hpp:80-97 ; agent ctor agent.cpp:24-65
static void spawn_agents(Env &e, const std::vector<C3> &spawns, const float *yaw = nullptr)   // yaw: HexMemory passes the angles (no draw)
{
    for (int i = 0; i < e.numAgents; ++i) {
        const C3 sp = spawns[i < int(spawns.size()) ? i : 0];
        Agent &a = e.agents[i];
        float keepShaping[MAX_SHAPING];
        std::memcpy(keepShaping, a.shaping, sizeof keepShaping);
        const float randomRotation = yaw ? yaw[i] : frand(e.rng) * PI_F * 2;
        float c, s;
        yaw_matrix(randomRotation, &c, &s);
        a.pos = v3(float(sp.x) + 0.5f, float(sp.y) + 0.0f + AGENT_HEIGHT, float(sp.z) + 0.5f);
        a.m00 = c; a.m02 = s; a.m20 = -s; a.m22 = c;
        a.pitch = 0;
        a.hvx = a.hvz = 0; a.vvel = 0; a.voffset = 0; a.step_offset = 0; a.jump_speed = 10.0f; a.was_jumping = 0;
        a.carrying = -1; a.picked_up = 0; a.visited_zone = 0;
        a.spawn[0] = sp.x; a.spawn[1] = sp.y; a.spawn[2] = sp.z;
        a.last_reward = 0; a.total_reward = 0; a.action = 0;
        std::memcpy(a.shaping, keepShaping, sizeof keepShaping);
    }
}

struct F3 { float x, y, z; };
// same as spawn_agents for starting positions that are not voxel corners (Sokoban)
static void spawn_agents_at(Env &e, const std::vector<F3> &positions, const float *yaw = nullptr)
{
    std::vector<C3> cells;
    for (const F3 &p : positions) cells.push_back(C3{int(floorf(p.x)), int(floorf(p.y)), int(floorf(p.z))});
    spawn_agents(e, cells, yaw);
    for (int i = 0; i < e.numAgents; ++i) {
        const F3 p = positions[i < int(positions.size()) ? i : 0];
        e.agents[i].pos = v3(p.x + 0.5f, p.y + 0.0f + AGENT_HEIGHT, p.z + 0.5f);
    }
}

static void obstacles_generate(Env &e);

static void tower_generate(Env &e)
{
    // ---- TowerBuildingScenario::reset, scenario_tower_building.cpp:129-154
    std::fill(e.chunk.begin(), e.chunk.end(), 0);
    unsigned layoutColor = randomLayoutColor(e.rng);
    while (layoutColor == COLOR_BUILDING_ZONE) layoutColor = randomLayoutColor(e.rng);

    // ---- TowerBuildingPlatform::init, :19-76
    Rng &rng = e.rng;
    const int A = e.numAgents;
    int height = randRange(5, 7, rng);
    int length = randRange(12, 30, rng);
    int width = randRange(12, 25, rng);
    const int bzL = randRange(3, 9, rng), bzW = randRange(3, 9, rng);
    const int matL = randRange(2, 8, rng), matW = randRange(2, 8, rng);
    length = std::max(bzL + matL + 3, length);
    width = std::max(bzW + matW + 3, width);
    const int bzX = randRange(1, length - bzL - 1, rng), bzZ = randRange(1, width - bzW - 1, rng);
    const int matX = randRange(1, length - matL - 1, rng), matZ = randRange(1, width - matW - 1, rng);

    std::vector<C3> cand;
    for (int x = 1; x < length - 1; ++x)
        for (int z = 1; z < width - 1; ++z) cand.push_back(C3{x, 2, z});
    std::shuffle(cand.begin(), cand.end(), rng);

    const int nAgentSpawns = std::min(A, int(cand.size()));
    const int maxRandomObjects = std::min(int(cand.size()) - A, 25);
    const int spawnObjects = randRange(0, std::max(1, maxRandomObjects), rng);
    std::vector<C3> objs(cand.begin() + nAgentSpawns, cand.begin() + nAgentSpawns + spawnObjects);
    for (auto &c : objs) {
        if (c.x >= matX && c.x < matX + matL && c.z >= matZ && c.z < matZ + matW) continue;
        c.y -= 1;
    }
    for (int x = matX; x < matX + matL; ++x)
        for (int z = matZ; z < matZ + matW; ++z) objs.push_back(C3{x, 1, z});

    e.L = length; e.H = height; e.W = width;
    e.bz[0] = bzX; e.bz[1] = bzX + bzL; e.bz[2] = bzZ; e.bz[3] = bzZ + bzW;

    // ---- vg.addPlatform(platform, layoutColor, randomLayoutColor(rng), randomBool(rng)), :145.
    // GCC evaluates the arguments right to left: randomBool first.
    const bool drawWalls = randomBool(rng);
    const unsigned wallColor = randomLayoutColor(rng);
    e.layoutColor = layoutColor; e.wallColor = wallColor; e.drawWalls = drawWalls;

    // floor (platforms.hpp:167-176) then walls S,N,E,W (:178-190); later fills override earlier
    // ones exactly like grid.set() does (component_voxel_grid.hpp:73-90).
    const uint8_t vFloor = VX_SOLID | VX_OPAQUE | (0 << VX_COLOR_SHIFT);
    const uint8_t vWall = VX_SOLID | (drawWalls ? VX_OPAQUE : 0) | (1 << VX_COLOR_SHIFT);
    fill_box(e, 0, 0, 0, length, 1, width, vFloor);
    fill_box(e, 0, 0, 0, 1, height, width, vWall);
    fill_box(e, length - 1, 0, 0, length, height, width, vWall);
    fill_box(e, 0, 0, 0, length, height, 1, vWall);
    fill_box(e, 0, 0, width - 1, length, height, width, vWall);
    // building zone terrain box has min.y == max.y == 1 -> writes no voxels (:83-88 + SURVEY A.4)

    merge_boxes(e);

    // ---- addEpisodeDrawables :161-177 + ObjectStackingComponent::addDrawablesAndCollisions
    e.numObjects = std::min(int(objs.size()), int(MAX_OBJECTS));
    for (int i = 0; i < e.numObjects; ++i) {
        e.objects[i] = Object{objs[i].x, objs[i].y, objs[i].z, 0};
        if (Env::inChunk(objs[i].x, objs[i].y, objs[i].z)) e.chunk[Env::cell(objs[i].x, objs[i].y, objs[i].z)] |= VX_OBJECT;
    }
    e.highestTower = 0;
    e.bzReward = tower_reward(e);
    // episodeLengthSec :263-266
    e.episodeLen = e.p_episodeLengthSec + 4.0f * float(e.numObjects);
    e.barHalfWidth = 0.24f;

    e.numTerrain = 0; e.numRewards = 0; e.numPlatforms = 0; e.solved = 0;
    spawn_agents(e, std::vector<C3>(cand.begin(), cand.begin() + nAgentSpawns));
}


// ------------------------------------------------------------------------------------------------
// Obstacles family -- scenario_obstacles.cpp:51-195 + platforms.hpp:137-557.
// The reference places platforms with a Magnum scene graph (90-degree rotations about Y and integer
// translations) and reads boxes back with lround(); restated with exact integer rigid transforms.
// ------------------------------------------------------------------------------------------------
struct Xf {   // p -> R^rot(p) + t,  R = +90 deg about Y: (x, y, z) -> (z, y, -x)
    int rot = 0;
    int t[3] = {0, 0, 0};
};
static void xf_rot(int rot, const int p[3], int out[3])
{
    int x = p[0], y = p[1], z = p[2];
    for (int i = 0; i < (rot & 3); ++i) { const int nx = z, nz = -x; x = nx; z = nz; }
    out[0] = x; out[1] = y; out[2] = z;
}
static void xf_apply(const Xf &a, const int p[3], int out[3])
{
    int r[3];
    xf_rot(a.rot, p, r);
    for (int k = 0; k < 3; ++k) out[k] = r[k] + a.t[k];
}
static Xf xf_compose(const Xf &a, const Xf &b)   // p -> a(b(p))
{
    Xf c;
    c.rot = (a.rot + b.rot) & 3;
    xf_apply(a, b.t, c.t);
    return c;
}
// Object3D::rotateYLocal(deg) followed by translateLocal(v): p -> R(p + v)
static Xf xf_local(int rot, int vx, int vy, int vz)
{
    Xf c;
    c.rot = rot & 3;
    const int v[3] = {vx, vy, vz};
    xf_rot(c.rot, v, c.t);
    return c;
}

struct BoxI { int min[3], max[3]; };
static BoxI box_abs(const Xf &root, const BoxI &local)
{   // MagnumAABB::boundingBox, platforms.hpp:126-133: transform min/max, then sort
    BoxI b;
    xf_apply(root, local.min, b.min);
    xf_apply(root, local.max, b.max);
    for (int k = 0; k < 3; ++k) if (b.min[k] > b.max[k]) std::swap(b.min[k], b.max[k]);
    return b;
}
enum { WALLS_SOUTH = 1, WALLS_NORTH = 2, WALLS_WEST = 4, WALLS_EAST = 8 };

struct Plat {
    int kind = PT_EMPTY, walls = 0, length = 0, height = 0, width = -1;
    Xf parent, local, root;          // root = parent o local
    int anchorY = 0;                 // nextPlatformAnchor = root o translate(length, anchorY, 0)
    std::vector<BoxI> layout, wallsB;            // root-local boxes
    std::vector<std::pair<int, BoxI>> terrain;   // (type, root-local box)
    std::map<std::pair<int, int>, int> occupancy;
    int wallHeight = 0, lavaLength = 0, stepHeight = 0, gap = 0, gapX = 0;

    Xf anchor() const { return xf_compose(root, xf_local(0, length, anchorY, 0)); }
    void updateRoot() { root = xf_compose(parent, local); }
    void addFloor() { layout.push_back(BoxI{{0, 0, 0}, {length, 1, width}}); }   // platforms.hpp:167-176
    void addWalls()
    {   // :178-190
        if (walls & WALLS_SOUTH) wallsB.push_back(BoxI{{0, 0, 0}, {1, height, width}});
        if (walls & WALLS_NORTH) wallsB.push_back(BoxI{{length - 1, 0, 0}, {length, height, width}});
        if (walls & WALLS_EAST) wallsB.push_back(BoxI{{0, 0, 0}, {length, height, 1}});
        if (walls & WALLS_WEST) wallsB.push_back(BoxI{{0, 0, width - 1}, {length, height, width}});
    }
    BoxI outerBox() const
    {   // platformBoundingBox :192-214
        BoxI o{{0, 0, 0}, {0, 0, 0}};
        if (!layout.empty()) o = box_abs(root, layout.front());
        else if (!wallsB.empty()) o = box_abs(root, wallsB.front());
        auto add = [&](const BoxI &b) {
            for (int k = 0; k < 3; ++k) { o.min[k] = std::min(o.min[k], std::min(b.min[k], b.max[k])); o.max[k] = std::max(o.max[k], std::max(b.min[k], b.max[k])); }
        };
        for (auto &b : layout) add(box_abs(root, b));
        for (auto &b : wallsB) add(box_abs(root, b));
        return o;
    }
    bool isMaxDifficulty(const ObstacleParams &op) const
    {
        if (kind == PT_WALL) return wallHeight >= op.maxHeight;
        if (kind == PT_LAVA) return lavaLength >= op.maxLava;
        if (kind == PT_STEP) return stepHeight >= op.maxHeight;
        return false;
    }
    int requiresBoxes() const
    {
        auto tri = [](int n) { return triangular_number(n); };
        if (kind == PT_WALL) return tri(wallHeight - 1);
        if (kind == PT_LAVA) return std::max(1, lavaLength - 1);
        if (kind == PT_STEP) return tri(stepHeight - 1);
        if (kind == PT_GAP) return tri(std::max(0, gap - 2));
        return 0;
    }
    void init(Rng &rng, const ObstacleParams &op)
    {
        if (kind == PT_TRANSITION) { height = 5; return; }   // :556
        length = randRange(4, 10, rng);                       // EmptyPlatform::init :314-321
        if (width == -1) width = randRange(5, 9, rng);
        height = 5;
        if (kind == PT_WALL) {
            wallHeight = randRange(op.minHeight, op.maxHeight + 1, rng);
            height = randRange(wallHeight + 4, wallHeight + 6, rng);
        } else if (kind == PT_LAVA) {
            length = randRange(6, 12, rng);
            const int minLava = std::min(op.minLava, length - 2), maxLava = std::min(op.maxLava + 1, length - 1);
            lavaLength = randRange(minLava, maxLava, rng);
        } else if (kind == PT_STEP) {
            stepHeight = randRange(op.minHeight, op.maxHeight + 1, rng);
            height = randRange(stepHeight + 2, stepHeight + 5, rng);
        } else if (kind == PT_GAP) {
            gap = randRange(op.minGap, std::min(op.maxGap + 1, length - 1), rng);
            gapX = randRange(1, length - gap, rng);
        }
    }
    void generate(Rng &rng)
    {
        if (kind == PT_STEP) {   // :440-461
            const int stepX = randRange(1, length, rng);
            layout.push_back(BoxI{{0, 0, 0}, {stepX + 1, 1, width}});
            layout.push_back(BoxI{{stepX, stepHeight, 0}, {length, stepHeight + 1, width}});
            layout.push_back(BoxI{{stepX, 0, 0}, {stepX + 1, stepHeight + 1, width}});
            anchorY = stepHeight;
            addWalls();
            for (int x = stepX + 1; x < length; ++x)
                for (int z = 1; z < width; ++z) occupancy[{x, z}] = stepHeight;
            return;
        }
        if (kind == PT_GAP) {   // :490-501
            layout.push_back(BoxI{{0, 0, 0}, {gapX, 1, width}});
            layout.push_back(BoxI{{gapX + gap, 0, 0}, {length, 1, width}});
            addWalls();
            return;
        }
        addFloor();
        addWalls();
        if (kind == PT_WALL) {   // :352-366
            const int wallX = randRange(1, length, rng);
            const int wallThickness = randRange(1, length - wallX + 1, rng);
            layout.push_back(BoxI{{wallX, 1, 1}, {wallX + wallThickness, 1 + wallHeight, width - 1}});
            for (int x = wallX; x < wallX + wallThickness; ++x)
                for (int z = 1; z < width; ++z) occupancy[{x, z}] = wallHeight;
        } else if (kind == PT_LAVA) {   // :401-409
            const int lavaX = randRange(1, length - lavaLength, rng);
            terrain.push_back({TERRAIN_LAVA, BoxI{{lavaX, 1, 1}, {lavaX + lavaLength, 2, width - 1}}});
        } else if (kind == PT_EXIT) {   // :535-541
            terrain.push_back({TERRAIN_EXIT, BoxI{{length - 3, 1, 1}, {length - 1, 3, width - 1}}});
        }
    }
    C3 adjust(int x, int y, int z) const
    {   // adjustTransformation :280-288: voxel centre through the root transform, then floor
        const int rx = (root.rot & 3);
        // centre (x+.5, y+.5, z+.5) -> R^rot -> + t -> floor, done in doubled integer coordinates
        int p2[3] = {2 * x + 1, 2 * y + 1, 2 * z + 1}, r2[3];
        xf_rot(rx, p2, r2);
        auto fl = [](int twice) { return (twice >= 0) ? twice / 2 : -((-twice + 1) / 2); };
        return C3{fl(r2[0] + 2 * root.t[0]), fl(r2[1] + 2 * root.t[1]), fl(r2[2] + 2 * root.t[2])};
    }
    std::vector<C3> objectPositions(int n, Rng &rng)
    {
        std::vector<C3> out;
        if (kind == PT_GAP) {   // :505-523
            std::vector<C3> cand;
            for (int x = 0; x < length; ++x)
                for (int z = 1; z < width - 1; ++z) {
                    if (x >= gapX && x < gapX + gap) continue;
                    cand.push_back(C3{x, 1, z});
                }
            for (int i = 0; i < n; ++i) {
                const C3 v = cand[randRange(0, int(cand.size()), rng)];
                const int y = ++occupancy[{v.x, v.z}];
                out.push_back(adjust(v.x, y, v.z));
            }
            return out;
        }
        for (int i = 0; i < n; ++i)   // Platform::generateObjectPositions :261-278
            for (int attempt = 0; attempt < 10; ++attempt) {
                const int x = randRange(1, length - 1, rng);
                const int z = randRange(1, width - 1, rng);
                if (occupancy[{x, z}] < 2 || attempt >= 9) {
                    const int y = ++occupancy[{x, z}];
                    out.push_back(adjust(x, y, z));
                    break;
                }
            }
        return out;
    }
};

static bool boxes_collide(const BoxI &a, const BoxI &b)
{   // BoundingBox::collidesWith :93-104
    for (int k = 0; k < 3; ++k) if (a.max[k] <= b.min[k] || a.min[k] >= b.max[k]) return false;
    return true;
}

// Canonical greedy merge on a dense grid spanning the level's bounding box (same rule as merge_boxes).
static void merge_dense(const std::vector<uint8_t> &g, const int org[3], const int dim[3], Env &e)
{
    auto cell = [&](int x, int y, int z) { return (size_t(y) * dim[2] + z) * dim[0] + x; };
    std::vector<uint8_t> visited(g.size(), 0);
    e.numBoxes = 0;
    for (int key = 0; key < 16; ++key) {
        const int type = key >> 2, slot = key & 3;
        if (type == 0) continue;
        auto match = [&](int x, int y, int z) {
            if (x < 0 || y < 0 || z < 0 || x >= dim[0] || y >= dim[1] || z >= dim[2]) return false;
            const size_t c = cell(x, y, z);
            return !visited[c] && (g[c] & 3) == type && (g[c] >> VX_COLOR_SHIFT) == slot;
        };
        for (int y = 0; y < dim[1]; ++y)
            for (int z = 0; z < dim[2]; ++z)
                for (int x = 0; x < dim[0]; ++x) {
                    if (!match(x, y, z)) continue;
                    int x1 = x;
                    while (match(x1 + 1, y, z)) ++x1;
                    int z1 = z;
                    for (;;) { bool ok = true; for (int xx = x; xx <= x1 && ok; ++xx) ok = match(xx, y, z1 + 1); if (!ok) break; ++z1; }
                    int y1 = y;
                    for (;;) {
                        bool ok = true;
                        for (int zz = z; zz <= z1 && ok; ++zz) for (int xx = x; xx <= x1 && ok; ++xx) ok = match(xx, y1 + 1, zz);
                        if (!ok) break;
                        ++y1;
                    }
                    for (int yy = y; yy <= y1; ++yy) for (int zz = z; zz <= z1; ++zz) for (int xx = x; xx <= x1; ++xx) visited[cell(xx, yy, zz)] = 1;
                    if (e.numBoxes < MAX_BOXES) {
                        Box &b = e.boxes[e.numBoxes++];
                        b.min[0] = x + org[0]; b.min[1] = y + org[1]; b.min[2] = z + org[2];
                        b.max[0] = x1 + 1 + org[0]; b.max[1] = y1 + 1 + org[1]; b.max[2] = z1 + 1 + org[2];
                        b.type = type; b.slot = slot;
                    }
                }
    }
}

static void obstacles_generate(Env &e)
{
    Rng &rng = e.rng;
    const ObstacleParams &op = e.op;
    const bool drawWalls = randRange(0, 2, rng);   // scenario_obstacles.cpp:63
    std::vector<Plat> platforms;
    int numPlatforms = 0;
    for (int attempt = 0; attempt < 20; ++attempt) {   // :68-156
        platforms.clear();
        numPlatforms = randRange(op.minPlatforms, op.maxPlatforms + 1, rng);
        Plat start; start.kind = PT_START; start.walls = WALLS_SOUTH | WALLS_EAST | WALLS_WEST;
        start.init(rng, op); start.updateRoot(); start.generate(rng);
        int requiredWidth = start.width;
        platforms.push_back(start);
        size_t prev = 0;
        int numMax = 0;
        for (int i = 0; i < numPlatforms; ++i) {
            const int orientation = randRange(0, 3, rng);   // randomSample({STRAIGHT, LEFT, RIGHT})
            requiredWidth = orientation == 0 ? requiredWidth : -1;
            Plat np;
            bool have = false;
            while (!have || (np.isMaxDifficulty(op) && numMax >= op.numAllowedMaxDifficulty)) {
                np = Plat();
                np.kind = op.platformTypes[randRange(0, int(op.platformTypes.size()), rng)];
                np.walls = WALLS_WEST | WALLS_EAST; np.width = requiredWidth;
                np.init(rng, op);
                have = true;
            }
            if (np.isMaxDifficulty(op)) ++numMax;
            np.parent = platforms[prev].anchor();
            np.updateRoot();
            np.generate(rng);
            if (orientation == 1) np.local = xf_local(1, -1, 0, -1);                                   // rotateCCW :149-153
            else if (orientation == 2) np.local = xf_local(3, platforms[prev].width - 1, 0, -np.width + 1);   // rotateCW :155-159
            np.updateRoot();
            platforms.push_back(np);
            const size_t cur = platforms.size() - 1;
            if (orientation != 0) {   // :128-138
                Plat tr; tr.kind = PT_TRANSITION;
                tr.walls = WALLS_NORTH | (orientation == 1 ? WALLS_WEST : WALLS_EAST);
                tr.length = platforms[cur].width - 1; tr.width = platforms[prev].width;
                tr.parent = platforms[prev].anchor(); tr.updateRoot();
                tr.init(rng, op); tr.generate(rng);
                platforms.push_back(tr);
            }
            prev = cur;
            requiredWidth = platforms[cur].width;
        }
        Plat ex; ex.kind = PT_EXIT; ex.walls = WALLS_NORTH | WALLS_EAST | WALLS_WEST; ex.width = requiredWidth;
        ex.parent = platforms[prev].anchor();
        ex.init(rng, op); ex.updateRoot(); ex.generate(rng);
        platforms.push_back(ex);

        bool selfCollision = false;   // :145-153
        for (int j = 0; j < int(platforms.size()) && !selfCollision; ++j)
            for (int k = 0; k < j - 2; ++k)
                if (boxes_collide(platforms[j].outerBox(), platforms[k].outerBox())) { selfCollision = true; break; }
        if (!selfCollision) break;
    }
    const unsigned layoutColor = randomLayoutColor(rng), wallColor = randomLayoutColor(rng);   // :158-159
    e.layoutColor = layoutColor; e.wallColor = wallColor; e.drawWalls = drawWalls; e.numPlatforms = numPlatforms;

    // ---- voxelise (vg.addPlatform per platform, component_voxel_grid.hpp:73-84) over the level's bounding box
    int lo[3] = {1 << 20, 1 << 20, 1 << 20}, hi[3] = {-(1 << 20), -(1 << 20), -(1 << 20)};
    for (auto &p : platforms) {
        auto grow = [&](const BoxI &b) { for (int k = 0; k < 3; ++k) { lo[k] = std::min(lo[k], b.min[k]); hi[k] = std::max(hi[k], b.max[k]); } };
        for (auto &b : p.layout) grow(box_abs(p.root, b));
        for (auto &b : p.wallsB) grow(box_abs(p.root, b));
    }
    const int dim[3] = {hi[0] - lo[0], hi[1] - lo[1], hi[2] - lo[2]};
    std::vector<uint8_t> grid(size_t(dim[0]) * dim[1] * dim[2], 0);
    auto fill = [&](const BoxI &b, uint8_t v) {
        for (int x = b.min[0]; x < b.max[0]; ++x) for (int y = b.min[1]; y < b.max[1]; ++y) for (int z = b.min[2]; z < b.max[2]; ++z)
            grid[(size_t(y - lo[1]) * dim[2] + (z - lo[2])) * dim[0] + (x - lo[0])] = v;
    };
    e.numTerrain = 0;
    for (auto &p : platforms) {
        for (auto &b : p.layout) fill(box_abs(p.root, b), VX_SOLID | VX_OPAQUE);
        for (auto &b : p.wallsB) fill(box_abs(p.root, b), VX_SOLID | (drawWalls ? VX_OPAQUE : 0) | (1 << VX_COLOR_SHIFT));
    }
    for (auto &p : platforms)   // addEpisodeDrawables :244-248 order: platforms, then map<TerrainType> order (EXIT < LAVA)
        for (int type : {TERRAIN_EXIT, TERRAIN_LAVA})
            for (auto &tb : p.terrain)
                if (tb.first == type && e.numTerrain < MAX_TERRAIN) {
                    const BoxI b = box_abs(p.root, tb.second);
                    TerrainBox &t = e.terrain[e.numTerrain++];
                    for (int k = 0; k < 3; ++k) { t.min[k] = b.min[k]; t.max[k] = b.max[k]; }
                    t.type = type;
                }
    merge_dense(grid, lo, dim, e);
    std::fill(e.chunk.begin(), e.chunk.end(), 0);   // the 32x16x32 chunk is a TowerBuilding structure; unused here
    e.L = dim[0]; e.H = dim[1]; e.W = dim[2];
    e.bz[0] = lo[0]; e.bz[1] = lo[1]; e.bz[2] = lo[2]; e.bz[3] = 0;   // level origin (informational)

    // ---- agent spawn points on the start platform (Platform::agentSpawnPoints :221-243)
    std::vector<C3> spawns;
    {
        Plat &sp = platforms[0];
        std::set<std::pair<int, int>> used;
        for (int i = 0; i < e.numAgents; ++i)
            for (int attempt = 0; attempt < 10; ++attempt) {
                const int x = randRange(1, sp.length - 1, rng), z = randRange(1, sp.width - 1, rng);
                if (used.count({x, z})) continue;
                const int y = sp.occupancy[{x, z}] + 1;
                sp.occupancy[{x, z}] += 2;
                spawns.push_back(C3{x, y, z});
                used.insert({x, z});
                break;
            }
        if (spawns.empty()) spawns.push_back(C3{1, 1, 1});
    }

    // ---- movable boxes (:166-188) and reward objects (:190-194)
    std::vector<int> numBoxes(platforms.size(), 0);
    for (int i = 1; i < int(platforms.size()); ++i) {
        const int n = platforms[i].requiresBoxes();
        for (int b = 0; b < n; ++b) ++numBoxes[randRange(std::max(0, i - 2), i, rng)];
    }
    std::vector<C3> objs, rews;
    for (int i = 0; i < int(platforms.size()); ++i) {
        const float randomBoxesFraction = frand(rng) * 0.5f;
        const int randomBoxes = int(lroundf(randomBoxesFraction * float(numBoxes[i]))) + randRange(0, 2, rng);
        const auto c = platforms[i].objectPositions(numBoxes[i] + randomBoxes, rng);
        objs.insert(objs.end(), c.begin(), c.end());
    }
    for (int i = 1; i < int(platforms.size()) - 1; ++i) {
        const int n = randRange(0, 2, rng);
        const auto c = platforms[i].objectPositions(n, rng);
        rews.insert(rews.end(), c.begin(), c.end());
    }
    e.numObjects = std::min(int(objs.size()), int(MAX_OBJECTS));
    for (int i = 0; i < e.numObjects; ++i) e.objects[i] = Object{objs[i].x, objs[i].y, objs[i].z, 0};
    e.numRewards = std::min(int(rews.size()), int(MAX_REWARDS));
    for (int i = 0; i < e.numRewards; ++i) e.rewards[i] = RewardObj{rews[i].x, rews[i].y, rews[i].z, 1};
    e.solved = 0; e.highestTower = 0; e.bzReward = 0;
    // episodeLengthSec :262-266 counts every generated position, clipped or not
    e.episodeLen = std::max(e.p_episodeLengthSec, float(numPlatforms) * 35 + float(objs.size()) * 1);
    e.barHalfWidth = 0.24f;
    spawn_agents(e, spawns);
}


// ------------------------------------------------------------------------------------------------
// Collect: Perlin heightfield landscape.  siv::PerlinNoise<double> (vendored by the reference as
// util/include/util/perlin_noise.hpp; restated here, pinned against that header by oracle/_ref):
// reseed :118-126, noise3D :171-197, Fade/Lerp/Grad :59-79, accumulatedOctaveNoise2D :244-259,
// accumulatedOctaveNoise2D_0_1 :315-318.
// ------------------------------------------------------------------------------------------------
struct Perlin {
    uint8_t p[512];
    explicit Perlin(uint32_t seed)
    {
        for (int i = 0; i < 256; ++i) p[i] = uint8_t(i);
        std::shuffle(p, p + 256, std::default_random_engine(seed));
        for (int i = 0; i < 256; ++i) p[256 + i] = p[i];
    }
    static double fade(double t) { return t * t * t * (t * (t * 6 - 15) + 10); }
    static double mix(double t, double a, double b) { return a + t * (b - a); }
    static double grad(uint8_t hash, double x, double y, double z)
    {
        const int h = hash & 15;
        const double u = h < 8 ? x : y;
        const double v = h < 4 ? y : (h == 12 || h == 14 ? x : z);
        return ((h & 1) == 0 ? u : -u) + ((h & 2) == 0 ? v : -v);
    }
    double noise3(double x, double y, double z) const
    {
        const int X = int(std::floor(x)) & 255, Y = int(std::floor(y)) & 255, Z = int(std::floor(z)) & 255;
        x -= std::floor(x); y -= std::floor(y); z -= std::floor(z);
        const double u = fade(x), v = fade(y), w = fade(z);
        const int A = p[X] + Y, AA = p[A] + Z, AB = p[A + 1] + Z;
        const int B = p[X + 1] + Y, BA = p[B] + Z, BB = p[B + 1] + Z;
        const double x1 = x - 1, y1 = y - 1, z1 = z - 1;
        const double front = mix(v, mix(u, grad(p[AA], x, y, z), grad(p[BA], x1, y, z)),
                                    mix(u, grad(p[AB], x, y1, z), grad(p[BB], x1, y1, z)));
        const double back = mix(v, mix(u, grad(p[AA + 1], x, y, z1), grad(p[BA + 1], x1, y, z1)),
                                   mix(u, grad(p[AB + 1], x, y1, z1), grad(p[BB + 1], x1, y1, z1)));
        return mix(w, front, back);
    }
    double octaves2_01(double x, double y, int octaves) const
    {
        double result = 0, amp = 1;
        for (int i = 0; i < octaves; ++i) {
            result += noise3(x, y, 0) * amp;
            x *= 2; y *= 2; amp /= 2;
        }
        return std::clamp<double>(result * 0.5 + 0.5, 0, 1);
    }
};

// CollectScenario::reset / createLandscape (scenario_collect.cpp:20-161), spawnAgents, then the reward draws of
// addEpisodeDrawables (:190-214) -- in the order Env::reset makes them (env.cpp:69-75).
static void collect_generate(Env &e)
{
    Rng &rng = e.rng;
    static const unsigned landscapeColors[7] = {0xffffff, 0xccffcc, 0xe6ecff, 0xd9d9d9, 0xffebcc, 0xb3b3b3, 0x555555};
    static const unsigned floorColors[3] = {0xb3b3b3, 0x555555, 0x555555};
    const unsigned landscapeColor = landscapeColors[randRange(0, 7, rng)];
    const unsigned floorColor = floorColors[randRange(0, 3, rng)];
    const int maxWidth = HM_DIM, maxLength = HM_DIM;
    const int width = randRange(8, maxWidth, rng);     // z extent
    const int length = randRange(8, maxLength, rng);   // x extent
    std::vector<int> spawnHeight(size_t(length) * width, 1);
    const double frequency = double(randRange(1, 100, rng)) / 10.0;
    const int octaves = randRange(1, 10, rng);
    const uint32_t seed = uint32_t(randRange(0, 1000000000, rng));
    const Perlin perlin(seed);
    const double fx = maxLength / frequency, fz = maxWidth / frequency;
    const int intensity = randRange(5, 18, rng);
    const float groundLevel = frand(rng) * 0.5f + 0.2f;

    std::fill(e.heightmap.begin(), e.heightmap.end(), int8_t(-1));
    for (int x = 0; x < length; ++x)
        for (int z = 0; z < width; ++z) e.heightmap[x * HM_DIM + z] = 0;   // floor :100-103
    int top = 0;
    for (int x = 1; x < length - 1; ++x)
        for (int z = 1; z < width - 1; ++z) {
            const double noise = perlin.octaves2_01(x / fx, z / fz, octaves);
            const double yCoord = intensity * (noise - groundLevel);
            if (yCoord >= 1) {
                const int r = int(lround(yCoord));
                e.heightmap[x * HM_DIM + z] = int8_t(r);
                spawnHeight[size_t(x) * width + z] = r + 1;
                top = std::max(top, r);
            }
        }

    // canonical merge.  toBoundingBoxes groups by (voxelType, colour) in std::map order (component_voxel_grid.hpp:33-47):
    // both classes are SOLID|OPAQUE, so the lower colour value comes first; equal colours are ONE class.
    const unsigned c0 = std::min(landscapeColor, floorColor), c1 = std::max(landscapeColor, floorColor);
    const int landSlot = landscapeColor == c0 ? 0 : 1, floorSlot = floorColor == c0 ? 0 : 1;
    {
        const int org[3] = {0, 0, 0}, dim[3] = {length, top + 1, width};
        std::vector<uint8_t> g(size_t(dim[0]) * dim[1] * dim[2], 0);
        for (int x = 0; x < length; ++x)
            for (int z = 0; z < width; ++z) {
                g[(size_t(0) * dim[2] + z) * dim[0] + x] = uint8_t(3 | (floorSlot << VX_COLOR_SHIFT));
                for (int y = 1; y <= e.heightmap[x * HM_DIM + z]; ++y) g[(size_t(y) * dim[2] + z) * dim[0] + x] = uint8_t(3 | (landSlot << VX_COLOR_SHIFT));
            }
        merge_dense(g, org, dim, e);
    }
    e.layoutColor = c0; e.wallColor = c1; e.drawWalls = 1;
    e.L = length; e.H = top + 1; e.W = width;
    e.bz[0] = e.bz[1] = e.bz[2] = e.bz[3] = 0;
    e.numTerrain = 0;
    std::fill(e.chunk.begin(), e.chunk.end(), 0);

    std::vector<C3> sp;
    for (int x = 1; x < length - 1; ++x)
        for (int z = 1; z < width - 1; ++z) sp.push_back(C3{x, spawnHeight[size_t(x) * width + z], z});
    std::shuffle(sp.begin(), sp.end(), rng);
    int offset = 0;
    const std::vector<C3> agentCells(sp.begin(), sp.begin() + e.numAgents);
    offset += e.numAgents;
    int numRewards = randRange(1, int(lround(0.05 * width * length)) + 2, rng);
    numRewards = std::min(numRewards, int(sp.size()) - offset);
    const int placedRandomly = std::max(numRewards / 2, 1);
    std::vector<C3> rews(sp.begin() + offset, sp.begin() + offset + placedRandomly);
    offset += placedRandomly;
    std::sort(sp.begin() + offset, sp.end(), [&](const C3 &a, const C3 &b) {   // highest cells first (:135-142)
        const int ha = spawnHeight[size_t(a.x) * width + a.z], hb = spawnHeight[size_t(b.x) * width + b.z];
        return ha != hb ? ha > hb : false;
    });
    rews.insert(rews.end(), sp.begin() + offset, sp.begin() + offset + (numRewards - placedRandomly));
    offset += numRewards - placedRandomly;
    std::shuffle(sp.begin() + offset, sp.end(), rng);
    const int objectsMin = std::max(3, int(length * width * 0.04));
    const int objectsMax = std::min(objectsMin + 1, int(lround(0.07 * width * length)) + 2);
    const int numObjects = std::min(randRange(objectsMin, objectsMax, rng), int(sp.size()) - offset);
    std::vector<C3> objs;
    // (:153-156) when the condition fails the reference keeps the PREVIOUS episode's objectPositions; with
    // (L-2)(W-2) >= 36 cells, <= 8 agents and these count formulas it always holds, so that path is not modelled.
    if (offset + numObjects < int(sp.size())) objs.assign(sp.begin() + offset, sp.begin() + offset + numObjects);

    e.numObjects = std::min(int(objs.size()), int(MAX_OBJECTS));
    for (int i = 0; i < e.numObjects; ++i) e.objects[i] = Object{objs[i].x, objs[i].y, objs[i].z, 0};
    e.solved = 0; e.highestTower = 0; e.bzReward = 0; e.numPlatforms = 0;
    e.episodeLen = e.p_episodeLengthSec + 2.0f * float(rews.size());   // scenario_collect.hpp:55-59
    e.barHalfWidth = 0.24f;
    spawn_agents(e, agentCells);
    e.numRewards = std::min(int(rews.size()), int(MAX_REWARDS));
    for (int i = 0; i < int(rews.size()); ++i) {   // one frand per reward, drawn after the agents' rotations
        const bool good = frand(rng) > 0.3f;
        if (i < e.numRewards) e.rewards[i] = RewardObj{rews[i].x, rews[i].y, rews[i].z, good ? 1 : 2};
        if (good) ++e.numPlatforms;   // numPositiveRewards
    }
}

// ------------------------------------------------------------------------------------------------
// Rearrange -- scenario_rearrange.{hpp,cpp}.  A fixed 19 x 14 room with a raised floor and two pedestals: the target
// arrangement (2..7 items: cylinder / capsule / box / sphere in random colours, offsets within +-1, stacked up to
// two high) stands on the left one, the same items -- partly displaced -- on the right one have to be rebuilt.
// ------------------------------------------------------------------------------------------------
static const int RE_LEFT[3] = {5, 2, 5}, RE_RIGHT[3] = {13, 2, 5};   // scenario_rearrange.hpp:130-131
static const unsigned OBJECT_COLORS[14] = {0xffdd3c, 0x3bb372, 0x50c878, 0x2eb5d0, 0xadd8e6, 0x3a7fa6, 0xffb400,
                                           0xb3b3b3, 0x555555, 0xffffff, 0xff0000, 0xffa770, 0xd468ee, 0xffe6e6};   // const.hpp:96-111

static Env::ArrItem random_item(Rng &rng, int ox, int oy, int oz)
{   // ArrangementItem::random, scenario_rearrange.hpp:31-44
    static const int shapes[4] = {SHAPE_CYLINDER, SHAPE_CAPSULE, SHAPE_BOX, SHAPE_SPHERE};
    Env::ArrItem it;
    it.shape = shapes[randRange(0, 4, rng)];
    it.color = OBJECT_COLORS[randRange(0, 14, rng)];
    it.off[0] = ox; it.off[1] = oy; it.off[2] = oz;
    return it;
}

static int rearrange_matching(const Env &e)
{   // countMatchingObjects :136-151: movable items that stand (not carried) where the target has the same shape + colour
    int matching = 0;
    for (int i = 0; i < e.numItems; ++i) {
        const Object &o = e.objects[i];
        if (o.state > 0) continue;
        const int off[3] = {o.x - RE_RIGHT[0], o.y - RE_RIGHT[1], o.z - RE_RIGHT[2]};
        for (int k = 0; k < e.numItems; ++k) {
            const Env::ArrItem &t = e.items[k];
            if (t.shape == e.items[i].shape && t.color == e.items[i].color && t.off[0] == off[0] && t.off[1] == off[1] && t.off[2] == off[2]) { ++matching; break; }
        }
    }
    return matching;
}

static void rearrange_generate(Env &e)
{
    Rng &rng = e.rng;
    std::fill(e.chunk.begin(), e.chunk.end(), 0);
    // RearrangePlatform::init :22-27 ; vg.addPlatform(*platform, DARK_GREY, DARK_GREY, randomBool(rng)) :61
    const int height = randRange(4, 7, rng), length = 19, width = 14;
    const bool drawWalls = randomBool(rng);
    e.L = length; e.H = height; e.W = width;
    e.bz[0] = e.bz[1] = e.bz[2] = e.bz[3] = 0;
    e.layoutColor = 0x555555; e.wallColor = 0x555555; e.drawWalls = drawWalls;
    // floor and walls share the colour: when the walls are drawn they are ONE (type, colour) class and merge together
    const uint8_t vFloor = VX_SOLID | VX_OPAQUE, vWall = uint8_t(VX_SOLID | (drawWalls ? VX_OPAQUE : 0));
    fill_box(e, 0, 0, 0, length, 1, width, vFloor);
    fill_box(e, 0, 0, 0, 1, height, width, vWall);
    fill_box(e, length - 1, 0, 0, length, height, width, vWall);
    fill_box(e, 0, 0, 0, length, height, 1, vWall);
    fill_box(e, 0, 0, width - 1, length, height, width, vWall);
    merge_boxes(e);

    // generateArrangement :68-127 (breadth-first growth; `directions` keeps being re-shuffled in place)
    const int arrangementSize = randRange(2, 8, rng);
    e.numItems = 0;
    std::vector<Env::ArrItem> queue;
    size_t head = 0;
    auto used = [&](int x, int y, int z) {
        for (int i = 0; i < e.numItems; ++i) if (e.items[i].off[0] == x && e.items[i].off[1] == y && e.items[i].off[2] == z) return true;
        return false;
    };
    e.items[e.numItems++] = random_item(rng, 0, 0, 0);
    queue.push_back(e.items[0]);
    std::vector<C3> directions{{-1, 0, 0}, {1, 0, 0}, {0, 1, 0}, {0, 0, -1}, {0, 0, 1}};
    while (head < queue.size()) {
        const Env::ArrItem curr = queue[head++];
        int maxBranches = randRange(1, int(directions.size()) + 1, rng);
        maxBranches = randRange(1, maxBranches + 1, rng);
        int numBranches = 0;
        std::shuffle(directions.begin(), directions.end(), rng);
        for (const C3 dir : directions) {
            const int nx = curr.off[0] + dir.x, ny = curr.off[1] + dir.y, nz = curr.off[2] + dir.z;
            if (ny >= 2 || std::abs(nx) >= 2 || std::abs(nz) >= 2) continue;
            if (used(nx, ny, nz)) continue;
            if (!(ny == 0 || used(nx, ny - 1, nz))) continue;   // on the floor or on top of another item
            const Env::ArrItem item = random_item(rng, nx, ny, nz);
            queue.push_back(item);
            e.items[e.numItems++] = item;
            ++numBranches;
            if (numBranches >= maxBranches) break;
            if (e.numItems >= arrangementSize) break;
        }
        if (e.numItems >= arrangementSize) break;
    }

    // agentStartingPositions :182-201 (drawn by spawnAgents BEFORE the spawn rotations)
    std::vector<C3> spawns(e.numAgents, C3{0, 0, 0});
    for (int i = 0; i < e.numAgents; ++i)
        for (int attempt = 0; attempt < 20; ++attempt) {
            const int ax = randRange(2, length - 1, rng), az = randRange(2, width - 1, rng);
            if (std::abs(ax - RE_LEFT[0]) < 2 && std::abs(az - RE_LEFT[2]) < 2) continue;
            if (std::abs(ax - RE_RIGHT[0]) < 2 && std::abs(az - RE_RIGHT[2]) < 2) continue;
            spawns[i] = C3{ax, 2, az};
            break;
        }
    spawn_agents(e, spawns);

    // addEpisodeDrawables :265-299: solid (undrawn, uncollided) voxels under both work areas so that drops stop at y == 2
    for (int dx = -3; dx <= 3; ++dx)
        for (int dz = -3; dz <= 3; ++dz)
            for (const int *c : {RE_LEFT, RE_RIGHT})
                if (Env::inChunk(c[0] + dx, 1, c[2] + dz)) e.chunk[Env::cell(c[0] + dx, 1, c[2] + dz)] |= VX_SOLID;
    // arrangementDrawables(right, interactive) :203-263: the first numUnmovedItems stay, the others go to random free floor cells
    const int numUnmoved = randRange(0, e.numItems, rng);
    std::vector<C3> occupied;
    for (int i = 0; i < e.numItems; ++i) occupied.push_back(C3{e.items[i].off[0], e.items[i].off[1], e.items[i].off[2]});
    auto is_occupied = [&](const C3 &c) { for (const C3 &o : occupied) if (o.x == c.x && o.y == c.y && o.z == c.z) return true; return false; };
    e.numObjects = e.numItems;
    for (int i = 0; i < e.numItems; ++i) {
        C3 off{e.items[i].off[0], e.items[i].off[1], e.items[i].off[2]};
        if (i >= numUnmoved) {
            while (is_occupied(off)) off = C3{randRange(-2, 3, rng), 0, randRange(-2, 3, rng)};
            occupied.push_back(off);
        }
        e.objects[i] = Object{off.x + RE_RIGHT[0], off.y + RE_RIGHT[1], off.z + RE_RIGHT[2], 0};
        if (Env::inChunk(e.objects[i].x, e.objects[i].y, e.objects[i].z)) e.chunk[Env::cell(e.objects[i].x, e.objects[i].y, e.objects[i].z)] |= VX_OBJECT;
    }
    e.numPlatforms = rearrange_matching(e);   // maxMatchingObjects :272

    // static colliding boxes :276-298: centre +- half extents, raised floor first, then the two pedestals
    e.numStatic = 0;
    auto add_static = [&](V3 half, V3 centre, unsigned color) {
        Env::StaticBox &b = e.statics[e.numStatic++];
        b.lo = v3(centre.x - half.x, centre.y - half.y, centre.z - half.z);
        b.hi = v3(centre.x + half.x, centre.y + half.y, centre.z + half.z);
        b.color = color;
    };
    add_static(v3(8.35f, 0.5f, 5.65f), v3(9.5f + 0.0f, 0.0f + 1.0f, 7.0f + 0.0f), 0x555555);
    const V3 lc = v3(float(RE_LEFT[0]), float(RE_LEFT[1]), float(RE_LEFT[2])), rc = v3(float(RE_RIGHT[0]), float(RE_RIGHT[1]), float(RE_RIGHT[2]));
    add_static(v3(3.0f, 0.5f, 3.0f), v3(lc.x + 0.5f, lc.y + -0.5f, lc.z + 0.5f), 0xffffff);
    add_static(v3(1.5f, 0.5f, 1.5f), v3(lc.x + 0.5f, lc.y + -0.45f, lc.z + 0.5f), 0x555555);
    add_static(v3(3.0f, 0.5f, 3.0f), v3(lc.x + 1.0f, lc.y + -0.66f, lc.z + 1.0f), 0xffffff);
    add_static(v3(3.0f, 0.5f, 3.0f), v3(lc.x + 1.5f, lc.y + -0.82f, lc.z + 1.5f), 0xffffff);
    add_static(v3(3.0f, 0.5f, 3.0f), v3(rc.x + 0.5f, rc.y + -0.5f, rc.z + 0.5f), 0x2eb5d0);
    add_static(v3(1.5f, 0.5f, 1.5f), v3(rc.x + 0.5f, rc.y + -0.45f, rc.z + 0.5f), 0x555555);
    add_static(v3(3.0f, 0.5f, 3.0f), v3(rc.x + 0.0f, rc.y + -0.66f, rc.z + 1.0f), 0x2eb5d0);
    add_static(v3(3.0f, 0.5f, 3.0f), v3(rc.x + -0.5f, rc.y + -0.82f, rc.z + 1.5f), 0x2eb5d0);

    e.numTerrain = 0; e.numRewards = 0; e.solved = 0; e.highestTower = 0; e.bzReward = 0;
    e.episodeLen = e.p_episodeLengthSec;
    e.barHalfWidth = 0.24f;
}

// ------------------------------------------------------------------------------------------------
// Honeycomb mazes -- src/libs/mazes (vendored in the reference tree, self-contained): HoneyCombMaze::InitialiseGraph
// (honeycombmaze.cpp:11-43), Kruskal::SpanningTree (kruskal.cpp:6-27), Maze::RemoveBorders (maze.cpp:20-37).  Restated with the same
// container orders, because the order of a cell's adjacency list is the order HexagonalMazeComponent walks the walls in
// (component_hexagonal_maze.cpp:52-58), i.e. the order of its RNG draws.  Pinned against the library itself compiled in place
// (oracle/_ref, tests/test_oracle_hex.py).  The reference seeds Kruskal's mt19937 from std::random_device
// (spanningtreealgorithm.cpp:3-5): its mazes are not reproducible from the env seed; here the generator takes a seed.
// ------------------------------------------------------------------------------------------------
struct HexBorder { int to; double x1, y1, x2, y2; };   // adjacent cell (-1: outside) and the border segment, maze units
struct HexMaze {
    int size = 0, cells = 0;
    std::vector<std::vector<HexBorder>> adj;
    std::vector<std::pair<double, double>> centers;
    double xlim = 0, ylim = 0;   // GetCoordinateBounds: [-xlim, xlim] x [-ylim, ylim]
};

static std::pair<int, int> hex_vextent(int size, int u) { return u < 0 ? std::make_pair(-size - u + 1, size - 1) : std::make_pair(-size + 1, size - 1 - u); }
static bool hex_valid(int size, int u, int v)
{
    if (u <= -size || u >= size) return false;
    const auto e = hex_vextent(size, u);
    return v >= e.first && v <= e.second;
}
static int hex_vertex_index(int size, int u, int v)
{
    if (u <= 0) return ((3 * size + u) * (size + u - 1)) / 2 + v;
    return (3 * size * (size - 1) + (4 * size - u - 1) * u) / 2 + v;
}

static void hex_maze_generate(HexMaze &m, int size, uint32_t kruskal_seed)
{
    static const int neigh[6][2] = {{-1, 0}, {-1, 1}, {0, 1}, {1, 0}, {1, -1}, {0, -1}};
    m.size = size;
    m.cells = 3 * size * (size - 1) + 1;
    m.adj.assign(m.cells, {});
    m.centers.assign(m.cells, {0.0, 0.0});
    const int startvertex = 0, endvertex = 3 * size * (size - 1);
    const bool bordersForEntranceAndExit = true;   // honeycombmaze.h:15
    for (int u = -size + 1; u < size; ++u) {
        const auto ve = hex_vextent(size, u);
        for (int v = ve.first; v <= ve.second; ++v) {
            const int node = hex_vertex_index(size, u, v);
            // the border end points are cos / sin of (n - 2.5) pi / 3 and of that + pi / 3 (honeycombmaze.cpp:63-68).  They are literal
            // doubles here -- the same table as in the product's host generator -- because a compiler may call sincos() where another calls
            // cos(): one ulp of a double in a sum that cancels.  (The reference library agrees to 1e-15: tests/test_oracle_hex.py.)
            static const double HEX_C1[6] = {-0x1.bb67ae8584cabp-1, 0x1.1a62633145c07p-54, 0x1.bb67ae8584cabp-1, 0x1.bb67ae8584cabp-1, 0x1.1a62633145c07p-54, -0x1.bb67ae8584cabp-1};
            static const double HEX_S1[6] = {-0x1.fffffffffffffp-2, -0x1.0000000000000p+0, -0x1.fffffffffffffp-2, 0x1.fffffffffffffp-2, 0x1.0000000000000p+0, 0x1.fffffffffffffp-2};
            static const double HEX_C2[6] = {-0x1.72cece675d1fcp-53, 0x1.bb67ae8584caap-1, 0x1.bb67ae8584cabp-1, 0x1.1a62633145c07p-54, -0x1.bb67ae8584ca9p-1, -0x1.bb67ae8584caap-1};
            static const double HEX_S2[6] = {-0x1.0000000000000p+0, -0x1.0000000000000p-1, 0x1.fffffffffffffp-2, 0x1.0000000000000p+0, 0x1.0000000000003p-1, -0x1.0000000000001p-1};
            const double dxu = 0x1.bb67ae8584caap-1 /* sqrt(3) / 2 */, dyu = 1.5, dxv = 0x1.bb67ae8584caap+0 /* sqrt(3) */, dyv = 0;
            const double cx = dxu * u + dxv * v, cy = dyu * u + dyv * v;
            m.centers[node] = {cx, cy};
            for (int n = 0; n < 6; ++n) {
                const int uu = u + neigh[n][0], vv = v + neigh[n][1];
                const HexBorder b{-1, cx + HEX_C1[n], cy + HEX_S1[n], cx + HEX_C2[n], cy + HEX_S2[n]};
                if (hex_valid(size, uu, vv)) {
                    const int nnode = hex_vertex_index(size, uu, vv);
                    if (nnode > node) continue;
                    HexBorder a = b; a.to = nnode;
                    HexBorder c = b; c.to = node;
                    m.adj[node].push_back(a);
                    m.adj[nnode].push_back(c);
                } else {
                    if (!bordersForEntranceAndExit && ((node == startvertex && n == 0) || (node == endvertex && n == 3))) continue;
                    m.adj[node].push_back(b);
                }
            }
        }
    }
    // Kruskal: every inner edge once (i < neighbour), std::shuffle with the algorithm's own mt19937, union-find with path compression
    std::vector<std::pair<int, int>> edges;
    for (int i = 0; i < m.cells; ++i)
        for (const HexBorder &e : m.adj[i])
            if (e.to > i) edges.push_back({i, e.to});
    std::mt19937 generator(kruskal_seed);
    std::shuffle(edges.begin(), edges.end(), generator);
    std::vector<int> parent(m.cells);
    std::iota(parent.begin(), parent.end(), 0);
    std::function<int(int)> root = [&](int u) { return parent[u] == u ? u : (parent[u] = root(parent[u])); };
    for (const auto &e : edges) {
        const int a = root(e.first), b = root(e.second);
        if (a == b) continue;
        parent[a] = b;
        // RemoveBorders: the first matching entry of either list
        for (int side = 0; side < 2; ++side) {
            auto &lst = m.adj[side == 0 ? e.first : e.second];
            const int other = side == 0 ? e.second : e.first;
            for (size_t i = 0; i < lst.size(); ++i)
                if (lst[i].to == other) { lst.erase(lst.begin() + i); break; }
        }
    }
    m.xlim = 0x1.bb67ae8584caap+0 * (size - 0.5);
    m.ylim = 1.5 * size - 0.5;
}

// ------------------------------------------------------------------------------------------------
// Sokoban -- scenario_sokoban.{hpp,cpp}: Boxoban levels (text files under $BOXOBAN_LEVELS/unfiltered/train), voxel size 2.
// ------------------------------------------------------------------------------------------------
// The three wall orientations.  layoutBox.rotateY(-atanf(dz / dx)) (component_hexagonal_maze.cpp:84-90) of a honeycomb border is +30,
// -30 or 90 degrees; (cos, sin) are literals so that neither libm nor constant folding is involved (the reference's 90 degrees is
// float(M_PI_2), cos = -4.4e-8: we use exactly 0).  Frame k: local = Ry(r)^T world, Ry(r) = [[c, 0, s], [0, 1, 0], [-s, 0, c]].
static const float HEX_ROT[HEX_FRAMES][2] = {{0.8660254f, 0.5f}, {0.8660254f, -0.5f}, {0.0f, 1.0f}};
static inline V3 hex_to_local(int k, V3 p)
{
    const float c = HEX_ROT[k][0], s = HEX_ROT[k][1];
    return v3(c * p.x - s * p.z, p.y, s * p.x + c * p.z);
}
static inline V3 hex_to_world(int k, V3 p)
{
    const float c = HEX_ROT[k][0], s = HEX_ROT[k][1];
    return v3(c * p.x + s * p.z, p.y, c * p.z - s * p.x);
}

struct HexParams {   // HexagonalMazeComponent members after reset(), component_hexagonal_maze.cpp:20-44
    int size;
    float scale, wallHeight, omitWalls, landmarkProb;
    unsigned bottomEdging, topEdging;
    double xMin, xMax, yMin, yMax;
};

static void hex_maze_reset(Env &e, HexMaze &m, HexParams &hp, int minSize, int maxSize, float omitMin, float omitMax, uint32_t kruskalSeed)
{
    hp.size = randRange(minSize, maxSize, e.rng);
    // `Kruskal algorithm;` seeds its own mt19937 from std::random_device (mazes/spanningtreealgorithm.h:19-20): not reproducible
    // in the reference.  Ours: seeded with the episode seed Env::reset drew, no extra draw from the env's generator.
    hex_maze_generate(m, hp.size, kruskalSeed);
    hp.scale = 3.5f;
    hp.wallHeight = frand(e.rng) * 0.55f + 0.85f;
    hp.omitWalls = frand(e.rng) * (omitMax - omitMin) + omitMin;
    hp.landmarkProb = frand(e.rng) * 0.15f + 0.15f;
    hp.bottomEdging = ALL_COLORS[randRange(0, 22, e.rng)];
    hp.topEdging = ALL_COLORS[randRange(0, 22, e.rng)];
    hp.xMin = -m.xlim * hp.scale; hp.xMax = m.xlim * hp.scale; hp.yMin = -m.ylim * hp.scale; hp.yMax = m.ylim * hp.scale;
}

// HexagonalMazeComponent::addDrawablesAndCollisions, component_hexagonal_maze.cpp:46-133
static void hex_add_maze(Env &e, const HexMaze &m, const HexParams &hp)
{
    {   // floor: addStaticCollidingBox(scale, translation): the unit cube [-1, 1]^3 scaled
        const V3 sc = v3(float(hp.xMax - hp.xMin), 0.0001f, float(hp.yMax - hp.yMin));
        const V3 tr = v3(float(hp.xMax + hp.xMin) / 2, 0.0f, float(hp.yMax + hp.yMin) / 2);
        Env::HexBox b; b.frame = -1; b.collide = 1; b.color = randomLayoutColor(e.rng);
        b.lo = v3(tr.x - sc.x, tr.y - sc.y, tr.z - sc.z); b.hi = v3(tr.x + sc.x, tr.y + sc.y, tr.z + sc.z);
        e.hexBoxes.push_back(b);
    }
    std::set<std::pair<int, int>> existingWalls;
    for (int cellIdx = 0; cellIdx < m.cells; ++cellIdx)
        for (const HexBorder &border : m.adj[cellIdx]) {
            std::pair<int, int> cellPair(cellIdx, border.to);
            if (cellPair.first > cellPair.second) std::swap(cellPair.first, cellPair.second);
            if (border.to != -1) {
                if (existingWalls.count(cellPair)) continue;
                if (frand(e.rng) < hp.omitWalls) continue;
            }
            existingWalls.insert(cellPair);
            const double x1 = border.x1 * hp.scale, z1 = border.y1 * hp.scale, x2 = border.x2 * hp.scale, z2 = border.y2 * hp.scale;
            const float length = 0.5f * sqrtf(float((x1 - x2) * (x1 - x2) + (z1 - z2) * (z1 - z2)));
            const V3 wallT = v3(float(x1 + x2) / 2, hp.wallHeight, float(z1 + z2) / 2);
            const double deltaX = x1 - x2, deltaZ = z1 - z2;
            int k = 2;                                           // rotationY = M_PI_2
            if (std::fabs(deltaX) > 1e-5f) k = (deltaZ / deltaX) < 0 ? 0 : 1;   // -atanf(tanAlpha): +30 / -30 degrees (EPSILON 1e-5f, util/macro.hpp:12)
            const V3 wc = hex_to_local(k, wallT);
            if (frand(e.rng) < hp.landmarkProb) {
                const float landmarkWidth = 0.15f, landmarkHeight = landmarkWidth * length / hp.wallHeight;
                const int numLandmarks = randRange(2, 5, e.rng);
                for (int li = 0; li < numLandmarks; ++li) {
                    const float depth = frand(e.rng) * 1.2f + 1.5f;
                    const float tx = float(li % 2 == 1) * landmarkWidth * 2, ty = float(li > 1) * landmarkHeight * 2 - 0.2f;
                    // child of the wall box: wallScale * (landmarkTranslation + landmarkScale * cube)
                    const V3 c = v3(wc.x + length * tx, wc.y + hp.wallHeight * ty, wc.z);
                    const V3 h = v3(length * landmarkWidth, hp.wallHeight * landmarkHeight, 0.15f * depth);
                    Env::HexBox b; b.frame = k; b.collide = 0; b.color = ALL_COLORS[randRange(0, 22, e.rng)];
                    b.lo = v3(c.x - h.x, c.y - h.y, c.z - h.z); b.hi = v3(c.x + h.x, c.y + h.y, c.z + h.z);
                    e.hexBoxes.push_back(b);
                }
            }
            {   // the wall: drawn DARK_BLUE, btBoxShape(1, 1, 1) under the same transformation
                const V3 h = v3(length, hp.wallHeight, 0.15f);
                Env::HexBox b; b.frame = k; b.collide = 1; b.color = 0x3a7fa6;
                b.lo = v3(wc.x - h.x, wc.y - h.y, wc.z - h.z); b.hi = v3(wc.x + h.x, wc.y + h.y, wc.z + h.z);
                e.hexBoxes.push_back(b);
            }
            {   // bottom edging (the top one is commented out in the reference)
                const V3 h = v3(length * 1.02f, hp.wallHeight * 0.12f, 0.2f);
                const V3 c = hex_to_local(k, v3(wallT.x, h.y, wallT.z));
                Env::HexBox b; b.frame = k; b.collide = 0; b.color = hp.bottomEdging;
                b.lo = v3(c.x - h.x, c.y - h.y, c.z - h.z); b.hi = v3(c.x + h.x, c.y + h.y, c.z + h.z);
                e.hexBoxes.push_back(b);
            }
        }
    // our list order (== depth-tie order; the reference's draw order is creation order, its ties are the GL rasteriser's): the
    // colliding boxes -- floor, walls -- first, so that the device's broadphase streams a prefix of the list
    std::stable_partition(e.hexBoxes.begin(), e.hexBoxes.end(), [](const Env::HexBox &b) { return b.collide != 0; });
}

static Env::HexObj hex_object(int shape, unsigned color, V3 loc, V3 scale, int good, V3 gridCoord)
{
    Env::HexObj o;
    o.pos = loc; o.scale = scale; o.shape = shape; o.good = good; o.alive = 1; o.color = color;
    o.vox[0] = (int)lroundf(floorf(gridCoord.x)); o.vox[1] = (int)lroundf(floorf(gridCoord.y)); o.vox[2] = (int)lroundf(floorf(gridCoord.z));
    return o;
}

static void hex_common_reset(Env &e)
{
    std::fill(e.chunk.begin(), e.chunk.end(), 0);
    e.numBoxes = 0; e.numObjects = 0; e.numTerrain = 0; e.numRewards = 0; e.numItems = 0; e.numStatic = 0;
    e.L = e.H = e.W = 0; e.bz[0] = e.bz[1] = e.bz[2] = e.bz[3] = 0; e.drawWalls = 0;
    e.solved = 0; e.highestTower = 0; e.numPlatforms = 0; e.bzReward = 0; e.barHalfWidth = 0.24f;
    e.hexBoxes.clear(); e.hexObjs.clear();
}

// HexExploreScenario: reset :21-41, agentStartingPositions :60-101, addEpisodeDrawables :103-110
static void hex_explore_generate(Env &e, uint32_t kruskalSeed)
{
    hex_common_reset(e);
    HexMaze m; HexParams hp;
    hex_maze_reset(e, m, hp, 2, 8, 0.1f, 0.4f, kruskalSeed);
    const int randomCellIdx = randRange(0, m.cells, e.rng);
    e.hexTarget = v3(float(m.centers[randomCellIdx].first) * hp.scale, 0.0f, float(m.centers[randomCellIdx].second) * hp.scale);

    // spawnAgents (scenario_default.hpp:80-97): starting positions, then one frand per agent
    std::vector<int> cellIndices(m.cells, 0);
    std::iota(cellIndices.begin(), cellIndices.end(), 0);
    std::shuffle(cellIndices.begin(), cellIndices.end(), e.rng);
    std::vector<F3> positions;
    float furthest = 0;
    for (int cellIdx : cellIndices) {
        const V3 spawnPos = v3(float(m.centers[cellIdx].first) * hp.scale, 0.1f, float(m.centers[cellIdx].second) * hp.scale);
        const float distance = sqrtf(len2(e.hexTarget - spawnPos));
        const float rotation = float(2 * M_PI / e.numAgents);
        if (distance > furthest) {
            positions.clear();
            for (int i = 0; i < e.numAgents; ++i)
                positions.push_back(F3{spawnPos.x + sinf(float(i) * rotation), spawnPos.y + 0.0f, spawnPos.z + cosf(float(i) * rotation)});
            furthest = distance;
        }
        if (distance > float(hp.size) * hp.scale) break;
    }
    if (positions.empty()) positions.assign(size_t(e.numAgents), F3{0, 1, 0});
    spawn_agents_at(e, positions);

    hex_add_maze(e, m, hp);
    const float sc = 1.9f;   // the reward object: VIOLET diamond
    e.hexObjs.push_back(hex_object(HEX_DIAMOND, 0xd468ee, v3(e.hexTarget.x + 0.0f, e.hexTarget.y + 1.2f, e.hexTarget.z + 0.0f),
                                   v3(0.17f * sc, 0.35f * sc, 0.17f * sc), 1, e.hexTarget));
    e.episodeLen = e.p_episodeLengthSec;
}

// HexMemoryScenario: reset :21-82, spawnAgents :131-160, addEpisodeDrawables :163-216
static void hex_memory_generate(Env &e, uint32_t kruskalSeed)
{
    hex_common_reset(e);
    HexMaze m; HexParams hp;
    hex_maze_reset(e, m, hp, 2, 8, 0.1f, 0.95f, kruskalSeed);

    double minDistanceToCenter = 1e9f;
    int centerCellIdx = 0;
    for (int cellIdx = 0; cellIdx < m.cells; ++cellIdx) {
        const double d = sqrt(m.centers[cellIdx].first * m.centers[cellIdx].first + m.centers[cellIdx].second * m.centers[cellIdx].second);
        if (d < minDistanceToCenter) { centerCellIdx = cellIdx; minDistanceToCenter = float(d); }   // float minDistanceToCenter
    }
    const V3 landmarkLocation = v3(float(m.centers[centerCellIdx].first * hp.scale), 1.0f, float(m.centers[centerCellIdx].second * hp.scale));

    std::vector<F3> objectCoordinates;
    for (int cellIdx = 0; cellIdx < m.cells; ++cellIdx) {
        if (cellIdx == centerCellIdx) continue;
        // Vector3(frand - 0.5f, 0, frand - 0.5f): GCC evaluates the arguments right to left, z first
        const float oz = frand(e.rng) - 0.5f;
        const float ox = frand(e.rng) - 0.5f;
        const float cx = float(m.centers[cellIdx].first) + ox, cz = float(m.centers[cellIdx].second) + oz;
        objectCoordinates.push_back(F3{cx * hp.scale, 0.5f + 0.0f, cz * hp.scale});
    }
    std::shuffle(objectCoordinates.begin(), objectCoordinates.end(), e.rng);
    const float fraction = frand(e.rng) * 0.25f + 0.2f;
    const long numGood = std::lround(ceilf(fraction * objectCoordinates.size()));
    const long numBad = numGood;
    const bool haveBad = long(objectCoordinates.size()) >= numGood + numBad;

    // spawnAgents: no draws; agents in a circle around the origin, each looking along its own direction
    {
        const float rotationBetweenAgents = float(2 * M_PI / e.numAgents);
        std::vector<F3> positions;
        std::vector<float> yaw;
        for (int i = 0; i < e.numAgents; ++i) {
            positions.push_back(F3{1.5f * sinf(rotationBetweenAgents * float(i)), 1.5f * 0.3f, 1.5f * cosf(rotationBetweenAgents * float(i))});
            yaw.push_back(rotationBetweenAgents * i);
        }
        spawn_agents_at(e, positions, yaw.data());
    }

    // addEpisodeDrawables
    unsigned goodColor = OBJECT_COLORS[randRange(0, 14, e.rng)], badColor = goodColor;
    int goodShape = randRange(0, 3, e.rng), badShape = goodShape;
    while (badColor == goodColor && badShape == goodShape) {
        badColor = OBJECT_COLORS[randRange(0, 14, e.rng)];
        badShape = randRange(0, 3, e.rng);
    }
    hex_add_maze(e, m, hp);

    auto scaleOf = [](int shape) {
        if (shape == HEX_SPHERE) return v3(0.75f, 0.75f, 0.75f);
        if (shape == HEX_PILLAR) return v3(0.5f, 2.0f, 0.5f);
        return v3(0.17f * 2.2f, 0.45f * 2.2f, 0.17f * 2.2f);
    };
    auto shiftOf = [](int shape) {
        if (shape == HEX_SPHERE) return v3(0.5f, 0.1f, 0.5f);
        if (shape == HEX_PILLAR) return v3(0.5f, 0.05f, 0.5f);
        return v3(0.5f, 0.6f, 0.5f);
    };
    {   // the landmark object in the centre cell shows what to collect; it is not in the grid
        Env::HexObj o = hex_object(goodShape, goodColor, landmarkLocation + shiftOf(goodShape), scaleOf(goodShape), 1, landmarkLocation);
        o.alive = 0;
        e.hexObjs.push_back(o);
    }
    const float objScale = 0.6f;
    for (int pass = 0; pass < 2; ++pass) {
        const int shape = pass == 0 ? goodShape : badShape;
        const unsigned color = pass == 0 ? goodColor : badColor;
        const long from = pass == 0 ? 0 : numGood, to = pass == 0 ? numGood : (haveBad ? numGood + numBad : numGood);
        for (long i = from; i < to; ++i) {
            const V3 coord = v3(objectCoordinates[i].x, objectCoordinates[i].y, objectCoordinates[i].z);
            e.hexObjs.push_back(hex_object(shape, color, coord + shiftOf(shape) * objScale, scaleOf(shape) * objScale, pass == 0, coord));
        }
    }
    e.numPlatforms = int(numGood);
    e.episodeLen = e.p_episodeLengthSec + 3.0f * float(numGood);   // scenario_hex_memory.hpp:45-49
}

static std::vector<std::string> split_tokens(const std::string &text, char delim)
{   // splitString (util/src/string_utils.cpp:10-25) is strtok_r: empty tokens are skipped
    std::vector<std::string> out;
    std::string cur;
    for (char c : text) {
        if (c == delim) { if (!cur.empty()) out.push_back(cur); cur.clear(); }
        else cur.push_back(c);
    }
    if (!cur.empty()) out.push_back(cur);
    return out;
}

static bool read_file(const std::string &path, std::string &out)
{
    FILE *f = fopen(path.c_str(), "rb");
    if (!f) return false;
    char buf[4096];
    size_t n;
    out.clear();
    while ((n = fread(buf, 1, sizeof buf, f)) > 0) out.append(buf, n);
    fclose(f);
    return true;
}

static void sokoban_reload_levels(Env &e)
{   // reloadLevels :80-102: one random file; a level is stored when the NEXT ';' line is met (the file's last level never is)
    const std::vector<std::string> &files = *e.sokoFiles;
    const std::string &path = files[randRange(0, int(files.size()), e.rng)];
    std::string content;
    read_file(path, content);
    const std::vector<std::string> lines = split_tokens(content, '\n');
    std::vector<std::string> level;
    for (int i = 0; i < int(lines.size()); ++i) {
        if (lines[i].find(';') == 0) {
            if (i > 0) e.sokoLevels.push_back(level);
            level.clear();
        } else level.push_back(lines[i]);
    }
    std::shuffle(e.sokoLevels.begin(), e.sokoLevels.end(), e.rng);
}

static void sokoban_generate(Env &e)
{
    Rng &rng = e.rng;
    e.voxelSize = 2.0f;
    std::fill(e.chunk.begin(), e.chunk.end(), 0);
    std::fill(e.soko.begin(), e.soko.end(), 0);
    if (e.sokoLevels.empty()) sokoban_reload_levels(e);
    const std::vector<std::string> rows = e.sokoLevels.back();   // reset :104-120
    e.sokoLevels.pop_back();

    // createLayout :122-170
    static const unsigned floorColors[5] = {0xffffff, 0xffffe6, 0xe6ecff, 0xffebcc, 0x555555};
    const unsigned floorColor = floorColors[randRange(0, 5, rng)];
    const int length = std::min(int(rows.size()), int(SOKO_DIM));
    int width = 0;
    std::vector<F3> agentPositions;
    std::vector<C3> boxes;
    for (int x = 0; x < length; ++x) width = std::max(width, std::min(int(rows[x].size()), int(SOKO_DIM)));
    const int org[3] = {0, 0, 0}, dim[3] = {length, 3, std::max(width, 1)};
    std::vector<uint8_t> g(size_t(dim[0]) * dim[1] * dim[2], 0);
    auto cell = [&](int x, int y, int z) -> uint8_t & { return g[(size_t(y) * dim[2] + z) * dim[0] + x]; };
    for (int x = 0; x < length; ++x) {
        const std::string &row = rows[x];
        for (int z = 0; z < std::min(int(row.size()), int(SOKO_DIM)); ++z) {
            cell(x, 0, z) = VX_SOLID | VX_OPAQUE;   // floor
            if (row[z] == '#') {
                cell(x, 1, z) = VX_SOLID; cell(x, 2, z) = VX_SOLID;   // solid, not drawn
                e.soko[x * SOKO_DIM + z] |= SOKO_WALL;
            }
            if (row[z] == '@' || row[z] == '+')
                for (int k = 0; k < e.numAgents; ++k) {
                    const float ax = float(x) + float(k % 2) * 0.5f, az = float(z) + float(k % 4 > 1) * 0.5f;
                    agentPositions.push_back(F3{ax * e.voxelSize, float(e.voxelSize + 0.3 * float(k) * e.voxelSize), az * e.voxelSize});
                }
            if (row[z] == '.' || row[z] == '+') e.soko[x * SOKO_DIM + z] = SOKO_GOAL;   // g.set(...): replaces the cell (a goal is never a wall)
            if (row[z] == '$' || row[z] == '*') boxes.push_back(C3{x, 1, z});           // ('*' puts a box down but, unlike '.', no goal)
        }
    }
    merge_dense(g, org, dim, e);
    e.L = length; e.H = 3; e.W = width;
    e.bz[0] = e.bz[1] = e.bz[2] = e.bz[3] = 0;
    e.layoutColor = floorColor; e.wallColor = floorColor; e.drawWalls = 0;
    e.numTerrain = 0; e.numRewards = 0; e.numPlatforms = 0; e.numItems = 0; e.numStatic = 0;
    e.numObjects = std::min(int(boxes.size()), int(MAX_OBJECTS));
    for (int i = 0; i < e.numObjects; ++i) e.objects[i] = Object{boxes[i].x, boxes[i].y, boxes[i].z, 0};
    e.solved = 0; e.highestTower = 0; e.bzReward = 0;
    e.episodeLen = e.p_episodeLengthSec;
    e.barHalfWidth = 0.24f;
    if (agentPositions.empty()) agentPositions.push_back(F3{0, 0, 0});
    spawn_agents_at(e, agentPositions);
}


// Empty -- scenario_empty.{hpp,cpp}: the scenario of the reference's own performance test (README.md:243-247).  reset() draws nothing;
// one static colliding box, scale (10, 1, 10) at (5, 0, 5) (addStaticCollidingBox, layout_utils.cpp:70-83) = the integer slab
// [-5, 15) x [-1, 1) x [-5, 15), BLUE; every agent starts at (1, 1, 1); no object stacking, no fall detection, no rewards.
static void empty_generate(Env &e)
{
    std::fill(e.chunk.begin(), e.chunk.end(), 0);
    e.numBoxes = 1;
    e.boxes[0] = Box{{-5, -1, -5}, {15, 1, 15}, VX_SOLID | VX_OPAQUE, 0};
    e.L = 20; e.H = 2; e.W = 20;
    e.bz[0] = e.bz[1] = e.bz[2] = e.bz[3] = 0;
    e.layoutColor = 0x2eb5d0; e.wallColor = 0x2eb5d0; e.drawWalls = 0;   // ColorRgb::BLUE
    e.numTerrain = 0; e.numRewards = 0; e.numPlatforms = 0; e.numItems = 0; e.numStatic = 0; e.numObjects = 0;
    e.solved = 0; e.highestTower = 0; e.bzReward = 0;
    e.episodeLen = e.p_episodeLengthSec;
    e.barHalfWidth = 0.24f;
    spawn_agents(e, std::vector<C3>(size_t(e.numAgents), C3{1, 1, 1}));   // agentStartingPositions, scenario_empty.cpp:20-23
}

static void env_reset(Env &e)
{
    // ---- Env::reset, env/src/env.cpp:57-76 ; EnvState::reset env.hpp:135-151
    e.done = 0; e.episodeSec = 0; e.numFrames = 0;
    const int seed = randRange(0, 1 << 30, e.rng);
    e.rng.seed((unsigned long)seed);
    if (e.scenario == SCN_TOWER) tower_generate(e);
    else if (e.scenario == SCN_OBSTACLES) obstacles_generate(e);
    else if (e.scenario == SCN_COLLECT) collect_generate(e);
    else if (e.scenario == SCN_REARRANGE) rearrange_generate(e);
    else if (e.scenario == SCN_EMPTY) empty_generate(e);
    else if (e.scenario == SCN_HEX_MEMORY) hex_memory_generate(e, uint32_t(seed));
    else if (e.scenario == SCN_HEX_EXPLORE) hex_explore_generate(e, uint32_t(seed));
    else sokoban_generate(e);
}

// ------------------------------------------------------------------------------------------------
// Collision geometry.  A vertical capsule (segment half-length CAP_HH, radius r) against an
// axis-aligned box == the capsule CENTRE against the box grown by CAP_HH in +-y, rounded by r.
// ------------------------------------------------------------------------------------------------
struct Collider {
    int kind;    // 0 none, 1 box (lo/hi already grown in y by CAP_HH), 2 vertical capsule (other agent), 3 box in hex wall frame `frame`
    int frame;
    V3 lo, hi;   // box bounds; capsule: lo = centre, hi.x = segment half-length
};

struct Closest {
    float dist;  // signed distance of the capsule surface to the collider (negative = penetrating)
    V3 n;        // unit normal from the collider towards the capsule
};

static Closest closest_box(V3 p, V3 lo, V3 hi, float r)
{
    const float qx = std::min(std::max(p.x, lo.x), hi.x);
    const float qy = std::min(std::max(p.y, lo.y), hi.y);
    const float qz = std::min(std::max(p.z, lo.z), hi.z);
    const V3 v = v3(p.x - qx, p.y - qy, p.z - qz);
    const float d2 = len2(v);
    Closest c;
    if (d2 > 0.0f) {
        const float d = sqrtf(d2);
        const float inv = 1.0f / d;
        c.dist = d - r;
        c.n = v * inv;
    } else {  // centre inside the grown box: exit through the nearest face
        float m = p.x - lo.x; V3 n = v3(-1, 0, 0);
        float t = hi.x - p.x; if (t < m) { m = t; n = v3(1, 0, 0); }
        t = p.y - lo.y; if (t < m) { m = t; n = v3(0, -1, 0); }
        t = hi.y - p.y; if (t < m) { m = t; n = v3(0, 1, 0); }
        t = p.z - lo.z; if (t < m) { m = t; n = v3(0, 0, -1); }
        t = hi.z - p.z; if (t < m) { m = t; n = v3(0, 0, 1); }
        c.dist = -m - r;
        c.n = n;
    }
    return c;
}

static Closest closest_capsule(V3 p, V3 centre, float halfLen, float r)
{
    const float qy = std::min(std::max(p.y, centre.y - halfLen), centre.y + halfLen);
    const V3 v = v3(p.x - centre.x, p.y - qy, p.z - centre.z);
    const float d2 = len2(v);
    Closest c;
    if (d2 > 1e-12f) {
        const float d = sqrtf(d2);
        const float inv = 1.0f / d;
        c.dist = d - r;
        c.n = v * inv;
    } else {
        c.dist = -r;
        c.n = v3(1, 0, 0);
    }
    return c;
}

// rBox: capsule radius against boxes; rCap: summed radii against another capsule
static Closest closest(const Collider &col, V3 p, float rBox, float rCap)
{
    if (col.kind == 1) return closest_box(p, col.lo, col.hi, rBox);
    if (col.kind == 3) {   // rotated about Y only: the capsule is vertical in the wall's frame too
        Closest c = closest_box(hex_to_local(col.frame, p), col.lo, col.hi, rBox);
        c.n = hex_to_world(col.frame, c.n);
        return c;
    }
    return closest_capsule(p, col.lo, col.hi.x, rCap);
}

// The closest-point query without the normalisation: v = p minus the collider's closest point (not unit), d = |v|, dist = the signed surface
// distance; centre inside the grown box / on the capsule's axis: v = the unit exit normal, d = 1.  closest().n == v * (1 / d).
struct Raw {
    V3 v;
    float d, dist;
};

static Raw raw_box(V3 p, V3 lo, V3 hi, float r)
{
    const float qx = std::min(std::max(p.x, lo.x), hi.x);
    const float qy = std::min(std::max(p.y, lo.y), hi.y);
    const float qz = std::min(std::max(p.z, lo.z), hi.z);
    const V3 v = v3(p.x - qx, p.y - qy, p.z - qz);
    const float d2 = len2(v);
    Raw c;
    if (d2 > 0.0f) {
        c.d = sqrtf(d2);
        c.dist = c.d - r;
        c.v = v;
    } else {
        float m = p.x - lo.x; V3 n = v3(-1, 0, 0);
        float t = hi.x - p.x; if (t < m) { m = t; n = v3(1, 0, 0); }
        t = p.y - lo.y; if (t < m) { m = t; n = v3(0, -1, 0); }
        t = hi.y - p.y; if (t < m) { m = t; n = v3(0, 1, 0); }
        t = p.z - lo.z; if (t < m) { m = t; n = v3(0, 0, -1); }
        t = hi.z - p.z; if (t < m) { m = t; n = v3(0, 0, 1); }
        c.d = 1.0f;
        c.dist = -m - r;
        c.v = n;
    }
    return c;
}

static Raw raw_capsule(V3 p, V3 centre, float halfLen, float r)
{
    const float qy = std::min(std::max(p.y, centre.y - halfLen), centre.y + halfLen);
    const V3 v = v3(p.x - centre.x, p.y - qy, p.z - centre.z);
    const float d2 = len2(v);
    Raw c;
    if (d2 > 1e-12f) {
        c.d = sqrtf(d2);
        c.dist = c.d - r;
        c.v = v;
    } else {
        c.d = 1.0f;
        c.dist = -r;
        c.v = v3(1, 0, 0);
    }
    return c;
}

// [3P] Bullet 2.89 btContinuousConvexCollision::calcTimeOfImpact restated for a translating
// convex shape with exact closest points (conservative advancement).  Call sites in the
// reference: kinematic_character_controller.cpp:252,363,425 (ghost convexSweepTest).
// Bullet advances by dist / (-(d . n)) with the unit normal n = v / |v|; the same quotient is formed here as (dist |v|) / (-(d . v)) -- one
// correctly rounded divide per iteration instead of two -- and the normal is normalised once, on a hit (the device's slowest tick is a
// grazing cast that runs all 64 iterations).  A box in a hex wall frame is cast in that frame (start and direction rotated in once, the
// normal rotated back once).  megaverse_amd/csrc/mv_physics.h: convex_cast is the same arithmetic, operation for operation.
static bool convex_cast(const Collider &col, V3 p, V3 d, float *fraction, V3 *normal)
{
    const bool boxLike = col.kind != 2;
    if (col.kind == 3) { p = hex_to_local(col.frame, p); d = hex_to_local(col.frame, d); }
    float lambda = 0.0f, lastLambda = 0.0f;
    int numIter = 0;
    Raw c = boxLike ? raw_box(p, col.lo, col.hi, CAP_R) : raw_capsule(p, col.lo, col.hi.x, 2 * CAP_R);
    float dist = c.dist + ALLOWED_CCD_PEN;
    float proj = -dot(d, c.v);   // |v| times Bullet's projected velocity
    if (proj <= SIMD_EPS * c.d) return false;
    while (dist > CAST_RADIUS) {
        proj = -dot(d, c.v);
        if (proj <= SIMD_EPS * c.d) return false;
        lambda = lambda + (dist * c.d) / proj;
        if (lambda > 1.0f) return false;
        if (lambda < 0.0f) return false;
        if (lambda <= lastLambda) return false;
        lastLambda = lambda;
        const V3 x = v3(p.x + lambda * d.x, p.y + lambda * d.y, p.z + lambda * d.z);
        c = boxLike ? raw_box(x, col.lo, col.hi, CAP_R) : raw_capsule(x, col.lo, col.hi.x, 2 * CAP_R);
        dist = c.dist + ALLOWED_CCD_PEN;
        if (++numIter > CAST_MAX_ITER) return false;
    }
    *fraction = lambda;
    const float inv = 1.0f / c.d;
    V3 n = c.v * inv;
    if (col.kind == 3) n = hex_to_world(col.frame, n);
    *normal = n;
    return true;
}

struct Colliders {
    int n = 0;
    Collider c[MAX_BOXES + MAX_STATIC + MAX_ITEMS + MAX_OBJECTS + MAX_AGENTS];   // (Hex*: floor + walls <= 1 + 294)   // empty slots are never emitted: only the relative order matters
};

// Rearrange items (arrangementDrawables, scenario_rearrange.cpp:203-263): drawable scale = scales[shape] * 0.45, collision
// box = that scale times the collision scale (cylinder (1, 0.5, 1), capsule (1, 2, 1)) -- 0.45 high for every shape.
static V3 item_draw_scale(int shape)
{
    const float s = 0.45f;
    if (shape == SHAPE_CAPSULE) return v3(0.8f * s, 0.5f * s, 0.8f * s);
    if (shape == SHAPE_CYLINDER) return v3(0.9f * s, 2.0f * s, 0.9f * s);
    return v3(1.0f * s, 1.0f * s, 1.0f * s);
}
static V3 item_collision_half(int shape)
{
    const V3 d = item_draw_scale(shape);
    if (shape == SHAPE_CAPSULE) return v3(d.x * 1.0f, d.y * 2.0f, d.z * 1.0f);
    if (shape == SHAPE_CYLINDER) return v3(d.x * 1.0f, d.y * 0.5f, d.z * 1.0f);
    return d;
}

// Collider order == index order used for tie-breaks: layout boxes, (Rearrange: static boxes, target items), movable boxes, agents.
static void build_colliders(const Env &e, int self, Colliders &out)
{
    out.n = 0;
    for (int i = 0; i < e.numBoxes; ++i) {
        if (!(e.boxes[i].type & VX_SOLID)) continue;  // layout_utils.cpp:42-49
        Collider &c = out.c[out.n++];
        const Box &b = e.boxes[i];
        c.kind = 1;
        const float vs = e.voxelSize;   // addBoundingBoxes scales by the voxel size (layout_utils.cpp:22-34)
        c.lo = v3(float(b.min[0]) * vs, float(b.min[1]) * vs - CAP_HH, float(b.min[2]) * vs);
        c.hi = v3(float(b.max[0]) * vs, float(b.max[1]) * vs + CAP_HH, float(b.max[2]) * vs);
    }
    if (e.scenario == SCN_HEX_MEMORY || e.scenario == SCN_HEX_EXPLORE)
        for (const Env::HexBox &b : e.hexBoxes) {   // floor (addStaticCollidingBox) and walls (btBoxShape(1,1,1) child of the wall box)
            if (!b.collide) continue;
            Collider &c = out.c[out.n++];
            c.kind = b.frame < 0 ? 1 : 3; c.frame = b.frame;
            c.lo = v3(b.lo.x, b.lo.y - CAP_HH, b.lo.z);
            c.hi = v3(b.hi.x, b.hi.y + CAP_HH, b.hi.z);
        }
    if (e.scenario == SCN_SOKOBAN)
        for (int i = 0; i < e.numObjects; ++i) {   // pushable boxes, scenario_sokoban.cpp:275-293: collision scale (1.15, 3, 1.15), offset (0, 0.6, 0)
            const Object &o = e.objects[i];
            const float vs = e.voxelSize;
            const float sx = (vs / 2) * 0.8f, sy = 0.45f * 0.8f;
            const float cx = (float(o.x) + 0.5f) * vs, cy = (float(o.y) + 0.2f) * vs + 0.6f, cz = (float(o.z) + 0.5f) * vs;
            Collider &c = out.c[out.n++];
            c.kind = 1;
            c.lo = v3(cx - sx * 1.15f, (cy - sy * 3.0f) - CAP_HH, cz - sx * 1.15f);
            c.hi = v3(cx + sx * 1.15f, (cy + sy * 3.0f) + CAP_HH, cz + sx * 1.15f);
        }
    if (e.scenario == SCN_REARRANGE) {
        for (int i = 0; i < e.numStatic; ++i) {   // addStaticCollidingBox, layout_utils.cpp:70-83
            Collider &c = out.c[out.n++];
            c.kind = 1;
            c.lo = v3(e.statics[i].lo.x, e.statics[i].lo.y - CAP_HH, e.statics[i].lo.z);
            c.hi = v3(e.statics[i].hi.x, e.statics[i].hi.y + CAP_HH, e.statics[i].hi.z);
        }
        for (int side = 0; side < 2; ++side)      // target items (static), then the movable ones that are not being carried
            for (int i = 0; i < e.numItems; ++i) {
                if (side == 1 && e.objects[i].state > 0) continue;
                const V3 h = item_collision_half(e.items[i].shape);
                const float cx = float(side == 0 ? e.items[i].off[0] + RE_LEFT[0] : e.objects[i].x) + 0.5f;
                const float cy = float(side == 0 ? e.items[i].off[1] + RE_LEFT[1] : e.objects[i].y) + 0.5f;
                const float cz = float(side == 0 ? e.items[i].off[2] + RE_LEFT[2] : e.objects[i].z) + 0.5f;
                Collider &c = out.c[out.n++];
                c.kind = 1;
                c.lo = v3(cx - h.x, (cy - h.y) - CAP_HH, cz - h.z);
                c.hi = v3(cx + h.x, (cy + h.y) + CAP_HH, cz + h.z);
            }
    }
    for (int i = 0; i < e.numObjects && e.scenario != SCN_REARRANGE && e.scenario != SCN_SOKOBAN; ++i) {
        if (e.objects[i].state > 0) continue;  // carried boxes: CF_NO_CONTACT_RESPONSE physics.hpp:76-85
        Collider &c = out.c[out.n++];
        const Object &o = e.objects[i];
        const float cx = float(o.x) + 0.5f, cy = float(o.y) + 0.5f + OBJ_COLL_YOFF, cz = float(o.z) + 0.5f;
        c.kind = 1;
        c.lo = v3(cx - OBJ_COLL_HALF, (cy - OBJ_COLL_HALF) - CAP_HH, cz - OBJ_COLL_HALF);
        c.hi = v3(cx + OBJ_COLL_HALF, (cy + OBJ_COLL_HALF) + CAP_HH, cz + OBJ_COLL_HALF);
    }
    for (int i = 0; i < e.numAgents; ++i) {
        if (i == self) continue;  // collision filter agent.cpp:63
        Collider &c = out.c[out.n++];
        c.kind = 2;
        c.lo = e.agents[i].pos;
        c.hi = v3(2 * CAP_HH, 0, 0);
    }
}

// [3P] btCollisionWorld::objectQuerySingle + ClosestConvexResultCallback, with the controller's
// KinematicClosestNotMeConvexResultCallback filter (kinematic_character_controller.cpp:53-93).
static bool sweep(const Colliders &cs, V3 from, V3 to, V3 up, float minSlopeDot, float *fraction, V3 *normal)
{
    const V3 d = to - from;
    float best = 1.0f;
    bool hit = false;
    for (int i = 0; i < cs.n; ++i) {
        if (cs.c[i].kind == 0) continue;
        float f; V3 n;
        if (!convex_cast(cs.c[i], from, d, &f, &n)) continue;
        if (!(len2(n) > 0.0001f)) continue;
        if (!(f < best)) continue;
        if (dot(up, n) < minSlopeDot) continue;
        best = f; *normal = n; hit = true;
    }
    *fraction = best;
    return hit;
}

// kinematic_character_controller.cpp:156-221.  [3P] the manifold is restated as one exact
// closest-point pair per overlapping object.
static bool recover_from_penetration(const Colliders &cs, V3 &pos)
{
    for (int i = 0; i < cs.n; ++i) {
        if (cs.c[i].kind == 0) continue;
        const Closest c = closest(cs.c[i], pos, CAP_R, 2 * CAP_R);
        if (c.dist < -MAX_PEN_DEPTH) {
            const float push = -c.dist;
            pos = v3(pos.x + c.n.x * push, pos.y + c.n.y * push, pos.z + c.n.z * push);
            return true;
        }
    }
    return false;
}

static bool on_ground(const Agent &a)
{   // kinematic_character_controller.cpp:679-682
    return (fabsf(a.vvel) < SIMD_EPS) && (fabsf(a.voffset) < SIMD_EPS);
}

// kinematic_character_controller.cpp:753-792
static void set_acceleration(Agent &a, V3 acc, float dt)
{
    const bool isOnGround = on_ground(a);
    const float accMag = sqrtf(len2(acc));
    const float currMax = isOnGround ? MAX_ACCEL : MAX_AIR_ACCEL;
    if (!(len2(acc) < SIMD_EPS * SIMD_EPS)) {  // !fuzzyZero
        const float k = currMax / accMag;
        acc = acc * k;
    }
    if (isOnGround) {
        a.hvx += acc.x * dt;
        a.hvz += acc.z * dt;
        const float speed = sqrtf(a.hvx * a.hvx + a.hvz * a.hvz);
        if (speed > MAX_H_SPEED) {
            const float dv = EXCEED_DECEL * dt;
            const float k = (speed - dv > MAX_H_SPEED) ? (speed - dv) / speed : MAX_H_SPEED / speed;
            a.hvx *= k; a.hvz *= k;
        }
    } else {
        const float speed = sqrtf(a.hvx * a.hvx + a.hvz * a.hvz);
        const float nx = a.hvx + acc.x * dt, nz = a.hvz + acc.z * dt;
        const float newSpeed = sqrtf(nx * nx + nz * nz);
        if (newSpeed <= MAX_AIR_SPEED || newSpeed < speed) { a.hvx = nx; a.hvz = nz; }
    }
}

static V3 lerp3(V3 a, V3 b, float rt)
{   // [3P] btVector3::setInterpolate3
    const float s = 1.0f - rt;
    return v3(s * a.x + rt * b.x, s * a.y + rt * b.y, s * a.z + rt * b.z);
}

// kinematic_character_controller.cpp:519-602 (preStep + playerStep) with stepUp :223-304,
// stepForwardAndStrafe :337-393, stepDown :400-442, updateTargetPositionBasedOnCollision :313-329
static void player_step(Env &e, int idx, float dt)
{
    Agent &a = e.agents[idx];
    Colliders cs;
    build_colliders(e, idx, cs);

    V3 cur = a.pos, target = a.pos;
    const V3 original = cur;
    const V3 UP = v3(0, 1, 0);

    const bool wasOnGround = on_ground(a);
    a.vvel -= GRAVITY * dt;
    if (a.vvel > 0.0f && a.vvel > a.jump_speed) a.vvel = a.jump_speed;
    if (a.vvel < 0.0f && fabsf(a.vvel) > fabsf(FALL_SPEED)) a.vvel = -fabsf(FALL_SPEED);
    a.voffset = a.vvel * dt;

    // ---- stepUp
    {
        const float stepHeight = (a.vvel < 0.0f) ? STEP_HEIGHT : 0.0f;
        const V3 start = cur;
        target = v3(cur.x, cur.y + stepHeight + (a.voffset > 0.0f ? a.voffset : 0.0f), cur.z);
        cur = target;
        float f; V3 n;
        if (sweep(cs, start, target, v3(0, -1, 0), MAX_SLOPE_COS, &f, &n)) {
            if (dot(n, UP) > 0.0f) {
                a.step_offset = stepHeight * f;
                cur = lerp3(cur, target, f);
            }
            int loops = 0;
            while (recover_from_penetration(cs, cur)) {
                if (++loops > 4) break;
            }
            target = cur;
            if (a.voffset > 0) { a.voffset = 0.0f; a.vvel = 0.0f; a.step_offset = STEP_HEIGHT; }
        } else {
            a.step_offset = stepHeight;
            cur = target;
        }
    }

    // ---- stepForwardAndStrafe
    {
        const V3 hv = v3(a.hvx, 0, a.hvz);
        target = v3(cur.x + hv.x * dt, cur.y + hv.y * dt, cur.z + hv.z * dt);
        int maxIter = 10;
        while (maxIter-- > 0) {
            const V3 negDir = cur - target;
            float f = 1.0f; V3 n = v3(0, 0, 0);
            bool hit = false;
            if (!(cur.x == target.x && cur.y == target.y && cur.z == target.z)) hit = sweep(cs, cur, target, negDir, 0.0f, &f, &n);
            if (!hit) break;
            // updateTargetPositionBasedOnCollision
            V3 dir = target - cur;
            const float movLen = sqrtf(len2(dir));
            if (movLen > SIMD_EPS) {
                dir = dir * (1.0f / movLen);
                const float mag = dot(dir, n);
                const V3 par = n * mag;
                const V3 perp = dir - par;
                target = cur;
                target = target + perp * movLen;
                target = target + par * (movLen * f);
            }
            V3 cd = target - cur;
            const float dist2 = len2(cd);
            if (dist2 > 0.0001f) {
                cd = cd * (1.0f / sqrtf(dist2));
                if (dot(cd, hv) <= 0.0f) { target = cur; break; }
            } else { target = cur; break; }
        }
        cur = target;
    }

    // ---- stepDown
    {
        float downVel = (a.vvel < 0.0f) ? -a.vvel : 0.0f;
        if (downVel > 0.0f && downVel > FALL_SPEED && (wasOnGround || !a.was_jumping)) downVel = FALL_SPEED;
        target = v3(target.x, target.y - (a.step_offset + downVel * dt), target.z);
        float f; V3 n;
        if (sweep(cs, cur, target, UP, MAX_SLOPE_COS, &f, &n)) {
            cur = lerp3(cur, target, f);
            a.vvel = 0.0f; a.voffset = 0.0f; a.was_jumping = 0;
        } else cur = target;
    }

    a.hvx = (cur.x - original.x) / dt;
    a.hvz = (cur.z - original.z) / dt;

    int loops = 0;
    while (recover_from_penetration(cs, cur)) {
        if (++loops > 4) break;
    }
    a.pos = cur;

    const float speed = sqrtf(a.hvx * a.hvx + a.hvz * a.hvz);
    if (on_ground(a)) {
        if (speed - NORMAL_DECEL * dt < 0) { a.hvx = 0; a.hvz = 0; }
        else { const float k = (speed - NORMAL_DECEL * dt) / speed; a.hvx *= k; a.hvz *= k; }
    }
}

// ------------------------------------------------------------------------------------------------
// Agent frames
// ------------------------------------------------------------------------------------------------
struct Cam {
    V3 eye;
    float c[3][3];  // columns = camera right / up / back in world space
};

static Cam camera_of(const Agent &a)
{   // agent.cpp:33 (+0.41), :95 (+0.05), :110-126 (pitch about local X)
    Cam cam;
    cam.eye = v3(a.pos.x, (a.pos.y + 0.05f) + 0.41f, a.pos.z);
    float sp, cp;
    mv_sincos(a.pitch, &sp, &cp);
    cam.c[0][0] = a.m00; cam.c[0][1] = a.m02 * sp; cam.c[0][2] = a.m02 * cp;
    cam.c[1][0] = 0.0f;  cam.c[1][1] = cp;         cam.c[1][2] = -sp;
    cam.c[2][0] = a.m20; cam.c[2][1] = a.m22 * sp; cam.c[2][2] = a.m22 * cp;
    return cam;
}

static V3 cam_to_world(const Cam &cam, V3 v)
{
    return v3((cam.c[0][0] * v.x + cam.c[0][1] * v.y) + cam.c[0][2] * v.z + cam.eye.x,
              (cam.c[1][0] * v.x + cam.c[1][1] * v.y) + cam.c[1][2] * v.z + cam.eye.y,
              (cam.c[2][0] * v.x + cam.c[2][1] * v.y) + cam.c[2][2] * v.z + cam.eye.z);
}

static void voxel_of(V3 p, int out[3])
{   // voxel_grid.hpp:18-21,144-149 (origin 0, voxelSize 1)
    out[0] = (int)lroundf(floorf(p.x)); out[1] = (int)lroundf(floorf(p.y)); out[2] = (int)lroundf(floorf(p.z));
}

// ------------------------------------------------------------------------------------------------
// Rewards -- scenario.hpp:259-298
// ------------------------------------------------------------------------------------------------
static void reward_agent(Env &e, int key, int idx, float mult) { e.agents[idx].last_reward += e.agents[idx].shaping[key] * mult; }
static void reward_team(Env &e, int key, int idx, float mult)
{
    reward_agent(e, key, idx, mult * (1 - e.agents[idx].shaping[0]));
    for (int i = 0; i < e.numAgents; ++i)
        e.agents[i].last_reward += e.agents[i].shaping[key] * e.agents[i].shaping[0] * mult / float(e.numAgents);
}

// VoxelGrid lookups.  TowerBuilding: dense chunk.  Obstacles: the level is a long chain, its voxels are
// answered from the merged layout boxes / terrain boxes (same set of voxels the reference's hash map holds).
static bool solid_at(const Env &e, int x, int y, int z)
{
    if (e.scenario == SCN_TOWER || e.scenario == SCN_REARRANGE) return (e.vox(x, y, z) & VX_SOLID) != 0;
    if (e.scenario == SCN_COLLECT)   // heightfield: floor at y == 0, landscape columns above it
        return x >= 0 && x < HM_DIM && z >= 0 && z < HM_DIM && y >= 0 && y <= e.heightmap[x * HM_DIM + z];
    for (int i = 0; i < e.numBoxes; ++i) {
        const Box &b = e.boxes[i];
        if ((b.type & VX_SOLID) && x >= b.min[0] && x < b.max[0] && y >= b.min[1] && y < b.max[1] && z >= b.min[2] && z < b.max[2]) return true;
    }
    return false;
}
static int terrain_at(const Env &e, int x, int y, int z)
{
    int t = 0;
    for (int i = 0; i < e.numTerrain; ++i) {
        const TerrainBox &b = e.terrain[i];
        if (x >= b.min[0] && x < b.max[0] && y >= b.min[1] && y < b.max[1] && z >= b.min[2] && z < b.max[2]) t |= b.type;
    }
    return t;
}

static int object_at(const Env &e, int x, int y, int z)
{
    for (int i = 0; i < e.numObjects; ++i)
        if (e.objects[i].state == 0 && e.objects[i].x == x && e.objects[i].y == y && e.objects[i].z == z) return i;
    return -1;
}

// component_object_stacking.hpp:58-168 + TowerBuilding callbacks scenario_tower_building.cpp:201-225
// RearrangeScenario::checkDone, scenario_rearrange.cpp:165-180
static void rearrange_check_done(Env &e, int idx)
{
    const int matches = rearrange_matching(e);
    if (matches > e.numPlatforms) {
        reward_team(e, 1, idx, 1);
        e.numPlatforms = matches;   // maxMatchingObjects
    }
    if (matches >= e.numItems && !e.solved) {
        e.solved = 1;
        reward_team(e, 2, idx, 1);
        e.episodeSec = std::max(e.episodeSec, e.episodeLen - 0.3f);   // doneWithTimer()
    }
}

static void on_interact(Env &e, int idx)
{
    Agent &a = e.agents[idx];
    const Cam cam = camera_of(a);
    if (a.carrying >= 0) {
        const V3 t = cam_to_world(cam, v3(0.0f, -0.44f + -0.3f, -1.0f));
        int vox[3];
        voxel_of(t, vox);
        bool collidesWithAgent = false;
        for (int j = 0; j < e.numAgents; ++j) {
            if (j == idx) continue;
            int c[3];
            voxel_of(v3(e.agents[j].pos.x, e.agents[j].pos.y + 0.05f, e.agents[j].pos.z), c);
            if (c[0] == vox[0] && c[1] == vox[1] && c[2] == vox[2]) { collidesWithAgent = true; break; }
        }
        bool placeable, empty, canPlace;
        const bool chunked = e.scenario == SCN_TOWER || e.scenario == SCN_REARRANGE;
        if (chunked) {
            // Dense chunk instead of the reference's unbounded hash map: cells outside the chunk in
            // x/z/+y are refused (deviation); below the chunk everything is empty.
            placeable = vox[0] >= 0 && vox[0] < CX && vox[2] >= 0 && vox[2] < CZ && vox[1] < CY;
            const uint8_t v = e.vox(vox[0], vox[1], vox[2]);
            empty = !(v & VX_SOLID) && !(v & VX_OBJECT);
            if (e.scenario == SCN_TOWER) canPlace = in_building_zone(e, vox[0], vox[2]);   // scenario_tower_building.cpp:201-204
            else canPlace = std::abs(vox[0] - RE_RIGHT[0]) <= 2 && std::abs(vox[2] - RE_RIGHT[2]) <= 2;   // scenario_rearrange.cpp:130-134
        } else {
            placeable = vox[1] > -120 && vox[1] < 120;         // int8 object coordinates
            empty = !solid_at(e, vox[0], vox[1], vox[2]) && object_at(e, vox[0], vox[1], vox[2]) < 0;
            canPlace = true;                                   // ObjectStackingCallbacks default
        }
        if (placeable && empty && !collidesWithAgent && canPlace) {
            for (;;) {
                const int by = vox[1] - 1;
                if (by < -30) break;
                if (chunked) {
                    const uint8_t vb = e.vox(vox[0], by, vox[2]);
                    if ((vb & VX_SOLID) || (vb & VX_OBJECT)) break;
                } else if (solid_at(e, vox[0], by, vox[2]) || object_at(e, vox[0], by, vox[2]) >= 0) break;
                vox[1] = by;
            }
            Object &o = e.objects[a.carrying];
            o.x = vox[0]; o.y = vox[1]; o.z = vox[2]; o.state = 0;
            if (Env::inChunk(vox[0], vox[1], vox[2])) e.chunk[Env::cell(vox[0], vox[1], vox[2])] |= VX_OBJECT;
            a.carrying = -1;
            if (e.scenario == SCN_TOWER) {   // placedObject :206-214
                const float newReward = tower_reward(e);
                const float delta = newReward - e.bzReward;
                e.bzReward = newReward;
                reward_team(e, 3, idx, delta);
                e.highestTower = std::max(e.highestTower, vox[1] - 1 + 1);
            }
            if (e.scenario == SCN_REARRANGE) rearrange_check_done(e, idx);   // placedObject :153-157
        }
    } else {
        const V3 pickup = cam_to_world(cam, v3(0.0f, -0.44f, -1.0f));
        int vox[3];
        voxel_of(pickup, vox);
        for (int h = 0; h <= 1; ++h) {
            const int oi = object_at(e, vox[0], vox[1], vox[2]);
            const bool above = object_at(e, vox[0], vox[1] + 1, vox[2]) >= 0;
            if (oi >= 0 && !above) {
                e.objects[oi].state = 1 + idx;
                if (Env::inChunk(vox[0], vox[1], vox[2])) e.chunk[Env::cell(vox[0], vox[1], vox[2])] &= ~VX_OBJECT;
                a.carrying = oi;
                // pickedObject :216-225 (TowerBuilding only; the Obstacles callbacks are no-ops)
                if (e.scenario == SCN_TOWER && !a.picked_up) { reward_agent(e, 1, idx, 1); a.picked_up = 1; }
                if (e.scenario == SCN_REARRANGE) rearrange_check_done(e, idx);   // pickedObject :159-163
                break;
            }
            vox[1] += 1;
        }
    }
}

static int sokoban_box_at(const Env &e, int x, int y, int z)
{
    for (int i = 0; i < e.numObjects; ++i)
        if (e.objects[i].x == x && e.objects[i].y == y && e.objects[i].z == z) return i;
    return -1;
}
static int sokoban_terrain(const Env &e, int x, int y, int z)
{
    if (y != 1 || x < 0 || x >= SOKO_DIM || z < 0 || z >= SOKO_DIM) return 0;
    return e.soko[x * SOKO_DIM + z];
}

// SokobanScenario::step :172-236
static void sokoban_step(Env &e)
{
    auto cell_of = [&](V3 p, int out[3]) { out[0] = int(floorf(p.x / e.voxelSize)); out[1] = int(floorf(p.y / e.voxelSize)); out[2] = int(floorf(p.z / e.voxelSize)); };
    for (int i = 0; i < e.numAgents; ++i) {
        Agent &a = e.agents[i];
        if (!(a.action & (1 << 8))) continue;
        const Cam cam = camera_of(a);
        const V3 t = cam_to_world(cam, v3(0.0f, -0.44f, -1.0f));   // interactLocation
        int boxPos[3], agentPos[3];
        cell_of(t, boxPos);
        const int bi = sokoban_box_at(e, boxPos[0], boxPos[1], boxPos[2]);
        if (bi < 0) continue;
        cell_of(v3(a.pos.x, a.pos.y + 0.05f, a.pos.z), agentPos);
        const int d[3] = {boxPos[0] - agentPos[0], boxPos[1] - agentPos[1], boxPos[2] - agentPos[2]};
        if (std::abs(d[0]) + std::abs(d[1]) + std::abs(d[2]) != 1) continue;   // only from the adjacent cell
        const int want[3] = {boxPos[0] + d[0], boxPos[1] + d[1], boxPos[2] + d[2]};
        bool occupied = false;
        for (int j = 0; j < e.numAgents; ++j) {
            int c[3];
            cell_of(v3(e.agents[j].pos.x, e.agents[j].pos.y + 0.05f, e.agents[j].pos.z), c);
            if (c[0] == want[0] && c[1] == want[1] && c[2] == want[2]) { occupied = true; break; }
        }
        if (occupied) continue;
        const int fromTerrain = sokoban_terrain(e, boxPos[0], boxPos[1], boxPos[2]), toTerrain = sokoban_terrain(e, want[0], want[1], want[2]);
        if (toTerrain == SOKO_WALL || sokoban_box_at(e, want[0], want[1], want[2]) >= 0) continue;
        e.objects[bi].x = want[0]; e.objects[bi].y = want[1]; e.objects[bi].z = want[2];
        if (fromTerrain != SOKO_GOAL && toTerrain == SOKO_GOAL) {
            ++e.highestTower;   // numBoxesOnGoal
            reward_team(e, 1, i, 1);
            if (e.highestTower == e.numObjects && !e.solved) {
                e.solved = 1;
                reward_team(e, 3, i, 1);
                e.episodeSec = std::max(e.episodeSec, e.episodeLen - 0.3f);   // doneWithTimer()
            }
        } else if (fromTerrain == SOKO_GOAL && toTerrain != SOKO_GOAL) {
            --e.highestTower;
            reward_team(e, 2, i, 1);
        }
    }
}

// ------------------------------------------------------------------------------------------------
// Env::step -- env/src/env.cpp:83-152
// ------------------------------------------------------------------------------------------------
static void env_step(Env &e)
{
    const float dt = DT;
    for (int i = 0; i < e.numAgents; ++i) e.agents[i].last_reward = 0.0f;

    for (int i = 0; i < e.numAgents; ++i) {
        Agent &a = e.agents[i];
        const int act = a.action;
        // forwardDirection/strafeLeftDirection agent.cpp:135-150
        V3 fwd = v3(a.m20, 0.0f, -a.m22);
        fwd = fwd * (1.0f / sqrtf(len2(fwd)));
        V3 left = v3(-a.m00, 0.0f, a.m02);
        left = left * (1.0f / sqrtf(len2(left)));
        V3 acc = v3(0, 0, 0);
        if (act & (1 << 3)) acc = acc + fwd;
        else if (act & (1 << 4)) acc = acc - fwd;
        if (act & (1 << 1)) acc = acc + left;
        else if (act & (1 << 2)) acc = acc - left;

        if (act & ((1 << 5) | (1 << 6))) {  // rotateYAxis agent.cpp:128-133
            float c, s;
            yaw_matrix(ROTATE_RAD * dt, &c, &s);
            if (act & (1 << 5)) { /* look left: +angle */ }
            else s = -s;
            const float n00 = a.m00 * c + a.m02 * (-s), n02 = a.m00 * s + a.m02 * c;
            const float n20 = a.m20 * c + a.m22 * (-s), n22 = a.m20 * s + a.m22 * c;
            a.m00 = n00; a.m02 = n02; a.m20 = n20; a.m22 = n22;
        }
        if (act & (1 << 10)) {  // lookUp agent.cpp:110-116
            a.pitch += ROTATE_X_RAD * dt;
            a.pitch = std::min(e.p_verticalLookLimitRad, a.pitch);
        } else if (act & (1 << 9)) {  // lookDown :118-126
            a.pitch -= ROTATE_X_RAD * dt * 1.1f;
            a.pitch = std::max(-e.p_verticalLookLimitRad, a.pitch);
        }

        set_acceleration(a, acc, dt);

        if ((act & (1 << 7)) && on_ground(a)) {  // agent.cpp:157-161, controller jump() :625-634
            a.jump_speed = sqrtf(6.2f * 6.2f);
            a.vvel = a.jump_speed;
            a.was_jumping = 1;
        }
    }

    // bWorld.stepSimulation(dt, 1, dt): controllers run in agent order (env.cpp:126)
    for (int i = 0; i < e.numAgents; ++i) player_step(e, i, dt);

    // scenario->step(): objectStacking, fallDetection, zone reward (scenario_tower_building.cpp:179-199)
    const bool hex = e.scenario == SCN_HEX_MEMORY || e.scenario == SCN_HEX_EXPLORE;
    for (int i = 0; i < e.numAgents; ++i)
        if (e.scenario != SCN_SOKOBAN && e.scenario != SCN_EMPTY && !hex && (e.agents[i].action & (1 << 8))) on_interact(e, i);   // (Sokoban, Empty, Hex*: no ObjectStackingComponent)

    auto reset_agent = [&](Agent &a) {   // FallDetectionComponent::resetAgent :45-55 + controller warp() :509-517
        int p[3] = {a.spawn[0], a.spawn[1], a.spawn[2]};
        while (solid_at(e, p[0], p[1], p[2]) && p[1] < 1000) ++p[1];
        a.pos = v3(float(p[0]) + 0.5f, float(p[1]) + 0.5f, float(p[2]) + 0.5f);
        a.m00 = 1; a.m02 = 0; a.m20 = 0; a.m22 = 1;  // warp(): xform.setIdentity()
        a.hvx = a.hvz = 0; a.vvel = 0;
    };
    for (int i = 0; i < e.numAgents; ++i)  // component_fall_detection.hpp:33-43
        if (e.scenario != SCN_REARRANGE && e.scenario != SCN_SOKOBAN && e.scenario != SCN_EMPTY && !hex && e.agents[i].pos.y + 0.05f < -20.0f) {   // (Rearrange, Sokoban, Empty, Hex*: no FallDetectionComponent)
            reset_agent(e.agents[i]);
            if (e.scenario == SCN_COLLECT) reward_agent(e, 2, i, 1);   // CollectScenario::agentFell, scenario_collect.cpp:214-218
        }

    if (e.scenario == SCN_COLLECT) {   // CollectScenario::step, scenario_collect.cpp:163-196
        for (int i = 0; i < e.numAgents; ++i) {
            Agent &a = e.agents[i];
            int vox[3];
            voxel_of(v3(a.pos.x, a.pos.y + 0.05f, a.pos.z), vox);
            for (int r = 0; r < e.numRewards; ++r) {
                RewardObj &ro = e.rewards[r];
                if (!ro.active || ro.x != vox[0] || ro.y != vox[1] || ro.z != vox[2]) continue;
                const int kind = ro.active;
                ro.active = 0;
                if (kind == 1) ++e.highestTower;                 // positiveRewardsCollected
                reward_team(e, kind == 1 ? 1 : 2, i, 1);
                if (e.highestTower >= e.numPlatforms && !e.solved) {
                    e.solved = 1;
                    e.episodeSec = std::max(e.episodeSec, e.episodeLen - 0.3f);   // doneWithTimer()
                    reward_team(e, 3, i, 1);
                }
                // vg.grid.remove(voxel) erases the whole cell: a movable box that was dropped into the diamond's cell
                // stays in the world (collision + drawable) but can no longer be found through the grid
                for (int o = 0; o < e.numObjects; ++o)
                    if (e.objects[o].state == 0 && e.objects[o].x == vox[0] && e.objects[o].y == vox[1] && e.objects[o].z == vox[2])
                        e.objects[o].state = -1;
            }
        }
    } else if (e.scenario == SCN_HEX_EXPLORE) {   // HexExploreScenario::step, scenario_hex_explore.cpp:43-58
        for (int i = 0; i < e.numAgents; ++i) {
            const Agent &a = e.agents[i];
            const V3 t = v3(a.pos.x, a.pos.y + 0.05f, a.pos.z);
            const float distance = sqrtf(len2(t - e.hexTarget));
            if (double(distance) < 1.2 && !e.solved) {
                e.solved = 1;
                e.episodeSec = std::max(e.episodeSec, e.episodeLen - 0.3f);   // doneWithTimer()
                reward_team(e, 1, i, 1);
                Env::HexObj &o = e.hexObjs[0];   // rewardObject->translate({1e3, 1e3, 1e3})
                o.pos = v3(o.pos.x + 1e3f, o.pos.y + 1e3f, o.pos.z + 1e3f);
                o.alive = 0;
                break;
            }
        }
    } else if (e.scenario == SCN_HEX_MEMORY) {    // HexMemoryScenario::step, scenario_hex_memory.cpp:84-127
        if (e.highestTower >= e.numPlatforms && !e.solved) {
            e.solved = 1;
            e.episodeSec = std::max(e.episodeSec, e.episodeLen - 0.3f);
        }
        for (int i = 0; i < e.numAgents; ++i) {
            const Agent &a = e.agents[i];
            const V3 t = v3(a.pos.x, a.pos.y + 0.05f, a.pos.z);
            int vox[3];
            voxel_of(t, vox);
            for (int dx = -1; dx <= 1; ++dx)
                for (int dz = -1; dz <= 1; ++dz)
                    for (Env::HexObj &o : e.hexObjs) {   // a voxel's list is in insertion (= index) order
                        if (!o.alive || o.vox[0] != vox[0] + dx || o.vox[1] != vox[1] || o.vox[2] != vox[2] + dz) continue;
                        const float distance = sqrtf(len2(o.pos - t));
                        if (!(distance < 1.0f)) continue;
                        reward_team(e, o.good ? 1 : 2, i, 1);
                        e.highestTower += o.good;
                        o.pos = v3(o.pos.x + 100.0f, o.pos.y + 100.0f, o.pos.z + 100.0f);   // still drawn, out of everybody's way
                        o.alive = 0;
                    }
        }
    } else if (e.scenario == SCN_EMPTY) {
        // EmptyScenario::step() {} (scenario_empty.hpp:22)
    } else if (e.scenario == SCN_SOKOBAN) {
        sokoban_step(e);
    } else if (e.scenario == SCN_REARRANGE) {
        // RearrangeScenario::step is objectStackingComponent.step only (:125-128)
    } else if (e.scenario == SCN_TOWER) {
        for (int i = 0; i < e.numAgents; ++i) {
            Agent &a = e.agents[i];
            if (a.carrying >= 0) {
                int vox[3];
                voxel_of(v3(a.pos.x, a.pos.y + 0.05f, a.pos.z), vox);
                if (in_building_zone(e, vox[0], vox[2]) && !a.visited_zone) {
                    reward_team(e, 2, i, 1);
                    a.visited_zone = 1;
                }
            }
        }
    } else {   // ObstaclesScenario::step, scenario_obstacles.cpp:197-239
        int numAgentsAtExit = 0;
        for (int i = 0; i < e.numAgents; ++i) {
            Agent &a = e.agents[i];
            int vox[3];
            voxel_of(v3(a.pos.x, a.pos.y + 0.05f, a.pos.z), vox);
            const int terrain = terrain_at(e, vox[0], vox[1], vox[2]);
            if (terrain & TERRAIN_EXIT) {
                ++numAgentsAtExit;
                if (!a.visited_zone) {
                    a.visited_zone = 1;   // agentReachedExit[i]
                    reward_team(e, 1, i, 1);
                    if (a.carrying >= 0) reward_team(e, 4, i, 1);
                }
            } else if (terrain & TERRAIN_LAVA) reset_agent(a);   // agentTouchedLava :274-278
            for (int r = 0; r < e.numRewards; ++r) {             // green diamonds :224-229
                RewardObj &ro = e.rewards[r];
                if (ro.active && ro.x == vox[0] && ro.y == vox[1] && ro.z == vox[2]) {
                    ro.active = 0;
                    reward_team(e, 3, i, 1);
                }
            }
        }
        if (numAgentsAtExit == e.numAgents && !e.solved) {
            e.solved = 1;
            e.episodeSec = std::max(e.episodeSec, e.episodeLen - 0.3f);   // doneWithTimer(), scenario.hpp:114-117
            for (int i = 0; i < e.numAgents; ++i) reward_agent(e, 2, i, 1);   // rewardAll
        }
    }

    e.episodeSec += dt;
    // updateUI scenario_default.hpp:164-169 ; remainingTimeFraction env.hpp:224-228
    e.barHalfWidth = std::max(0.0f, (e.episodeLen - e.episodeSec) / e.episodeLen) * 0.24f;
    if (e.episodeSec >= e.episodeLen) e.done = 1;

    for (int i = 0; i < e.numAgents; ++i) {
        e.agents[i].action = 0;
        e.agents[i].total_reward += e.agents[i].last_reward;
    }
    ++e.numFrames;
}

// ------------------------------------------------------------------------------------------------
// Software first-person renderer.  One primary ray per pixel centre; for opaque convex solids
// with depth test + back-face culling this selects the same surface as the GL rasteriser
// (magnum_env_renderer.cpp:288-340).  Shading = Magnum Shaders::Phong [3P] with the uniforms
// set at :200-203; projection = env_renderer.hpp:34-38 (hfov 100, aspect 128/72 regardless of
// the framebuffer size).
// ------------------------------------------------------------------------------------------------
static const float TAN_HALF_FOV = 1.19175359f;                      // tan(50 deg)
static const float TAN_HALF_FOV_Y = 1.19175359f / (128.0f / 72.0f);
static const float NEAR_Z = 0.01f, FAR_Z = 120.0f;

struct Prim {
    int kind;   // 1 = box in frame `frame`, 2 = vertical capsule in world, 3 = cone in world (lo = apex, hi = (radius, height, +1 apex up / -1 apex down)),
                // 4 / 5 / 6 = unit sphere / unit capsule (r 1, half-length 1) / unit capped cylinder (r 1, half-length 0.5), axis y,
                //             scaled by hi and centred at lo in frame `frame`
    int frame;  // -1 world axes, k in [0, MAX_AGENTS): camera frame of agent k, MAX_AGENTS + k: hex wall orientation k
    V3 lo, hi;  // box bounds in its frame; capsule: lo = centre, hi = (radius, halfLen, 0)
    unsigned color;
};

static void build_prims(const Env &e, int viewer, std::vector<Prim> &out)
{
    out.clear();
    for (int i = 0; i < e.numBoxes; ++i) {  // layout_utils.cpp:17-50
        const Box &b = e.boxes[i];
        if (!(b.type & VX_OPAQUE)) continue;
        Prim p; p.kind = 1; p.frame = -1;
        const float vs = e.voxelSize;
        p.lo = v3(float(b.min[0]) * vs, float(b.min[1]) * vs, float(b.min[2]) * vs);
        p.hi = v3(float(b.max[0]) * vs, float(b.max[1]) * vs, float(b.max[2]) * vs);
        p.color = b.slot == 0 ? e.layoutColor : e.wallColor;
        out.push_back(p);
    }
    if (e.scenario == SCN_SOKOBAN) {   // addEpisodeDrawables :243-294
        const float vs = e.voxelSize;
        for (int x = 0; x < e.L; ++x)            // wall caps (0.7 high, the walls themselves are not drawn) and goal pads
            for (int z = 0; z < e.W; ++z) {
                const int t = e.soko[x * SOKO_DIM + z];
                if (!t) continue;
                const float h = t == SOKO_WALL ? 0.35f : 0.025f;
                const V3 c = v3(vs * float(x) + vs / 2, vs + h, vs * float(z) + vs / 2);
                Prim p; p.kind = 1; p.frame = -1;
                p.lo = v3(c.x - 1.0f, c.y - h, c.z - 1.0f); p.hi = v3(c.x + 1.0f, c.y + h, c.z + 1.0f);
                p.color = t == SOKO_WALL ? 0xffa770 : 0x50c878;   // LIGHT_ORANGE / LIGHT_GREEN
                out.push_back(p);
            }
        for (int i = 0; i < e.numObjects; ++i) {   // the boxes
            const Object &o = e.objects[i];
            const float sx = (vs / 2) * 0.8f, sy = 0.45f * 0.8f;
            const V3 c = v3((float(o.x) + 0.5f) * vs, (float(o.y) + 0.2f) * vs, (float(o.z) + 0.5f) * vs);
            Prim p; p.kind = 1; p.frame = -1; p.color = 0x3a7fa6;   // DARK_BLUE
            p.lo = v3(c.x - sx, c.y - sy, c.z - sx); p.hi = v3(c.x + sx, c.y + sy, c.z + sx);
            out.push_back(p);
        }
    }
    if (e.scenario == SCN_HEX_MEMORY || e.scenario == SCN_HEX_EXPLORE) {
        for (const Env::HexBox &b : e.hexBoxes) {
            Prim p; p.kind = 1; p.frame = b.frame < 0 ? -1 : MAX_AGENTS + b.frame; p.lo = b.lo; p.hi = b.hi; p.color = b.color;
            out.push_back(p);
        }
        for (const Env::HexObj &o : e.hexObjs) {   // layout_utils.cpp:85-126 addSphere / addPillar / addDiamond
            Prim p; p.frame = -1; p.color = o.color;
            if (o.shape == HEX_SPHERE) { p.kind = 4; p.lo = o.pos; p.hi = o.scale; out.push_back(p); }
            else if (o.shape == HEX_PILLAR) {
                p.kind = 6; p.lo = o.pos; p.hi = o.scale; out.push_back(p);
                const float capY = 0.47f * o.scale.y;
                p.hi = v3(o.scale.x * 1.2f, 0.15f, o.scale.z * 1.2f);
                p.lo = v3(o.pos.x, o.pos.y + capY, o.pos.z); out.push_back(p);
                p.lo = v3(o.pos.x, o.pos.y - capY, o.pos.z); out.push_back(p);
            } else {
                p.kind = 3;
                p.lo = v3(o.pos.x, o.pos.y + 0.5f * o.scale.y, o.pos.z); p.hi = v3(o.scale.x, o.scale.y, 1.0f); out.push_back(p);
                p.lo = v3(o.pos.x, o.pos.y - 1.5f * o.scale.y, o.pos.z); p.hi = v3(o.scale.x, o.scale.y, -1.0f); out.push_back(p);
            }
        }
    }
    if (e.scenario == SCN_TOWER) {   // building zone slab, layout_utils.cpp:53-68
        Prim p; p.kind = 1; p.frame = -1;
        p.lo = v3(float(e.bz[0]), 1.0f, float(e.bz[2]));
        p.hi = v3(float(e.bz[1]), 1.0f + 0.05f, float(e.bz[3]));
        p.color = COLOR_BUILDING_ZONE;
        out.push_back(p);
    }
    for (int i = 0; i < e.numTerrain; ++i) {   // addTerrain, layout_utils.cpp:53-68: 0.05 thick slab on the box's floor
        const TerrainBox &t = e.terrain[i];
        Prim p; p.kind = 1; p.frame = -1;
        p.lo = v3(float(t.min[0]), float(t.min[1]), float(t.min[2]));
        p.hi = v3(float(t.max[0]), float(t.min[1]) + 0.05f, float(t.max[2]));
        p.color = t.type == TERRAIN_EXIT ? COLOR_EXIT_PAD : COLOR_RED;   // platforms.hpp:47-56
        out.push_back(p);
    }
    if (e.scenario == SCN_REARRANGE) {
        for (int i = 0; i < e.numStatic; ++i) {   // addStaticCollidingBox, layout_utils.cpp:70-83
            Prim p; p.kind = 1; p.frame = -1; p.lo = e.statics[i].lo; p.hi = e.statics[i].hi; p.color = e.statics[i].color;
            out.push_back(p);
        }
        for (int side = 0; side < 2; ++side)      // arrangementDrawables :203-263: target (left), then the movable copy (right)
            for (int i = 0; i < e.numItems; ++i) {
                const Env::ArrItem &it = e.items[i];
                V3 sc = item_draw_scale(it.shape);
                Prim p; p.color = it.color;
                V3 c;
                if (side == 0) { p.frame = -1; c = v3(float(it.off[0] + RE_LEFT[0]) + 0.5f, float(it.off[1] + RE_LEFT[1]) + 0.5f, float(it.off[2] + RE_LEFT[2]) + 0.5f); }
                else if (e.objects[i].state <= 0) { p.frame = -1; c = v3(float(e.objects[i].x) + 0.5f, float(e.objects[i].y) + 0.5f, float(e.objects[i].z) + 0.5f); }
                else {   // carried: scaled by 0.78 and hung in front of the carrier's camera (component_object_stacking.hpp:146-152)
                    p.frame = e.objects[i].state - 1;
                    c = v3(0.0f, -0.44f + -0.3f, -1.0f);
                    sc = v3(sc.x * CARRY_SCALE, sc.y * CARRY_SCALE, sc.z * CARRY_SCALE);
                }
                if (it.shape == SHAPE_BOX) { p.kind = 1; p.lo = v3(c.x - sc.x, c.y - sc.y, c.z - sc.z); p.hi = v3(c.x + sc.x, c.y + sc.y, c.z + sc.z); }
                else { p.kind = it.shape == SHAPE_SPHERE ? 4 : it.shape == SHAPE_CAPSULE ? 5 : 6; p.lo = c; p.hi = sc; }
                out.push_back(p);
            }
    }
    for (int i = 0; i < e.numObjects && e.scenario != SCN_REARRANGE && e.scenario != SCN_SOKOBAN; ++i) {  // component_object_stacking.hpp:170-198, :146-152
        const Object &o = e.objects[i];
        Prim p; p.kind = 1; p.color = COLOR_MOVABLE_BOX;
        if (o.state <= 0) {
            p.frame = -1;
            const V3 c = v3(float(o.x) + 0.5f, float(o.y) + 0.5f, float(o.z) + 0.5f);
            p.lo = v3(c.x - OBJ_HALF, c.y - OBJ_HALF, c.z - OBJ_HALF);
            p.hi = v3(c.x + OBJ_HALF, c.y + OBJ_HALF, c.z + OBJ_HALF);
        } else {
            p.frame = o.state - 1;
            const float hh = OBJ_HALF * CARRY_SCALE;
            const V3 c = v3(0.0f, -0.44f + -0.3f, -1.0f);
            p.lo = v3(c.x - hh, c.y - hh, c.z - hh);
            p.hi = v3(c.x + hh, c.y + hh, c.z + hh);
        }
        out.push_back(p);
    }
    for (int i = 0; i < e.numRewards; ++i) {   // addDiamond, layout_utils.cpp:114-126 + scenario_obstacles.cpp:254
        const RewardObj &r = e.rewards[i];
        if (!r.active) continue;               // collected diamonds are translated far away (:227)
        // Obstacles: scale (0.17, 0.45, 0.17) * 0.8 at y + 0.7; Collect (scenario_collect.cpp:192,208): unscaled at y + 0.8
        const bool collect = e.scenario == SCN_COLLECT;
        const float sx = collect ? 0.17f : 0.17f * 0.8f, sy = collect ? 0.45f : 0.45f * 0.8f;
        const V3 c = v3(float(r.x) + 0.5f, float(r.y) + (collect ? 0.8f : 0.7f), float(r.z) + 0.5f);
        Prim up; up.kind = 3; up.frame = -1; up.color = r.active == 2 ? COLOR_RED : COLOR_GREEN;
        up.lo = v3(c.x, c.y + 0.5f * sy, c.z); up.hi = v3(sx, sy, 1.0f);
        out.push_back(up);
        Prim dn = up;
        dn.lo = v3(c.x, c.y - 1.5f * sy, c.z); dn.hi = v3(sx, sy, -1.0f);
        out.push_back(dn);
    }
    for (int k = 0; k < e.numAgents; ++k) {  // scenario_default.hpp:99-170
        if (k != viewer) {
            Prim body; body.kind = 2; body.frame = -1;
            const Agent &a = e.agents[k];
            body.lo = v3(a.pos.x, (a.pos.y + 0.05f) + 0.09f, a.pos.z);
            body.hi = v3(0.35f, 0.36f, 0.0f);
            body.color = AGENT_COLORS[k % 7];
            out.push_back(body);
            Prim eyes; eyes.kind = 1; eyes.frame = k;
            eyes.lo = v3(-0.25f, -0.12f, -0.19f - 0.2f);
            eyes.hi = v3(0.25f, 0.12f, -0.19f + 0.2f);
            eyes.color = COLOR_AGENT_EYES;
            out.push_back(eyes);
        }
        Prim bar; bar.kind = 1; bar.frame = k;
        bar.lo = v3(-e.barHalfWidth, -0.131f - 0.0015f, -0.2f - 0.001f);
        bar.hi = v3(e.barHalfWidth, -0.131f + 0.0015f, -0.2f + 0.001f);
        bar.color = COLOR_UI_BAR;
        out.push_back(bar);
    }
}

// slab test; returns entry t and entry-face normal (back faces culled: the camera inside a
// box sees nothing of it)
static bool ray_box(V3 o, V3 d, V3 lo, V3 hi, float *t_out, V3 *n_out)
{
    float tEnter = -INFINITY, tExit = INFINITY;
    int axis = -1;
    const float oo[3] = {o.x, o.y, o.z}, dd[3] = {d.x, d.y, d.z}, l[3] = {lo.x, lo.y, lo.z}, h[3] = {hi.x, hi.y, hi.z};
    for (int k = 0; k < 3; ++k) {
        if (dd[k] == 0.0f) {
            if (oo[k] < l[k] || oo[k] > h[k]) return false;
        } else {
            const float inv = 1.0f / dd[k];
            const float t1 = (l[k] - oo[k]) * inv, t2 = (h[k] - oo[k]) * inv;
            const float tn = std::min(t1, t2), tf = std::max(t1, t2);
            if (tn > tEnter) { tEnter = tn; axis = k; }
            tExit = std::min(tExit, tf);
        }
    }
    if (axis < 0 || tEnter > tExit || tEnter < NEAR_Z || tEnter > FAR_Z) return false;
    V3 n = v3(0, 0, 0);
    const float sgn = dd[axis] > 0 ? -1.0f : 1.0f;
    if (axis == 0) n.x = sgn; else if (axis == 1) n.y = sgn; else n.z = sgn;
    *t_out = tEnter; *n_out = n;
    return true;
}

// vertical capsule (radius r, segment half-length hl): cylinder side + two spheres
static bool ray_capsule(V3 o, V3 d, V3 c, float r, float hl, float *t_out, V3 *n_out)
{
    bool hit = false;
    float best = INFINITY; V3 bn = v3(0, 0, 0);
    const float ox = o.x - c.x, oz = o.z - c.z;
    const float A = d.x * d.x + d.z * d.z;
    if (A > 0.0f) {
        const float B = ox * d.x + oz * d.z;
        const float C = (ox * ox + oz * oz) - r * r;
        const float disc = B * B - A * C;
        if (disc >= 0.0f) {
            const float t = (-B - sqrtf(disc)) / A;
            const float y = o.y + t * d.y;
            if (t >= NEAR_Z && t <= FAR_Z && y >= c.y - hl && y <= c.y + hl) {
                hit = true; best = t;
                const float inv = 1.0f / r;
                bn = v3((ox + t * d.x) * inv, 0.0f, (oz + t * d.z) * inv);
            }
        }
    }
    for (int s = 0; s < 2; ++s) {
        const float cy = s == 0 ? c.y - hl : c.y + hl;
        const float oy = o.y - cy;
        const float A3 = (d.x * d.x + d.y * d.y) + d.z * d.z;
        const float B3 = (ox * d.x + oy * d.y) + oz * d.z;
        const float C3 = ((ox * ox + oy * oy) + oz * oz) - r * r;
        const float disc = B3 * B3 - A3 * C3;
        if (disc >= 0.0f) {
            const float t = (-B3 - sqrtf(disc)) / A3;
            const float y = o.y + t * d.y;
            const bool capSide = s == 0 ? (y <= cy) : (y >= cy);
            if (t >= NEAR_Z && t <= FAR_Z && capSide && t < best) {
                hit = true; best = t;
                const float inv = 1.0f / r;
                bn = v3((ox + t * d.x) * inv, (oy + t * d.y) * inv, (oz + t * d.z) * inv);
            }
        }
    }
    if (hit) { *t_out = best; *n_out = bn; }
    return hit;
}

// Unit capped cylinder (Primitives::cylinderSolid(.., halfLength 0.5, CapEnds), render_utils.cpp:30): radius 1, |y| <= hl.
static bool ray_cylinder_unit(V3 o, V3 d, float hl, float *t_out, V3 *n_out)
{
    bool hit = false;
    float best = INFINITY; V3 bn = v3(0, 0, 0);
    const float A = d.x * d.x + d.z * d.z;
    if (A > 0.0f) {
        const float B = o.x * d.x + o.z * d.z;
        const float C = (o.x * o.x + o.z * o.z) - 1.0f;
        const float disc = B * B - A * C;
        if (disc >= 0.0f) {
            const float t = (-B - sqrtf(disc)) / A;
            const float y = o.y + t * d.y;
            if (t >= NEAR_Z && t <= FAR_Z && y >= -hl && y <= hl) { hit = true; best = t; bn = v3(o.x + t * d.x, 0.0f, o.z + t * d.z); }
        }
    }
    for (int s = 0; s < 2; ++s) {   // caps: entered from outside only (back faces are culled)
        const float cy = s == 0 ? -hl : hl;
        const bool entering = s == 0 ? (d.y > 0.0f && o.y < cy) : (d.y < 0.0f && o.y > cy);
        if (!entering) continue;
        const float t = (cy - o.y) / d.y;
        const float x = o.x + t * d.x, z = o.z + t * d.z;
        if (t >= NEAR_Z && t <= FAR_Z && x * x + z * z <= 1.0f && t < best) { hit = true; best = t; bn = v3(0.0f, s == 0 ? -1.0f : 1.0f, 0.0f); }
    }
    if (hit) { *t_out = best; *n_out = bn; }
    return hit;
}

// kinds 4 / 5 / 6: the ray is taken into the shape's unit space (the scale is diagonal, so t is unchanged), the unit-space
// normal comes back through the inverse-transpose scale
static bool ray_scaled_shape(int kind, V3 o, V3 d, V3 centre, V3 scale, float *t_out, V3 *n_out)
{
    const V3 oo = v3((o.x - centre.x) / scale.x, (o.y - centre.y) / scale.y, (o.z - centre.z) / scale.z);
    const V3 dd = v3(d.x / scale.x, d.y / scale.y, d.z / scale.z);
    V3 nl;
    bool hit;
    if (kind == 6) hit = ray_cylinder_unit(oo, dd, 0.5f, t_out, &nl);
    else hit = ray_capsule(oo, dd, v3(0, 0, 0), 1.0f, kind == 5 ? 1.0f : 0.0f, t_out, &nl);
    if (!hit) return false;
    V3 n = v3(nl.x / scale.x, nl.y / scale.y, nl.z / scale.z);
    n = n * (1.0f / sqrtf(len2(n)));
    *n_out = n;
    return true;
}

// Open cone (no base cap, Primitives::coneSolid without CapEnd): apex a, axis +-y (dirSign +1: apex up),
// height h, base radius r.  Only the outside is visible (back-face culling).
static bool ray_cone(V3 o, V3 d, V3 a, float r, float h, float dirSign, float *t_out, V3 *n_out)
{
    const float k = (r / h) * (r / h);
    const float ox = o.x - a.x, oz = o.z - a.z;
    const float s0 = dirSign * (a.y - o.y);      // distance from the apex along the axis at t = 0
    const float ds = -dirSign * d.y;             // its derivative
    const float A = (d.x * d.x + d.z * d.z) - k * (ds * ds);
    const float B = (ox * d.x + oz * d.z) - k * (s0 * ds);
    const float C = (ox * ox + oz * oz) - k * (s0 * s0);
    if (A == 0.0f) return false;
    const float disc = B * B - A * C;
    if (!(disc >= 0.0f)) return false;
    const float sq = sqrtf(disc);
    bool hit = false;
    float best = INFINITY; V3 bn = v3(0, 0, 0);
    for (int i = 0; i < 2; ++i) {
        const float t = (i == 0 ? (-B - sq) : (-B + sq)) / A;
        const float s = s0 + t * ds;
        if (!(t >= NEAR_Z && t <= FAR_Z && s >= 0.0f && s <= h && t < best)) continue;
        const float px = ox + t * d.x, pz = oz + t * d.z;
        V3 n = v3(px, dirSign * (k * s), pz);
        if (!(dot(n, d) < 0.0f)) continue;       // back face
        const float l2 = len2(n);
        if (!(l2 > 0.0f)) continue;
        n = n * (1.0f / sqrtf(l2));
        hit = true; best = t; bn = n;
    }
    if (hit) { *t_out = best; *n_out = bn; }
    return hit;
}

static inline V3 mat_mul(const float m[3][3], V3 v)
{
    return v3((m[0][0] * v.x + m[0][1] * v.y) + m[0][2] * v.z, (m[1][0] * v.x + m[1][1] * v.y) + m[1][2] * v.z,
              (m[2][0] * v.x + m[2][1] * v.y) + m[2][2] * v.z);
}
static inline V3 mat_tmul(const float m[3][3], V3 v)
{
    return v3((m[0][0] * v.x + m[1][0] * v.y) + m[2][0] * v.z, (m[0][1] * v.x + m[1][1] * v.y) + m[2][1] * v.z,
              (m[0][2] * v.x + m[1][2] * v.y) + m[2][2] * v.z);
}

static inline float pow300(float x)
{
    const float x2 = x * x, x4 = x2 * x2, x8 = x4 * x4, x16 = x8 * x8, x32 = x16 * x16, x64 = x32 * x32,
                x128 = x64 * x64, x256 = x128 * x128;
    return ((x256 * x32) * x8) * x4;
}

static inline uint8_t to_u8(float v)
{
    v = std::min(std::max(v, 0.0f), 1.0f);
    return (uint8_t)(int)floorf(v * 255.0f + 0.5f);
}

// Conservative screen rectangle of a primitive's bounding box for the tiled raster below: the projection of the part of the box in front of the
// plane w = NEAR_Z / 2 (camera depth) -- the corners in front of it plus the points where the box's edges pierce it -- widened by a pixel and a
// half.  It only ever culls; the pixels do not depend on it (the HIP frame setup has a rectangle of its own, mv_frame.h, for the same purpose).
struct ScreenRect { int x0, x1, y0, y1; bool any; };

static ScreenRect prim_screen_rect(const Prim &p, const Cam *cams, int viewer, int W, int H)
{
    V3 lo = p.lo, hi = p.hi;
    if (p.kind == 2) { const float r = p.hi.x, hl = p.hi.y; lo = v3(p.lo.x - r, p.lo.y - (hl + r), p.lo.z - r); hi = v3(p.lo.x + r, p.lo.y + (hl + r), p.lo.z + r); }
    else if (p.kind == 3) {
        const float r = p.hi.x, h = p.hi.y;
        lo = v3(p.lo.x - r, p.hi.z > 0.0f ? p.lo.y - h : p.lo.y, p.lo.z - r);
        hi = v3(p.lo.x + r, p.hi.z > 0.0f ? p.lo.y : p.lo.y + h, p.lo.z + r);
    } else if (p.kind >= 4) {
        const float ey = p.kind == 5 ? p.hi.y * 2.0f : p.kind == 6 ? p.hi.y * 0.5f : p.hi.y;
        lo = v3(p.lo.x - p.hi.x, p.lo.y - ey, p.lo.z - p.hi.z); hi = v3(p.lo.x + p.hi.x, p.lo.y + ey, p.lo.z + p.hi.z);
    }
    const Cam &cv = cams[viewer];
    double cx[8], cy[8], cw[8];
    for (int c = 0; c < 8; ++c) {
        V3 q = v3((c & 1) ? hi.x : lo.x, (c & 2) ? hi.y : lo.y, (c & 4) ? hi.z : lo.z);
        if (p.frame != viewer) {
            if (p.frame >= MAX_AGENTS) q = hex_to_world(p.frame - MAX_AGENTS, q);
            else if (p.frame >= 0) q = cam_to_world(cams[p.frame], q);
            q = mat_tmul(cv.c, q - cv.eye);
        }
        cx[c] = q.x; cy[c] = q.y; cw[c] = -double(q.z);
    }
    const double clipW = 0.5 * NEAR_Z;
    double xmin = 1e30, xmax = -1e30, ymin = 1e30, ymax = -1e30;
    bool any = false;
    auto add = [&](double x, double y, double w) { x /= w; y /= w; xmin = std::min(xmin, x); xmax = std::max(xmax, x); ymin = std::min(ymin, y); ymax = std::max(ymax, y); any = true; };
    for (int c = 0; c < 8; ++c) if (cw[c] >= clipW) add(cx[c], cy[c], cw[c]);
    for (int axis = 0; axis < 3; ++axis)
        for (int c = 0; c < 8; ++c) {
            const int d = c | (1 << axis);
            if (d == c || (cw[c] >= clipW) == (cw[d] >= clipW)) continue;
            const double t = (clipW - cw[c]) / (cw[d] - cw[c]);
            add(cx[c] + t * (cx[d] - cx[c]), cy[c] + t * (cy[d] - cy[c]), clipW);
        }
    ScreenRect r{0, 0, 0, 0, false};
    if (!any) return r;
    xmin = std::max(xmin / TAN_HALF_FOV, -4.0); xmax = std::min(xmax / TAN_HALF_FOV, 4.0);
    ymin = std::max(ymin / TAN_HALF_FOV_Y, -4.0); ymax = std::min(ymax / TAN_HALF_FOV_Y, 4.0);
    const double fx0 = (xmin * 0.5 + 0.5) * W - 1.5, fx1 = (xmax * 0.5 + 0.5) * W + 0.5, fy0 = (ymin * 0.5 + 0.5) * H - 1.5, fy1 = (ymax * 0.5 + 0.5) * H + 0.5;
    if (fx1 < 0.0 || fy1 < 0.0 || fx0 > W || fy0 > H) return r;
    r.x0 = (int)std::floor(std::max(fx0, 0.0)); r.x1 = (int)std::ceil(std::min(fx1, double(W - 1)));
    r.y0 = (int)std::floor(std::max(fy0, 0.0)); r.y1 = (int)std::ceil(std::min(fy1, double(H - 1)));
    r.any = true;
    return r;
}

static void render_agent(const Env &e, int viewer, int W, int H, uint8_t *out, bool tiled = false)
{
    std::vector<Prim> prims;
    build_prims(e, viewer, prims);
    Cam cams[MAX_AGENTS];
    for (int k = 0; k < e.numAgents; ++k) cams[k] = camera_of(e.agents[k]);
    const Cam &cam = cams[viewer];

    // ray origins/rotations into every agent frame are per-frame constants
    V3 originIn[MAX_AGENTS];
    for (int k = 0; k < e.numAgents; ++k) originIn[k] = mat_tmul(cams[k].c, cam.eye - cams[k].eye);

    V3 hexOrigin[HEX_FRAMES];
    for (int k = 0; k < HEX_FRAMES; ++k) hexOrigin[k] = hex_to_local(k, cam.eye);

    const V3 LIGHT = v3(0.0f, 4.0f, 2.0f);
    const float AMB = float(0x55) / 255.0f, DIF = float(0xbb) / 255.0f, LCOL = float(0xaa) / 255.0f;

    // one pixel against the primitives list[0 .. n) (indices into prims, ascending: ties keep the draw order)
    auto pixel = [&](int i, int j, const int *list, int n) {
            const float xn = ((float(i) + 0.5f) / float(W)) * 2.0f - 1.0f;
            const float yn = ((float(j) + 0.5f) / float(H)) * 2.0f - 1.0f;
            const V3 dc = v3(xn * TAN_HALF_FOV, yn * TAN_HALF_FOV_Y, -1.0f);
            const V3 dw = mat_mul(cam.c, dc);

            float best = INFINITY; V3 bestN = v3(0, 0, 0); unsigned bestColor = 0; bool any = false;
            for (int li = 0; li < n; ++li) {
                const Prim &p = prims[(size_t)list[li]];
                float t; V3 n;  // n ends up in the viewer's camera space
                bool hit;
                if (p.kind == 2) {
                    hit = ray_capsule(cam.eye, dw, p.lo, p.hi.x, p.hi.y, &t, &n);
                    if (hit) n = mat_tmul(cam.c, n);
                } else if (p.kind == 3) {
                    hit = ray_cone(cam.eye, dw, p.lo, p.hi.x, p.hi.y, p.hi.z, &t, &n);
                    if (hit) n = mat_tmul(cam.c, n);
                } else if (p.kind >= 4) {
                    if (p.frame < 0) {
                        hit = ray_scaled_shape(p.kind, cam.eye, dw, p.lo, p.hi, &t, &n);
                        if (hit) n = mat_tmul(cam.c, n);
                    } else if (p.frame == viewer) {
                        hit = ray_scaled_shape(p.kind, v3(0, 0, 0), dc, p.lo, p.hi, &t, &n);
                    } else {
                        const V3 dk = mat_tmul(cams[p.frame].c, dw);
                        hit = ray_scaled_shape(p.kind, originIn[p.frame], dk, p.lo, p.hi, &t, &n);
                        if (hit) n = mat_tmul(cam.c, mat_mul(cams[p.frame].c, n));
                    }
                } else if (p.frame >= MAX_AGENTS) {   // hex wall frame: a rotation about Y around the world origin
                    const int k = p.frame - MAX_AGENTS;
                    hit = ray_box(hexOrigin[k], hex_to_local(k, dw), p.lo, p.hi, &t, &n);
                    if (hit) n = mat_tmul(cam.c, hex_to_world(k, n));
                } else if (p.frame < 0) {
                    hit = ray_box(cam.eye, dw, p.lo, p.hi, &t, &n);
                    if (hit) n = mat_tmul(cam.c, n);
                } else if (p.frame == viewer) {
                    hit = ray_box(v3(0, 0, 0), dc, p.lo, p.hi, &t, &n);
                } else {
                    const V3 dk = mat_tmul(cams[p.frame].c, dw);
                    hit = ray_box(originIn[p.frame], dk, p.lo, p.hi, &t, &n);
                    if (hit) n = mat_tmul(cam.c, mat_mul(cams[p.frame].c, n));
                }
                if (hit && t < best) { best = t; bestN = n; bestColor = p.color; any = true; }
            }

            uint8_t *px = out + (size_t(j) * W + i) * 4;
            if (!any) { px[0] = px[1] = px[2] = 0; px[3] = 255; return; }
            const V3 P = dc * best;  // camera-space position
            V3 Ld = LIGHT - P;
            Ld = Ld * (1.0f / sqrtf(len2(Ld)));
            const float intensity = std::max(0.0f, dot(bestN, Ld));
            float spec = 0.0f;
            if (intensity > 0.001f) {
                // reflect(-L, N) = -L + 2 (N.L) N
                const float k2 = 2.0f * dot(bestN, Ld);
                const V3 R = v3(k2 * bestN.x - Ld.x, k2 * bestN.y - Ld.y, k2 * bestN.z - Ld.z);
                V3 Vd = v3(-P.x, -P.y, -P.z);
                Vd = Vd * (1.0f / sqrtf(len2(Vd)));
                spec = pow300(std::max(0.0f, dot(Vd, R)));
                spec = std::min(std::max(spec, 0.0f), 1.0f);
            }
            const float col[3] = {float((bestColor >> 16) & 255) / 255.0f, float((bestColor >> 8) & 255) / 255.0f,
                                  float(bestColor & 255) / 255.0f};
            for (int ch = 0; ch < 3; ++ch) {
                const float v = (AMB * col[ch] + (DIF * col[ch]) * LCOL * intensity) + spec;
                px[ch] = to_u8(v);
            }
            px[3] = 255;
    };

    std::vector<int> all(prims.size());
    std::iota(all.begin(), all.end(), 0);
    if (!tiled) {   // the checker: every primitive against every pixel
        for (int j = 0; j < H; ++j)
            for (int i = 0; i < W; ++i) pixel(i, j, all.data(), (int)all.size());
        return;
    }
    // the timing leg (bench.py cpu_baseline): 16 x 16 tiles, each against the primitives whose conservative screen rectangle meets it.  Culling only
    // removes primitives no pixel of the tile can hit, the lists stay in draw order: the image is the brute-force one, byte for byte
    // (tests/test_oracle_properties.py: test_tiled_raster_equals_brute_force)
    std::vector<ScreenRect> rects(prims.size());
    for (size_t pi = 0; pi < prims.size(); ++pi) rects[pi] = prim_screen_rect(prims[pi], cams, viewer, W, H);
    std::vector<int> list;
    for (int ty = 0; ty < H; ty += 16)
        for (int tx = 0; tx < W; tx += 16) {
            const int x1 = std::min(tx + 16, W) - 1, y1 = std::min(ty + 16, H) - 1;
            list.clear();
            for (size_t pi = 0; pi < prims.size(); ++pi) {
                const ScreenRect &r = rects[pi];
                if (r.any && r.x0 <= x1 && r.x1 >= tx && r.y0 <= y1 && r.y1 >= ty) list.push_back((int)pi);
            }
            for (int j = ty; j <= y1; ++j)
                for (int i = tx; i <= x1; ++i) pixel(i, j, list.data(), (int)list.size());
        }
}

// ------------------------------------------------------------------------------------------------
// VectorEnv + MegaverseGym -- env/src/vector_env.cpp, bindings/megaverse.cpp
// ------------------------------------------------------------------------------------------------
struct Gym {
    int w, h, numEnvs, numAgents, numThreads;
    bool tiledRaster = false;   // mvo_set_raster: 1 = the tiled software raster (same image; the timing leg), 0 = every primitive against every pixel
    std::vector<std::unique_ptr<Env>> envs;
    std::vector<uint8_t> done;
    std::vector<float> trueObjective;
    std::vector<uint8_t> obs;
    std::vector<std::string> sokoFiles;   // Sokoban: the level files found at construction (scenario_sokoban.cpp:40-78)
    Rng rng{std::random_device{}()};  // megaverse.cpp:253

    // ---- the worker pool of VectorEnv (vector_env.cpp:5-40,58-87): numThreads - 1 PERSISTENT background threads + the caller; a task is handed out
    // under a mutex + condition variable (one generation counter instead of the reference's per-thread task slots), every thread works through
    // its static block of envs (envsPerThread = ceil(numEnvs / numThreads), :12,65-68), the caller takes block 0 and then spins on an atomic
    // count of finished workers (:84-86 "just waiting on an atomic").  Started on first use; pinned one worker per allowed CPU (round-robin)
    // when MVO_PIN=1 (bench.py's cpu_baseline sets it: BASELINE.md 3).  Spawning and joining std::threads on EVERY call -- what this was until
    // round 3 -- cost more than the step itself beyond a few threads (T=1 4.9 ms, T=8 5.5 ms per 1024-env tick).
    std::vector<std::thread> pool;
    std::mutex poolMutex;
    std::condition_variable poolCv;
    std::function<void(int)> poolTask;
    unsigned long long poolGen = 0;
    bool poolStop = false;
    int poolT = 1;
    std::atomic<int> poolReady{0};

    void pool_block(int t)
    {
        const int per = numEnvs / poolT + (numEnvs % poolT != 0);
        const int s = t * per, en = std::min(s + per, numEnvs);
        for (int i = s; i < en; ++i) poolTask(i);
    }

    void pool_start()
    {
        poolT = std::max(1, std::min(numThreads, numEnvs));
        std::vector<int> cpus;
#ifdef __linux__
        if (const char *pin = getenv("MVO_PIN"); pin && atoi(pin)) {
            cpu_set_t set;
            if (sched_getaffinity(0, sizeof(set), &set) == 0)
                for (int c = 0; c < CPU_SETSIZE; ++c) if (CPU_ISSET(c, &set)) cpus.push_back(c);
        }
#endif
        for (int t = 1; t < poolT; ++t) {
            pool.emplace_back([this, t] {
                unsigned long long seen = 0;
                for (;;) {
                    {
                        std::unique_lock<std::mutex> lock(poolMutex);
                        poolCv.wait(lock, [&] { return poolStop || poolGen != seen; });
                        if (poolStop) return;
                        seen = poolGen;
                    }
                    pool_block(t);
                    poolReady.fetch_add(1, std::memory_order_release);
                }
            });
#ifdef __linux__
            if (!cpus.empty()) {
                cpu_set_t one; CPU_ZERO(&one); CPU_SET(cpus[(size_t)t % cpus.size()], &one);
                pthread_setaffinity_np(pool.back().native_handle(), sizeof(one), &one);
            }
#endif
        }
    }

    void pool_stop()
    {
        {
            std::lock_guard<std::mutex> lock(poolMutex);
            poolStop = true;
        }
        poolCv.notify_all();
        for (auto &t : pool) t.join();
        pool.clear();
    }

    template <typename F> void parallel_envs(F f)
    {
        if (pool.empty() && poolT == 1 && numThreads > 1 && numEnvs > 1) pool_start();
        poolTask = f;
        if (poolT == 1) { pool_block(0); return; }
        poolReady.store(0, std::memory_order_relaxed);
        {
            std::lock_guard<std::mutex> lock(poolMutex);
            ++poolGen;
        }
        poolCv.notify_all();
        pool_block(0);
        int spins = 0;
        while (poolReady.load(std::memory_order_acquire) < poolT - 1)
            if (++spins > 20000) std::this_thread::yield();   // (more threads than cores: do not starve the workers)
    }

    void render()
    {
        parallel_envs([&](int i) {
            for (int a = 0; a < numAgents; ++a)
                render_agent(*envs[i], a, w, h, obs.data() + (size_t(i) * numAgents + a) * size_t(w) * h * 4, tiledRaster);
        });
    }

    void step(bool doRender)
    {
        parallel_envs([&](int i) { env_step(*envs[i]); });
        for (int i = 0; i < numEnvs; ++i) {  // vector_env.cpp:93-105, serial
            if (envs[i]->done) {
                done[i] = 1;
                for (int a = 0; a < numAgents; ++a)
                    trueObjective[size_t(i) * numAgents + a] = envs[i]->scenario == SCN_TOWER ? float(envs[i]->highestTower) : float(envs[i]->solved);   // scenario_collect.hpp:42
                env_reset(*envs[i]);
            } else done[i] = 0;
        }
        if (doRender) render();
    }
};

}  // namespace mvo

// ------------------------------------------------------------------------------------------------
// C API
// ------------------------------------------------------------------------------------------------
using namespace mvo;
struct mvo_gym : Gym {};

extern "C" {

mvo_gym *mvo_create(const char *scenario, int w, int h, int num_envs, int num_agents_per_env, int num_threads,
                    const char *const *keys, const float *vals, int n_params)
{
    std::string s(scenario ? scenario : "");
    for (auto &ch : s) ch = (char)tolower(ch);
    int scen = SCN_TOWER;
    ObstacleParams op;
    float carriedDefault = 0.0f;
    if (s == "towerbuilding") scen = SCN_TOWER;
    else if (s == "obstacleseasy") { scen = SCN_OBSTACLES; }   // scenario_obstacles.hpp:112-138 (== base defaults)
    else if (s == "obstaclesmedium") { scen = SCN_OBSTACLES; op.minPlatforms = 2; op.maxPlatforms = 4; op.minLava = 2; op.maxLava = 5; }
    else if (s == "obstacleshard") {
        scen = SCN_OBSTACLES; op.minPlatforms = 2; op.maxPlatforms = 7; op.minGap = 2; op.maxGap = 3; op.minLava = 3; op.maxLava = 10;
        op.minHeight = 2; op.maxHeight = 4;
    } else if (s == "obstacleswalls" || s == "obstaclessteps" || s == "obstacleslava") {   // :190-268
        scen = SCN_OBSTACLES; op.minPlatforms = 1; op.maxPlatforms = 4; op.minGap = 1; op.maxGap = 3; op.minLava = 2; op.maxLava = 10;
        op.minHeight = 1; op.maxHeight = 3; carriedDefault = 1.0f;
        op.platformTypes = {s == "obstacleswalls" ? PT_WALL : s == "obstaclessteps" ? PT_STEP : PT_LAVA};
    } else if (s == "collect") scen = SCN_COLLECT;   // scenarios/init.hpp:45
    else if (s == "rearrange") scen = SCN_REARRANGE;   // scenarios/init.hpp:49
    else if (s == "sokoban") scen = SCN_SOKOBAN;       // scenarios/init.hpp:46
    else if (s == "empty") scen = SCN_EMPTY;           // scenarios/init.hpp:34
    else if (s == "hexmemory") scen = SCN_HEX_MEMORY;  // scenarios/init.hpp:47-48
    else if (s == "hexexplore") scen = SCN_HEX_EXPLORE;
    else { fprintf(stderr, "mv_oracle: unknown scenario %s\n", s.c_str()); return nullptr; }
    if (num_agents_per_env < 1 || num_agents_per_env > MAX_AGENTS || num_envs < 1) return nullptr;
    auto *g = new mvo_gym();
    if (scen == SCN_SOKOBAN) {   // $BOXOBAN_LEVELS (or ~/datasets/boxoban) / unfiltered / train / 000.txt .. 999.txt
        const char *dir = getenv("BOXOBAN_LEVELS");
        std::string base = dir && *dir ? dir : "~/datasets/boxoban";
        const size_t tilde = base.find('~');
        if (tilde != std::string::npos) { const char *home = getenv("HOME"); base.replace(tilde, 1, home ? home : ""); }
        for (int i = 0; i <= 999; ++i) {
            char name[16];
            snprintf(name, sizeof name, "%03d.txt", i);
            const std::string path = base + "/unfiltered/train/" + name;
            if (FILE *f = fopen(path.c_str(), "rb")) { fclose(f); g->sokoFiles.push_back(path); }
        }
        if (g->sokoFiles.empty()) { fprintf(stderr, "mv_oracle: no Boxoban levels under %s\n", base.c_str()); delete g; return nullptr; }
    }
    g->w = w; g->h = h; g->numEnvs = num_envs; g->numAgents = num_agents_per_env; g->numThreads = std::max(1, num_threads);
    for (int i = 0; i < num_envs; ++i) {
        auto e = std::make_unique<Env>();
        e->numAgents = num_agents_per_env;
        if (scen == SCN_SOKOBAN) { e->p_episodeLengthSec = 80.0f; e->sokoFiles = &g->sokoFiles; }   // scenario_sokoban.hpp:49-53
        for (int k = 0; k < n_params; ++k) {
            if (!strcmp(keys[k], "episodeLengthSec")) e->p_episodeLengthSec = vals[k];
            if (!strcmp(keys[k], "verticalLookLimitRad")) e->p_verticalLookLimitRad = vals[k];
            auto ip = [&](const char *name, int &dst) { if (!strcmp(keys[k], name)) dst = int(lroundf(vals[k])); };
            ip("obstaclesMinNumPlatforms", op.minPlatforms); ip("obstaclesMaxNumPlatforms", op.maxPlatforms);
            ip("obstaclesMinGap", op.minGap); ip("obstaclesMaxGap", op.maxGap); ip("obstaclesMinLava", op.minLava);
            ip("obstaclesMaxLava", op.maxLava); ip("obstaclesMinHeight", op.minHeight); ip("obstaclesMaxHeight", op.maxHeight);
            if (!strcmp(keys[k], "obstaclesNumAllowedMaxDifficulty")) op.numAllowedMaxDifficulty = int(vals[k]);
        }
        e->scenario = scen;
        e->op = op;
        e->numShaping = scen == SCN_TOWER || scen == SCN_SOKOBAN ? 4 : scen == SCN_REARRANGE || scen == SCN_HEX_MEMORY ? 3
                      : scen == SCN_HEX_EXPLORE ? 2 : scen == SCN_EMPTY ? 1 : 5;   // Empty: teamSpirit only
        e->shapingKeys = scen == SCN_TOWER ? SHAPING_KEYS_TOWER : scen == SCN_OBSTACLES ? SHAPING_KEYS_OBST
                       : scen == SCN_COLLECT ? SHAPING_KEYS_COLLECT : scen == SCN_SOKOBAN ? SHAPING_KEYS_SOKOBAN
                       : scen == SCN_HEX_MEMORY ? SHAPING_KEYS_HEX_MEMORY : scen == SCN_HEX_EXPLORE ? SHAPING_KEYS_HEX_EXPLORE : SHAPING_KEYS_REARRANGE;
        for (int a = 0; a < MAX_AGENTS; ++a) {
            std::memset(e->agents[a].shaping, 0, sizeof e->agents[a].shaping);
            for (int k = 0; k < e->numShaping; ++k)
                e->agents[a].shaping[k] = scen == SCN_TOWER ? SHAPING_DEFAULT_TOWER[k]
                                        : scen == SCN_COLLECT ? SHAPING_DEFAULT_COLLECT[k]
                                        : scen == SCN_REARRANGE ? SHAPING_DEFAULT_REARRANGE[k]
                                        : scen == SCN_SOKOBAN ? SHAPING_DEFAULT_SOKOBAN[k]
                                        : scen == SCN_HEX_MEMORY ? SHAPING_DEFAULT_HEX_MEMORY[k]
                                        : scen == SCN_HEX_EXPLORE ? SHAPING_DEFAULT_HEX_EXPLORE[k]
                                        : (k == 4 ? carriedDefault : SHAPING_DEFAULT_OBST[k]);
        }
        g->envs.push_back(std::move(e));
    }
    g->done.assign(num_envs, 0);
    g->trueObjective.assign(size_t(num_envs) * num_agents_per_env, 0.0f);
    g->obs.assign(size_t(num_envs) * num_agents_per_env * w * h * 4, 0);
    return g;
}

void mvo_close(mvo_gym *g) { if (g) { g->pool_stop(); delete g; } }

void mvo_seed(mvo_gym *g, int seed)
{   // megaverse.cpp:60-69
    g->rng.seed((unsigned long)seed);
    for (auto &e : g->envs) {
        const int noise = randRange(0, 1 << 30, g->rng);
        e->rng.seed((unsigned long)noise);
    }
}

void mvo_reset(mvo_gym *g)
{   // vector_env.cpp:110-120
    for (auto &e : g->envs) env_reset(*e);
    g->render();
}

int mvo_action_mask(const int *actions, int n)
{   // megaverse.cpp:100-116
    static const int spaces[6] = {3, 3, 3, 2, 2, 3};
    int actionIdx = 0, mask = 0;
    for (int i = 0; i < n && i < 6; ++i) {
        if (actions[i] > 0) mask |= 1 << (actionIdx + actions[i]);
        actionIdx += spaces[i] - 1;
    }
    return mask;
}

void mvo_set_actions(mvo_gym *g, int env, int agent, const int *actions, int n) { g->envs[env]->agents[agent].action = mvo_action_mask(actions, n); }
void mvo_set_action_mask(mvo_gym *g, int env, int agent, int mask) { g->envs[env]->agents[agent].action = mask; }
void mvo_set_action_masks(mvo_gym *g, const int *masks)   // [N*A] env-major, one call per tick (bench.py's CPU baseline)
{
    for (int e = 0; e < g->numEnvs; ++e)
        for (int a = 0; a < g->numAgents; ++a) g->envs[e]->agents[a].action = masks[(size_t)e * g->numAgents + a];
}
void mvo_step(mvo_gym *g) { g->step(true); }
void mvo_step_norender(mvo_gym *g) { g->step(false); }
void mvo_render(mvo_gym *g) { g->render(); }
void mvo_set_raster(mvo_gym *g, int tiled) { g->tiledRaster = tiled != 0; }
int mvo_is_done(mvo_gym *g, int env) { return g->done[env]; }
void mvo_get_dones(mvo_gym *g, uint8_t *out) { for (int i = 0; i < g->numEnvs; ++i) out[i] = g->done[i]; }   /* all envs' done flags with one call (full-size parity tests) */
void mvo_render_env(mvo_gym *g, int env)   /* the frames of ONE env's agents (full-size parity tests sample a few envs: the brute-force raster of all 1024 would take minutes) */
{
    for (int a = 0; a < g->numAgents; ++a)
        render_agent(*g->envs[env], a, g->w, g->h, g->obs.data() + (size_t(env) * g->numAgents + a) * size_t(g->w) * g->h * 4);
}

void mvo_get_last_rewards(mvo_gym *g, float *out)
{   // megaverse.cpp:128-137 (read after the auto-reset zero-filled them)
    int k = 0;
    for (int i = 0; i < g->numEnvs; ++i)
        for (int a = 0; a < g->numAgents; ++a) out[k++] = g->envs[i]->agents[a].last_reward;
}

float mvo_true_objective(mvo_gym *g, int env, int agent) { return g->trueObjective[size_t(env) * g->numAgents + agent]; }
const uint8_t *mvo_get_observation(mvo_gym *g, int env, int agent) { return g->obs.data() + (size_t(env) * g->numAgents + agent) * size_t(g->w) * g->h * 4; }

float mvo_get_reward_shaping(mvo_gym *g, int env, int agent, const char *key, int *found)
{
    const Env &e = *g->envs[env];
    for (int k = 0; k < e.numShaping; ++k)
        if (!strcmp(key, e.shapingKeys[k])) { if (found) *found = 1; return e.agents[agent].shaping[k]; }
    if (found) *found = 0;
    return 0.0f;
}
void mvo_set_reward_shaping(mvo_gym *g, int env, int agent, const char *key, float v)
{
    Env &e = *g->envs[env];
    for (int k = 0; k < e.numShaping; ++k)
        if (!strcmp(key, e.shapingKeys[k])) e.agents[agent].shaping[k] = v;
}

// ---- snapshot (layout: DESIGN.md "snapshot format") ----
#pragma pack(push, 4)
struct SnapAgent {
    float pos[3], basis[4], pitch, hv[2], vvel, voffset, step_offset, jump_speed;
    int32_t was_jumping, carrying, picked_up, visited_zone, spawn[3];
    float last_reward, total_reward, shaping[MAX_SHAPING];
};
struct SnapHeader {
    int32_t scenario, L, H, W, bz[4], layout_color, wall_color, draw_walls, num_objects, num_boxes, num_frames, done, highest_tower,
        num_agents, num_terrain, num_rewards, num_platforms, solved;
    float episode_sec, episode_len, bz_reward, bar_half_width;
    int32_t boxes[MAX_BOXES][8];
    int32_t terrain[MAX_TERRAIN][8];
    int8_t objects[MAX_OBJECTS][4];
    int8_t rewards[MAX_REWARDS][4];
    SnapAgent agents[MAX_AGENTS];
    uint8_t chunk[CHUNK];
    int8_t heightmap[HM_DIM * HM_DIM];
    int32_t num_items, items[MAX_ITEMS][5];   // Rearrange: shape, colour, offset x y z
    uint8_t soko[SOKO_DIM * SOKO_DIM];        // Sokoban: wall / goal cells
    // Hex*: boxes {lo[3], meta = (frame + 1) | collide << 4, hi[3], colour}, objects {pos[3], meta = shape | good << 4 | alive << 8 |
    // (vox.x + 128) << 12 | (vox.z + 128) << 20, scale[3], colour} -- the device's own records
    int32_t hex_num_boxes, hex_num_objs;
    float hex_target[3];
    struct HexRec { float a[3]; int32_t meta; float b[3]; int32_t color; } hex_boxes[HEX_MAX_BOXES], hex_objs[HEX_MAX_OBJS];
};
#pragma pack(pop)

/* test hook: put an agent somewhere (e.g. below the fall-detection threshold, component_fall_detection.hpp:33-55) */
void mvo_debug_set_agent_pos(mvo_gym *g, int env, int agent, float x, float y, float z) { g->envs[env]->agents[agent].pos = v3(x, y, z); }
/* test hooks for the canonical-pose tests: the yaw basis from (cos, sin) exactly like spawn_agents builds it, and the velocities */
void mvo_debug_set_agent_yaw(mvo_gym *g, int env, int agent, float c, float s)
{
    Agent &a = g->envs[env]->agents[agent];
    a.m00 = c; a.m02 = s; a.m20 = -s; a.m22 = c;
}
void mvo_debug_set_agent_velocity(mvo_gym *g, int env, int agent, float hvx, float hvz, float vvel)
{
    Agent &a = g->envs[env]->agents[agent];
    a.hvx = hvx; a.hvz = hvz; a.vvel = vvel;
}

int mvo_snapshot_size(mvo_gym *) { return (int)sizeof(SnapHeader); }

void mvo_snapshot(mvo_gym *g, int env, void *out)
{
    const Env &e = *g->envs[env];
    auto *s = new SnapHeader();
    std::memset(s, 0, sizeof *s);
    s->scenario = e.scenario; s->L = e.L; s->H = e.H; s->W = e.W;
    s->num_terrain = e.numTerrain; s->num_rewards = e.numRewards; s->num_platforms = e.numPlatforms; s->solved = e.solved;
    for (int i = 0; i < e.numTerrain; ++i) {
        const TerrainBox &t = e.terrain[i];
        int32_t *o = s->terrain[i];
        o[0] = t.min[0]; o[1] = t.min[1]; o[2] = t.min[2]; o[3] = t.max[0]; o[4] = t.max[1]; o[5] = t.max[2]; o[6] = t.type; o[7] = 0;
    }
    for (int i = 0; i < e.numRewards; ++i) {
        s->rewards[i][0] = (int8_t)e.rewards[i].x; s->rewards[i][1] = (int8_t)e.rewards[i].y; s->rewards[i][2] = (int8_t)e.rewards[i].z;
        s->rewards[i][3] = (int8_t)e.rewards[i].active;
    }
    for (int i = 0; i < 4; ++i) s->bz[i] = e.bz[i];
    s->layout_color = (int)e.layoutColor; s->wall_color = (int)e.wallColor; s->draw_walls = e.drawWalls;
    s->num_objects = e.numObjects; s->num_boxes = e.numBoxes; s->num_frames = e.numFrames; s->done = e.done;
    s->highest_tower = e.highestTower; s->num_agents = e.numAgents;
    s->episode_sec = e.episodeSec; s->episode_len = e.episodeLen; s->bz_reward = e.bzReward; s->bar_half_width = e.barHalfWidth;
    for (int i = 0; i < e.numBoxes; ++i) {
        const Box &b = e.boxes[i];
        int32_t *o = s->boxes[i];
        o[0] = b.min[0]; o[1] = b.min[1]; o[2] = b.min[2]; o[3] = b.max[0]; o[4] = b.max[1]; o[5] = b.max[2]; o[6] = b.type; o[7] = b.slot;
    }
    for (int i = 0; i < e.numObjects; ++i) {
        s->objects[i][0] = (int8_t)e.objects[i].x; s->objects[i][1] = (int8_t)e.objects[i].y;
        s->objects[i][2] = (int8_t)e.objects[i].z; s->objects[i][3] = (int8_t)e.objects[i].state;
    }
    for (int i = 0; i < e.numAgents; ++i) {
        const Agent &a = e.agents[i];
        SnapAgent &o = s->agents[i];
        o.pos[0] = a.pos.x; o.pos[1] = a.pos.y; o.pos[2] = a.pos.z;
        o.basis[0] = a.m00; o.basis[1] = a.m02; o.basis[2] = a.m20; o.basis[3] = a.m22;
        o.pitch = a.pitch; o.hv[0] = a.hvx; o.hv[1] = a.hvz; o.vvel = a.vvel; o.voffset = a.voffset;
        o.step_offset = a.step_offset; o.jump_speed = a.jump_speed; o.was_jumping = a.was_jumping; o.carrying = a.carrying;
        o.picked_up = a.picked_up; o.visited_zone = a.visited_zone;
        for (int k = 0; k < 3; ++k) o.spawn[k] = a.spawn[k];
        o.last_reward = a.last_reward; o.total_reward = a.total_reward;
        for (int k = 0; k < MAX_SHAPING; ++k) o.shaping[k] = a.shaping[k];
    }
    std::memcpy(s->chunk, e.chunk.data(), CHUNK);
    std::memcpy(s->heightmap, e.heightmap.data(), HM_DIM * HM_DIM);
    std::memcpy(s->soko, e.soko.data(), SOKO_DIM * SOKO_DIM);
    s->num_items = e.scenario == SCN_REARRANGE ? e.numItems : 0;
    for (int i = 0; i < s->num_items; ++i) {
        s->items[i][0] = e.items[i].shape; s->items[i][1] = (int32_t)e.items[i].color;
        s->items[i][2] = e.items[i].off[0]; s->items[i][3] = e.items[i].off[1]; s->items[i][4] = e.items[i].off[2];
    }
    if (e.scenario == SCN_HEX_MEMORY || e.scenario == SCN_HEX_EXPLORE) {
        s->hex_num_boxes = int(e.hexBoxes.size()); s->hex_num_objs = int(e.hexObjs.size());
        s->hex_target[0] = e.hexTarget.x; s->hex_target[1] = e.hexTarget.y; s->hex_target[2] = e.hexTarget.z;
        for (size_t i = 0; i < e.hexBoxes.size() && i < HEX_MAX_BOXES; ++i) {
            const Env::HexBox &b = e.hexBoxes[i];
            s->hex_boxes[i] = SnapHeader::HexRec{{b.lo.x, b.lo.y, b.lo.z}, (b.frame + 1) | (b.collide << 4), {b.hi.x, b.hi.y, b.hi.z}, int32_t(b.color)};
        }
        for (size_t i = 0; i < e.hexObjs.size() && i < HEX_MAX_OBJS; ++i) {
            const Env::HexObj &o = e.hexObjs[i];
            const int meta = o.shape | (o.good << 4) | (o.alive << 8) | (((o.vox[0] + 128) & 255) << 12) | (((o.vox[2] + 128) & 255) << 20);
            s->hex_objs[i] = SnapHeader::HexRec{{o.pos.x, o.pos.y, o.pos.z}, meta, {o.scale.x, o.scale.y, o.scale.z}, int32_t(o.color)};
        }
    }
    std::memcpy(out, s, sizeof *s);
    delete s;
}

// ---- spec helpers ----
void mvo_perlin_octave2_01(uint32_t seed, const double *xs, const double *ys, int n, int octaves, double *out)
{
    const Perlin perlin(seed);
    for (int i = 0; i < n; ++i) out[i] = perlin.octaves2_01(xs[i], ys[i], octaves);
}

uint32_t mvo_mt19937_nth(uint32_t seed, int n)
{
    Rng r(seed);
    uint32_t v = 0;
    for (int i = 0; i < n; ++i) v = (uint32_t)r();
    return v;
}
int mvo_rand_range_seq(uint32_t seed, const int *lo, const int *hi, int n, int *out)
{
    Rng r(seed);
    for (int i = 0; i < n; ++i) out[i] = randRange(lo[i], hi[i], r);
    return 0;
}
void mvo_frand_seq(uint32_t seed, int n, float *out)
{
    Rng r(seed);
    for (int i = 0; i < n; ++i) out[i] = frand(r);
}
void mvo_shuffle_iota(uint32_t seed, int n, int *out)
{
    Rng r(seed);
    std::vector<int> v(n);
    for (int i = 0; i < n; ++i) v[i] = i;
    std::shuffle(v.begin(), v.end(), r);
    for (int i = 0; i < n; ++i) out[i] = v[i];
}
void mvo_get_coords(const float *v, int *out) { voxel_of(v3(v[0], v[1], v[2]), out); }
float mvo_building_reward_coeff(float h) { return building_reward_coeff(h); }
int mvo_triangular_number(int n) { return triangular_number(n); }

/* honeycomb maze of `size` with Kruskal seeded by `seed`: per cell the number of borders left, then for every border (cell order, list
 * order) the adjacent cell and the four coordinates; returns the number of borders (out == NULL: count only).  centers: [cells][2]. */
int mvo_hex_maze(int size, uint32_t seed, int *cells_out, int *border_counts, int *border_to, double *border_xy, double *centers, double *bounds)
{
    HexMaze m;
    hex_maze_generate(m, size, seed);
    if (cells_out) *cells_out = m.cells;
    int n = 0;
    for (int i = 0; i < m.cells; ++i) {
        if (border_counts) border_counts[i] = int(m.adj[i].size());
        for (const HexBorder &b : m.adj[i]) {
            if (border_to) border_to[n] = b.to;
            if (border_xy) { border_xy[4 * n] = b.x1; border_xy[4 * n + 1] = b.y1; border_xy[4 * n + 2] = b.x2; border_xy[4 * n + 3] = b.y2; }
            ++n;
        }
        if (centers) { centers[2 * i] = m.centers[i].first; centers[2 * i + 1] = m.centers[i].second; }
    }
    if (bounds) { bounds[0] = -m.xlim; bounds[1] = -m.ylim; bounds[2] = m.xlim; bounds[3] = m.ylim; }
    return n;
}
void mvo_sincos(float x, float *s, float *c) { mv_sincos(x, s, c); }
/* the level-file tokeniser (split_tokens) as a test hook: tokens joined by '\x1f'; returns their number */
int mvo_split_tokens(const char *text, char delim, char *out, int cap)
{
    const std::vector<std::string> tokens = split_tokens(text, delim);
    std::string joined;
    for (size_t i = 0; i < tokens.size(); ++i) { if (i) joined += '\x1f'; joined += tokens[i]; }
    if (cap > 0) { strncpy(out, joined.c_str(), size_t(cap) - 1); out[cap - 1] = 0; }
    return int(tokens.size());
}

}  // extern "C"
