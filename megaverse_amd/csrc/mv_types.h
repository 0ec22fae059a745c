// megaverse_amd/csrc/mv_types.h -- HBM-resident state of the batched voxel-world simulator.
//
// Layout rule: one wavefront owns one env in the step/reset kernels, so everything a wave needs
// for an env is contiguous per env (array-of-structs per kind); the raster kernel reads the same
// arrays.  All sizes are multiples of 16 B so a wave's loads are aligned dwordx4.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace mv {

enum : int {
    CX = 32, CY = 16, CZ = 32, CHUNK_BYTES = CX * CY * CZ,   // dense voxel chunk, 1 B per cell
    MAX_BOXES = 128,          // merged layout parallelepipeds per env (TowerBuilding uses 5, ObstaclesHard up to ~60)
    TOWER_BOXES = 16,         // collider / primitive slots the TowerBuilding kernels reserve for them
    MAX_OBJECTS = 80, MAX_AGENTS = 8, MAX_TERRAIN = 16, MAX_REWARDS = 16,
    NUM_SHAPING = 8,
    // Collect: Perlin heightfield up to 42 x 42 columns (scenario_collect.cpp:63); its merged slabs are many
    COLLECT_MAX_BOXES = 1024, COLLECT_MAX_REWARDS = 96, HM_DIM = 42, HM_BYTES = 1792,
    // HexMemory / HexExplore: floor + walls + edgings + landmarks of a honeycomb maze of up to 127 cells (294 walls, each with an
    // edging and up to 4 landmarks: 1765 boxes at most), collectables (one per cell but the centre one, + the landmark object)
    HEX_MAX_BOXES = 2048, HEX_MAX_OBJS = 128, HEX_FRAMES = 3,
    MAX_CAMS = MAX_AGENTS + HEX_FRAMES,   // frames of reference a box can live in besides the world: agent cameras, hex wall orientations
};

enum : int { FRAME_HDR_BYTES = 1024 };
// Pipelining (mv_api.hip): a step's hand-over buffers -- frame lists, headers, cost lists, reward / done staging -- exist once per SLOT.
// Slots come in PIPE_GROUPS groups of `batch` slots: one call (mv_step: one tick, mv_step_n: up to `batch` ticks) takes the next group, and
// its step kernels may run while the observation passes of the previous PIPE_GROUPS - 1 calls are still reading theirs.  Cost histograms:
// one more than there are slots (a pass clears the next; GymView::lpt_hists).
enum : int { PIPE_GROUPS = 3, PIPE_BATCH_MAX = 16 };
// where a tick's actions come from (GymView::sample_on): the action array, or drawn inside the step kernel (mv_actions.h)
enum : int { POLICY_NONE = 0, POLICY_MULTIDISCRETE = 1, POLICY_SINGLE_BIT = 2 };

// error flags a kernel raises in episode_status[N + 1]; mv_step reports them (mv_api.hip: check_status_flags)
enum : int { ST_STARVED = 1, ST_CANDIDATES = 2, ST_VISIBLE = 4, ST_CHUNK = 8 };

enum : int { SCN_TOWER = 0, SCN_OBSTACLES = 1, SCN_COLLECT = 2, SCN_REARRANGE = 3, SCN_SOKOBAN = 4,
             SCN_EMPTY = 5,     // Empty runs on the Obstacles kernels (one slab, no terrain) with fall detection off
             SCN_HEX_MEMORY = 6, SCN_HEX_EXPLORE = 7 };
enum : int { HEX_PILLAR = 0, HEX_DIAMOND = 1, HEX_SPHERE = 2 };                       // scenario_hex_memory.cpp:163-168 ShapeType
enum : int { SOKO_DIM = 32, SOKO_WALL = 1, SOKO_GOAL = 2 };                            // Sokoban level cells (scenario_sokoban.cpp:28-33)
enum : int { MAX_ITEMS = 8, NUM_STATIC = 9 };                                         // Rearrange: arrangement items, static colliding boxes
enum : int { SHAPE_BOX = 0, SHAPE_CAPSULE = 1, SHAPE_SPHERE = 2, SHAPE_CYLINDER = 4 };   // DrawableType, env/include/env/env.hpp:58-69
enum : int { TERRAIN_EXIT = 1, TERRAIN_LAVA = 2, TERRAIN_BUILDING_ZONE = 4 };   // scenarios/platforms.hpp:28-34

// voxel cell byte (reference: env/include/env/voxel_state.hpp:10-37, scenarios/platforms.hpp:28-34)
enum : int { VX_SOLID = 1, VX_OPAQUE = 2, VX_OBJECT = 4, VX_TERRAIN_SHIFT = 3, VX_COLOR_SHIFT = 6 };

// action bits (reference: env/include/env/env.hpp:22-42)
enum : int {
    ACT_LEFT = 1 << 1, ACT_RIGHT = 1 << 2, ACT_FORWARD = 1 << 3, ACT_BACKWARD = 1 << 4, ACT_LOOK_LEFT = 1 << 5,
    ACT_LOOK_RIGHT = 1 << 6, ACT_JUMP = 1 << 7, ACT_INTERACT = 1 << 8, ACT_LOOK_DOWN = 1 << 9, ACT_LOOK_UP = 1 << 10,
};

struct alignas(16) EnvHeader {   // 128 B
    int32_t L, H, W;                    // room extents in voxels (x, y, z)
    int32_t bz[4];                      // building zone min x, max x, min z, max z (y == 1)
    int32_t layout_color, wall_color;   // 0xRRGGBB
    int32_t draw_walls;
    int32_t num_objects, num_boxes;
    int32_t num_frames, done, highest_tower;
    float episode_sec, episode_len, bz_reward, bar_half_width;
    uint32_t next_seed;                 // value the next Env::reset() re-seeds with (env.cpp:61-62)
    int32_t seed_is_env_seed;           // 1: next_seed is the Env::seed() value, reset must draw first
    float p_episode_len_sec, p_vertical_look_limit;   // float params (scenario.hpp:225-232)
    int32_t scenario, num_terrain, num_rewards, num_platforms, solved;   // Obstacles family; Rearrange: num_terrain = items,
                                                                          // num_platforms = maxMatchingObjects; Collect: num_platforms = number of +1
                                                                          // diamonds, highest_tower = how many of them were collected
    int32_t episodes_consumed;          // how many host-generated episodes this env has taken (refill protocol)
    int32_t starved;                    // set if a reset found no fresh episode (must never happen)
    float hex_target[2];                // HexExplore: x, z of rewardObjectCoords (y = 0)
};
static_assert(sizeof(EnvHeader) == 128, "EnvHeader must be 128 B");

struct alignas(16) LayoutBox {   // 32 B, merged layout parallelepiped (voxel units, max exclusive)
    int32_t min[3]; int32_t type;
    int32_t max[3]; int32_t slot;
};

struct alignas(4) MovableObject {   // 4 B
    int8_t x, y, z;
    int8_t state;   // 0 placed at (x,y,z); 1+k carried by agent k; -1 placed, but its grid cell was erased (Collect).
                    // Reward objects: 0 collected, 1 still there (Collect: 1 = +1 diamond, 2 = -1 diamond)
};

struct alignas(16) ArrangementItem {   // 32 B (scenario_rearrange.hpp:20-48): one item of the target arrangement; item i's movable copy is objects[i]
    int32_t shape, color;
    int32_t off[3];
    int32_t pad[3];
};

// Hex scenarios: one record type for boxes and collectables.
//   box:    a = lo, b = hi in the box's frame; meta = (frame + 1) | collide << 4, frame -1 = world, 0..2 = wall orientation HEX_ROT[k]
//   object: a = position, b = scale as given to addSphere / addPillar / addDiamond (layout_utils.cpp:85-126);
//           meta = shape | good << 4 | alive << 8 | (voxel x + 128) << 12 | (voxel z + 128) << 20  (its cell in the collect grid, y = 0)
struct alignas(16) HexRec {   // 32 B
    float a[3]; int32_t meta;
    float b[3]; int32_t color;
};

struct alignas(16) TerrainBox {   // 32 B, voxel units, max exclusive (platforms.hpp terrainBoxes)
    int32_t min[3]; int32_t type;
    int32_t max[3]; int32_t pad;
};

struct alignas(16) AgentState {   // 128 B
    float pos[3];                 // capsule centre == Bullet ghost origin
    float m00, m02, m20, m22;     // yaw basis
    float pitch;
    float hvx, hvz, vvel, voffset, step_offset, jump_speed;
    int32_t was_jumping, carrying, picked_up, visited_zone;
    int32_t spawn[3];
    float last_reward, total_reward;
    float shaping[NUM_SHAPING];   // per-scenario reward-shaping coefficients (scenario.hpp:184-215)
    int32_t pad[1];
};
static_assert(sizeof(AgentState) == 128, "AgentState must be 128 B");

// ---- cost bins of the observation pass (mv_frame.h fills them, mv_raster.hip reads them, mv_api.hip sizes them)
constexpr int MAX_STEP_TICKS = 16;   // ticks of one batched call = of its one observation launch (its step launches hold up to 8 each: StepTicksArgs8)
constexpr int LPT_BUCKETS = 256;
// Every cost bin has LPT_SUBS counters and lists, picked by frame index: the frames of a launch finish together and most of them fall into
// the same three or four bins -- one counter per bin made their returning atomics queue up at one L2 address (measured: 1.3 us of a 7 us frame setup).
constexpr int LPT_SUBS = 4;   // (one 16-byte read per bin in the raster prologue; 8 measured: the step kernel no faster, the prologue slower)
__host__ __device__ constexpr int lpt_sub_capacity(int frames) { return (frames + LPT_SUBS - 1) / LPT_SUBS; }   // frames that can share one (bin, sub) list


// Everything a kernel needs, passed by value.
struct GymView {
    int32_t num_envs, num_agents;
    int32_t box_stride, reward_stride;   // MAX_BOXES / MAX_REWARDS, or the COLLECT_ sizes
    int32_t scenario;                    // SCN_*: one scenario per gym
    int32_t env_offset, env_stride;      // job-wide index of local env j = env_offset + j * env_stride
    int32_t sample_on;                   // POLICY_*: != 0: this tick's actions are drawn inside the step kernel (mv_actions.h)
    uint32_t sample_seed, sample_step;
    EnvHeader *hdr;            // [N]
    LayoutBox *boxes;          // [N][box_stride]
    MovableObject *objects;    // [N][MAX_OBJECTS]
    AgentState *agents;        // [N][A]
    uint8_t *chunk;            // [N][CHUNK_BYTES]   (TowerBuilding)
    TerrainBox *terrain;       // [N][MAX_TERRAIN]   (Obstacles)
    MovableObject *rewards_obj;// [N][reward_stride] (Obstacles: green diamonds; Collect: green/red diamonds)
    int8_t *heightmap;         // [N][HM_BYTES]      (Collect: top solid y of column x * HM_DIM + z, -1 = no voxels)
    ArrangementItem *items;    // [N][MAX_ITEMS]     (Rearrange: the target arrangement; hdr.num_terrain holds the item count)
    uint8_t *soko_cells;       // [N][SOKO_DIM * SOKO_DIM] (Sokoban: SOKO_WALL / SOKO_GOAL per level cell, [x * SOKO_DIM + z])
    HexRec *hex_boxes;         // [N][HEX_MAX_BOXES] (Hex*: hdr.num_boxes boxes, the first hdr.num_terrain of them collide)
    HexRec *hex_objs;          // [N][HEX_MAX_OBJS]  (Hex*: hdr.num_rewards collectables)
    int32_t *episode_status;   // [N + 2] episodes consumed per env (host-generated scenarios), their total, error flags (ST_*)
    const void *blobs;         // [N][spares] resident next episodes (EpisodeBlob / CollectBlob / RearrangeBlob / SokobanBlob): episode
                               // number q (1-based) of an env lives in ring slot (q - 1) % spares
    int32_t spares;            // resident episodes per env (2: a second reset can follow the first before the host has refilled)
    int32_t *actions;          // [N][A] bitmasks
    const int32_t *md_actions; // != null: this tick's actions as multi-discrete [N*A][6] in the caller's device buffer (mv_set_actions_device)
    float *rewards;            // [N*A] as reported by get_last_rewards (0 on done steps)
    uint8_t *done;             // [N]
    float *true_objective;     // [N*A]
    // per-frame visible-primitive lists (mv_raster.hip: frame_setup_kernel -> raster_kernel)
    void *vis_prims;           // [N*A][vis_stride] 32-byte records
    void *vis_rects;           // [N*A][vis_stride] short4 screen rectangles
    int32_t *vis_count;        // [N*A]
    int32_t vis_stride;        // 256, or 1024 for Collect
    int32_t *lpt_bucket;       // [N*A] cost bin of every frame (raster scheduling)
    int32_t *lpt_order;        // [N*A] frames sorted by cost bin, most expensive first (exact raster kernel)
    int32_t *lpt_hist;         // [lpt_hists][256][LPT_SUBS] frames per cost bin and sub-list, rotating over the passes (fast raster kernel)
    int32_t lpt_hists;         // number of histograms (slots + 1)
    int32_t *lpt_list;         // [256][LPT_SUBS][ceil(N*A / LPT_SUBS)] the frames of every bin's sub-lists in arrival order
    int32_t lpt_parity;        // which of the histograms this observation pass uses
    uint8_t *vis_hdr;          // [N*A][FRAME_HDR_BYTES] per-frame header for raster_fast_kernel (cameras, light vectors, masks, count)
    unsigned long long *dbg;   // null, or (builds with -DMV_TICK_TIMING, MV_TICK_TIMING=1) [N][64] counters of the TowerBuilding tick
    int32_t debug_redo;        // tests (MV_DEBUG_FORCE_REDO=1): the multi-agent tick takes its "check failed" path every time
    // long lists (Collect, Hex), fast pixels: the frame setup leaves the list in depth classes, nearest first (mv_frame.h: DepthSortScratch), so that the
    // observation pass can stop walking it where everything nearer has covered a tile (mv_raster.hip: raster_glist_body)
    uint8_t *sort_scratch;     // [N*A][vis_stride] x (32 + 8) bytes: the list as found, before it is dealt into its depth classes (null: short lists)
    // 1: deal the list into depth classes (set per launch: fast pixel mode only -- the exact kernel resolves depth ties by list position)
    int32_t depth_sort;
    // 1: the frame setup does not clear the cost histogram of the next pass (a multi-tick step launch: the pass that draws from a histogram clears it,
    // mv_raster.hip: hist_done)
    int32_t lpt_no_clear;
    struct TowerGen *tower_gen;// [N] TowerBuilding: where each env's episode generator stands (mv_reset_device.h: tower_draw); its resident episodes are `blobs` (TowerBlob)
};

// The n <= 8 consecutive ticks of a multi-tick step launch (mv_step.hip: step_ticks_kernel, and every mv_step_*.hip), the same envs in all of them: their views
// BY VALUE as the launch's arguments -- 2.6 KB of the 4 KB kernel-argument segment; the kernels read a tick's view with scalar loads on demand, its fields do
// not live in registers across the tick.  A call of more than 8 ticks (MAX_STEP_TICKS: 16) is two such launches back to back.  (Built and measured in round 5,
// removed: one launch of 16 ticks with its views in device memory, written there by a small kernel in front of it -- that kernel, queued behind the previous
// step launch and beside an observation launch that had just taken the chip, took 10-44 us and made the step launch arrive late: Empty 27.6 M obs/s against
// 41.5 M with two launches of 8, r08t; and deriving tick j's view from tick 0's in the kernel: 100-140 bytes of scratch per lane more and 1.5-3.5 % of the
// rate.)
struct StepTicksArgs8 {
    int32_t n, pad;
    GymView gv[8];
    __host__ __device__ const GymView &view(int j) const { return gv[j]; }
};
static_assert(sizeof(StepTicksArgs8) + 16 <= 4096, "StepTicksArgs8 + (W, H) must fit the 4 KB kernel-argument segment");

// tick j's view from tick 0's: ten buffers one hand-over slot (slot_stride bytes) further per tick (mv_api.hip carves a gym's slots out of its arena one after
// the other), the action index, the cost histogram (consecutive, modulo their number).  Used where k x n views are too many to pass: the
// group kernels (mv_union.h) and the observation launch of a batched call (mv_raster.hip).
__host__ __device__ inline GymView tick_view(const GymView &base, int64_t slot_stride, int j)
{
    GymView v = base;
    const int64_t d = slot_stride * j;
    v.vis_prims = (uint8_t *)base.vis_prims + d;
    v.vis_rects = (uint8_t *)base.vis_rects + d;
    v.vis_count = (int32_t *)((uint8_t *)base.vis_count + d);
    v.lpt_bucket = (int32_t *)((uint8_t *)base.lpt_bucket + d);
    v.lpt_order = (int32_t *)((uint8_t *)base.lpt_order + d);
    v.vis_hdr = base.vis_hdr + d;
    v.lpt_list = (int32_t *)((uint8_t *)base.lpt_list + d);
    v.rewards = (float *)((uint8_t *)base.rewards + d);
    v.done = base.done + d;
    v.true_objective = (float *)((uint8_t *)base.true_objective + d);
    v.sample_step = base.sample_step + (uint32_t)j;
    v.lpt_parity = (base.lpt_parity + j) % base.lpt_hists;
    return v;
}
// the slot stride of k views that are one hand-over slot apart per tick (what mv_api.hip hands over); false: they are not
inline bool slot_stride_of(const GymView *views, int k, int64_t &stride)
{
    stride = k > 1 ? (int64_t)((const uint8_t *)views[1].vis_prims - (const uint8_t *)views[0].vis_prims) : 0;
    for (int j = 1; j < k; ++j) {
        const GymView w = tick_view(views[0], stride, j);
        if (views[j].vis_prims != w.vis_prims || views[j].vis_hdr != w.vis_hdr || views[j].lpt_list != w.lpt_list
            || views[j].rewards != w.rewards || views[j].done != w.done ||
            views[j].true_objective != w.true_objective || views[j].lpt_parity != w.lpt_parity || views[j].lpt_no_clear != w.lpt_no_clear)
            return false;
    }
    return true;
}

// TowerBuilding: one episode as its generator DREW it -- everything of Env::reset that consumes the env's random stream (scenario_tower_building.cpp:19-89,
// 129-154, scenario_default.hpp:80-97) -- resident in HBM ahead of the reset that will build it (mv_reset_device.h: tower_draw fills it in a kernel of its own,
// off the step path; tower_swap_in -- the tail of a finishing env's tick, or mv_reset -- turns it into chunk, boxes, header and agents).
struct alignas(16) TowerBlob {
    int32_t seq;                        // 1-based index of this episode for its env; 0 = empty
    int32_t L, H, W;                    // room extents
    int32_t bz[4];                      // building zone
    int32_t layout_color, wall_color, draw_walls, num_objects;
    float bz_reward;                    // initial tower reward, summed in object order
    int32_t pad[3];
    int32_t spawn[MAX_AGENTS];          // x << 8 | z of every agent's spawn cell (y = 2)
    float yaw_rot[MAX_AGENTS];          // frand * 2 pi as drawn for the agent's spawn rotation
    MovableObject objects[MAX_OBJECTS];
};
static_assert(sizeof(TowerBlob) == 448, "TowerBlob: 448 B");

// where an env's TowerBuilding generator stands: the seed the NEXT episode's Env::reset re-seeds with (env.cpp:61-62) and how many episodes were drawn
struct alignas(16) TowerGen {
    uint32_t seed;
    int32_t seed_is_env_seed;           // 1: `seed` is the Env::seed() value, the reset draws its seed from it first
    int32_t generated;                  // episodes drawn so far (episode `generated` lives in ring slot (generated - 1) % spares)
    int32_t pad;
};

// One host-generated episode (Obstacles family): everything Env::reset produces, ready to be swapped in by
// the reset kernel.  Fixed-size POD so that the host can fill a pinned staging copy and upload it as is.
struct alignas(16) EpisodeBlob {
    int32_t seq;                        // 1-based index of this episode for its env; 0 = empty
    int32_t num_boxes, num_terrain, num_objects, num_rewards, num_platforms;
    int32_t layout_color, wall_color, draw_walls;
    int32_t dim[3], org[3];
    float episode_len;
    int32_t spawn[MAX_AGENTS][3];
    float yaw_frand[MAX_AGENTS];        // the frand() drawn for each agent's spawn rotation (scenario_default.hpp:87)
    LayoutBox boxes[MAX_BOXES];
    TerrainBox terrain[MAX_TERRAIN];
    MovableObject objects[MAX_OBJECTS];
    MovableObject rewards[MAX_REWARDS];
};

// Rearrange episode (everything RearrangeScenario::reset + spawnAgents + addEpisodeDrawables draw)
struct alignas(16) RearrangeBlob {
    int32_t seq;
    int32_t num_boxes, num_items, max_matching, draw_walls;
    int32_t dim[3];
    float episode_len;
    int32_t pad[3];
    int32_t spawn[MAX_AGENTS][3];
    float yaw_frand[MAX_AGENTS];
    LayoutBox boxes[TOWER_BOXES];
    ArrangementItem items[MAX_ITEMS];
    MovableObject objects[MAX_ITEMS];
};

// Sokoban episode (mv_gen_sokoban.cpp -> mv_step_sokoban.hip)
struct alignas(16) SokobanBlob {
    int32_t seq;
    int32_t num_boxes, num_objects;
    int32_t dim[3];
    int32_t floor_color;
    float episode_len;
    float spawn[MAX_AGENTS][3];          // agentStartingPositions (not voxel corners: agents share the player's cell)
    float yaw_frand[MAX_AGENTS];
    LayoutBox boxes[MAX_BOXES];          // in voxels; one voxel is 2 units wide
    MovableObject objects[MAX_OBJECTS];  // the pushable boxes
    uint8_t cells[SOKO_DIM * SOKO_DIM];  // [x * SOKO_DIM + z]: SOKO_WALL / SOKO_GOAL
};

// HexMemory / HexExplore episode (mv_gen_hex.cpp -> mv_step_hex.hip).  `boxes` comes last: only the used prefix travels.
struct alignas(16) HexBlob {
    int32_t seq;
    int32_t num_boxes, num_colliders, num_objs, num_good;
    float episode_len;
    float target[2];                     // HexExplore: rewardObjectCoords x, z
    float spawn[MAX_AGENTS][3];          // agentStartingPositions (+ (0.5, 0, 0.5) and the agent height are added on the device)
    float yaw[MAX_AGENTS];               // spawn rotation in radians (HexExplore: frand * 2 pi; HexMemory: i * 2 pi / A)
    HexRec objs[HEX_MAX_OBJS];
    HexRec boxes[HEX_MAX_BOXES];
};

// Collect episode.  `boxes` comes last so that only the used prefix needs to travel.
struct alignas(16) CollectBlob {
    int32_t seq;
    int32_t num_boxes, num_objects, num_rewards, num_positive;
    int32_t layout_color, wall_color;
    int32_t dim[3];
    float episode_len;
    int32_t pad;
    int32_t spawn[MAX_AGENTS][3];
    float yaw_frand[MAX_AGENTS];
    MovableObject objects[MAX_OBJECTS];
    MovableObject rewards[COLLECT_MAX_REWARDS];
    int8_t heightmap[HM_BYTES];
    LayoutBox boxes[COLLECT_MAX_BOXES];
};

}  // namespace mv
