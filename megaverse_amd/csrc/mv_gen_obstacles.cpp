// megaverse_amd/csrc/mv_gen_obstacles.cpp -- host-side level generator of the Obstacles scenario family.
//
// Replaces ObstaclesScenario::reset + addEpisodeDrawables
//   (reference: src/libs/scenarios/src/scenario_obstacles.cpp:51-195,241-260), the platform classes
//   (src/libs/scenarios/include/scenarios/platforms.hpp:137-557), VoxelGridComponent::addPlatform /
//   toBoundingBoxes (component_voxel_grid.hpp:73-187) and DefaultScenario::spawnAgents' random draw
//   (scenario_default.hpp:87).
//
// Why on the host: the generator is branchy, retry-based code around std::map/std::set and draws from
// std::mt19937 through libstdc++ distributions; here it uses exactly those library templates, so the RNG
// stream is the reference's by construction.  An episode never depends on what happens during the previous
// one (nothing draws from the env rng while stepping), so the library always keeps ONE finished episode per
// env resident in HBM (EpisodeBlob) and the reset kernel just swaps it in; the host refills behind the GPU
// (mv_api.hip: refill_episodes).  TowerBuilding keeps its fully on-device generator (mv_reset.hip).
//
// Geometry: the reference parents every platform under the previous platform's anchor in a Magnum scene
// graph (90-degree turns about Y, integer offsets) and reads boxes back with lround(); all of that is
// exact integer rigid motion, represented here as a rotation count plus a translation.
#include <algorithm>
#include <cmath>
#include <cstring>
#include <map>
#include <random>
#include <set>
#include <vector>

#include "mv_gen.h"

#include <atomic>

namespace mv {

// per THREAD: a generator runs on a feeder worker (EpisodeFeeder::generate collects the flags into its own gym's feeder afterwards) or on
// the caller's thread (mv_debug_generate_episode) -- one gym's oversized level is never reported by another gym
static thread_local int t_generator_overflow = 0;
void generator_overflow_raise(int flags) { t_generator_overflow |= flags; }
int generator_overflow_take() { const int f = t_generator_overflow; t_generator_overflow = 0; return f; }
static inline bool fits_i8(int v) { return v >= -128 && v <= 127; }


namespace {

using Rng = std::mt19937;
inline int rand_range(int lo, int hi, Rng &rng) { return std::uniform_int_distribution<>{lo, hi - 1}(rng); }   // util.hpp:30-33
inline float frand01(Rng &rng) { return std::uniform_real_distribution<float>{0, 1}(rng); }                    // util.hpp:46-49

const unsigned kLayoutColors[14] = {0xffffff, 0xffffe6, 0xccffcc, 0xe6ecff, 0xd9d9d9, 0xffebcc, 0xb3b3b3,
                                    0xb3b3b3, 0xb3b3b3, 0xb3b3b3, 0x555555, 0x555555, 0x555555, 0x555555};   // env/const.hpp:121-136

struct Int3 { int x, y, z; };

struct Rigid {   // q = rot^k(p) + off, rot = quarter turn about +Y: (x, y, z) -> (z, y, -x)
    int k = 0;
    Int3 off{0, 0, 0};
    static Int3 turn(int k, Int3 p)
    {
        switch (k & 3) {
            case 1: return Int3{p.z, p.y, -p.x};
            case 2: return Int3{-p.x, p.y, -p.z};
            case 3: return Int3{-p.z, p.y, p.x};
            default: return p;
        }
    }
    Int3 map(Int3 p) const { const Int3 r = turn(k, p); return Int3{r.x + off.x, r.y + off.y, r.z + off.z}; }
    Rigid then(const Rigid &inner) const   // this o inner
    {
        Rigid r;
        r.k = (k + inner.k) & 3;
        r.off = map(inner.off);
        return r;
    }
    // Object3D::rotateYLocal(quarters) followed by translateLocal(v): p -> rot(p + v)
    static Rigid rotate_then_shift(int quarters, Int3 v)
    {
        Rigid r;
        r.k = quarters & 3;
        r.off = turn(quarters, v);
        return r;
    }
};

struct Aabb { Int3 lo, hi; };

Aabb placed(const Rigid &root, const Aabb &local)
{   // MagnumAABB::boundingBox (platforms.hpp:126-133)
    const Int3 a = root.map(local.lo), b = root.map(local.hi);
    return Aabb{{std::min(a.x, b.x), std::min(a.y, b.y), std::min(a.z, b.z)}, {std::max(a.x, b.x), std::max(a.y, b.y), std::max(a.z, b.z)}};
}

enum Kind { EMPTY, WALL, LAVA, STEP, GAP, START, EXIT, TRANSITION };
enum { SOUTH = 1, NORTH = 2, WEST = 4, EAST = 8 };

struct Platform {
    Kind kind = EMPTY;
    int walls = 0, length = 0, height = 0, width = -1;
    Rigid parent, own, root;
    int anchor_rise = 0;
    std::vector<Aabb> floor_boxes, wall_boxes;
    std::vector<std::pair<int, Aabb>> terrain;
    std::map<std::pair<int, int>, int> occupancy;
    int wall_h = 0, lava_len = 0, step_h = 0, gap = 0, gap_x = 0;

    void place() { root = parent.then(own); }
    Rigid next_anchor() const { return root.then(Rigid::rotate_then_shift(0, Int3{length, anchor_rise, 0})); }

    void roll(Rng &rng, const ObstacleConfig &c)
    {   // the init() chain of the platform classes
        if (kind == TRANSITION) { height = 5; return; }
        length = rand_range(4, 10, rng);
        if (width == -1) width = rand_range(5, 9, rng);
        height = 5;
        switch (kind) {
            case WALL:
                wall_h = rand_range(c.min_height, c.max_height + 1, rng);
                height = rand_range(wall_h + 4, wall_h + 6, rng);
                break;
            case LAVA: {
                length = rand_range(6, 12, rng);
                const int lo = std::min(c.min_lava, length - 2), hi = std::min(c.max_lava + 1, length - 1);
                lava_len = rand_range(lo, hi, rng);
                break;
            }
            case STEP:
                step_h = rand_range(c.min_height, c.max_height + 1, rng);
                height = rand_range(step_h + 2, step_h + 5, rng);
                break;
            case GAP:
                gap = rand_range(c.min_gap, std::min(c.max_gap + 1, length - 1), rng);
                gap_x = rand_range(1, length - gap, rng);
                break;
            default: break;
        }
    }

    void side_walls()
    {
        if (walls & SOUTH) wall_boxes.push_back(Aabb{{0, 0, 0}, {1, height, width}});
        if (walls & NORTH) wall_boxes.push_back(Aabb{{length - 1, 0, 0}, {length, height, width}});
        if (walls & EAST) wall_boxes.push_back(Aabb{{0, 0, 0}, {length, height, 1}});
        if (walls & WEST) wall_boxes.push_back(Aabb{{0, 0, width - 1}, {length, height, width}});
    }

    void build(Rng &rng)
    {   // the generate() chain
        if (kind == STEP) {
            const int sx = rand_range(1, length, rng);
            floor_boxes.push_back(Aabb{{0, 0, 0}, {sx + 1, 1, width}});
            floor_boxes.push_back(Aabb{{sx, step_h, 0}, {length, step_h + 1, width}});
            floor_boxes.push_back(Aabb{{sx, 0, 0}, {sx + 1, step_h + 1, width}});
            anchor_rise = step_h;
            side_walls();
            for (int x = sx + 1; x < length; ++x)
                for (int z = 1; z < width; ++z) occupancy[{x, z}] = step_h;
            return;
        }
        if (kind == GAP) {
            floor_boxes.push_back(Aabb{{0, 0, 0}, {gap_x, 1, width}});
            floor_boxes.push_back(Aabb{{gap_x + gap, 0, 0}, {length, 1, width}});
            side_walls();
            return;
        }
        floor_boxes.push_back(Aabb{{0, 0, 0}, {length, 1, width}});
        side_walls();
        if (kind == WALL) {
            const int wx = rand_range(1, length, rng);
            const int thick = rand_range(1, length - wx + 1, rng);
            floor_boxes.push_back(Aabb{{wx, 1, 1}, {wx + thick, 1 + wall_h, width - 1}});
            for (int x = wx; x < wx + thick; ++x)
                for (int z = 1; z < width; ++z) occupancy[{x, z}] = wall_h;
        } else if (kind == LAVA) {
            const int lx = rand_range(1, length - lava_len, rng);
            terrain.push_back({TERRAIN_LAVA, Aabb{{lx, 1, 1}, {lx + lava_len, 2, width - 1}}});
        } else if (kind == EXIT) {
            terrain.push_back({TERRAIN_EXIT, Aabb{{length - 3, 1, 1}, {length - 1, 3, width - 1}}});
        }
    }

    bool hardest(const ObstacleConfig &c) const
    {
        return (kind == WALL && wall_h >= c.max_height) || (kind == LAVA && lava_len >= c.max_lava) || (kind == STEP && step_h >= c.max_height);
    }
    int boxes_needed() const
    {
        auto tri = [](int n) { return n * (n + 1) / 2; };
        switch (kind) {
            case WALL: return tri(wall_h - 1);
            case LAVA: return std::max(1, lava_len - 1);
            case STEP: return tri(step_h - 1);
            case GAP: return tri(std::max(0, gap - 2));
            default: return 0;
        }
    }
    Aabb hull() const
    {
        Aabb h{{0, 0, 0}, {0, 0, 0}};
        bool first = true;
        auto eat = [&](const Aabb &b) {
            if (first) { h = b; first = false; return; }
            h.lo = Int3{std::min(h.lo.x, b.lo.x), std::min(h.lo.y, b.lo.y), std::min(h.lo.z, b.lo.z)};
            h.hi = Int3{std::max(h.hi.x, b.hi.x), std::max(h.hi.y, b.hi.y), std::max(h.hi.z, b.hi.z)};
        };
        for (const auto &b : floor_boxes) eat(placed(root, b));
        for (const auto &b : wall_boxes) eat(placed(root, b));
        return h;
    }
    // voxel whose centre is the image of the local voxel centre (adjustTransformation, platforms.hpp:280-288)
    Int3 to_world_voxel(int x, int y, int z) const
    {
        const Int3 c2 = root.map(Int3{2 * x + 1, 2 * y + 1, 2 * z + 1});   // doubled coordinates, offset applied once too many below
        const Int3 t = root.off;
        auto half_floor = [](int v) { return v >= 0 ? v / 2 : -((-v + 1) / 2); };
        return Int3{half_floor(c2.x + t.x), half_floor(c2.y + t.y), half_floor(c2.z + t.z)};
    }
    std::vector<Int3> scatter(int n, Rng &rng)
    {
        std::vector<Int3> out;
        if (kind == GAP) {
            std::vector<Int3> cand;
            for (int x = 0; x < length; ++x)
                for (int z = 1; z < width - 1; ++z)
                    if (!(x >= gap_x && x < gap_x + gap)) cand.push_back(Int3{x, 1, z});
            for (int i = 0; i < n; ++i) {
                const Int3 v = cand[rand_range(0, int(cand.size()), rng)];
                const int y = ++occupancy[{v.x, v.z}];
                out.push_back(to_world_voxel(v.x, y, v.z));
            }
            return out;
        }
        for (int i = 0; i < n; ++i)
            for (int attempt = 0; attempt < 10; ++attempt) {
                const int x = rand_range(1, length - 1, rng);
                const int z = rand_range(1, width - 1, rng);
                if (occupancy[{x, z}] < 2 || attempt >= 9) {
                    const int y = ++occupancy[{x, z}];
                    out.push_back(to_world_voxel(x, y, z));
                    break;
                }
            }
        return out;
    }
};

bool overlap(const Aabb &a, const Aabb &b)
{
    return !(a.hi.x <= b.lo.x || a.lo.x >= b.hi.x || a.hi.y <= b.lo.y || a.lo.y >= b.hi.y || a.hi.z <= b.lo.z || a.lo.z >= b.hi.z);
}

}  // namespace

void generate_obstacles_episode(std::mt19937 &rng, const ObstacleConfig &cfg, int num_agents, float base_episode_len, EpisodeBlob &out)
{
    std::memset(&out, 0, sizeof out);
    // Env::reset (env.cpp:61-62): re-seed from the env's own stream
    const int seed = rand_range(0, 1 << 30, rng);
    rng.seed((unsigned long)seed);

    const bool draw_walls = rand_range(0, 2, rng) != 0;
    std::vector<Platform> chain;
    int num_platforms = 0;
    for (int attempt = 0; attempt < 20; ++attempt) {
        chain.clear();
        num_platforms = rand_range(cfg.min_platforms, cfg.max_platforms + 1, rng);
        Platform first;
        first.kind = START; first.walls = SOUTH | EAST | WEST;
        first.roll(rng, cfg); first.place(); first.build(rng);
        int want_width = first.width;
        chain.push_back(first);
        size_t last = 0;
        int hardest_so_far = 0;
        for (int i = 0; i < num_platforms; ++i) {
            const int turn = rand_range(0, 3, rng);   // 0 straight, 1 left, 2 right
            if (turn != 0) want_width = -1;
            Platform p;
            for (bool drawn = false; !drawn || (p.hardest(cfg) && hardest_so_far >= cfg.num_allowed_max_difficulty); drawn = true) {
                p = Platform();
                p.kind = Kind(cfg.platform_types[rand_range(0, cfg.num_platform_types, rng)]);
                p.walls = WEST | EAST;
                p.width = want_width;
                p.roll(rng, cfg);
            }
            if (p.hardest(cfg)) ++hardest_so_far;
            p.parent = chain[last].next_anchor();
            p.build(rng);
            if (turn == 1) p.own = Rigid::rotate_then_shift(1, Int3{-1, 0, -1});
            else if (turn == 2) p.own = Rigid::rotate_then_shift(3, Int3{chain[last].width - 1, 0, -p.width + 1});
            p.place();
            chain.push_back(p);
            const size_t me = chain.size() - 1;
            if (turn != 0) {
                Platform corner;
                corner.kind = TRANSITION;
                corner.walls = NORTH | (turn == 1 ? WEST : EAST);
                corner.length = chain[me].width - 1;
                corner.width = chain[last].width;
                corner.parent = chain[last].next_anchor();
                corner.place(); corner.roll(rng, cfg); corner.build(rng);
                chain.push_back(corner);
            }
            last = me;
            want_width = chain[me].width;
        }
        Platform goal;
        goal.kind = EXIT; goal.walls = NORTH | EAST | WEST; goal.width = want_width;
        goal.parent = chain[last].next_anchor();
        goal.roll(rng, cfg); goal.place(); goal.build(rng);
        chain.push_back(goal);

        bool clash = false;
        for (int j = 0; j < int(chain.size()) && !clash; ++j)
            for (int k = 0; k < j - 2; ++k)
                if (overlap(chain[j].hull(), chain[k].hull())) { clash = true; break; }
        if (!clash) break;
    }
    out.layout_color = (int)kLayoutColors[rand_range(0, 14, rng)];
    out.wall_color = (int)kLayoutColors[rand_range(0, 14, rng)];
    out.draw_walls = draw_walls ? 1 : 0;
    out.num_platforms = num_platforms;

    // ---- voxelise over the level's bounding box; later boxes overwrite earlier ones (grid.set)
    Int3 lo{1 << 20, 1 << 20, 1 << 20}, hi{-(1 << 20), -(1 << 20), -(1 << 20)};
    for (const auto &p : chain) {
        const Aabb h = p.hull();
        lo = Int3{std::min(lo.x, h.lo.x), std::min(lo.y, h.lo.y), std::min(lo.z, h.lo.z)};
        hi = Int3{std::max(hi.x, h.hi.x), std::max(hi.y, h.hi.y), std::max(hi.z, h.hi.z)};
    }
    const int nx = hi.x - lo.x, ny = hi.y - lo.y, nz = hi.z - lo.z;
    std::vector<uint8_t> vox(size_t(nx) * ny * nz, 0);
    auto at = [&](int x, int y, int z) -> uint8_t & { return vox[(size_t(y - lo.y) * nz + (z - lo.z)) * nx + (x - lo.x)]; };
    auto paint = [&](const Aabb &b, uint8_t v) {
        for (int x = b.lo.x; x < b.hi.x; ++x)
            for (int y = b.lo.y; y < b.hi.y; ++y)
                for (int z = b.lo.z; z < b.hi.z; ++z) at(x, y, z) = v;
    };
    const uint8_t v_floor = VX_SOLID | VX_OPAQUE, v_wall = uint8_t(VX_SOLID | (draw_walls ? VX_OPAQUE : 0) | (1 << VX_COLOR_SHIFT));
    // (the merge below visits the classes in key order; only these two values are ever painted: the ten other classes are not scanned for -- the merge,
    // twelve passes over ~10^5 cells, was 5/6 of this generator's time)
    const unsigned present = (1u << (((v_floor & 3) << 2) | (v_floor >> VX_COLOR_SHIFT))) | (1u << (((v_wall & 3) << 2) | (v_wall >> VX_COLOR_SHIFT)));
    for (const auto &p : chain) {
        for (const auto &b : p.floor_boxes) paint(placed(p.root, b), v_floor);
        for (const auto &b : p.wall_boxes) paint(placed(p.root, b), v_wall);
    }
    out.dim[0] = nx; out.dim[1] = ny; out.dim[2] = nz;
    out.org[0] = lo.x; out.org[1] = lo.y; out.org[2] = lo.z;

    // ---- canonical greedy merge: keys by (type, colour slot); scan y, z, x; grow x, then z, then y
    {
        std::vector<uint8_t> used(vox.size(), 0);
        auto idx = [&](int x, int y, int z) { return (size_t(y) * nz + z) * nx + x; };
        int nb = 0;
        for (int key = 4; key < 16; ++key) {
            if (!((present >> key) & 1u)) continue;
            const int type = key >> 2, slot = key & 3;
            auto is = [&](int x, int y, int z) {
                if (x < 0 || y < 0 || z < 0 || x >= nx || y >= ny || z >= nz) return false;
                const size_t i = idx(x, y, z);
                return !used[i] && (vox[i] & 3) == type && (vox[i] >> VX_COLOR_SHIFT) == slot;
            };
            for (int y = 0; y < ny; ++y)
                for (int z = 0; z < nz; ++z)
                    for (int x = 0; x < nx; ++x) {
                        if (!is(x, y, z)) continue;
                        int x1 = x, z1 = z, y1 = y;
                        while (is(x1 + 1, y, z)) ++x1;
                        for (bool ok = true; ok;) {
                            for (int xx = x; xx <= x1 && ok; ++xx) ok = is(xx, y, z1 + 1);
                            if (ok) ++z1;
                        }
                        for (bool ok = true; ok;) {
                            for (int zz = z; zz <= z1 && ok; ++zz)
                                for (int xx = x; xx <= x1 && ok; ++xx) ok = is(xx, y1 + 1, zz);
                            if (ok) ++y1;
                        }
                        for (int yy = y; yy <= y1; ++yy)
                            for (int zz = z; zz <= z1; ++zz)
                                for (int xx = x; xx <= x1; ++xx) used[idx(xx, yy, zz)] = 1;
                        if (nb >= MAX_BOXES) generator_overflow_raise(GEN_SLABS);
                        if (nb < MAX_BOXES) {
                            LayoutBox &b = out.boxes[nb++];
                            b.min[0] = x + lo.x; b.min[1] = y + lo.y; b.min[2] = z + lo.z;
                            b.max[0] = x1 + 1 + lo.x; b.max[1] = y1 + 1 + lo.y; b.max[2] = z1 + 1 + lo.z;
                            b.type = type; b.slot = slot;
                        }
                    }
        }
        out.num_boxes = nb;
    }

    // ---- terrain boxes in drawable order: per platform, exit pads before lava (std::map<TerrainType,...> order)
    for (const auto &p : chain)
        for (int type : {int(TERRAIN_EXIT), int(TERRAIN_LAVA)})
            for (const auto &t : p.terrain)
                if (t.first == type && out.num_terrain >= MAX_TERRAIN) generator_overflow_raise(GEN_TERRAIN);
                else if (t.first == type) {
                    const Aabb b = placed(p.root, t.second);
                    TerrainBox &o = out.terrain[out.num_terrain++];
                    o.min[0] = b.lo.x; o.min[1] = b.lo.y; o.min[2] = b.lo.z; o.max[0] = b.hi.x; o.max[1] = b.hi.y; o.max[2] = b.hi.z;
                    o.type = type; o.pad = 0;
                }

    // ---- agent spawn cells on the start platform (Platform::agentSpawnPoints)
    {
        Platform &sp = chain[0];
        std::set<std::pair<int, int>> taken;
        int n = 0;
        for (int i = 0; i < num_agents; ++i)
            for (int attempt = 0; attempt < 10; ++attempt) {
                const int x = rand_range(1, sp.length - 1, rng), z = rand_range(1, sp.width - 1, rng);
                if (taken.count({x, z})) continue;
                const int y = sp.occupancy[{x, z}] + 1;
                sp.occupancy[{x, z}] += 2;
                out.spawn[n][0] = x; out.spawn[n][1] = y; out.spawn[n][2] = z;
                ++n;
                taken.insert({x, z});
                break;
            }
        if (n == 0) { out.spawn[0][0] = 1; out.spawn[0][1] = 1; out.spawn[0][2] = 1; n = 1; }
        for (int i = n; i < num_agents; ++i) std::memcpy(out.spawn[i], out.spawn[0], sizeof out.spawn[0]);
    }

    // ---- movable boxes and reward objects
    std::vector<int> share(chain.size(), 0);
    for (int i = 1; i < int(chain.size()); ++i) {
        const int need = chain[i].boxes_needed();
        for (int b = 0; b < need; ++b) ++share[rand_range(std::max(0, i - 2), i, rng)];
    }
    size_t total_objects = 0;
    for (int i = 0; i < int(chain.size()); ++i) {
        const float fraction = frand01(rng) * 0.5f;
        const int extra = int(std::lround(fraction * float(share[i]))) + rand_range(0, 2, rng);
        for (const Int3 &c : chain[i].scatter(share[i] + extra, rng)) {
            if (out.num_objects >= MAX_OBJECTS) generator_overflow_raise(GEN_OBJECTS);
            if (!fits_i8(c.x) || !fits_i8(c.y) || !fits_i8(c.z)) generator_overflow_raise(GEN_COORDS);
            if (out.num_objects < MAX_OBJECTS) out.objects[out.num_objects++] = MovableObject{(int8_t)c.x, (int8_t)c.y, (int8_t)c.z, 0};
            ++total_objects;
        }
    }
    for (int i = 1; i < int(chain.size()) - 1; ++i) {
        const int n = rand_range(0, 2, rng);
        for (const Int3 &c : chain[i].scatter(n, rng)) {
            if (out.num_rewards >= MAX_REWARDS) generator_overflow_raise(GEN_REWARDS);
            if (!fits_i8(c.x) || !fits_i8(c.y) || !fits_i8(c.z)) generator_overflow_raise(GEN_COORDS);
            if (out.num_rewards < MAX_REWARDS) out.rewards[out.num_rewards++] = MovableObject{(int8_t)c.x, (int8_t)c.y, (int8_t)c.z, 1};
        }
    }
    out.episode_len = std::max(base_episode_len, float(num_platforms) * 35 + float(total_objects) * 1);

    for (int i = 0; i < num_agents; ++i) out.yaw_frand[i] = frand01(rng);
}

void generate_empty_episode(std::mt19937 &rng, int num_agents, float base_episode_len, EpisodeBlob &out)
{
    std::memset(&out, 0, sizeof out);
    const int episode_seed = rand_range(0, 1 << 30, rng);   // Env::reset: re-seed from the env's own stream (env.cpp:61-62)
    rng.seed((unsigned long)episode_seed);
    // EmptyScenario::reset() {} ; addStaticCollidingBox(scale (10, 1, 10), translation (5, 0, 5), BLUE), scenario_empty.cpp:25-28
    out.num_boxes = 1;
    LayoutBox &b = out.boxes[0];
    b.min[0] = -5; b.min[1] = -1; b.min[2] = -5; b.max[0] = 15; b.max[1] = 1; b.max[2] = 15;
    b.type = VX_SOLID | VX_OPAQUE; b.slot = 0;
    out.layout_color = 0x2eb5d0; out.wall_color = 0x2eb5d0; out.draw_walls = 0;   // ColorRgb::BLUE
    out.dim[0] = 20; out.dim[1] = 2; out.dim[2] = 20;
    out.episode_len = base_episode_len;
    for (int k = 0; k < num_agents; ++k) { out.spawn[k][0] = out.spawn[k][1] = out.spawn[k][2] = 1; }   // agentStartingPositions :20-23
    for (int i = 0; i < num_agents; ++i) out.yaw_frand[i] = frand01(rng);
}

}  // namespace mv
