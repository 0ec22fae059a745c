// megaverse_amd/csrc/mv_gen_hex.cpp -- host-side episode generator of the HexMemory and HexExplore scenarios.
//
// Replaces, per episode (reference paths relative to src/libs):
//   HoneyCombMaze::InitialiseGraph + Kruskal::SpanningTree + Maze::RemoveBorders   mazes/src/honeycombmaze.cpp:11-107, kruskal.cpp:6-38,
//                                                                                  maze.cpp:17-45 (vendored, dependency-free)
//   HexagonalMazeComponent::reset / addDrawablesAndCollisions                      scenarios/src/component_hexagonal_maze.cpp:20-133
//   HexExploreScenario::reset / agentStartingPositions / addEpisodeDrawables       scenarios/src/scenario_hex_explore.cpp:21-41,60-110
//   HexMemoryScenario::reset / spawnAgents / addEpisodeDrawables                   scenarios/src/scenario_hex_memory.cpp:21-82,131-216
//   DefaultScenario::spawnAgents (one frand per agent)                             scenarios/include/scenarios/scenario_default.hpp:80-97
// Host-side for the same reason as the other generators: the draws go through libstdc++'s mt19937 / shuffle templates, the maze
// geometry is double precision, and the result is swapped in by the reset kernel from a resident blob.
//
// What the device gets (mv_types.h: HexBlob): the floor, walls, edgings and landmarks as boxes, each axis-aligned in the world or in one
// of three frames rotated about Y -- a honeycomb's borders only come in three directions, and a rotation about Y keeps the agents'
// capsules vertical, so neither the physics nor the rays ever see an oriented box -- and the collectables.  The reference's Kruskal
// seeds itself from std::random_device (spanningtreealgorithm.h:19-20); here it is seeded with the episode seed Env::reset drew.
#include <algorithm>
#include <cmath>
#include <cstddef>
#include <cstring>
#include <numeric>
#include <set>
#include <utility>

#include "mv_gen.h"

namespace mv {

namespace {

using Rng = std::mt19937;
inline int rand_range(int lo, int hi, Rng &rng) { return std::uniform_int_distribution<>{lo, hi - 1}(rng); }   // util.hpp:30-33
inline float frand01(Rng &rng) { return std::uniform_real_distribution<float>{0, 1}(rng); }                    // util.hpp:46-49

const unsigned ALL_COLORS[22] = {   // env/include/env/const.hpp:58-83 allColors
    0xffdd3c, 0x3bb372, 0x50c878, 0x2eb5d0, 0xadd8e6, 0x3a7fa6, 0x2c3e50, 0xffb400, 0xb3b3b3, 0x555555, 0x222222,
    0xffffff, 0xff0000, 0xffa770, 0xd468ee, 0xffe6e6, 0xffffe6, 0xccffcc, 0xe6ecff, 0xd9d9d9, 0xf2e6ff, 0xffebcc};
const unsigned OBJECT_COLORS[14] = {   // const.hpp:96-111 objectColors
    0xffdd3c, 0x3bb372, 0x50c878, 0x2eb5d0, 0xadd8e6, 0x3a7fa6, 0xffb400, 0xb3b3b3, 0x555555, 0xffffff, 0xff0000, 0xffa770, 0xd468ee, 0xffe6e6};
const unsigned LAYOUT_COLORS[14] = {   // const.hpp:121-136 layoutColors
    0xffffff, 0xffffe6, 0xccffcc, 0xe6ecff, 0xd9d9d9, 0xffebcc, 0xb3b3b3, 0xb3b3b3, 0xb3b3b3, 0xb3b3b3, 0x555555, 0x555555, 0x555555, 0x555555};

// ---- the honeycomb: cells (u, v) on a skewed grid, |u| < size, numbered row by row ------------------------------------------------------
struct Wall { int other; double ax, ay, bx, by; };   // what is on the far side (-1: outside) and the segment, maze units

class Honeycomb {
public:
    Honeycomb(int size, uint32_t kruskal_seed) : size_{size}, walls_(size_t(3 * size * (size - 1) + 1)), centre_(walls_.size())
    {
        lay_out();
        carve(kruskal_seed);
    }
    int size() const { return size_; }
    int cells() const { return int(walls_.size()); }
    const std::vector<Wall> &walls(int cell) const { return walls_[cell]; }
    std::pair<double, double> centre(int cell) const { return centre_[cell]; }
    double half_width() const { return SQRT3 * (size_ - 0.5); }    // GetCoordinateBounds: [-w, w] x [-h, h]
    double half_height() const { return 1.5 * size_ - 0.5; }

private:
    static constexpr double SQRT3 = 0x1.bb67ae8584caap+0;

    int first_v(int u) const { return u < 0 ? -size_ - u + 1 : -size_ + 1; }
    int last_v(int u) const { return u < 0 ? size_ - 1 : size_ - 1 - u; }
    bool inside(int u, int v) const { return u > -size_ && u < size_ && v >= first_v(u) && v <= last_v(u); }
    int index(int u, int v) const
    {
        return u <= 0 ? ((3 * size_ + u) * (size_ + u - 1)) / 2 + v : (3 * size_ * (size_ - 1) + (4 * size_ - u - 1) * u) / 2 + v;
    }

    // every cell lists its six sides counter-clockwise from the lower left one; a side shared with an already numbered cell is listed
    // by both when the later cell comes along
    void lay_out()
    {
        static const int STEP[6][2] = {{-1, 0}, {-1, 1}, {0, 1}, {1, 0}, {1, -1}, {0, -1}};
        // corner offsets: cos / sin of (n - 2.5) pi / 3 and of (n - 1.5) pi / 3 as the reference's build evaluates them, as literals (a
        // compiler may use sincos() where another calls cos(): one ulp of a double in a cancelling sum)
        static const double FROM_X[6] = {-0x1.bb67ae8584cabp-1, 0x1.1a62633145c07p-54, 0x1.bb67ae8584cabp-1,
                0x1.bb67ae8584cabp-1, 0x1.1a62633145c07p-54, -0x1.bb67ae8584cabp-1};
        static const double FROM_Y[6] = {-0x1.fffffffffffffp-2, -0x1.0000000000000p+0, -0x1.fffffffffffffp-2,
                0x1.fffffffffffffp-2, 0x1.0000000000000p+0, 0x1.fffffffffffffp-2};
        static const double TO_X[6] = {-0x1.72cece675d1fcp-53, 0x1.bb67ae8584caap-1, 0x1.bb67ae8584cabp-1,
                0x1.1a62633145c07p-54, -0x1.bb67ae8584ca9p-1, -0x1.bb67ae8584caap-1};
        static const double TO_Y[6] = {-0x1.0000000000000p+0, -0x1.0000000000000p-1, 0x1.fffffffffffffp-2,
                0x1.0000000000000p+0, 0x1.0000000000003p-1, -0x1.0000000000001p-1};
        for (int u = -size_ + 1; u < size_; ++u)
            for (int v = first_v(u); v <= last_v(u); ++v) {
                const int cell = index(u, v);
                const double x = (SQRT3 / 2) * u + SQRT3 * v, y = 1.5 * u + 0.0 * v;
                centre_[cell] = {x, y};
                for (int side = 0; side < 6; ++side) {
                    const int nu = u + STEP[side][0], nv = v + STEP[side][1];
                    const Wall w{-1, x + FROM_X[side], y + FROM_Y[side], x + TO_X[side], y + TO_Y[side]};
                    if (!inside(nu, nv)) { walls_[cell].push_back(w); continue; }   // the rim keeps all its walls (entrance and exit included)
                    const int neighbour = index(nu, nv);
                    if (neighbour > cell) continue;
                    walls_[cell].push_back(Wall{neighbour, w.ax, w.ay, w.bx, w.by});
                    walls_[neighbour].push_back(Wall{cell, w.ax, w.ay, w.bx, w.by});
                }
            }
    }

    // randomised Kruskal: shuffle the inner walls, knock down every one that joins two separate regions
    void carve(uint32_t seed)
    {
        std::vector<std::pair<int, int>> inner;
        for (int c = 0; c < cells(); ++c)
            for (const Wall &w : walls_[c])
                if (w.other > c) inner.emplace_back(c, w.other);
        std::mt19937 generator(seed);
        std::shuffle(inner.begin(), inner.end(), generator);
        std::vector<int> region(walls_.size());
        std::iota(region.begin(), region.end(), 0);
        auto find = [&](int c) {
            int r = c;
            while (region[r] != r) r = region[r];
            while (region[c] != r) { const int next = region[c]; region[c] = r; c = next; }
            return r;
        };
        for (const auto &[a, b] : inner) {
            const int ra = find(a), rb = find(b);
            if (ra == rb) continue;
            region[ra] = rb;
            knock_down(a, b);
            knock_down(b, a);
        }
    }
    void knock_down(int cell, int other)
    {
        std::vector<Wall> &list = walls_[cell];
        const auto it = std::find_if(list.begin(), list.end(), [&](const Wall &w) { return w.other == other; });
        if (it != list.end()) list.erase(it);
    }

    const int size_;
    std::vector<std::vector<Wall>> walls_;
    std::vector<std::pair<double, double>> centre_;
};

// ---- HexagonalMazeComponent ---------------------------------------------------------------------------------------------------------------
struct MazeLook {   // what reset() draws after the maze itself
    float scale = 3.5f, wall_height, omit_walls, landmark_chance;
    unsigned bottom_edging, top_edging;
};

MazeLook draw_maze_look(Rng &rng, float omit_min, float omit_max)
{
    MazeLook k;
    k.wall_height = frand01(rng) * 0.55f + 0.85f;
    k.omit_walls = frand01(rng) * (omit_max - omit_min) + omit_min;
    k.landmark_chance = frand01(rng) * 0.15f + 0.15f;
    k.bottom_edging = ALL_COLORS[rand_range(0, 22, rng)];
    k.top_edging = ALL_COLORS[rand_range(0, 22, rng)];   // drawn, never shown (the top edging is commented out in the reference)
    return k;
}

// (cos, sin) of the three directions a border of a honeycomb can have once rotateY(-atan(dz / dx)) is applied: +30, -30, 90 degrees
const float FRAME_COS[HEX_FRAMES] = {0.8660254f, 0.8660254f, 0.0f}, FRAME_SIN[HEX_FRAMES] = {0.5f, -0.5f, 1.0f};

struct P3 { float x, y, z; };
inline P3 into_frame(int k, P3 p) { return P3{FRAME_COS[k] * p.x - FRAME_SIN[k] * p.z, p.y, FRAME_SIN[k] * p.x + FRAME_COS[k] * p.z}; }

inline HexRec box_record(int frame, bool collide, P3 centre, P3 half, unsigned color)
{
    HexRec r;
    r.a[0] = centre.x - half.x; r.a[1] = centre.y - half.y; r.a[2] = centre.z - half.z;
    r.b[0] = centre.x + half.x; r.b[1] = centre.y + half.y; r.b[2] = centre.z + half.z;
    r.meta = (frame + 1) | (collide ? 16 : 0);
    r.color = int32_t(color);
    return r;
}

// floor, then per standing wall: its landmarks, the wall, its edging -- and finally the colliding ones moved to the front (stable)
void build_maze_boxes(Rng &rng, const Honeycomb &maze, const MazeLook &look, std::vector<HexRec> &boxes)
{
    const double s = look.scale;
    const double x_min = -maze.half_width() * s, x_max = maze.half_width() * s, y_min = -maze.half_height() * s, y_max = maze.half_height() * s;
    {
        const P3 half{float(x_max - x_min), 0.0001f, float(y_max - y_min)};   // addStaticCollidingBox: the cube [-1, 1]^3 scaled, i.e. twice the maze
        const P3 centre{float(x_max + x_min) / 2, 0.0f, float(y_max + y_min) / 2};
        boxes.push_back(box_record(-1, true, centre, half, LAYOUT_COLORS[rand_range(0, 14, rng)]));
    }
    std::set<std::pair<int, int>> standing;
    for (int cell = 0; cell < maze.cells(); ++cell)
        for (const Wall &w : maze.walls(cell)) {
            const std::pair<int, int> key = std::minmax(cell, w.other);
            if (w.other != -1) {
                if (standing.count(key)) continue;              // seen from the other side already
                if (frand01(rng) < look.omit_walls) continue;   // randomly left out
            }
            standing.insert(key);
            const double x1 = w.ax * s, z1 = w.ay * s, x2 = w.bx * s, z2 = w.by * s;
            const float length = 0.5f * std::sqrt(float((x1 - x2) * (x1 - x2) + (z1 - z2) * (z1 - z2)));
            const P3 mid{float(x1 + x2) / 2, look.wall_height, float(z1 + z2) / 2};
            const double dx = x1 - x2, dz = z1 - z2;
            const int frame = std::fabs(dx) > 1e-5f ? ((dz / dx) < 0 ? 0 : 1) : 2;   // EPSILON, util/macro.hpp:12
            const P3 centre = into_frame(frame, mid);
            if (frand01(rng) < look.landmark_chance) {
                const float lw = 0.15f, lh = lw * length / look.wall_height;
                const int count = rand_range(2, 5, rng);
                for (int i = 0; i < count; ++i) {
                    const float depth = frand01(rng) * 1.2f + 1.5f;
                    const float tx = float(i % 2 == 1) * lw * 2, ty = float(i > 1) * lh * 2 - 0.2f;
                    const P3 c{centre.x + length * tx, centre.y + look.wall_height * ty, centre.z};
                    const P3 h{length * lw, look.wall_height * lh, 0.15f * depth};
                    boxes.push_back(box_record(frame, false, c, h, ALL_COLORS[rand_range(0, 22, rng)]));
                }
            }
            boxes.push_back(box_record(frame, true, centre, P3{length, look.wall_height, 0.15f}, 0x3a7fa6));   // DARK_BLUE
            const P3 eh{length * 1.02f, look.wall_height * 0.12f, 0.2f};
            boxes.push_back(box_record(frame, false, into_frame(frame, P3{mid.x, eh.y, mid.z}), eh, look.bottom_edging));
        }
    std::stable_partition(boxes.begin(), boxes.end(), [](const HexRec &r) { return (r.meta & 16) != 0; });
}

inline HexRec object_record(int shape, unsigned color, P3 at, P3 scale, bool good, bool alive, P3 grid)
{
    HexRec r;
    r.a[0] = at.x; r.a[1] = at.y; r.a[2] = at.z;
    r.b[0] = scale.x; r.b[1] = scale.y; r.b[2] = scale.z;
    const int vx = int(std::lround(std::floor(grid.x))), vz = int(std::lround(std::floor(grid.z)));
    r.meta = shape | (good ? 16 : 0) | (alive ? 256 : 0) | (((vx + 128) & 255) << 12) | (((vz + 128) & 255) << 20);
    r.color = int32_t(color);
    return r;
}

void store(HexBlob &out, const std::vector<HexRec> &boxes, const std::vector<HexRec> &objs, const std::vector<P3> &spawn, const std::vector<float> &yaw)
{
    int nb = int(boxes.size()), no = int(objs.size());
    if (nb > HEX_MAX_BOXES) { generator_overflow_raise(GEN_SLABS); nb = HEX_MAX_BOXES; }
    if (no > HEX_MAX_OBJS) { generator_overflow_raise(GEN_REWARDS); no = HEX_MAX_OBJS; }
    out.num_boxes = nb; out.num_objs = no;
    out.num_colliders = int(std::count_if(boxes.begin(), boxes.begin() + nb, [](const HexRec &r) { return (r.meta & 16) != 0; }));
    std::memcpy(out.boxes, boxes.data(), size_t(nb) * sizeof(HexRec));
    std::memcpy(out.objs, objs.data(), size_t(no) * sizeof(HexRec));
    for (size_t i = 0; i < spawn.size() && i < MAX_AGENTS; ++i) {
        out.spawn[i][0] = spawn[i].x; out.spawn[i][1] = spawn[i].y; out.spawn[i][2] = spawn[i].z;
        out.yaw[i] = yaw[i];
    }
}

}  // namespace

void generate_hex_explore_episode(std::mt19937 &rng, int num_agents, float base_episode_len, HexBlob &out)
{
    std::memset(&out, 0, offsetof(HexBlob, boxes));
    const int seed = rand_range(0, 1 << 30, rng);   // Env::reset, env.cpp:61-62
    rng.seed((unsigned long)seed);

    const int size = rand_range(2, 8, rng);
    const Honeycomb maze(size, uint32_t(seed));
    const MazeLook look = draw_maze_look(rng, 0.1f, 0.4f);
    const int goal = rand_range(0, maze.cells(), rng);
    const P3 target{float(maze.centre(goal).first) * look.scale, 0.0f, float(maze.centre(goal).second) * look.scale};

    // agentStartingPositions: the first cell of a shuffled list that is further than `size` cells from the reward object -- or the
    // furthest one met on the way -- with the agents on a unit circle around its centre
    std::vector<int> order(size_t(maze.cells()), 0);
    std::iota(order.begin(), order.end(), 0);
    std::shuffle(order.begin(), order.end(), rng);
    std::vector<P3> spawn;
    float furthest = 0;
    const float turn = float(2 * M_PI / num_agents);
    for (int cell : order) {
        const P3 at{float(maze.centre(cell).first) * look.scale, 0.1f, float(maze.centre(cell).second) * look.scale};
        const float dx = target.x - at.x, dy = target.y - at.y, dz = target.z - at.z;
        const float distance = std::sqrt((dx * dx + dy * dy) + dz * dz);
        if (distance > furthest) {
            spawn.clear();
            for (int i = 0; i < num_agents; ++i) spawn.push_back(P3{at.x + sinf(float(i) * turn), at.y + 0.0f, at.z + cosf(float(i) * turn)});
            furthest = distance;
        }
        if (distance > float(size) * look.scale) break;
    }
    if (spawn.empty()) spawn.assign(size_t(num_agents), P3{0, 1, 0});
    std::vector<float> yaw;
    for (int i = 0; i < num_agents; ++i) yaw.push_back(frand01(rng) * 3.14159274f * 2);   // scenario_default.hpp:87

    std::vector<HexRec> boxes, objs;
    build_maze_boxes(rng, maze, look, boxes);
    const float sc = 1.9f;   // the reward object: a VIOLET diamond above the goal cell
    objs.push_back(object_record(HEX_DIAMOND, 0xd468ee, P3{target.x + 0.0f, target.y + 1.2f, target.z + 0.0f},
                   P3{0.17f * sc, 0.35f * sc, 0.17f * sc}, true, true, target));

    store(out, boxes, objs, spawn, yaw);
    out.num_good = 0;
    out.target[0] = target.x; out.target[1] = target.z;
    out.episode_len = base_episode_len;
}

void generate_hex_memory_episode(std::mt19937 &rng, int num_agents, float base_episode_len, HexBlob &out)
{
    std::memset(&out, 0, offsetof(HexBlob, boxes));
    const int seed = rand_range(0, 1 << 30, rng);
    rng.seed((unsigned long)seed);

    const int size = rand_range(2, 8, rng);
    const Honeycomb maze(size, uint32_t(seed));
    const MazeLook look = draw_maze_look(rng, 0.1f, 0.95f);

    int middle = 0;   // "hacky way to find the cell closest to center"
    float best = 1e9f;
    for (int cell = 0; cell < maze.cells(); ++cell) {
        const double d = std::sqrt(maze.centre(cell).first * maze.centre(cell).first + maze.centre(cell).second * maze.centre(cell).second);
        if (d < best) { middle = cell; best = float(d); }
    }
    const P3 landmark{float(maze.centre(middle).first * look.scale), 1.0f, float(maze.centre(middle).second * look.scale)};

    std::vector<P3> places;   // one candidate spot per other cell, jittered
    for (int cell = 0; cell < maze.cells(); ++cell) {
        if (cell == middle) continue;
        const float jz = frand01(rng) - 0.5f;   // Vector3(frand - 0.5f, 0, frand - 0.5f): g++ evaluates call arguments right to left
        const float jx = frand01(rng) - 0.5f;
        const float x = float(maze.centre(cell).first) + jx, z = float(maze.centre(cell).second) + jz;
        places.push_back(P3{x * look.scale, 0.5f + 0.0f, z * look.scale});
    }
    std::shuffle(places.begin(), places.end(), rng);
    const float fraction = frand01(rng) * 0.25f + 0.2f;
    const long good_count = std::lround(std::ceil(fraction * places.size()));
    const long bad_count = long(places.size()) >= 2 * good_count ? good_count : 0;

    std::vector<P3> spawn;   // spawnAgents: a circle of radius 1.5 around the origin, everybody facing along their own angle; no draws
    std::vector<float> yaw;
    const float turn = float(2 * M_PI / num_agents);
    for (int i = 0; i < num_agents; ++i) {
        spawn.push_back(P3{1.5f * sinf(turn * float(i)), 1.5f * 0.3f, 1.5f * cosf(turn * float(i))});
        yaw.push_back(turn * i);
    }

    unsigned good_color = OBJECT_COLORS[rand_range(0, 14, rng)], bad_color = good_color;
    int good_shape = rand_range(0, 3, rng), bad_shape = good_shape;
    while (bad_color == good_color && bad_shape == good_shape) {
        bad_color = OBJECT_COLORS[rand_range(0, 14, rng)];
        bad_shape = rand_range(0, 3, rng);
    }
    std::vector<HexRec> boxes, objs;
    build_maze_boxes(rng, maze, look, boxes);

    auto scale_of = [](int shape) {
        return shape == HEX_SPHERE ? P3{0.75f, 0.75f, 0.75f} : shape == HEX_PILLAR ? P3{0.5f, 2.0f, 0.5f} : P3{0.17f * 2.2f, 0.45f * 2.2f, 0.17f * 2.2f};
    };
    auto shift_of = [](int shape) { return shape == HEX_SPHERE ? P3{0.5f, 0.1f, 0.5f} : shape == HEX_PILLAR ? P3{0.5f, 0.05f, 0.5f} : P3{0.5f, 0.6f, 0.5f}; };
    {   // the object in the middle cell shows what to collect; it cannot be collected itself
        const P3 sh = shift_of(good_shape);
        objs.push_back(object_record(good_shape, good_color, P3{landmark.x + sh.x, landmark.y + sh.y, landmark.z + sh.z},
                       scale_of(good_shape), true, false, landmark));
    }
    const float shrink = 0.6f;
    for (long i = 0; i < good_count + bad_count; ++i) {
        const bool good = i < good_count;
        const int shape = good ? good_shape : bad_shape;
        const P3 sh = shift_of(shape), sc = scale_of(shape), at = places[size_t(i)];
        objs.push_back(object_record(shape, good ? good_color : bad_color, P3{at.x + sh.x * shrink, at.y + sh.y * shrink, at.z + sh.z * shrink},
                                     P3{sc.x * shrink, sc.y * shrink, sc.z * shrink}, good, true, at));
    }
    store(out, boxes, objs, spawn, yaw);
    out.num_good = int(good_count);
    out.episode_len = base_episode_len + 3.0f * float(good_count);   // scenario_hex_memory.hpp:45-49
}

}  // namespace mv
