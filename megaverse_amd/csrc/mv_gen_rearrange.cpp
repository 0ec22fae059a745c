// megaverse_amd/csrc/mv_gen_rearrange.cpp -- host-side episode generator of the Rearrange scenario.
//
// Replaces RearrangeScenario::reset / generateArrangement / agentStartingPositions and the draws of
// arrangementDrawables (reference: src/libs/scenarios/src/scenario_rearrange.cpp:50-127,182-263), the room of
// RearrangePlatform (:11-34, platforms.hpp:167-190) with VoxelGridComponent::toBoundingBoxes
// (component_voxel_grid.hpp:108-187) and the spawn rotation draw of DefaultScenario::spawnAgents (scenario_default.hpp:87).
// Host-side for the same reason as the other generators (libstdc++ distributions and shuffle); the room is fixed
// (19 x height x 14), so its merged slabs are written down analytically in canonical order.
#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <random>
#include <vector>

#include "mv_gen.h"

namespace mv {

namespace {

using Rng = std::mt19937;
inline int rand_range(int lo, int hi, Rng &rng) { return std::uniform_int_distribution<>{lo, hi - 1}(rng); }   // util.hpp:30-33
inline float frand01(Rng &rng) { return std::uniform_real_distribution<float>{0, 1}(rng); }                    // util.hpp:46-49

constexpr int kLength = 19, kWidth = 14;                  // RearrangePlatform::init
constexpr int kLeft[3] = {5, 2, 5}, kRight[3] = {13, 2, 5};   // scenario_rearrange.hpp:130-131
const unsigned kObjectColors[14] = {0xffdd3c, 0x3bb372, 0x50c878, 0x2eb5d0, 0xadd8e6, 0x3a7fa6, 0xffb400,
                                    0xb3b3b3, 0x555555, 0xffffff, 0xff0000, 0xffa770, 0xd468ee, 0xffe6e6};   // env/const.hpp:96-111

struct Offset { int x, y, z; };
inline bool same(const Offset &a, const Offset &b) { return a.x == b.x && a.y == b.y && a.z == b.z; }

ArrangementItem draw_item(Rng &rng, Offset at)
{   // shape first, colour second (ArrangementItem::random)
    static const int kShapes[4] = {SHAPE_CYLINDER, SHAPE_CAPSULE, SHAPE_BOX, SHAPE_SPHERE};
    ArrangementItem it{};
    it.shape = kShapes[rand_range(0, 4, rng)];
    it.color = int(kObjectColors[rand_range(0, 14, rng)]);
    it.off[0] = at.x; it.off[1] = at.y; it.off[2] = at.z;
    return it;
}

void slab(RearrangeBlob &out, int x0, int y0, int z0, int x1, int y1, int z1, int type)
{
    if (x0 >= x1 || y0 >= y1 || z0 >= z1) return;
    LayoutBox &b = out.boxes[out.num_boxes++];
    b.min[0] = x0; b.min[1] = y0; b.min[2] = z0; b.max[0] = x1; b.max[1] = y1; b.max[2] = z1;
    b.type = type; b.slot = 0;
}

}  // namespace

void generate_rearrange_episode(std::mt19937 &rng, int num_agents, float base_episode_len, RearrangeBlob &out)
{
    std::memset(&out, 0, sizeof out);

    // Env::reset: re-seed from the env's own stream (env.cpp:61-62)
    const int episode_seed = rand_range(0, 1 << 30, rng);
    rng.seed((unsigned long)episode_seed);

    const int height = rand_range(4, 7, rng);
    const bool draw_walls = rand_range(0, 2, rng) != 0;   // randomBool: the last argument of vg.addPlatform (:61)
    out.dim[0] = kLength; out.dim[1] = height; out.dim[2] = kWidth;
    out.draw_walls = draw_walls ? 1 : 0;

    // ---- the room's slabs in canonical merge order (class by type; seeds in (y, z, x) order; grown along x, then z, then y).
    // Floor and walls have the same colour.  Walls drawn: ONE class (solid + opaque), so the first seed swallows the whole
    // floor layer and the walls are what is left above it.  Walls not drawn: the solid-only wall class comes first
    // (type 1 < 3) and owns its floor cells (later grid.set calls overwrite the floor's).
    const int L = kLength, W = kWidth, H = height;
    if (draw_walls) {
        const int t = VX_SOLID | VX_OPAQUE;
        slab(out, 0, 0, 0, L, 1, W, t);
        slab(out, 0, 1, 0, L, H, 1, t);
        slab(out, 0, 1, 1, 1, H, W, t);
        slab(out, L - 1, 1, 1, L, H, W, t);
        slab(out, 1, 1, W - 1, L - 1, H, W, t);
    } else {
        const int tw = VX_SOLID, tf = VX_SOLID | VX_OPAQUE;
        slab(out, 0, 0, 0, L, H, 1, tw);
        slab(out, 0, 0, 1, 1, H, W, tw);
        slab(out, L - 1, 0, 1, L, H, W, tw);
        slab(out, 1, 0, W - 1, L - 1, H, W, tw);
        slab(out, 1, 0, 1, L - 1, 1, W - 1, tf);
    }

    // ---- generateArrangement: breadth-first growth; the direction list keeps being shuffled in place
    const int wanted = rand_range(2, 8, rng);
    std::vector<ArrangementItem> items;
    auto taken = [&](const Offset &o) {
        for (const auto &it : items) if (it.off[0] == o.x && it.off[1] == o.y && it.off[2] == o.z) return true;
        return false;
    };
    items.push_back(draw_item(rng, Offset{0, 0, 0}));
    std::vector<Offset> dirs{{-1, 0, 0}, {1, 0, 0}, {0, 1, 0}, {0, 0, -1}, {0, 0, 1}};
    for (size_t head = 0; head < items.size() && int(items.size()) < wanted; ++head) {
        const Offset from{items[head].off[0], items[head].off[1], items[head].off[2]};
        int limit = rand_range(1, int(dirs.size()) + 1, rng);
        limit = rand_range(1, limit + 1, rng);
        std::shuffle(dirs.begin(), dirs.end(), rng);
        int grown = 0;
        for (const Offset d : dirs) {
            const Offset to{from.x + d.x, from.y + d.y, from.z + d.z};
            if (to.y >= 2 || std::abs(to.x) >= 2 || std::abs(to.z) >= 2) continue;
            if (taken(to)) continue;
            if (to.y != 0 && !taken(Offset{to.x, to.y - 1, to.z})) continue;
            items.push_back(draw_item(rng, to));
            if (++grown >= limit) break;
            if (int(items.size()) >= wanted) break;
        }
    }
    out.num_items = int(items.size());
    for (int i = 0; i < out.num_items; ++i) out.items[i] = items[i];

    // ---- agentStartingPositions: up to 20 tries per agent to land outside both work areas, else (0, 0, 0)
    for (int i = 0; i < num_agents; ++i)
        for (int attempt = 0; attempt < 20; ++attempt) {
            const int x = rand_range(2, kLength - 1, rng), z = rand_range(2, kWidth - 1, rng);
            if (std::abs(x - kLeft[0]) < 2 && std::abs(z - kLeft[2]) < 2) continue;
            if (std::abs(x - kRight[0]) < 2 && std::abs(z - kRight[2]) < 2) continue;
            out.spawn[i][0] = x; out.spawn[i][1] = 2; out.spawn[i][2] = z;
            break;
        }
    for (int i = 0; i < num_agents; ++i) out.yaw_frand[i] = frand01(rng);

    // ---- the movable copy on the right pedestal: the first `unmoved` items in place, the rest on random free floor cells
    const int unmoved = rand_range(0, out.num_items, rng);
    std::vector<Offset> occupied;
    for (const auto &it : items) occupied.push_back(Offset{it.off[0], it.off[1], it.off[2]});
    auto is_occupied = [&](const Offset &o) { return std::any_of(occupied.begin(), occupied.end(), [&](const Offset &p) { return same(p, o); }); };
    for (int i = 0; i < out.num_items; ++i) {
        Offset at{items[i].off[0], items[i].off[1], items[i].off[2]};
        if (i >= unmoved) {
            while (is_occupied(at)) at = Offset{rand_range(-2, 3, rng), 0, rand_range(-2, 3, rng)};
            occupied.push_back(at);
        }
        out.objects[i] = MovableObject{int8_t(at.x + kRight[0]), int8_t(at.y + kRight[1]), int8_t(at.z + kRight[2]), 0};
    }
    // countMatchingObjects at the start of the episode (maxMatchingObjects, :272)
    for (int i = 0; i < out.num_items; ++i) {
        const Offset at{out.objects[i].x - kRight[0], out.objects[i].y - kRight[1], out.objects[i].z - kRight[2]};
        for (int k = 0; k < out.num_items; ++k)
            if (items[k].shape == items[i].shape && items[k].color == items[i].color && items[k].off[0] == at.x
                && items[k].off[1] == at.y && items[k].off[2] == at.z) {
                ++out.max_matching;
                break;
            }
    }
    out.episode_len = base_episode_len;
}

}  // namespace mv
