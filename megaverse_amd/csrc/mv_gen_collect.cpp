// megaverse_amd/csrc/mv_gen_collect.cpp -- host-side landscape generator of the Collect scenario.
//
// Replaces CollectScenario::reset / createLandscape and the reward draws of addEpisodeDrawables
//   (reference: src/libs/scenarios/src/scenario_collect.cpp:20-161,190-214), siv::PerlinNoise as vendored in
//   src/libs/util/include/util/perlin_noise.hpp (reseed :118-126, noise3D :171-197, octaves :244-259,315-318),
//   VoxelGridComponent::toBoundingBoxes (component_voxel_grid.hpp:108-187) and the spawn rotation draw of
//   DefaultScenario::spawnAgents (scenario_default.hpp:87).
//
// Host-side for the same reason as mv_gen_obstacles.cpp: the draws go through libstdc++'s mt19937 /
// minstd_rand0 / shuffle / sort templates, the noise is double precision, and the result is swapped in by the
// reset kernel from a resident blob.  The landscape is a heightfield, so it is kept as one: the kernels answer
// voxel questions from `heightmap`, and collide / draw the merged slabs.
#include <algorithm>
#include <cmath>
#include <cstring>
#include <random>
#include <vector>

#include "mv_gen.h"

namespace mv {

namespace {

using Rng = std::mt19937;
inline int rand_range(int lo, int hi, Rng &rng) { return std::uniform_int_distribution<>{lo, hi - 1}(rng); }   // util.hpp:30-33
inline float frand01(Rng &rng) { return std::uniform_real_distribution<float>{0, 1}(rng); }                    // util.hpp:46-49

// improved Perlin noise over a seed-shuffled permutation, value type double
class Noise {
public:
    explicit Noise(uint32_t seed)
    {
        for (int i = 0; i < 256; ++i) perm_[i] = uint8_t(i);
        std::shuffle(perm_, perm_ + 256, std::default_random_engine(seed));
        std::memcpy(perm_ + 256, perm_, 256);
    }

    // sum of `octaves` layers, each twice the frequency and half the weight, mapped to [0, 1]
    double layered01(double x, double y, int octaves) const
    {
        double sum = 0, weight = 1;
        for (int o = 0; o < octaves; ++o) {
            sum += at_z0(x, y) * weight;
            x *= 2; y *= 2; weight /= 2;
        }
        return std::clamp<double>(sum * 0.5 + 0.5, 0, 1);
    }

    // One axis of a grid of samples: for sample i and octave o the coordinate i / step * 2^o (the doublings are exact, so the table holds the very values
    // layered01 walks through) split into lattice cell, offset and smoothed offset -- a landscape's nx x nz samples share nx + nz of these per octave instead
    // of computing them nx x nz times.
    struct Axis { int cell; double t, s; };
    static void axis_table(int n, double step, int octaves, std::vector<Axis> &out)
    {
        out.resize(size_t(n) * octaves);
        for (int i = 0; i < n; ++i) {
            double c = i / step;
            for (int o = 0; o < octaves; ++o, c *= 2) {
                const double f = std::floor(c);
                Axis &a = out[size_t(i) * octaves + o];
                a.cell = int(f) & 255; a.t = c - f; a.s = smooth(a.t);
            }
        }
    }
    // layered01(x / step_x, z / step_z, octaves) from the two axes' tables: the same operations on the same values, in the same order
    double layered01(const Axis *ax, const Axis *az, int octaves) const
    {
        double sum = 0, weight = 1;
        for (int o = 0; o < octaves; ++o) {
            const int ix = ax[o].cell, iy = az[o].cell;
            const double x = ax[o].t, y = az[o].t, u = ax[o].s, v = az[o].s;
            const int a = perm_[ix] + iy, b = perm_[ix + 1] + iy;
            const int aa = perm_[a], ab = perm_[a + 1], ba = perm_[b], bb = perm_[b + 1];
            const double n = blend(v, blend(u, corner0(perm_[aa], x, y), corner0(perm_[ba], x - 1, y)),
                                      blend(u, corner0(perm_[ab], x, y - 1), corner0(perm_[bb], x - 1, y - 1)));
            sum += n * weight;
            weight /= 2;
        }
        return std::clamp<double>(sum * 0.5 + 0.5, 0, 1);
    }

private:
    uint8_t perm_[512];

    static double smooth(double t) { return t * t * t * (t * (t * 6 - 15) + 10); }
    static double blend(double t, double a, double b) { return a + t * (b - a); }
    // corner(h, x, y, 0) without the branches (h is as good as random: every one of them mispredicts half of the time): which of x, y, 0 the two terms are, and
    // their signs, from a table -- the same two values, negated or not, and their sum
    static double corner0(uint8_t h, double x, double y)
    {
        static const uint8_t kU[16] = {0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 1, 1, 1, 1};
        static const uint8_t kV[16] = {1, 1, 1, 1, 2, 2, 2, 2, 2, 2, 2, 2, 0, 2, 0, 2};
        h &= 15;
        const double val[3] = {x, y, 0.0};
        uint64_t ub, vb;
        std::memcpy(&ub, &val[kU[h]], 8); std::memcpy(&vb, &val[kV[h]], 8);
        ub ^= uint64_t(h & 1) << 63; vb ^= uint64_t(h & 2) << 62;
        double u, v;
        std::memcpy(&u, &ub, 8); std::memcpy(&v, &vb, 8);
        return u + v;
    }
    static double corner(uint8_t h, double x, double y, double z)
    {
        h &= 15;
        const double u = h < 8 ? x : y, v = h < 4 ? y : (h == 12 || h == 14) ? x : z;
        return ((h & 1) ? -u : u) + ((h & 2) ? -v : v);
    }
    // at(x, y, 0): the plane the landscape samples (perlin_noise.hpp:315-318 calls the 3-D noise with z = 0).  There w = smooth(0) = 0 and the result
    // is near + 0 * (far - near) = near: the four corners of the far layer are not evaluated (they were half of this generator's arithmetic).  The only
    // thing lost is the sign of a zero result, which the sum over the octaves does not see.
    double at_z0(double x, double y) const
    {
        const double fx = std::floor(x), fy = std::floor(y);
        const int ix = int(fx) & 255, iy = int(fy) & 255;
        x -= fx; y -= fy;
        const double u = smooth(x), v = smooth(y);
        const int a = perm_[ix] + iy, b = perm_[ix + 1] + iy;
        const int aa = perm_[a], ab = perm_[a + 1], ba = perm_[b], bb = perm_[b + 1];
        return blend(v, blend(u, corner(perm_[aa], x, y, 0.0), corner(perm_[ba], x - 1, y, 0.0)),
                        blend(u, corner(perm_[ab], x, y - 1, 0.0), corner(perm_[bb], x - 1, y - 1, 0.0)));
    }
    double at(double x, double y, double z) const
    {
        const double fx = std::floor(x), fy = std::floor(y), fz = std::floor(z);
        const int ix = int(fx) & 255, iy = int(fy) & 255, iz = int(fz) & 255;
        x -= fx; y -= fy; z -= fz;
        const double u = smooth(x), v = smooth(y), w = smooth(z);
        const int a = perm_[ix] + iy, b = perm_[ix + 1] + iy;
        const int aa = perm_[a] + iz, ab = perm_[a + 1] + iz, ba = perm_[b] + iz, bb = perm_[b + 1] + iz;
        const double near = blend(v, blend(u, corner(perm_[aa], x, y, z), corner(perm_[ba], x - 1, y, z)),
                                     blend(u, corner(perm_[ab], x, y - 1, z), corner(perm_[bb], x - 1, y - 1, z)));
        const double far = blend(v, blend(u, corner(perm_[aa + 1], x, y, z - 1), corner(perm_[ba + 1], x - 1, y, z - 1)),
                                    blend(u, corner(perm_[ab + 1], x, y - 1, z - 1), corner(perm_[bb + 1], x - 1, y - 1, z - 1)));
        return blend(w, near, far);
    }
};

struct Cell { int x, y, z; };

}  // namespace

void generate_collect_episode(std::mt19937 &rng, int num_agents, float base_episode_len, CollectBlob &out)
{
    out.seq = 0; out.num_boxes = out.num_objects = out.num_rewards = out.num_positive = 0; out.pad = 0;
    std::memset(out.spawn, 0, sizeof out.spawn);

    // Env::reset: re-seed from the env's own stream (env.cpp:61-62)
    const int episode_seed = rand_range(0, 1 << 30, rng);
    rng.seed((unsigned long)episode_seed);

    static const unsigned kLandscape[7] = {0xffffff, 0xccffcc, 0xe6ecff, 0xd9d9d9, 0xffebcc, 0xb3b3b3, 0x555555};   // :39-47, env/const.hpp
    static const unsigned kFloor[3] = {0xb3b3b3, 0x555555, 0x555555};                                              // :48-52
    const unsigned land_color = kLandscape[rand_range(0, 7, rng)];
    const unsigned floor_color = kFloor[rand_range(0, 3, rng)];
    const int nz = rand_range(8, HM_DIM, rng);   // "width"
    const int nx = rand_range(8, HM_DIM, rng);   // "length"
    const double frequency = double(rand_range(1, 100, rng)) / 10.0;
    const int octaves = rand_range(1, 10, rng);
    const Noise noise(uint32_t(rand_range(0, 1000000000, rng)));
    const double step_x = HM_DIM / frequency, step_z = HM_DIM / frequency;
    const int intensity = rand_range(5, 18, rng);
    const float ground = frand01(rng) * 0.5f + 0.2f;

    // ---- heightfield: floor everywhere, hills inside the one-cell rim
    std::memset(out.heightmap, 0xff, sizeof out.heightmap);
    auto height = [&](int x, int z) -> int8_t & { return out.heightmap[x * HM_DIM + z]; };
    int top = 0;
    std::vector<Noise::Axis> axis_x, axis_z;
    Noise::axis_table(nx, step_x, octaves, axis_x);
    Noise::axis_table(nz, step_z, octaves, axis_z);
    for (int x = 0; x < nx; ++x)
        for (int z = 0; z < nz; ++z) {
            int h = 0;
            if (x >= 1 && x < nx - 1 && z >= 1 && z < nz - 1) {
                const double elevation = intensity * (noise.layered01(&axis_x[size_t(x) * octaves], &axis_z[size_t(z) * octaves], octaves) - ground);
                if (elevation >= 1) h = int(std::lround(elevation));
            }
            height(x, z) = int8_t(h);
            top = std::max(top, h);
        }

    // ---- merged slabs.  Two voxel classes, both solid + opaque: the floor layer (y == 0) and the hills; the
    // reference groups by (type, colour) in std::map order, so the lower colour value is emitted first and
    // equal colours form a single class.  Per class: seeds in (y, z, x) order, grown along x, then z, then y.
    const unsigned color_lo = std::min(land_color, floor_color), color_hi = std::max(land_color, floor_color);
    out.layout_color = int(color_lo); out.wall_color = int(color_hi);
    {
        const int ny = top + 1;
        std::vector<uint8_t> taken(size_t(nx) * ny * nz, 0);
        auto id = [&](int x, int y, int z) { return (size_t(y) * nz + z) * nx + x; };
        for (int slot = 0; slot < 2; ++slot) {
            const unsigned want = slot == 0 ? color_lo : color_hi;
            if (slot == 1 && color_hi == color_lo) break;
            auto free_cell = [&](int x, int y, int z) {
                if (x < 0 || x >= nx || z < 0 || z >= nz || y < 0 || y > height(x, z)) return false;
                return (y == 0 ? floor_color : land_color) == want && !taken[id(x, y, z)];
            };
            for (int y = 0; y < ny; ++y)
                for (int z = 0; z < nz; ++z)
                    for (int x = 0; x < nx; ++x) {
                        if (!free_cell(x, y, z)) continue;
                        int x_end = x + 1, z_end = z + 1, y_end = y + 1;
                        while (free_cell(x_end, y, z)) ++x_end;
                        auto row_free = [&](int yy, int zz) { for (int xx = x; xx < x_end; ++xx) if (!free_cell(xx, yy, zz)) return false; return true; };
                        while (row_free(y, z_end)) ++z_end;
                        auto layer_free = [&](int yy) { for (int zz = z; zz < z_end; ++zz) if (!row_free(yy, zz)) return false; return true; };
                        while (layer_free(y_end)) ++y_end;
                        for (int yy = y; yy < y_end; ++yy)
                            for (int zz = z; zz < z_end; ++zz)
                                for (int xx = x; xx < x_end; ++xx) taken[id(xx, yy, zz)] = 1;
                        if (out.num_boxes >= COLLECT_MAX_BOXES) generator_overflow_raise(GEN_SLABS);
                        if (out.num_boxes < COLLECT_MAX_BOXES) {
                            LayoutBox &b = out.boxes[out.num_boxes++];
                            b.min[0] = x; b.min[1] = y; b.min[2] = z; b.max[0] = x_end; b.max[1] = y_end; b.max[2] = z_end;
                            b.type = VX_SOLID | VX_OPAQUE; b.slot = slot;
                        }
                    }
        }
    }
    out.dim[0] = nx; out.dim[1] = top + 1; out.dim[2] = nz;

    // ---- one shuffled list of free cells feeds agents, diamonds and movable boxes (:105-160)
    auto free_y = [&](int x, int z) { return height(x, z) >= 1 ? height(x, z) + 1 : 1; };
    std::vector<Cell> cells;
    for (int x = 1; x < nx - 1; ++x)
        for (int z = 1; z < nz - 1; ++z) cells.push_back(Cell{x, free_y(x, z), z});
    std::shuffle(cells.begin(), cells.end(), rng);
    size_t next = 0;
    for (int i = 0; i < num_agents; ++i, ++next) { out.spawn[i][0] = cells[next].x; out.spawn[i][1] = cells[next].y; out.spawn[i][2] = cells[next].z; }

    int num_rewards = rand_range(1, int(std::lround(0.05 * nz * nx)) + 2, rng);
    num_rewards = std::min(num_rewards, int(cells.size() - next));
    const int scattered = std::max(num_rewards / 2, 1);
    std::vector<Cell> reward_cells(cells.begin() + next, cells.begin() + next + scattered);
    next += scattered;
    // the other half goes to the highest ground: an (unstable) std::sort by descending free height
    std::sort(cells.begin() + next, cells.end(), [&](const Cell &a, const Cell &b) {
        const int ha = free_y(a.x, a.z), hb = free_y(b.x, b.z);
        return ha != hb ? ha > hb : false;
    });
    reward_cells.insert(reward_cells.end(), cells.begin() + next, cells.begin() + next + (num_rewards - scattered));
    next += num_rewards - scattered;

    std::shuffle(cells.begin() + next, cells.end(), rng);
    const int objects_min = std::max(3, int(nx * nz * 0.04));
    const int objects_max = std::min(objects_min + 1, int(std::lround(0.07 * nz * nx)) + 2);
    const int num_objects = std::min(rand_range(objects_min, objects_max, rng), int(cells.size() - next));
    if (num_objects > MAX_OBJECTS) generator_overflow_raise(GEN_OBJECTS);
    if (next + num_objects < cells.size())   // always true for <= 8 agents (see oracle note on :153-156)
        for (int i = 0; i < num_objects && i < MAX_OBJECTS; ++i, ++next)
            out.objects[out.num_objects++] = MovableObject{int8_t(cells[next].x), int8_t(cells[next].y), int8_t(cells[next].z), 0};

    out.episode_len = base_episode_len + 2.0f * float(reward_cells.size());   // scenario_collect.hpp:55-59

    for (int i = 0; i < num_agents; ++i) out.yaw_frand[i] = frand01(rng);
    for (size_t i = 0; i < reward_cells.size(); ++i) {   // 70 % of the diamonds are worth +1 (:198)
        const bool good = frand01(rng) > 0.3f;
        if (good) ++out.num_positive;
        if (int(i) >= COLLECT_MAX_REWARDS) generator_overflow_raise(GEN_REWARDS);
        if (int(i) < COLLECT_MAX_REWARDS)
            out.rewards[out.num_rewards++] = MovableObject{int8_t(reward_cells[i].x), int8_t(reward_cells[i].y),
                                                                  int8_t(reward_cells[i].z), int8_t(good ? 1 : 2)};
    }
}

}  // namespace mv
