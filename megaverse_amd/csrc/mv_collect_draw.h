// megaverse_amd/csrc/mv_collect_draw.h -- the Collect scenario's episode generator as code that runs on the device (and, for the CPU test, on the host).
//
// Same episodes as mv_gen_collect.cpp, byte for byte (tests/test_collect_draw.py, tests/test_collect_draw_gpu.py): CollectScenario::reset /
// createLandscape / addEpisodeDrawables (reference: src/libs/scenarios/src/scenario_collect.cpp:20-161,190-214), siv::PerlinNoise
// (src/libs/util/include/util/perlin_noise.hpp:118-126,171-197,244-259,315-318), VoxelGridComponent::toBoundingBoxes
// (component_voxel_grid.hpp:108-187), DefaultScenario::spawnAgents (scenario_default.hpp:87), Env::reset's seed draw (env.cpp:61-62).
//
// mv_gen_collect.cpp reaches the reference's draws through libstdc++ (11.4 in this image): std::mt19937, std::minstd_rand0, uniform_int_distribution,
// generate_canonical, std::shuffle and -- for the diamonds on the highest ground -- an UNSTABLE std::sort.  Those are third-party; their published algorithms
// are restated here so that one GPU lane can run them:
//   * MT19937 (Matsumoto & Nishimura 1998), sequential form;
//   * minstd_rand0: x <- 16807 x mod (2^31 - 1) (Lewis, Goodman, Miller 1969), seed 0 -> 1;
//   * uniform_int_distribution: Lemire's nearly-divisionless method on the 32-bit generator (bits/uniform_int_dist.h, _S_nd), the two-division
//     rejection form on any other generator range (same file, "fallback case");
//   * std::shuffle: one draw for two swap positions while range^2 fits the generator's range (bits/stl_algo.h);
//   * std::sort: introsort -- median-of-three quicksort down to 16 elements under a depth limit of 2 lg n, heapsort beyond it, one final insertion
//     sort (bits/stl_algo.h, bits/stl_heap.h).  The order of equal keys is part of the episode (which cells the diamonds take), so the restatement
//     follows the element moves exactly, not just the result's sortedness.
// The noise is double precision; the library is built with -ffp-contract=off, so host and device round every operation alike.
//
// Execution model on the device: one wavefront per episode.  Everything that consumes a random stream is serial by nature and runs on LANE 0 ALONE
// (plain loads and stores to LDS; the other lanes wait at a wave barrier); the heightfield -- nx x nz samples of up to nine octaves of noise -- is
// spread over the 64 lanes, and so is the slab merge (mv_collect_draw.hip: draw_slabs_wave; the serial form below is the host's and the reference the tests
// hold it against).  An episode costs 0.7 ms of one wavefront on average, off the step path (mv_feeder.cpp: device mode).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "mv_types.h"

namespace mv {
namespace cdraw {

#define MV_HD __host__ __device__ inline

enum : int { MAX_CELLS = (HM_DIM - 2) * (HM_DIM - 2), TAKEN_WORDS = (HM_DIM * HM_DIM * 20 + 31) / 32, SORT_STACK = 128 };
enum : int { FLAG_SLABS = 1, FLAG_OBJECTS = 4, FLAG_REWARDS = 8 };   // = GEN_SLABS, GEN_OBJECTS, GEN_REWARDS (mv_gen.h)

// scratch of one episode: LDS on the device, plain memory on the host
struct Scratch {
    uint32_t mt[624];
    uint32_t cells[MAX_CELLS];      // x | y << 8 | z << 16
    uint32_t taken[TAKEN_WORDS];    // the slab merge's "cell already covered" bits
    uint32_t stack[SORT_STACK];     // introsort's pending ranges: first | last << 11 | depth << 22
    uint8_t perm[512];
    int8_t hm[HM_BYTES];
};

// where an env's generator stands (TowerGen's twin): the seed the NEXT episode's Env::reset re-seeds with, and the sequence number it will get
struct alignas(16) GenState {
    uint32_t seed;
    int32_t seed_is_env_seed;   // 1: `seed` is the Env::seed() value, the reset draws its seed from it first
    int32_t next_seq;
    int32_t flags;              // FLAG_* raised by this env's generator since the host last looked
};

struct Params {
    int nx, nz, octaves, intensity;
    double step;
    float ground;
    unsigned land_color, floor_color;
};

// ---- MT19937, sequential -----------------------------------------------------------------------------------------------------------------
struct Mt {
    uint32_t *s;
    int idx;
};
MV_HD void mt_seed1(Mt &g, uint32_t seed)
{
    uint32_t x = seed;
    g.s[0] = x;
    for (int i = 1; i < 624; ++i) {
        x = 1812433253u * (x ^ (x >> 30)) + (uint32_t)i;
        g.s[i] = x;
    }
    g.idx = 624;
}
MV_HD uint32_t mt_next1(Mt &g)
{
    if (g.idx >= 624) {
        uint32_t *s = g.s;
        for (int k = 0; k < 624; ++k) {
            const uint32_t y = (s[k] & 0x80000000u) | (s[k == 623 ? 0 : k + 1] & 0x7fffffffu);
            s[k] = s[k < 227 ? k + 397 : k - 227] ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
        }
        g.idx = 0;
    }
    uint32_t y = g.s[g.idx++];
    y ^= y >> 11;
    y ^= (y << 7) & 0x9d2c5680u;
    y ^= (y << 15) & 0xefc60000u;
    y ^= y >> 18;
    return y;
}
// unbiased integer in [0, range), range >= 1 (Lemire)
MV_HD uint32_t mt_below1(Mt &g, uint32_t range)
{
    uint64_t product = (uint64_t)mt_next1(g) * (uint64_t)range;
    uint32_t low = (uint32_t)product;
    if (low < range) {
        const uint32_t threshold = (0u - range) % range;
        while (low < threshold) {
            product = (uint64_t)mt_next1(g) * (uint64_t)range;
            low = (uint32_t)product;
        }
    }
    return (uint32_t)(product >> 32);
}
MV_HD int rand_range1(Mt &g, int lo, int hi) { return lo + (int)mt_below1(g, (uint32_t)(hi - lo)); }   // util.hpp:30-33
MV_HD float frand1(Mt &g)                                                                                 // util.hpp:46-49
{
    float ret = (float)mt_next1(g) / 4294967296.0f;
    if (ret >= 1.0f) ret = 0.99999994f;
    return ret;
}

// ---- minstd_rand0 through uniform_int_distribution's two-division form ----------------------------------------------------------------------
struct Minstd {
    uint64_t x;
};
MV_HD uint64_t minstd_next(Minstd &g) { g.x = (g.x * 16807ull) % 2147483647ull; return g.x; }
MV_HD uint64_t minstd_below(Minstd &g, uint64_t range)   // integer in [0, range)
{
    const uint64_t urngrange = 2147483645ull, scaling = urngrange / range, past = range * scaling;
    uint64_t ret;
    do ret = minstd_next(g) - 1ull; while (ret >= past);
    return ret / scaling;
}

// std::shuffle's walk, two swap positions per draw; `below(range)` draws, `swap(i, j)` exchanges
template <class Below, class Swap>
MV_HD void shuffle_walk(int n, Below below, Swap swap)
{
    if (n <= 0) return;
    int i = 1;
    if ((n % 2) == 0) {
        const int j = (int)below(2u);
        swap(i, j);
        ++i;
    }
    while (i < n) {
        const uint32_t r = (uint32_t)i + 1u;
        const uint32_t x = (uint32_t)below(r * (r + 1u));
        swap(i, (int)(x / (r + 1u)));
        ++i;
        swap(i, (int)(x % (r + 1u)));
        ++i;
    }
}

// ---- std::sort of cells by DESCENDING y -------------------------------------------------------------------------------------------------
MV_HD bool before(uint32_t a, uint32_t b) { return ((a >> 8) & 255u) > ((b >> 8) & 255u); }   // the comparator: higher free cell first, equal: false

MV_HD void sift(uint32_t *a, int first, int hole, int len, uint32_t value)   // __adjust_heap + __push_heap
{
    const int top = hole;
    int child = hole;
    while (child < (len - 1) / 2) {
        child = 2 * (child + 1);
        if (before(a[first + child], a[first + child - 1])) --child;
        a[first + hole] = a[first + child];
        hole = child;
    }
    if ((len & 1) == 0 && child == (len - 2) / 2) {
        child = 2 * (child + 1);
        a[first + hole] = a[first + child - 1];
        hole = child - 1;
    }
    int parent = (hole - 1) / 2;
    while (hole > top && before(a[first + parent], value)) {
        a[first + hole] = a[first + parent];
        hole = parent;
        parent = (hole - 1) / 2;
    }
    a[first + hole] = value;
}
MV_HD void heap_sort(uint32_t *a, int first, int last)   // __partial_sort(first, last, last): make_heap, then sort_heap
{
    const int len = last - first;
    if (len >= 2)
        for (int parent = (len - 2) / 2;; --parent) {
            sift(a, first, parent, len, a[first + parent]);
            if (parent == 0) break;
        }
    for (int end = last; end - first > 1;) {
        --end;
        const uint32_t value = a[end];
        a[end] = a[first];
        sift(a, first, 0, end - first, value);
    }
}
MV_HD void linear_insert(uint32_t *a, int last)   // __unguarded_linear_insert
{
    const uint32_t val = a[last];
    int next = last - 1;
    while (before(val, a[next])) {
        a[last] = a[next];
        last = next;
        --next;
    }
    a[last] = val;
}
MV_HD void insertion_sort(uint32_t *a, int first, int last)
{
    if (first == last) return;
    for (int i = first + 1; i != last; ++i) {
        if (before(a[i], a[first])) {
            const uint32_t val = a[i];
            for (int k = i; k > first; --k) a[k] = a[k - 1];
            a[first] = val;
        } else linear_insert(a, i);
    }
}
MV_HD void swap_cells(uint32_t *a, int i, int j) { const uint32_t t = a[i]; a[i] = a[j]; a[j] = t; }
MV_HD int partition_pivot(uint32_t *a, int first, int last)   // __unguarded_partition_pivot
{
    const int mid = first + (last - first) / 2, ia = first + 1, ib = mid, ic = last - 1;
    if (before(a[ia], a[ib])) {
        if (before(a[ib], a[ic])) swap_cells(a, first, ib);
        else if (before(a[ia], a[ic])) swap_cells(a, first, ic);
        else swap_cells(a, first, ia);
    } else if (before(a[ia], a[ic])) swap_cells(a, first, ia);
    else if (before(a[ib], a[ic])) swap_cells(a, first, ic);
    else swap_cells(a, first, ib);
    int lo = first + 1, hi = last;
    while (true) {
        while (before(a[lo], a[first])) ++lo;
        --hi;
        while (before(a[first], a[hi])) --hi;
        if (!(lo < hi)) return lo;
        swap_cells(a, lo, hi);
        ++lo;
    }
}
// a[0 .. n): the introsort loop with its recursion as a stack of pending ranges (they are disjoint: the order they are finished in changes nothing)
MV_HD void sort_cells(uint32_t *a, int n, uint32_t *stack)
{
    if (n <= 0) return;
    int lg = 0;
    while ((n >> (lg + 1)) != 0) ++lg;
    int sp = 0;
    stack[sp++] = 0u | ((uint32_t)n << 11) | ((uint32_t)(2 * lg) << 22);
    while (sp > 0) {
        const uint32_t f = stack[--sp];
        const int first = (int)(f & 2047u);
        int last = (int)((f >> 11) & 2047u), depth = (int)(f >> 22);
        while (last - first > 16) {
            if (depth == 0) { heap_sort(a, first, last); break; }
            --depth;
            const int cut = partition_pivot(a, first, last);
            if (sp < SORT_STACK) stack[sp++] = (uint32_t)cut | ((uint32_t)last << 11) | ((uint32_t)depth << 22);
            last = cut;
        }
    }
    if (n > 16) {
        insertion_sort(a, 0, 16);
        for (int i = 16; i != n; ++i) linear_insert(a, i);
    } else insertion_sort(a, 0, n);
}

// ---- the noise -------------------------------------------------------------------------------------------------------------------------------
MV_HD double smooth(double t) { return t * t * t * (t * (t * 6 - 15) + 10); }
MV_HD double blend(double t, double a, double b) { return a + t * (b - a); }
MV_HD double corner0(uint8_t h, double x, double y)   // grad(h, x, y, 0): two of x, y, 0, each negated or not, and their sum
{
    h &= 15;
    double u = h < 8 ? x : y, v = h < 4 ? y : (h == 12 || h == 14) ? x : 0.0;
    if (h & 1) u = -u;
    if (h & 2) v = -v;
    return u + v;
}
// intensity * (octave noise at (x / step, z / step), mapped to [0, 1], - ground) -> the column's height (0: floor only)
MV_HD int column_height(const Params &p, const uint8_t *perm, int x, int z)
{
    if (!(x >= 1 && x < p.nx - 1 && z >= 1 && z < p.nz - 1)) return 0;
    double cx = x / p.step, cz = z / p.step, sum = 0, weight = 1;
    for (int o = 0; o < p.octaves; ++o) {
        const double fx = floor(cx), fz = floor(cz);
        const int ix = (int)fx & 255, iy = (int)fz & 255;
        const double tx = cx - fx, ty = cz - fz, u = smooth(tx), v = smooth(ty);
        const int a = perm[ix] + iy, b = perm[ix + 1] + iy;
        const int aa = perm[a], ab = perm[a + 1], ba = perm[b], bb = perm[b + 1];
        const double n = blend(v, blend(u, corner0(perm[aa], tx, ty), corner0(perm[ba], tx - 1, ty)),
                                  blend(u, corner0(perm[ab], tx, ty - 1), corner0(perm[bb], tx - 1, ty - 1)));
        sum += n * weight;
        weight /= 2;
        cx *= 2; cz *= 2;
    }
    double l = sum * 0.5 + 0.5;
    l = l < 0.0 ? 0.0 : 1.0 < l ? 1.0 : l;
    const double elevation = p.intensity * (l - p.ground);
    return elevation >= 1 ? (int)floor(elevation + 0.5) : 0;   // lround: elevation is positive and small, + 0.5 is exact below the next integer
}
MV_HD int lround_pos(double v) { return (int)floor(v + 0.5); }

// ---- the episode, in three parts -----------------------------------------------------------------------------------------------------------
// head (serial): Env::reset's seed draw, the landscape's parameters, the noise's permutation.  Leaves the generator `g` where createLandscape's draws end.
MV_HD void draw_head(Mt &g, const GenState &st, Scratch &w, Params &p)
{
    g.s = w.mt;
    uint32_t seed = st.seed;
    if (st.seed_is_env_seed) {   // Env::seed, then Env::reset: seed = randRange(0, 1 << 30, rng); rng.seed(seed)
        mt_seed1(g, seed);
        seed = (uint32_t)rand_range1(g, 0, 1 << 30);
    }
    mt_seed1(g, seed);
    const unsigned kLandscape[7] = {0xffffff, 0xccffcc, 0xe6ecff, 0xd9d9d9, 0xffebcc, 0xb3b3b3, 0x555555};
    const unsigned kFloor[3] = {0xb3b3b3, 0x555555, 0x555555};
    p.land_color = kLandscape[rand_range1(g, 0, 7)];
    p.floor_color = kFloor[rand_range1(g, 0, 3)];
    p.nz = rand_range1(g, 8, HM_DIM);
    p.nx = rand_range1(g, 8, HM_DIM);
    const double frequency = double(rand_range1(g, 1, 100)) / 10.0;
    p.octaves = rand_range1(g, 1, 10);
    const uint32_t noiseSeed = (uint32_t)rand_range1(g, 0, 1000000000);
    p.step = HM_DIM / frequency;
    p.intensity = rand_range1(g, 5, 18);
    p.ground = frand1(g) * 0.5f + 0.2f;
    // siv::PerlinNoise::reseed: iota, shuffle with default_random_engine(seed), doubled
    for (int i = 0; i < 256; ++i) w.perm[i] = (uint8_t)i;
    Minstd m{noiseSeed % 2147483647ull};
    if (m.x == 0) m.x = 1;
    uint8_t *perm = w.perm;
    shuffle_walk(256, [&](uint32_t range) { return minstd_below(m, range); }, [&](int i, int j) { const uint8_t t = perm[i]; perm[i] = perm[j]; perm[j] = t; });
    for (int i = 0; i < 256; ++i) perm[256 + i] = perm[i];
}

// tail (serial): the merged slabs from the heightfield in w.hm, then every draw behind the landscape.  `out`'s heightmap is written by the caller.
// -> the value the env's NEXT Env::reset will draw for its seed (nothing consumes the env's stream in between)
struct NoMark { MV_HD void operator()(int) const {} };
// SLABS = false: the caller has merged the slabs already (collect_draw_kernel: all 64 lanes, draw_slabs_wave) and set out->num_boxes
template <class Mark = NoMark, bool SLABS = true>   // mark(k): phase k of the tail ends here (the timing build of the test hook reads a clock there)
MV_HD uint32_t draw_tail(Mt &g, const Params &p, Scratch &w, int top, int num_agents, float base_episode_len, CollectBlob *out, int &flags, Mark mark = Mark())
{
    const int nx = p.nx, nz = p.nz, ny = top + 1;
    const int8_t *hm = w.hm;
    if (SLABS) out->num_boxes = 0;
    out->num_objects = 0; out->num_rewards = 0; out->num_positive = 0; out->pad = 0;
    for (int i = 0; i < MAX_AGENTS; ++i) { out->spawn[i][0] = out->spawn[i][1] = out->spawn[i][2] = 0; }
    // ---- merged slabs: the floor layer (y == 0) and the hills, the lower colour value first, equal colours one class; seeds in (y, z, x) order, grown
    // along x, then z, then y
    const unsigned colorLo = p.land_color < p.floor_color ? p.land_color : p.floor_color, colorHi = p.land_color < p.floor_color ? p.floor_color : p.land_color;
    out->layout_color = (int)colorLo; out->wall_color = (int)colorHi;
    if (SLABS) {
        uint32_t *taken = w.taken;
        const int words = (nx * ny * nz + 31) / 32;
        for (int i = 0; i < words; ++i) taken[i] = 0;
        int numBoxes = 0;
        for (int slot = 0; slot < 2; ++slot) {
            const unsigned want = slot == 0 ? colorLo : colorHi;
            if (slot == 1 && colorHi == colorLo) break;
            auto free_cell = [&](int x, int y, int z) {
                if (x < 0 || x >= nx || z < 0 || z >= nz || y < 0 || y > hm[x * HM_DIM + z]) return false;
                const int id = (y * nz + z) * nx + x;
                return (y == 0 ? p.floor_color : p.land_color) == want && !((taken[id >> 5] >> (id & 31)) & 1u);
            };
            for (int y = 0; y < ny; ++y)
                for (int z = 0; z < nz; ++z)
                    for (int x = 0; x < nx; ++x) {
                        if (!free_cell(x, y, z)) continue;
                        int xEnd = x + 1, zEnd = z + 1, yEnd = y + 1;
                        while (free_cell(xEnd, y, z)) ++xEnd;
                        auto row_free = [&](int yy, int zz) { for (int xx = x; xx < xEnd; ++xx) if (!free_cell(xx, yy, zz)) return false; return true; };
                        while (row_free(y, zEnd)) ++zEnd;
                        auto layer_free = [&](int yy) { for (int zz = z; zz < zEnd; ++zz) if (!row_free(yy, zz)) return false; return true; };
                        while (layer_free(yEnd)) ++yEnd;
                        for (int yy = y; yy < yEnd; ++yy)
                            for (int zz = z; zz < zEnd; ++zz)
                                for (int xx = x; xx < xEnd; ++xx) { const int id = (yy * nz + zz) * nx + xx; taken[id >> 5] |= 1u << (id & 31); }
                        if (numBoxes >= COLLECT_MAX_BOXES) flags |= FLAG_SLABS;
                        else {
                            LayoutBox b;
                            b.min[0] = x; b.min[1] = y; b.min[2] = z; b.max[0] = xEnd; b.max[1] = yEnd; b.max[2] = zEnd;
                            b.type = VX_SOLID | VX_OPAQUE; b.slot = slot;
                            out->boxes[numBoxes++] = b;
                        }
                    }
        }
        out->num_boxes = numBoxes;
    }
    out->dim[0] = nx; out->dim[1] = top + 1; out->dim[2] = nz;
    mark(0);

    // ---- one shuffled list of free cells feeds agents, diamonds and movable boxes (scenario_collect.cpp:105-160)
    uint32_t *cells = w.cells;
    int n = 0;
    for (int x = 1; x < nx - 1; ++x)
        for (int z = 1; z < nz - 1; ++z) {
            const int h = hm[x * HM_DIM + z], y = h >= 1 ? h + 1 : 1;
            cells[n++] = (uint32_t)x | ((uint32_t)y << 8) | ((uint32_t)z << 16);
        }
    auto below = [&](uint32_t range) { return mt_below1(g, range); };
    shuffle_walk(n, below, [&](int i, int j) { swap_cells(cells, i, j); });
    mark(1);
    int next = 0;
    for (int i = 0; i < num_agents; ++i, ++next) {
        out->spawn[i][0] = (int)(cells[next] & 255u); out->spawn[i][1] = (int)((cells[next] >> 8) & 255u); out->spawn[i][2] = (int)((cells[next] >> 16) & 255u);
    }
    int numRewards = rand_range1(g, 1, lround_pos(0.05 * nz * nx) + 2);
    if (numRewards > n - next) numRewards = n - next;
    const int scattered = numRewards / 2 > 1 ? numRewards / 2 : 1;
    // reward cells: `scattered` of the shuffled list as it stands, the others from the highest ground -- an (unstable) std::sort by descending free height
    const int rewardFirst = next;
    next += scattered;
    sort_cells(cells + next, n - next, w.stack);
    mark(2);
    next += numRewards - scattered;   // (the two runs are adjacent in `cells`: rewardFirst .. next)
    const int rewardCount = next - rewardFirst;

    uint32_t *rest = cells + next;
    shuffle_walk(n - next, below, [&](int i, int j) { swap_cells(rest, i, j); });
    const int objectsMin = 3 > (int)(nx * nz * 0.04) ? 3 : (int)(nx * nz * 0.04);
    const int objectsCap = lround_pos(0.07 * nz * nx) + 2, objectsMax = objectsMin + 1 < objectsCap ? objectsMin + 1 : objectsCap;
    int numObjects = rand_range1(g, objectsMin, objectsMax);
    if (numObjects > n - next) numObjects = n - next;
    if (numObjects > MAX_OBJECTS) flags |= FLAG_OBJECTS;
    int placed = 0;
    if (next + numObjects < n)
        for (int i = 0; i < numObjects && i < MAX_OBJECTS; ++i, ++next) {
            MovableObject o;
            o.x = (int8_t)(cells[next] & 255u); o.y = (int8_t)((cells[next] >> 8) & 255u); o.z = (int8_t)((cells[next] >> 16) & 255u); o.state = 0;
            out->objects[placed++] = o;
        }
    out->num_objects = placed;
    for (int i = placed; i < MAX_OBJECTS; ++i) out->objects[i] = MovableObject{0, 0, 0, 0};

    out->episode_len = base_episode_len + 2.0f * float(rewardCount);   // scenario_collect.hpp:55-59

    for (int i = 0; i < num_agents; ++i) out->yaw_frand[i] = frand1(g);
    for (int i = num_agents; i < MAX_AGENTS; ++i) out->yaw_frand[i] = 0.0f;
    int numPositive = 0, kept = 0;
    for (int i = 0; i < rewardCount; ++i) {   // 70 % of the diamonds are worth +1 (:198)
        const bool good = frand1(g) > 0.3f;
        if (good) ++numPositive;
        if (i >= COLLECT_MAX_REWARDS) { flags |= FLAG_REWARDS; continue; }
        const uint32_t c = cells[rewardFirst + i];
        MovableObject o;
        o.x = (int8_t)(c & 255u); o.y = (int8_t)((c >> 8) & 255u); o.z = (int8_t)((c >> 16) & 255u); o.state = (int8_t)(good ? 1 : 2);
        out->rewards[kept++] = o;
    }
    out->num_rewards = kept; out->num_positive = numPositive;
    for (int i = kept; i < COLLECT_MAX_REWARDS; ++i) out->rewards[i] = MovableObject{0, 0, 0, 0};
    return (uint32_t)rand_range1(g, 0, 1 << 30);
}

}  // namespace cdraw
}  // namespace mv
