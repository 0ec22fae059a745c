// megaverse_amd/csrc/mv_rng.h -- device restatement of the reference's RNG helpers.
//
// The reference draws its procedural generation from std::mt19937 through libstdc++'s
// uniform_int_distribution / uniform_real_distribution<float> / std::shuffle
// (reference: src/libs/util/include/util/util.hpp:25-56; scenario_tower_building.cpp:46).
// Those are third-party (libstdc++ 11.4 in this image); their published algorithms are restated
// here for one wavefront:
//   * MT19937 (Matsumoto & Nishimura 1998), state in LDS, regeneration ("twist") done by all 64
//     lanes in three dependency-free phases;
//   * uniform_int_distribution on a 32-bit URBG: Lemire's nearly-divisionless method
//     (bits/uniform_int_dist.h, _S_nd);
//   * generate_canonical<float,24>: one draw / 2^32, clamped below 1 (bits/random.tcc);
//   * std::shuffle: two swap positions per draw while range^2 fits in 32 bits (bits/stl_algo.h).
// tests/test_rng_parity.py checks every one of them against the reference's util.hpp compiled in
// place (the _ref build of the test tree) and against the standard's mt19937 known answer.
//
// Execution model: every lane of the wave runs the same scalar code on the same values ("uniform
// execution"); LDS reads are broadcasts, LDS writes are done by lane 0 only.  The block is exactly
// one wave, so wave_sync() (mv_math.h) is the ordering point.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace mv {

struct Mt19937 {
    uint32_t *mt;   // 624 words in LDS
    int idx;        // wave-uniform
};

__device__ __forceinline__ void mt_seed(Mt19937 &g, uint32_t seed)
{
    const int lane = (int)(threadIdx.x & 63);
    uint32_t x = seed;
    if (lane == 0) g.mt[0] = x;
    for (int i = 1; i < 624; ++i) {
        x = 1812433253u * (x ^ (x >> 30)) + (uint32_t)i;
        if (lane == 0) g.mt[i] = x;
    }
    g.idx = 624;
    wave_sync();
}

__device__ __forceinline__ uint32_t mt_mix(uint32_t a, uint32_t b, uint32_t m)
{
    const uint32_t y = (a & 0x80000000u) | (b & 0x7fffffffu);
    return m ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
}

__device__ __forceinline__ void mt_twist_range(uint32_t *mt, int k0, int k1, int lane)
{   // new mt[k] for k in [k0,k1): inputs mt[k], mt[k+1] (old) and mt[(k+397)%624] (old for k<227, new otherwise)
    for (int base = k0; base < k1; base += 64) {
        const int k = base + lane;
        uint32_t v = 0;
        if (k < k1) v = mt_mix(mt[k], mt[k + 1], mt[k < 227 ? k + 397 : k - 227]);
        wave_sync();
        if (k < k1) mt[k] = v;
        wave_sync();
    }
}

__device__ __forceinline__ void mt_twist(Mt19937 &g)
{
    const int lane = (int)(threadIdx.x & 63);
    mt_twist_range(g.mt, 0, 227, lane);     // needs old [0..227] and old [397..623]
    mt_twist_range(g.mt, 227, 454, lane);   // needs new [0..226]
    mt_twist_range(g.mt, 454, 623, lane);   // needs new [227..395]
    const uint32_t v = mt_mix(g.mt[623], g.mt[0], g.mt[396]);
    wave_sync();
    if (lane == 0) g.mt[623] = v;
    wave_sync();
    g.idx = 0;
}

__device__ __forceinline__ uint32_t mt_next(Mt19937 &g)
{
    if (g.idx >= 624) mt_twist(g);
    uint32_t y = g.mt[g.idx++];
    y ^= y >> 11;
    y ^= (y << 7) & 0x9d2c5680u;
    y ^= (y << 15) & 0xefc60000u;
    y ^= y >> 18;
    return y;
}

// unbiased integer in [0, range), range >= 1
__device__ __forceinline__ uint32_t mt_below(Mt19937 &g, uint32_t range)
{
    uint64_t product = (uint64_t)mt_next(g) * (uint64_t)range;
    uint32_t low = (uint32_t)product;
    if (low < range) {
        const uint32_t threshold = (0u - range) % range;
        while (low < threshold) {
            product = (uint64_t)mt_next(g) * (uint64_t)range;
            low = (uint32_t)product;
        }
    }
    return (uint32_t)(product >> 32);
}

// reference randRange(low, high): integer in [low, high)
__device__ __forceinline__ int rand_range(Mt19937 &g, int low, int high) { return low + (int)mt_below(g, (uint32_t)(high - low)); }
__device__ __forceinline__ bool random_bool(Mt19937 &g) { return rand_range(g, 0, 2) != 0; }

// reference frand(): float in [0,1)
__device__ __forceinline__ float frand(Mt19937 &g)
{
    const float sum = (float)mt_next(g);
    float ret = sum / 4294967296.0f;
    if (ret >= 1.0f) ret = 0.99999994f;   // nextafter(1.0f, 0.0f)
    return ret;
}

// std::shuffle of n 16-bit items in LDS
__device__ __forceinline__ void shuffle_u16(Mt19937 &g, uint16_t *a, int n)
{
    if (n <= 0) return;
    const int lane = (int)(threadIdx.x & 63);
    auto swap_items = [&](int i, int j) {
        const uint16_t vi = a[i], vj = a[j];
        wave_sync();
        if (lane == 0) { a[i] = vj; a[j] = vi; }
        wave_sync();
    };
    // n*n <= 2^32-1 for every n this simulator uses (n <= 65535), i.e. the paired path
    int i = 1;
    if ((n % 2) == 0) {
        const int j = (int)mt_below(g, 2u);
        swap_items(i, j);
        ++i;
    }
    while (i < n) {
        const uint32_t r = (uint32_t)i + 1u;
        const uint32_t x = mt_below(g, r * (r + 1u));
        const uint32_t p0 = x / (r + 1u), p1 = x % (r + 1u);
        swap_items(i, (int)p0);
        ++i;
        swap_items(i, (int)p1);
        ++i;
    }
}

// std::shuffle of the n items item(0) .. item(n - 1) when only the FIRST `need` (<= 64) entries of the result are looked at afterwards: the same
// permutation as shuffle_u16, without its chain of n dependent LDS swaps (two round trips each: ~75 us for the 600 spawn candidates of a large room,
// and a launch lasts as long as its slowest env).
//   1. The swap partners of all steps come from consecutive generator outputs (step i swaps positions i and j_i <= i; two steps share a draw): one
//      draw per lane, 64 at a time, written to `steps`.  That needs every draw to be accepted at once (Lemire's method rejects ~range / 2^32 of
//      them) and the outputs to come from the current state (no regeneration in between): otherwise -> false, nothing consumed, and the caller
//      runs shuffle_u16.
//   2. Entry p of the result is found by walking the steps BACKWARDS from position p -- "who was here before step i" -- down to the original index;
//      one lane per entry, all lanes read the same step (an LDS broadcast).
// `steps`: n 16-bit words of LDS; out[p], p < need, receives item(original index).
template <class Item>
__device__ __forceinline__ bool shuffle_prefix_u16(Mt19937 &g, int n, int need, uint16_t *steps, uint16_t *out, Item item)
{
    const int lane = (int)(threadIdx.x & 63);
    if (n <= 1 || need > 64) return false;
    if (g.idx >= 624) mt_twist(g);   // (what the first draw would do)
    const int single = (n % 2) == 0 ? 1 : 0;          // an even n starts with one step of its own (i = 1)
    const int pairs = (n - 1 - single + 1) / 2;       // then two steps per draw: (i, i + 1), i = 1 + single, 3 + single, ...; the last pair may hold one step
    const int draws = single + pairs;
    if (g.idx + draws > 624) return false;
    bool redo = false;
    for (int base = 0; base < draws; base += 64) {
        const int d = base + lane;
        if (d < draws) {
            uint32_t y = g.mt[g.idx + d];
            y ^= y >> 11;
            y ^= (y << 7) & 0x9d2c5680u;
            y ^= (y << 15) & 0xefc60000u;
            y ^= y >> 18;
            if (d < single) {
                const uint64_t product = (uint64_t)y * 2ull;
                redo = redo || (uint32_t)product < 2u;
                steps[1] = (uint16_t)(product >> 32);
            } else {
                const int i = 1 + single + 2 * (d - single);
                const uint32_t r = (uint32_t)i + 1u, range = r * (r + 1u);
                const uint64_t product = (uint64_t)y * (uint64_t)range;
                redo = redo || (uint32_t)product < range;   // (mt_below would look at its threshold now, and perhaps draw again)
                const uint32_t x = (uint32_t)(product >> 32);
                const uint32_t p0 = x / (r + 1u), p1 = x - p0 * (r + 1u);
                steps[i] = (uint16_t)p0;
                if (i + 1 < n) steps[i + 1] = (uint16_t)p1;
            }
        }
    }
    if (__ballot(redo) != 0ull) return false;
    wave_sync();
    int p = lane;
#pragma unroll 8
    for (int i = n - 1; i >= 1; --i) {
        const int j = (int)steps[i];
        p = p == i ? j : (p == j ? i : p);
    }
    if (lane < need) out[lane] = item(p);
    g.idx += draws;
    wave_sync();
    return true;
}

}  // namespace mv
