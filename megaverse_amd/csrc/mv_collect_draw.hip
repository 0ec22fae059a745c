// megaverse_amd/csrc/mv_collect_draw.hip -- Collect episodes generated on the device (mv_collect_draw.h): the kernel, its launcher, and the test hooks that
// hold it against mv_gen_collect.cpp (the host generator, which tests/test_host_generators.py holds against the oracle and tests/test_oracle_collect.py
// against the reference's perlin_noise.hpp).
#include <cstring>
#include <string>
#include <vector>

#include "mv_api_internal.h"
#include "mv_collect_draw.h"
#include "mv_math.h"

namespace mv {

using namespace cdraw;

// One wavefront per entry of `envs` (or per env when it is null): the env's next episode into its staging slot, its generator state advanced.
// The slot's `seq` is written last, behind a release: whoever finds it set finds the whole episode (the host orders its copies behind this launch anyway).
// The slab merge of draw_tail (mv_collect_draw.h) by all 64 lanes of the episode's wave -- the same greedy walk, box for box: the scan for the next free cell
// takes 64 cells per round (a ballot), a box's growth along x, then z, then y one ballot each (lane k tests the k-th cell / row / layer beyond the seed, the run
// of ones from lane 0 is the growth), the box's cells are marked by the lanes together.  The serial form scanned up to 25 k cells one LDS round trip at a time:
// 880 us of an episode's 1.4 ms (r12c).
__device__ __forceinline__ int trailing_ones(unsigned long long m) { return m == ~0ull ? 64 : __ffsll((long long)~m) - 1; }
__device__ __forceinline__ void draw_slabs_wave(const Params &p, Scratch &w, int top, CollectBlob *out, int &flags)
{
    const int lane = lane_id();
    const int nx = p.nx, nz = p.nz, ny = top + 1, total = nx * ny * nz;
    const int8_t *hm = w.hm;
    uint32_t *taken = w.taken;
    for (int i = lane; i < (total + 31) / 32; i += 64) taken[i] = 0;
    wave_sync();
    const unsigned colorLo = p.land_color < p.floor_color ? p.land_color : p.floor_color, colorHi = p.land_color < p.floor_color ? p.floor_color : p.land_color;
    int numBoxes = 0;
    for (int slot = 0; slot < 2; ++slot) {
        const unsigned want = slot == 0 ? colorLo : colorHi;
        if (slot == 1 && colorHi == colorLo) break;
        auto free_cell = [&](int x, int y, int z) {
            if (x < 0 || x >= nx || z < 0 || z >= nz || y < 0 || y > hm[x * HM_DIM + z]) return false;
            const int id = (y * nz + z) * nx + x;
            return (y == 0 ? p.floor_color : p.land_color) == want && !((taken[id >> 5] >> (id & 31)) & 1u);
        };
        for (int base = 0; base < total; base += 64) {   // cells in scan order: id = (y nz + z) nx + x
            const int id = base + lane, mx = id % nx, mt = id / nx, mz = mt % nz, my = mt / nz;
            unsigned long long m = __ballot(id < total && free_cell(mx, my, mz));
            while (m) {
                const int seed = base + __ffsll((long long)m) - 1;
                const int x = seed % nx, t = seed / nx, z = t % nz, y = t / nz;
                const int xEnd = x + 1 + trailing_ones(__ballot(free_cell(x + 1 + lane, y, z)));
                bool ok = z + 1 + lane < nz;
                for (int xx = x; ok && xx < xEnd; ++xx) ok = free_cell(xx, y, z + 1 + lane);
                const int zEnd = z + 1 + trailing_ones(__ballot(ok));
                ok = y + 1 + lane < ny;
                for (int zz = z; ok && zz < zEnd; ++zz)
                    for (int xx = x; ok && xx < xEnd; ++xx) ok = free_cell(xx, y + 1 + lane, zz);
                const int yEnd = y + 1 + trailing_ones(__ballot(ok));
                const int wx = xEnd - x, wz = zEnd - z, vol = wx * wz * (yEnd - y);
                wave_sync();   // (every lane's reads of the bits before they change)
                for (int i = lane; i < vol; i += 64) {
                    const int cx = x + i % wx, ct = i / wx, cz = z + ct % wz, cy = y + ct / wz, cid = (cy * nz + cz) * nx + cx;
                    atomicOr(&taken[cid >> 5], 1u << (cid & 31));
                }
                wave_sync();
                if (numBoxes >= COLLECT_MAX_BOXES) flags |= FLAG_SLABS;
                else {
                    if (lane == 0) {
                        LayoutBox b;
                        b.min[0] = x; b.min[1] = y; b.min[2] = z; b.max[0] = xEnd; b.max[1] = yEnd; b.max[2] = zEnd;
                        b.type = VX_SOLID | VX_OPAQUE; b.slot = slot;
                        out->boxes[numBoxes] = b;
                    }
                    ++numBoxes;
                }
                m = __ballot(id < total && free_cell(mx, my, mz));
            }
        }
    }
    if (lane == 0) out->num_boxes = numBoxes;
}

// DRAW_WAVES episodes per workgroup, one per wave (nothing in the kernel crosses a wave).  Measured both ways (r12h / r12i: 65 one-wave workgroups against 17
// four-wave ones per launch): no difference to the observation launches beside them -- what a draw launch costs them it costs by being IN FLIGHT (mv_feeder.cpp:
// the batches gather), not by where its waves sit.  Four keeps the grid small.
#ifndef MV_DRAW_WAVES
#define MV_DRAW_WAVES 4
#endif
constexpr int DRAW_WAVES = MV_DRAW_WAVES;
static_assert(sizeof(Scratch) * DRAW_WAVES <= 65000, "static LDS of collect_draw_kernel");
__global__ __launch_bounds__(64 * DRAW_WAVES) void collect_draw_kernel(GenState *states, const int32_t *envs, int count, uint8_t *slots, size_t slot_bytes, int num_agents,
                                                          float base_episode_len, int32_t *flag_word, long long *timing = nullptr)
{
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int k = (int)blockIdx.x * DRAW_WAVES + wave;   // (the waves of a workgroup draw episodes of their own: nothing below crosses a wave)
    if (k >= count) return;
    // One lane's chain of dependent instructions beside the observation passes' seven or eight waves per SIMD: top priority, so that the few dozen waves of a
    // launch are done -- and the launch out of flight -- as soon as their own latencies allow (r12f: no measurable difference to the passes either way).
    __builtin_amdgcn_s_setprio(3);
    const int env = envs ? envs[k] : k;
    const int lane = lane_id();
    __shared__ Scratch s_w[DRAW_WAVES];
    __shared__ Params s_p[DRAW_WAVES];
    __shared__ int s_tops[DRAW_WAVES];
    Scratch &w = s_w[wave];
    Params &sp = s_p[wave];
    int &s_top = s_tops[wave];
    CollectBlob *out = reinterpret_cast<CollectBlob *>(slots + (size_t)env * slot_bytes);
    GenState st = states[env];
    Mt g{w.mt, 624};
    long long *tm = timing ? timing + (size_t)k * 8 : nullptr;   // (test hook: clock reads at the phases' ends, lane 0)
    if (tm && lane == 0) tm[0] = wall_clock64();
    if (lane == 0) {
        Params p;
        draw_head(g, st, w, p);
        sp = p;
        s_top = 0;
    }
    for (int i = lane; i < HM_BYTES / 4; i += 64) reinterpret_cast<uint32_t *>(w.hm)[i] = 0xffffffffu;
    wave_sync();
    if (tm && lane == 0) tm[1] = wall_clock64();
    {   // the heightfield: one column per lane and round
        const Params p = sp;
        const int n = p.nx * p.nz;
        int top = 0;
        for (int i = lane; i < n; i += 64) {
            const int x = i / p.nz, z = i - x * p.nz;
            const int h = column_height(p, w.perm, x, z);
            w.hm[x * HM_DIM + z] = (int8_t)h;
            top = max(top, h);
        }
        if (top > 0) atomicMax(&s_top, top);
    }
    wave_sync();
    int flags = 0;
    uint32_t nextSeed = 0;
    if (tm && lane == 0) tm[2] = wall_clock64();
    draw_slabs_wave(sp, w, s_top, out, flags);
    if (tm && lane == 0) tm[3] = wall_clock64();
    if (lane == 0) {
        if (tm) {
            auto mark = [tm](int q) { if (q > 0) tm[3 + q] = wall_clock64(); };
            nextSeed = draw_tail<decltype(mark), false>(g, sp, w, s_top, num_agents, base_episode_len, out, flags, mark);
        } else nextSeed = draw_tail<NoMark, false>(g, sp, w, s_top, num_agents, base_episode_len, out, flags);
    }
    if (tm && lane == 0) tm[6] = wall_clock64();
    for (int i = lane; i < HM_BYTES / 16; i += 64) reinterpret_cast<uint4 *>(out->heightmap)[i] = reinterpret_cast<const uint4 *>(w.hm)[i];
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    wave_sync();
    if (lane == 0) {
        __hip_atomic_store(&out->seq, st.next_seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        st.seed = nextSeed; st.seed_is_env_seed = 0; st.next_seq += 1; st.flags |= flags;
        states[env] = st;
        if (flags && flag_word) atomicOr(flag_word, flags);   // (a capacity of the episode record was hit: reported once by the next stepping call)
    }
}

void launch_collect_draw(void *states, const int32_t *envs, int count, uint8_t *slots, size_t slot_bytes, int num_agents, float base_episode_len,
                         hipStream_t stream, int32_t *flag_word)
{
    if (count <= 0) return;
    hipLaunchKernelGGL(collect_draw_kernel, dim3((count + DRAW_WAVES - 1) / DRAW_WAVES), dim3(64 * DRAW_WAVES), 0, stream, static_cast<GenState *>(states), envs, count, slots, slot_bytes, num_agents,
                       base_episode_len, flag_word);
}

// Device-drawn episodes, staging slot -> ring slot: one workgroup per episode, the record's used prefix (the slab list comes last), `seq` last of all --
// one launch where hipMemcpyAsync would be one small copy kernel per episode on the simulation stream (r12b: 17 of them per call of 16 ticks).
struct BlobCopyArgs {
    int32_t count;
    int32_t env[64], slot[64];
};
__global__ __launch_bounds__(256) void collect_blob_copy_kernel(BlobCopyArgs a, const uint8_t *staging, uint8_t *ring, size_t blob_bytes, int spares)
{
    const int k = blockIdx.x;
    if (k >= a.count) return;
    const int env = a.env[k];
    const CollectBlob *src = reinterpret_cast<const CollectBlob *>(staging + (size_t)env * blob_bytes);
    CollectBlob *dst = reinterpret_cast<CollectBlob *>(ring + ((size_t)env * spares + (size_t)a.slot[k]) * blob_bytes);
    const int nb = min(max(src->num_boxes, 0), (int)COLLECT_MAX_BOXES);
    const int words = (int)((offsetof(CollectBlob, boxes) + (size_t)nb * sizeof(LayoutBox)) / 16);
    const uint4 *s4 = reinterpret_cast<const uint4 *>(src);
    uint4 *d4 = reinterpret_cast<uint4 *>(dst);
    for (int i = (int)threadIdx.x + 1; i < words; i += 256) d4[i] = s4[i];
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    __syncthreads();
    if (threadIdx.x == 0) d4[0] = s4[0];   // (seq and the counts: a reader that finds seq set finds the rest -- the step launches are ordered behind this one anyway)
}
static_assert(offsetof(CollectBlob, boxes) % 16 == 0 && sizeof(LayoutBox) % 16 == 0, "the record's used prefix is copied in 16-byte words");

void launch_collect_blob_copy(const int32_t *envs, const int32_t *slots, int count, const uint8_t *staging, uint8_t *ring, size_t blob_bytes, int spares,
                              hipStream_t stream)
{
    for (int first = 0; first < count; first += 64) {
        BlobCopyArgs a;
        a.count = std::min(64, count - first);
        for (int i = 0; i < a.count; ++i) { a.env[i] = envs[first + i]; a.slot[i] = slots[first + i]; }
        hipLaunchKernelGGL(collect_blob_copy_kernel, dim3(a.count), dim3(256), 0, stream, a, staging, ring, blob_bytes, spares);
    }
}

// the same episode on the host: the code of mv_collect_draw.h compiled for the CPU
static uint32_t collect_draw_host(GenState &st, Scratch &w, int num_agents, float base_episode_len, CollectBlob &out)
{
    Mt g{w.mt, 624};
    Params p;
    draw_head(g, st, w, p);
    std::memset(w.hm, 0xff, sizeof w.hm);
    int top = 0;
    for (int x = 0; x < p.nx; ++x)
        for (int z = 0; z < p.nz; ++z) {
            const int h = column_height(p, w.perm, x, z);
            w.hm[x * HM_DIM + z] = (int8_t)h;
            top = std::max(top, h);
        }
    int flags = 0;
    const uint32_t next = draw_tail(g, p, w, top, num_agents, base_episode_len, &out, flags);
    std::memcpy(out.heightmap, w.hm, sizeof out.heightmap);
    out.seq = st.next_seq;
    st.seed = next; st.seed_is_env_seed = 0; st.next_seq += 1; st.flags |= flags;
    return next;
}

}  // namespace mv

using namespace mv;
using namespace mvapi;

extern "C" {

// Host-only test hook (no device needed): the n-th episode of an env seeded with `env_seed` as mv_collect_draw.h's code generates it ON THE HOST -- the
// same record mv_debug_generate_episode("Collect", ...) returns (its `seq` is n here, 0 there; the records' unused tails are zero here).
int mv_debug_collect_draw_host(int32_t num_agents, int32_t env_seed, int32_t n, float base_episode_len, void *out, int32_t out_bytes)
{
    if (!out) return (int)sizeof(CollectBlob);
    if (num_agents < 1 || num_agents > MAX_AGENTS || n < 1) return fail("mv_debug_collect_draw_host: bad arguments");
    if ((size_t)out_bytes < sizeof(CollectBlob)) return fail("mv_debug_collect_draw_host: buffer too small");
    std::vector<uint8_t> buf(sizeof(CollectBlob), 0);
    auto w = std::make_unique<cdraw::Scratch>();
    cdraw::GenState st{(uint32_t)env_seed, 1, 1, 0};
    for (int i = 0; i < n; ++i) {
        std::memset(buf.data(), 0, buf.size());
        collect_draw_host(st, *w, num_agents, base_episode_len, *reinterpret_cast<CollectBlob *>(buf.data()));
    }
    std::memcpy(out, buf.data(), buf.size());
    return (int)buf.size();
}

// Test hook: `count` envs seeded with env_seeds[i], each drawing its first n episodes ON THE DEVICE (n launches of collect_draw_kernel, `count` wavefronts
// each); out receives the envs' n-th episodes, count x sizeof(CollectBlob) bytes; *ms_per_launch (may be null) the mean time of a launch.
int mv_debug_collect_draw_device(int32_t device, int32_t num_agents, const int32_t *env_seeds, int32_t count, int32_t n, float base_episode_len, void *out,
                                 int64_t out_bytes, float *ms_per_launch)
{
    if (num_agents < 1 || num_agents > MAX_AGENTS || n < 1 || count < 1 || !env_seeds || !out) return fail("mv_debug_collect_draw_device: bad arguments");
    if ((size_t)out_bytes < (size_t)count * sizeof(CollectBlob)) return fail("mv_debug_collect_draw_device: buffer too small");
    HIP_TRY(hipSetDevice(device));
    std::vector<cdraw::GenState> st((size_t)count);
    for (int i = 0; i < count; ++i) st[(size_t)i] = cdraw::GenState{(uint32_t)env_seeds[i], 1, 1, 0};
    cdraw::GenState *dSt = nullptr;
    uint8_t *dSlots = nullptr;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    const size_t bytes = (size_t)count * sizeof(CollectBlob);
    bool ok = hipMalloc((void **)&dSt, (size_t)count * sizeof(cdraw::GenState)) == hipSuccess && hipMalloc((void **)&dSlots, bytes) == hipSuccess &&
              hipMemset(dSlots, 0, bytes) == hipSuccess &&
              hipMemcpy(dSt, st.data(), (size_t)count * sizeof(cdraw::GenState), hipMemcpyHostToDevice) == hipSuccess &&
              hipEventCreate(&e0) == hipSuccess && hipEventCreate(&e1) == hipSuccess;
    if (ok) {
        ok = hipEventRecord(e0, nullptr) == hipSuccess;
        long long *dTm = nullptr;
        const bool timing = getenv("MV_DRAW_TIMING") != nullptr;   // phases of the LAST launch's episodes to stderr (100 MHz clock)
        if (timing) ok = hipMalloc((void **)&dTm, (size_t)count * 8 * sizeof(long long)) == hipSuccess;
        for (int i = 0; i < n && ok; ++i) {
            if (timing) hipLaunchKernelGGL(collect_draw_kernel, dim3((count + DRAW_WAVES - 1) / DRAW_WAVES), dim3(64 * DRAW_WAVES), 0, nullptr, dSt, (const int32_t *)nullptr, count, dSlots, sizeof(CollectBlob),
                                           num_agents, base_episode_len, (int32_t *)nullptr, dTm);
            else launch_collect_draw(dSt, nullptr, count, dSlots, sizeof(CollectBlob), num_agents, base_episode_len, nullptr, nullptr);
        }
        if (timing && ok) {
            std::vector<long long> tm((size_t)count * 8);
            ok = hipMemcpy(tm.data(), dTm, tm.size() * sizeof(long long), hipMemcpyDeviceToHost) == hipSuccess;
            double ph[6] = {0, 0, 0, 0, 0, 0}, worst = 0;
            for (int e = 0; e < count && ok; ++e) {
                for (int q = 0; q < 6; ++q) ph[q] += double(tm[(size_t)e * 8 + q + 1] - tm[(size_t)e * 8 + q]) * 0.01;
                worst = std::max(worst, double(tm[(size_t)e * 8 + 6] - tm[(size_t)e * 8]) * 0.01);
            }
            std::fprintf(stderr, "[mv draw timing] %d episodes, us per episode: head %.0f, heights %.0f, slabs %.0f, cells + shuffle %.0f, sort %.0f, rest %.0f; "
                         "slowest episode %.0f\n", count, ph[0] / count, ph[1] / count, ph[2] / count, ph[3] / count, ph[4] / count, ph[5] / count, worst);
            (void)hipFree(dTm);
        }
        ok = ok && hipGetLastError() == hipSuccess && hipEventRecord(e1, nullptr) == hipSuccess && hipEventSynchronize(e1) == hipSuccess &&
             hipMemcpy(out, dSlots, bytes, hipMemcpyDeviceToHost) == hipSuccess;
        float ms = 0.0f;
        if (ok && ms_per_launch && hipEventElapsedTime(&ms, e0, e1) == hipSuccess) *ms_per_launch = ms / float(n);
    }
    if (e0) (void)hipEventDestroy(e0);
    if (e1) (void)hipEventDestroy(e1);
    if (dSt) (void)hipFree(dSt);
    if (dSlots) (void)hipFree(dSlots);
    return ok ? 0 : fail("mv_debug_collect_draw_device: a HIP call failed");
}

}  // extern "C"
