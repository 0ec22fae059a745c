// megaverse_amd/csrc/mv_math.h -- fp32 device math for the simulator kernels.
//
// Numerics contract (DESIGN.md "numerics"): the library is built with -ffp-contract=off and no
// fast-math, so every fp32 +,-,*,/ and sqrt below is one IEEE-754 round-to-nearest operation in
// source order (hipcc's default -fhip-fp32-correctly-rounded-divide-sqrt keeps / and sqrtf exact).
// That is what lets the parity tests demand bit-equal state and pixels.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace mv {

struct V3 {
    float x, y, z;
};
__device__ __forceinline__ V3 v3(float x, float y, float z) { return V3{x, y, z}; }
__device__ __forceinline__ V3 operator+(V3 a, V3 b) { return v3(a.x + b.x, a.y + b.y, a.z + b.z); }
__device__ __forceinline__ V3 operator-(V3 a, V3 b) { return v3(a.x - b.x, a.y - b.y, a.z - b.z); }
__device__ __forceinline__ V3 operator*(V3 a, float s) { return v3(a.x * s, a.y * s, a.z * s); }
__device__ __forceinline__ float dot(V3 a, V3 b) { return (a.x * b.x + a.y * b.y) + a.z * b.z; }
__device__ __forceinline__ float len2(V3 a) { return dot(a, a); }

// select-based min/max with std::min/std::max operand semantics (no fminf NaN/-0 rules)
__device__ __forceinline__ float fmin_sel(float a, float b) { return (b < a) ? b : a; }
__device__ __forceinline__ float fmax_sel(float a, float b) { return (a < b) ? b : a; }
__device__ __forceinline__ float clampf(float v, float lo, float hi) { return fmin_sel(fmax_sel(v, lo), hi); }

// sin/cos on |x| <~ 8: two-step pi/2 reduction + cephes-style minimax polynomials
__device__ __forceinline__ void sincos_poly(float x, float &s, float &c)
{
    const float TWO_OVER_PI = 0.636619772f;
    const float PIO2_HI = 1.57079625f;
    const float PIO2_LO = 7.54978942e-08f;
    const float kf = floorf(x * TWO_OVER_PI + 0.5f);
    const int k = (int)kf;
    float r = x - kf * PIO2_HI;
    r = r - kf * PIO2_LO;
    const float r2 = r * r;
    const float sp = ((-1.9515295891e-4f * r2 + 8.3321608736e-3f) * r2 - 1.6666654611e-1f) * r2 * r + r;
    const float cp = ((2.443315711809948e-5f * r2 - 1.388731625493765e-3f) * r2 + 4.166664568298827e-2f) * r2 * r2 -
                     0.5f * r2 + 1.0f;
    const int q = k & 3;
    s = (q == 0) ? sp : (q == 1) ? cp : (q == 2) ? -sp : -cp;
    c = (q == 0) ? cp : (q == 1) ? -sp : (q == 2) ? -cp : sp;
}

// (cos, sin) entries of the y-rotation matrix built from an axis-angle quaternion
__device__ __forceinline__ void yaw_matrix(float angle, float &c_out, float &s_out)
{
    float sh, ch;
    sincos_poly(angle * 0.5f, sh, ch);
    const float d = sh * sh + ch * ch;
    const float s = 2.0f / d;
    const float ys = sh * s;
    const float wy = ch * ys;
    const float yy = sh * ys;
    c_out = 1.0f - yy;
    s_out = wy;
}

// Per-env records have a wave-uniform address, so hipcc fetches them with scalar (s_load) instructions.
// The scalar data cache is built for kernel constants shared by every wave; with one private record
// per wave its miss path serialises (measured: 16 us for a 128-byte header at 1024 waves versus 0.4 us
// through the vector path).  Laundering the pointer through a VGPR makes the loads vector loads.
template <class T>
__device__ __forceinline__ T *vector_path(T *p)
{
    asm volatile("" : "+v"(p));
    return p;
}

// ---- wave64 helpers ------------------------------------------------------------------------
// Ordering point for LDS / global traffic ONE wavefront exchanges between its own lanes: what __syncthreads() is to a one-wave
// workgroup minus the s_barrier, so that the tick code (one wave per env) can run as wave 0 of a larger workgroup whose other waves
// are parked at a real barrier (fused step + frame setup kernels).  A wave's LDS operations execute in order.
__device__ __forceinline__ void wave_sync()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
}

__device__ __forceinline__ int lane_id() { return (int)(threadIdx.x & 63); }

// value of lane `src_lane`, which every caller passes wave-uniform (a lane found by a ballot, a loop counter): one v_readlane_b32 instead of
// a ds_bpermute_b32 round trip through the LDS crossbar (~100 cycles on the tick's critical path, three to five of them per sweep)
__device__ __forceinline__ float bcast_f(float v, int src_lane)
{
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), __builtin_amdgcn_readfirstlane(src_lane)));
}
__device__ __forceinline__ int bcast_i(int v, int src_lane) { return __builtin_amdgcn_readlane(v, __builtin_amdgcn_readfirstlane(src_lane)); }

// Wave-wide reductions over all 64 lanes (the callers run with a full exec mask) on the DPP path: within quads, within rows of 16 (rotations),
// then across rows (row_bcast:15, row_bcast:31); lane 63 ends up with the result.  Six VALU steps of a few cycles each -- the __shfl_xor
// butterfly they replace is six dependent LDS-crossbar round trips.  (old == src: a lane without a source for the row broadcasts keeps its value.)
#define MV_DPP(v, ctrl) __builtin_amdgcn_update_dpp((int)(v), (int)(v), ctrl, 0xf, 0xf, false)
__device__ __forceinline__ unsigned wave_min_u32(unsigned v)
{
    v = min(v, (unsigned)MV_DPP(v, 0xb1));    // quad_perm:[1,0,3,2]
    v = min(v, (unsigned)MV_DPP(v, 0x4e));    // quad_perm:[2,3,0,1]
    v = min(v, (unsigned)MV_DPP(v, 0x124));   // row_ror:4
    v = min(v, (unsigned)MV_DPP(v, 0x128));   // row_ror:8
    v = min(v, (unsigned)MV_DPP(v, 0x142));   // row_bcast:15
    v = min(v, (unsigned)MV_DPP(v, 0x143));   // row_bcast:31
    return (unsigned)__builtin_amdgcn_readlane((int)v, 63);
}
__device__ __forceinline__ unsigned wave_or_u32(unsigned v)
{
    v |= (unsigned)MV_DPP(v, 0xb1);
    v |= (unsigned)MV_DPP(v, 0x4e);
    v |= (unsigned)MV_DPP(v, 0x124);
    v |= (unsigned)MV_DPP(v, 0x128);
    v |= (unsigned)MV_DPP(v, 0x142);
    v |= (unsigned)MV_DPP(v, 0x143);
    return (unsigned)__builtin_amdgcn_readlane((int)v, 63);
}
#undef MV_DPP
__device__ __forceinline__ unsigned long long wave_or_u64(unsigned long long v)
{
    return ((unsigned long long)wave_or_u32((unsigned)(v >> 32)) << 32) | wave_or_u32((unsigned)v);
}

// Phase timing of a tick (an instrumented build only: -DMV_TICK_TIMING; results are printed by mv_close): MV_T(k) adds the shader cycles
// since the previous mark to slot k of the env's record
#ifdef MV_TICK_TIMING
#define MV_T_BEGIN unsigned long long mv_t_last_ = __builtin_amdgcn_s_memtime();
#define MV_T(k)                                                                                     \
    do {                                                                                            \
        const unsigned long long mv_t_now_ = __builtin_amdgcn_s_memtime();                          \
        if (gv.dbg && (threadIdx.x & 63) == 0) {                                                    \
            gv.dbg[(size_t)env * 64 + (k)] += mv_t_now_ - mv_t_last_;                               \
            gv.dbg[(size_t)env * 64 + 16 + (k)] = mv_t_now_ - mv_t_last_;                           \
        }                                                                                           \
        mv_t_last_ = mv_t_now_;                                                                     \
    } while (0)
#else
#define MV_T_BEGIN
#define MV_T(k) do { } while (0)
#endif

}  // namespace mv
