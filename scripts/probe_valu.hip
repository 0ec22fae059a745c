// scripts/probe_valu.hip -- issue rate of wave64 VALU instruction classes on gfx950 (cycles per instruction per SIMD)
#include <hip/hip_runtime.h>
#include <cstdio>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
template <int OP>
__global__ __launch_bounds__(256) void k(float *out, int iters, float a, float b)
{
    float x[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) x[i] = a + threadIdx.x * 1e-3f + i;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                if (OP == 0) x[i] = __builtin_fmaf(x[i], a, b);
                else if (OP == 1) x[i] = x[i] * a;
                else if (OP == 2) x[i] = __builtin_fminf(x[i], x[(i + 1) & 7] + 0.0f);
                else if (OP == 3) x[i] = __builtin_amdgcn_rcpf(x[i]);
                else if (OP == 4) x[i] = __builtin_fmaxf(__builtin_fmaxf(x[i], x[(i + 1) & 7]), x[(i + 2) & 7]);
                else if (OP == 5) x[i] = x[i] > b ? x[(i + 1) & 7] : x[i];
                else if (OP == 6) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(*(double *)&x[i & 6]) : "v"(*(double *)&x[(i + 2) & 6]), "v"(*(double *)&x[(i + 4) & 6]));
            }
    }
    float s = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) s += x[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int OP> int run(const char *name, int per_iter)
{
    float *d; CHECK(hipMalloc(&d, 4 * 256 * 2048));
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    const int iters = 4096, grid = 256 * 8;   // 8 workgroups of 4 waves per CU = 8 waves per SIMD
    k<OP><<<grid, 256>>>(d, 16, 1.0001f, 0.5f);
    CHECK(hipEventRecord(e0));
    k<OP><<<grid, 256>>>(d, iters, 1.0001f, 0.5f);
    CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
    float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
    const double instr_per_simd = (double)iters * per_iter * 8;   // 8 waves per SIMD
    printf("%-28s %8.3f ms  -> %.2f ns per wave-instruction per SIMD (= %.2f cycles at 2.4 GHz)\n", name, ms, ms * 1e6 / instr_per_simd, ms * 1e6 / instr_per_simd * 2.4);
    return 0;
}
int main()
{
    run<0>("v_fma_f32", 32); run<1>("v_mul_f32", 32); run<2>("v_add+v_min_f32", 64); run<3>("v_rcp_f32", 32); run<4>("v_max3_f32", 32);
    run<5>("v_cmp+v_cndmask", 64); run<6>("v_pk_fma_f32", 32);
    return 0;
}
