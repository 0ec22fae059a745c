// scripts/probe_numerics.hip -- what v_cvt_pk_u8_f32 does with fractions, negatives and overflow (run on the GPU box once;
// the answer decides whether raster_fast_kernel may pack its bytes with it).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
__global__ void k(const float *a, unsigned *o, int n)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) o[i] = __builtin_amdgcn_cvt_pk_u8_f32(a[i], 0, 0u);
}
int main()
{
    const float xs[] = {-3.f, -0.6f, -0.5f, -0.4f, 0.f, 0.4f, 0.5f, 0.6f, 1.49f, 1.5f, 1.51f, 2.5f, 3.5f, 126.5f, 127.5f, 254.4f, 254.5f, 254.6f, 255.f, 255.4f, 255.5f, 256.f, 300.f, 1e9f, NAN, INFINITY};
    const int n = sizeof(xs) / sizeof(xs[0]);
    float *d; unsigned *o; unsigned h[64];
    hipMalloc(&d, sizeof xs); hipMalloc(&o, n * 4);
    hipMemcpy(d, xs, sizeof xs, hipMemcpyHostToDevice);
    k<<<1, 64>>>(d, o, n);
    hipMemcpy(h, o, n * 4, hipMemcpyDeviceToHost);
    for (int i = 0; i < n; ++i) printf("cvt_pk_u8_f32(%g) = %u\n", xs[i], h[i]);
    return 0;
}
