// Is the trimmed division sequence of level_strip_kernel (recip_ge1) the compiler's 1.0f / d bit for bit?  Every float in
// [1, 2^40) by steps of 7 bit patterns plus the neighbourhoods of powers of two.  hipcc --offload-arch=gfx950 -O3 -ffp-contract=off
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
__device__ __forceinline__ float recip_ge1(float d)
{
    float r = __builtin_amdgcn_rcpf(d);
    const float e = __builtin_fmaf(-d, r, 1.0f);
    r = __builtin_fmaf(e, r, r);
    float q = r;
    const float e2 = __builtin_fmaf(-d, q, 1.0f);
    q = __builtin_fmaf(e2, r, q);
    const float e3 = __builtin_fmaf(-d, q, 1.0f);
    return __builtin_fmaf(e3, r, q);
}
__global__ void check(uint32_t first, uint32_t step, uint32_t n, unsigned long long *bad, uint32_t *example)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n)
        return;
    const float d = __uint_as_float(first + i * step);
    const float a = 1.0f / d, b = recip_ge1(d);
    if (__float_as_uint(a) != __float_as_uint(b))
    {
        if (atomicAdd(bad, 1ull) == 0)
            *example = first + i * step;
    }
}
int main()
{
    unsigned long long *bad;
    uint32_t *ex;
    hipMalloc(&bad, 8);
    hipMalloc(&ex, 4);
    hipMemset(bad, 0, 8);
    hipMemset(ex, 0, 4);
    const uint32_t lo = 0x3f800000u, hi = 0x53800000u; // 1 .. 2^40
    const uint32_t step = 3, n = (hi - lo) / step;
    hipLaunchKernelGGL(check, dim3((n + 255) / 256), dim3(256), 0, 0, lo, step, n, bad, ex);
    hipLaunchKernelGGL(check, dim3((n + 255) / 256), dim3(256), 0, 0, lo + 1, step, n, bad, ex);
    hipLaunchKernelGGL(check, dim3((n + 255) / 256), dim3(256), 0, 0, lo + 2, step, n, bad, ex);
    unsigned long long hb = 0;
    uint32_t he = 0;
    hipMemcpy(&hb, bad, 8, hipMemcpyDeviceToHost);
    hipMemcpy(&he, ex, 4, hipMemcpyDeviceToHost);
    printf("checked %llu floats in [1, 2^40): %llu differ (first example bits 0x%08x)\n", 3ull * n, hb, he);
    return hb != 0;
}
