// Issue rate of the block-scaled MFMAs with FP4 / FP8 operands on gfx950: N back-to-back instructions on independent
// accumulators per wave, one wave per SIMD on every CU; prints shader cycles per instruction.
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/ubench_mfma_fp4 scripts/ubench_mfma_fp4.hip && /tmp/ubench_mfma_fp4
#include <hip/hip_runtime.h>
#include <cstdio>
typedef int v8i __attribute__((ext_vector_type(8)));
typedef float v16f __attribute__((ext_vector_type(16)));
typedef float v4f __attribute__((ext_vector_type(4)));

template <int MODE> __global__ __launch_bounds__(256) void k(float *out, unsigned long long *cyc, int iters)
{
    v8i a = {0x22222222, 0x22222222, 0x22222222, 0x22222222, 0, 0, 0, 0}, b = a;
    if (MODE == 1 || MODE == 3) // fp8 operands use all eight registers
        a = b = v8i{0x38383838, 0x38383838, 0x38383838, 0x38383838, 0x38383838, 0x38383838, 0x38383838, 0x38383838};
    v16f c[4] = {};
    v4f d[4] = {};
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int i = 0; i < iters; i++)
    {
#pragma unroll
        for (int j = 0; j < 4; j++)
        {
            if (MODE == 0)
                c[j] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c[j], 4, 4, 0, 127, 0, 127);
            else if (MODE == 1)
                c[j] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c[j], 0, 0, 0, 127, 0, 127);
            else if (MODE == 2)
                d[j] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a, b, d[j], 4, 4, 0, 127, 0, 127);
            else
                d[j] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a, b, d[j], 0, 0, 0, 127, 0, 127);
        }
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    float s = 0;
    for (int j = 0; j < 4; j++)
        s += c[j][0] + d[j][0];
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0)
        *cyc = t1 - t0;
}

int main()
{
    float *out;
    unsigned long long *cyc, h;
    hipMalloc(&out, 256 * 256 * 4);
    hipMalloc(&cyc, 8);
    const int iters = 20000;
    const char *names[4] = {"32x32x64 fp4", "32x32x64 fp8", "16x16x128 fp4", "16x16x128 fp8"};
    for (int m = 0; m < 4; m++)
    {
        hipEvent_t e0, e1;
        hipEventCreate(&e0);
        hipEventCreate(&e1);
        for (int rep = 0; rep < 2; rep++)
        {
            hipEventRecord(e0);
            if (m == 0) hipLaunchKernelGGL(k<0>, dim3(256), dim3(256), 0, 0, out, cyc, iters);
            if (m == 1) hipLaunchKernelGGL(k<1>, dim3(256), dim3(256), 0, 0, out, cyc, iters);
            if (m == 2) hipLaunchKernelGGL(k<2>, dim3(256), dim3(256), 0, 0, out, cyc, iters);
            if (m == 3) hipLaunchKernelGGL(k<3>, dim3(256), dim3(256), 0, 0, out, cyc, iters);
            hipEventRecord(e1);
            hipEventSynchronize(e1);
        }
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
        const double n = 4.0 * iters;
        const double flop = (m < 2 ? 2.0 * 32 * 32 * 64 : 2.0 * 16 * 16 * 128) * n * 1024;
        printf("%-14s %.1f shader cycles per instruction (one wave per SIMD), %.3f ms, %.0f TFLOP/s\n", names[m], (double)h / n, ms, flop / ms / 1e9);
    }
    return 0;
}
