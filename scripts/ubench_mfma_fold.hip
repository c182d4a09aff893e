// How many independent accumulator chains the block-scaled FP4 MFMA (32x32x64) needs to run at its issue rate, and how many
// top-2 fold instructions (v_med3_f32 + v_max_i32 on OTHER registers) hide in its gaps: the numbers the matcher's loop is
// built on.  One wave per SIMD on every CU; prints shader cycles per MFMA.
//   hipcc --offload-arch=gfx950 -O3 -mllvm -amdgpu-mfma-vgpr-form -o /tmp/ubench_mfma_fold scripts/ubench_mfma_fold.hip && /tmp/ubench_mfma_fold
#include <hip/hip_runtime.h>
#include <cstdio>
typedef int v8i __attribute__((ext_vector_type(8)));
typedef float v16f __attribute__((ext_vector_type(16)));

template <int CHAINS, int FILL> __global__ __launch_bounds__(256) void k(float *out, unsigned long long *cyc, int iters)
{
    v8i a = {0x22222222, 0x22222222, 0x22222222, 0x22222222, 0, 0, 0, 0}, b = a;
    // two sets of CHAINS accumulators: one is computed (8 matrix instructions per chain) while the other, finished in the phase
    // before, is folded - FILL (v_med3_f32, v_max_i32) pairs behind every matrix instruction, a (best, second) pair per chain
    v16f c[2][CHAINS] = {};
    float best[CHAINS], second[CHAINS];
    for (int j = 0; j < CHAINS; j++)
        best[j] = second[j] = -1.0f;
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int i = 0; i < iters; i++)
    {
#pragma unroll
        for (int set = 0; set < 2; set++)
#pragma unroll
            for (int step = 0; step < 8; step++)
#pragma unroll
                for (int j = 0; j < CHAINS; j++)
                {
                    c[set][j] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c[set][j], 4, 4, 0, 127, 0, 127);
#pragma unroll
                    for (int f = 0; f < FILL; f++)
                    {
                        const float v = c[set ^ 1][j][(step * FILL + f) & 15];
                        second[j] = __builtin_amdgcn_fmed3f(best[j], second[j], v);
                        best[j] = __int_as_float(max(__float_as_int(best[j]), __float_as_int(v)));
                    }
                }
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    float s = 0;
    for (int j = 0; j < CHAINS; j++)
        s += c[0][j][0] + c[1][j][0] + best[j] + second[j];
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0)
        *cyc = t1 - t0;
}

template <int CHAINS, int FILL> void run(float *out, unsigned long long *cyc)
{
    const int iters = 2000;
    unsigned long long h;
    for (int rep = 0; rep < 2; rep++)
        hipLaunchKernelGGL((k<CHAINS, FILL>), dim3(256), dim3(256), 0, 0, out, cyc, iters);
    hipDeviceSynchronize();
    hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
    printf("%d chain(s), %d fold pairs (%d vector instructions) per MFMA: %.1f cycles per MFMA\n", CHAINS, FILL, 2 * FILL, (double)h / ((double)CHAINS * 16 * iters));
}

int main()
{
    float *out;
    unsigned long long *cyc;
    hipMalloc(&out, 256 * 256 * 4);
    hipMalloc(&cyc, 8);
    run<1, 0>(out, cyc);
    run<2, 0>(out, cyc);
    run<4, 0>(out, cyc);
    run<2, 1>(out, cyc);
    run<2, 2>(out, cyc);
    run<2, 3>(out, cyc);
    run<4, 1>(out, cyc);
    run<4, 2>(out, cyc);
    return 0;
}
