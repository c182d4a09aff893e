// Micro-benchmark: sustained issue rate of the integer VALU ops the Hamming kernel is made of.
// Build+run on the GPU box: hipcc --offload-arch=gfx950 -O3 scripts/ubench_valu.hip -o /tmp/ub && /tmp/ub
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>

#define REP16(X) X X X X X X X X X X X X X X X X

template <int OP> __global__ __launch_bounds__(256) void k(uint32_t *out, uint32_t seed, int iters)
{
    uint32_t a0 = threadIdx.x + seed, a1 = a0 * 3, a2 = a0 * 5, a3 = a0 * 7, a4 = a0 * 11, a5 = a0 * 13, a6 = a0 * 17,
             a7 = a0 * 19;
    uint32_t x = seed * 2654435761u + threadIdx.x;
    for (int i = 0; i < iters; i++)
    {
        if (OP == 0)
        {
            REP16(asm volatile("v_bcnt_u32_b32 %0, %8, %0\n v_bcnt_u32_b32 %1, %8, %1\n v_bcnt_u32_b32 %2, %8, %2\n"
                               "v_bcnt_u32_b32 %3, %8, %3\n v_bcnt_u32_b32 %4, %8, %4\n v_bcnt_u32_b32 %5, %8, %5\n"
                               "v_bcnt_u32_b32 %6, %8, %6\n v_bcnt_u32_b32 %7, %8, %7\n"
                               : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)
                               : "v"(x));)
        }
        else if (OP == 1)
        {
            REP16(asm volatile("v_xor_b32 %0, %8, %0\n v_xor_b32 %1, %8, %1\n v_xor_b32 %2, %8, %2\n"
                               "v_xor_b32 %3, %8, %3\n v_xor_b32 %4, %8, %4\n v_xor_b32 %5, %8, %5\n"
                               "v_xor_b32 %6, %8, %6\n v_xor_b32 %7, %8, %7\n"
                               : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)
                               : "v"(x));)
        }
        else if (OP == 2)
        {
            REP16(asm volatile("v_med3_u32 %0, %8, %0, %1\n v_med3_u32 %1, %8, %1, %2\n v_med3_u32 %2, %8, %2, %3\n"
                               "v_med3_u32 %3, %8, %3, %4\n v_med3_u32 %4, %8, %4, %5\n v_med3_u32 %5, %8, %5, %6\n"
                               "v_med3_u32 %6, %8, %6, %7\n v_med3_u32 %7, %8, %7, %0\n"
                               : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)
                               : "v"(x));)
        }
        else if (OP == 3) // xor with an SGPR operand + bcnt: the kernel's actual pair
        {
            uint32_t s = __builtin_amdgcn_readfirstlane(x + i);
            REP16(asm volatile("v_xor_b32 %0, %8, %0\n v_bcnt_u32_b32 %1, %0, %1\n v_xor_b32 %2, %8, %2\n"
                               "v_bcnt_u32_b32 %3, %2, %3\n v_xor_b32 %4, %8, %4\n v_bcnt_u32_b32 %5, %4, %5\n"
                               "v_xor_b32 %6, %8, %6\n v_bcnt_u32_b32 %7, %6, %7\n"
                               : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)
                               : "s"(s));)
        }
        else if (OP == 4) // bfi (carry of a carry-save adder)
        {
            REP16(asm volatile("v_bfi_b32 %0, %8, %0, %1\n v_bfi_b32 %1, %8, %1, %2\n v_bfi_b32 %2, %8, %2, %3\n"
                               "v_bfi_b32 %3, %8, %3, %4\n v_bfi_b32 %4, %8, %4, %5\n v_bfi_b32 %5, %8, %5, %6\n"
                               "v_bfi_b32 %6, %8, %6, %7\n v_bfi_b32 %7, %8, %7, %0\n"
                               : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)
                               : "v"(x));)
        }
        else if (OP == 5) // 64-bit-wide lane ops do not exist for bcnt; try v_mbcnt as a cheaper popcount? (mask count)
        {
            REP16(asm volatile("v_mbcnt_lo_u32_b32 %0, %8, %0\n v_mbcnt_lo_u32_b32 %1, %8, %1\n"
                               "v_mbcnt_lo_u32_b32 %2, %8, %2\n v_mbcnt_lo_u32_b32 %3, %8, %3\n"
                               "v_mbcnt_lo_u32_b32 %4, %8, %4\n v_mbcnt_lo_u32_b32 %5, %8, %5\n"
                               "v_mbcnt_lo_u32_b32 %6, %8, %6\n v_mbcnt_lo_u32_b32 %7, %8, %7\n"
                               : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)
                               : "v"(x));)
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7;
}

template <int OP> void run(const char *name, int waves_per_simd)
{
    const int blocks = 256 * waves_per_simd; // 256-thread blocks = 4 waves = 1 per SIMD
    uint32_t *out;
    hipMalloc(&out, (size_t)blocks * 256 * 4);
    const int iters = 2000;
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, out, 1u, 10);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, out, 1u, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    const double instr_per_wave = (double)iters * 16 * 8;
    const double waves = blocks * 4.0;
    const double per_simd = instr_per_wave * waves / (256.0 * 4.0);
    printf("%-28s waves/SIMD %d: %.3f ms  -> %.2f ns per wave-instr per SIMD (2 cyc @2.4GHz = 0.83 ns)  %.1f Tlane-op/s\n",
           name, waves_per_simd, ms, ms * 1e6 / per_simd, instr_per_wave * waves * 64 / ms / 1e9);
    hipFree(out);
}

int main()
{
    for (int w : {1, 2, 4, 8})
    {
        run<0>("v_bcnt_u32_b32", w);
        run<1>("v_xor_b32", w);
        run<2>("v_med3_u32", w);
        run<3>("v_xor(sgpr)+v_bcnt", w);
        run<4>("v_bfi_b32", w);
        run<5>("v_mbcnt_lo", w);
    }
    return 0;
}
