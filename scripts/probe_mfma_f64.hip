// Empirical operand/result layout of v_mfma_f64_16x16x4_f64 on gfx950.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
typedef double v4f64 __attribute__((ext_vector_type(4)));
__global__ void k(const double *a, const double *b, double *d)
{
    const int l = threadIdx.x;
    v4f64 acc = {0, 0, 0, 0};
    acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a[l], b[l], acc, 0, 0, 0);
    for (int e = 0; e < 4; e++)
        d[l * 4 + e] = acc[e];
}
int main()
{
    double ha[64], hb[64], hd[256], *a, *b, *d;
    for (int l = 0; l < 64; l++)
    {
        ha[l] = 1.0 + l * 0.37 + (l % 7) * 0.011; // candidate: A[l%16][l/16]
        hb[l] = 2.0 + l * 0.53 + (l % 5) * 0.007; // candidate: B[l/16][l%16]
    }
    hipMalloc(&a, 512); hipMalloc(&b, 512); hipMalloc(&d, 2048);
    hipMemcpy(a, ha, 512, hipMemcpyHostToDevice); hipMemcpy(b, hb, 512, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, a, b, d);
    hipMemcpy(hd, d, 2048, hipMemcpyDeviceToHost);
    // hypothesis H1: A[i][k] = ha[16k+i], B[k][j] = hb[16k+j]
    double ref[16][16];
    for (int i = 0; i < 16; i++) for (int j = 0; j < 16; j++) { double s = 0; for (int kk = 0; kk < 4; kk++) s += ha[16*kk+i]*hb[16*kk+j]; ref[i][j] = s; }
    int ok = 0;
    for (int l = 0; l < 64; l++) for (int e = 0; e < 4; e++)
    {
        int fi = -1, fj = -1;
        for (int i = 0; i < 16; i++) for (int j = 0; j < 16; j++) if (fabs(ref[i][j] - hd[l*4+e]) < 1e-9 * fabs(ref[i][j])) { fi = i; fj = j; }
        if (l < 20 || l % 16 == 0) printf("lane %2d reg %d -> D[%d][%d]\n", l, e, fi, fj);
        ok += fi >= 0;
    }
    printf("matched %d of 256 under H1\n", ok);
    return 0;
}
