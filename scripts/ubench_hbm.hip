// Micro-benchmark: what streaming kernels reach on this part for the read : write mixes of the extract stage's passes
// (float4 per thread, grid-stride, 1 GiB per array, best of 5).  Build + run on the GPU box:
//   hipcc --offload-arch=gfx950 -O3 scripts/ubench_hbm.hip -o /tmp/ubh && /tmp/ubh
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>

__global__ __launch_bounds__(256) void k_read(const float4 *a, size_t n, float *sink)
{
    float s = 0;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256)
    {
        const float4 v = a[i];
        s += v.x + v.y + v.z + v.w;
    }
    if (s == 12345.678f)
        *sink = s;
}
__global__ __launch_bounds__(256) void k_write(float4 *a, size_t n)
{
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256)
        a[i] = make_float4(1.f, 2.f, 3.f, (float)i);
}
__global__ __launch_bounds__(256) void k_copy(const float4 *a, float4 *b, size_t n)
{
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256)
        b[i] = a[i];
}
__global__ __launch_bounds__(256) void k_2r1w(const float4 *a, const float4 *b, float4 *c, size_t n)
{
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256)
    {
        const float4 x = a[i], y = b[i];
        c[i] = make_float4(x.x + y.x, x.y + y.y, x.z + y.z, x.w + y.w);
    }
}
__global__ __launch_bounds__(256) void k_1r3w(const float4 *a, float4 *b, float4 *c, float4 *d, size_t n)
{
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256)
    {
        const float4 x = a[i];
        b[i] = x;
        c[i] = make_float4(x.y, x.x, x.w, x.z);
        d[i] = make_float4(x.w, x.z, x.y, x.x);
    }
}

int main()
{
    const size_t bytes = 1ull << 30, n = bytes / 16;
    float4 *a, *b, *c, *d;
    float *sink;
    hipMalloc(&a, bytes), hipMalloc(&b, bytes), hipMalloc(&c, bytes), hipMalloc(&d, bytes), hipMalloc(&sink, 4);
    hipMemset(a, 1, bytes), hipMemset(b, 1, bytes);
    hipEvent_t e0, e1;
    hipEventCreate(&e0), hipEventCreate(&e1);
    auto timeit = [&](const char *name, double traffic_bytes, auto launch) {
        float best = 1e30f;
        for (int rep = 0; rep < 6; rep++)
        {
            hipEventRecord(e0);
            launch();
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            float ms;
            hipEventElapsedTime(&ms, e0, e1);
            if (rep)
                best = ms < best ? ms : best;
        }
        printf("%-28s %7.3f ms  %6.2f TB/s\n", name, best, traffic_bytes / best / 1e9);
    };
    for (int grid : {256 * 8, 256 * 32, 256 * 128})
    {
        printf("grid %d workgroups of 256\n", grid);
        timeit("read only", bytes, [&] { hipLaunchKernelGGL(k_read, dim3(grid), dim3(256), 0, 0, a, n, sink); });
        timeit("write only", bytes, [&] { hipLaunchKernelGGL(k_write, dim3(grid), dim3(256), 0, 0, c, n); });
        timeit("copy (1 read : 1 write)", 2.0 * bytes, [&] { hipLaunchKernelGGL(k_copy, dim3(grid), dim3(256), 0, 0, a, c, n); });
        timeit("2 reads : 1 write (nld)", 3.0 * bytes, [&] { hipLaunchKernelGGL(k_2r1w, dim3(grid), dim3(256), 0, 0, a, b, c, n); });
        timeit("1 read : 3 writes (blur)", 4.0 * bytes, [&] { hipLaunchKernelGGL(k_1r3w, dim3(grid), dim3(256), 0, 0, a, b, c, d, n); });
    }
    return 0;
}
