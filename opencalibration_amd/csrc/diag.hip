// libochip.so — diagnostics: element-wise fp64 primitives exactly as the hot-path kernels compile
// them (-ffp-contract=off), so tests can assert that device division and square root are correctly
// rounded (bit-identical to the host's IEEE results), which the bit-exact RANSAC parity relies on.
#include "ctx.hpp"

#include <vector>

namespace
{
__global__ void fp64_op_kernel(int op, const double *__restrict__ x, const double *__restrict__ y,
                               double *__restrict__ out, size_t n)
{
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n)
        return;
    const double a = x[i], b = y[i];
    double r;
    switch (op)
    {
    case 0:
        r = a / b;
        break;
    case 1:
        r = sqrt(a);
        break;
    case 2:
        r = log(a);
        break;
    case 3:
        r = a * b + a; // must stay an unfused multiply then add
        break;
    default:
        r = a + b;
        break;
    }
    out[i] = r;
}
} // namespace

extern "C" int ochip_debug_fp64(ochip_ctx *ctx, int op, const double *x, const double *y, size_t n, double *out)
{
    if (!ctx || !x || !y || !out)
        return OCHIP_EINVAL;
    if (n == 0)
        return OCHIP_OK;
    OCHIP_HIP(ctx, hipSetDevice(ctx->device));
    double *d = nullptr;
    if (hipMalloc((void **)&d, 3 * n * sizeof(double)) != hipSuccess)
        return ochip_fail(ctx, OCHIP_ENOMEM, "hipMalloc failed");
    hipError_t e = hipMemcpy(d, x, n * 8, hipMemcpyHostToDevice);
    if (e == hipSuccess)
        e = hipMemcpy(d + n, y, n * 8, hipMemcpyHostToDevice);
    if (e == hipSuccess)
    {
        hipLaunchKernelGGL(fp64_op_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream, op, d, d + n,
                           d + 2 * n, n);
        e = hipStreamSynchronize(ctx->stream);
    }
    if (e == hipSuccess)
        e = hipMemcpy(out, d + 2 * n, n * 8, hipMemcpyDeviceToHost);
    (void)hipFree(d);
    if (e != hipSuccess)
        return ochip_fail(ctx, OCHIP_EHIP, "ochip_debug_fp64: %s", hipGetErrorString(e));
    return OCHIP_OK;
}
