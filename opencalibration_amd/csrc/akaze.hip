// libochip.so — extract: AKAZE features (nonlinear FED scale space, Hessian-determinant extrema, 486-bit
// 3-channel M-LDB) for a batch of equally sized images (gfx950).
//
// Replaces cv::AKAZE::detectAndCompute behind src/extract/extract_features.cpp:35-36 together with the
// grey conversion and INTER_AREA downscale in front of it (:25-27).  AKAZE lives in OpenCV (absent from
// this image): the algorithm is restated from its publication (Alcantarilla, Nuevo, Bartoli, BMVC 2013)
// in the structure of OpenCV's implementation; parity with OpenCV itself is unpinned and the
// kernels are checked against the CPU restatement, with which they share every table (Gaussian taps,
// FED step sizes, orientation weights are computed on the host in double) and every float32 operation
// order (no FMA, no device transcendentals), so level images, keypoints and descriptors are bit-identical.
//
// All planes are [image][y][x] float and one launch covers the whole batch (blockIdx.z = image), ~170 launches
// per batch.  The stencil passes are LDS-tiled and fused with their consumers (Gaussian + conductivity +
// derivatives; determinant + maxima), the diffusion steps run up to four at a time out of registers, the
// candidate list is built without global atomics in a space-filling tile order, and a wavefront per
// candidate does suppression and description.  DESIGN.md section 4.4 has the per-kernel numbers.
#include "ctx.hpp"

#include <algorithm>
#include <cmath>
#include <limits>
#include <cstring>
#include <type_traits>
#include <vector>

namespace
{

struct level_info
{
    int octave, w, h, sigma_size;
    float esigma;
    size_t off; // plane offset (floats) inside one image's pyramid
    int tile_off, tiles_x; // first id and row length of the level's 64 x 24 detection tiles
    int mask_off;          // first word of the level's maxima bit mask (one 64-bit word per row of a tile column)
    int sup_off, sup_tx;   // the suppression's masks: first word and words per row of the level's 8 x 8-pixel mask words
};

__device__ __forceinline__ int clampi(int v, int lo, int hi)
{
    return v < lo ? lo : (v > hi ? hi : v);
}
// BORDER_REFLECT_101 for an index at most one image size outside [0, n): no loop, no branch
__device__ __forceinline__ int reflect101_once(int v, int n)
{
    v = v < 0 ? -v : v;
    return v >= n ? 2 * (n - 1) - v : v;
}
__device__ __forceinline__ int reflect101(int v, int n)
{
    if (n == 1)
        return 0;
    while (v < 0 || v >= n)
        v = v < 0 ? -v : 2 * (n - 1) - v;
    return v;
}

// ---- grey + INTER_AREA (extract_features.cpp:25-27)
__global__ void gray_kernel(const uint8_t *__restrict__ bgr, uint8_t *__restrict__ gray, size_t n)
{
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n)
        gray[i] = (uint8_t)((bgr[3 * i] * 1868 + bgr[3 * i + 1] * 9617 + bgr[3 * i + 2] * 4899 + (1 << 13)) >> 14);
}

constexpr int RESIZE_ROWS = 8, RESIZE_MAX_TAPS = 8;
__global__ __launch_bounds__(256) void resize_area_kernel(const uint8_t *__restrict__ src, int sw, int sh, float *__restrict__ dst,
                                                          int dw, int dh, const int *__restrict__ xoff,
                                                          const int *__restrict__ xsi, const float *__restrict__ xal,
                                                          const int *__restrict__ yoff, const int *__restrict__ ysi,
                                                          const float *__restrict__ yal, int half_up_cols)
{
    // a thread keeps the horizontal taps of its destination column in registers and walks RESIZE_ROWS rows; the
    // vertical taps are uniform per row.  Sums run in table order exactly like the per-pixel form.
    const int x = blockIdx.x * blockDim.x + threadIdx.x;
    if (x >= dw)
        return;
    const uint8_t *s = src + (size_t)blockIdx.z * sw * sh;
    const int k0 = xoff[x], nk = xoff[x + 1] - k0;
    int si[RESIZE_MAX_TAPS];
    float al[RESIZE_MAX_TAPS];
#pragma unroll
    for (int k = 0; k < RESIZE_MAX_TAPS; k++)
    {
        si[k] = k < nk ? xsi[k0 + k] : 0;
        al[k] = k < nk ? xal[k0 + k] : 0.0f;
    }
    for (int r = 0; r < RESIZE_ROWS; r++)
    {
        const int y = blockIdx.y * RESIZE_ROWS + r;
        if (y >= dh)
            break;
        float acc = 0.0f;
        for (int e = yoff[y]; e < yoff[y + 1]; e++)
        {
            const uint8_t *row = s + (size_t)ysi[e] * sw;
            float rr = 0.0f;
            if (nk <= RESIZE_MAX_TAPS)
            {
#pragma unroll
                for (int k = 0; k < RESIZE_MAX_TAPS; k++)
                    if (k < nk)
                        rr += (float)row[si[k]] * al[k];
            }
            else // very large decimation factors: taps straight from the table
                for (int k = k0; k < k0 + nk; k++)
                    rr += (float)row[xsi[k]] * xal[k];
            acc += rr * yal[e];
        }
        // (half_up_cols: a scale of exactly 2 is cv::resize's integer path, whose vector body rounds (a + b + c + d + 2) >> 2 - acc
        // is exact there - and whose scalar tail, the last dw % 16 columns at most, rounds to even like the general path)
        const float v = x < half_up_cols ? floorf(acc + 0.5f) : rintf(acc);
        // saturate to the 8-bit working image, then the 1/255 float conversion in front of AKAZE
        dst[(size_t)blockIdx.z * dw * dh + (size_t)y * dw + x] = (float)(uint8_t)fminf(255.0f, fmaxf(0.0f, v)) * (1.0f / 255.0f);
    }
}

// The same resize with the source window of a workgroup (256 destination columns x RESIZE_ROWS rows) staged in LDS by
// dword loads: the per-tap single-byte global loads of the kernel above (about a dozen per destination pixel) were
// what it spent its time on.  Needs a source width that is a multiple of 4 and a window that fits the tile; the host
// checks both against the tables and falls back to the kernel above otherwise.  Same taps, same order, same sums.
// FROM_BGR: the staging converts the BGR source to grey on the way in (cvtColor's fixed-point weights, four pixels from
// three dwords as in gray4_kernel), so the full-resolution grey image is neither written nor read.
constexpr int RS_PITCH = 704, RS_ROWS = 32; // bytes per staged row, staged rows (2.5 : 1 needs 648 x 21)
// (round 5: the kernel was issue-bound.  MAXT: the host knows the longest run of horizontal taps - 4 at the 2.5 : 1 of a
// 4000-pixel view - and the tap loop is unrolled to it instead of to 8 predicated trips; the grey conversion takes a pixel's
// three bytes as one dword - the fourth byte meets a zero weight - through two v_dot4_u32_u8 with the 14-bit weights split
// into bytes: 1868 = 7 * 256 + 76, 9617 = 37 * 256 + 145, 4899 = 19 * 256 + 35; integers, the same value.)
template <bool FROM_BGR, int MAXT>
__global__ __launch_bounds__(256) void resize_area_lds_kernel(const uint8_t *__restrict__ src, int sw, int sh, float *__restrict__ dst,
                                                              int dw, int dh, const int *__restrict__ xoff,
                                                              const int *__restrict__ xsi, const float *__restrict__ xal,
                                                              const int *__restrict__ yoff, const int *__restrict__ ysi,
                                                              const float *__restrict__ yal, int half_up_cols)
{
    __shared__ unsigned int tile[RS_ROWS * RS_PITCH / 4];
    const int x0 = blockIdx.x * 256, y0 = blockIdx.y * RESIZE_ROWS;
    const int x1 = min(x0 + 256, dw), y1 = min(y0 + RESIZE_ROWS, dh);
    const int cx0 = xsi[xoff[x0]] & ~3, cx1 = xsi[xoff[x1] - 1]; // source columns cx0..cx1
    const int ry0 = ysi[yoff[y0]], ry1 = ysi[yoff[y1] - 1];      // source rows ry0..ry1
    const int nd = (cx1 - cx0) / 4 + 1, nr = ry1 - ry0 + 1;
    const uint8_t *s = src + (size_t)blockIdx.z * sw * sh * (FROM_BGR ? 3 : 1);
    // the window's loads go out eight trips at a time before the first of them is converted (one trip per loop turn waited
    // out a memory round trip each: fourteen in a row per thread at 2.5 : 1, which is what bounded this kernel)
    constexpr int STAGE = 8;
    // (idx -> (row, dword) of the window: by a multiply with ceil(2^32 / nd), exact for idx < 2^32 / nd - the window has a few
    // thousand dwords.  Written as idx / nd the compiler's 32-bit division, twice per dword, was half of this kernel's vector
    // instructions: 27 per source pixel, which is what kept it at 3.7 TB/s)
    const unsigned int nd_magic = nd > 1 ? 0xFFFFFFFFu / (unsigned int)nd + 1u : 0u;
    auto row_of = [&](int idx) { return nd > 1 ? (int)__umulhi((unsigned int)idx, nd_magic) : idx; };
    for (int base = threadIdx.x; base < nd * nr; base += 256 * STAGE)
    {
        uint32_t w0[STAGE], w1[STAGE], w2[STAGE];
#pragma unroll
        for (int u = 0; u < STAGE; u++)
        {
            const int idx = base + 256 * u;
            const int row = row_of(idx), d = idx - row * nd;
            w0[u] = w1[u] = w2[u] = 0;
            if (idx < nd * nr)
            {
                if (FROM_BGR)
                {
                    const uint32_t *q = reinterpret_cast<const uint32_t *>(s + 3 * ((size_t)(ry0 + row) * sw + cx0 + 4 * d));
                    w0[u] = q[0], w1[u] = q[1], w2[u] = q[2];
                }
                else
                    w0[u] = *reinterpret_cast<const unsigned int *>(s + (size_t)(ry0 + row) * sw + cx0 + 4 * d);
            }
        }
#pragma unroll
        for (int u = 0; u < STAGE; u++)
        {
            const int idx = base + 256 * u;
            const int row = row_of(idx), d = idx - row * nd;
            if (idx >= nd * nr)
                continue;
            if (FROM_BGR)
            {
                auto g = [](uint32_t bgrx) { // b 1868 + g 9617 + r 4899 + 2^13 >> 14 of the low three bytes
                    const uint32_t hi = __builtin_amdgcn_udot4(bgrx, 0x00132507u, 0u, false);
                    return __builtin_amdgcn_udot4(bgrx, 0x0023914Cu, (hi << 8) + (1u << 13), false) >> 14;
                };
                const uint32_t p0 = g(w0[u]), p1 = g(__builtin_amdgcn_alignbit(w1[u], w0[u], 24)),
                               p2 = g(__builtin_amdgcn_alignbit(w2[u], w1[u], 16)), p3 = g(w2[u] >> 8);
                tile[row * (RS_PITCH / 4) + d] = p0 | (p1 << 8) | (p2 << 16) | (p3 << 24);
            }
            else
                tile[row * (RS_PITCH / 4) + d] = w0[u];
        }
    }
    __syncthreads();
    const int x = x0 + threadIdx.x;
    if (x >= dw)
        return;
    const uint8_t *t8 = reinterpret_cast<const uint8_t *>(tile);
    const int k0 = xoff[x], nk = xoff[x + 1] - k0;
    int si[MAXT];
    float al[MAXT];
#pragma unroll
    for (int k = 0; k < MAXT; k++)
    {
        si[k] = k < nk ? xsi[k0 + k] - cx0 : 0;
        al[k] = k < nk ? xal[k0 + k] : 0.0f;
    }
    for (int y = y0; y < y1; y++)
    {
        float acc = 0.0f;
        for (int e = yoff[y]; e < yoff[y + 1]; e++)
        {
            const uint8_t *row = t8 + (ysi[e] - ry0) * RS_PITCH;
            float rr = 0.0f;
#pragma unroll
            for (int k = 0; k < MAXT; k++)
                if (k < nk)
                    rr += (float)row[si[k]] * al[k];
            acc += rr * yal[e];
        }
        const float v = x < half_up_cols ? floorf(acc + 0.5f) : rintf(acc); // (see resize_area_kernel)
        dst[(size_t)blockIdx.z * dw * dh + (size_t)y * dw + x] = (float)(uint8_t)fminf(255.0f, fmaxf(0.0f, v)) * (1.0f / 255.0f);
    }
}

// four pixels per thread: three 32-bit loads carry 4 BGR triplets, one 32-bit store carries 4 grey bytes
__global__ void gray4_kernel(const uint32_t *__restrict__ bgr, uint32_t *__restrict__ gray, size_t n4)
{
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n4)
        return;
    const uint32_t w0 = bgr[3 * i], w1 = bgr[3 * i + 1], w2 = bgr[3 * i + 2];
    auto g = [](uint32_t b, uint32_t gg, uint32_t r) { return (b * 1868u + gg * 9617u + r * 4899u + (1u << 13)) >> 14; };
    const uint32_t p0 = g(w0 & 255u, (w0 >> 8) & 255u, (w0 >> 16) & 255u);
    const uint32_t p1 = g(w0 >> 24, w1 & 255u, (w1 >> 8) & 255u);
    const uint32_t p2 = g((w1 >> 16) & 255u, w1 >> 24, w2 & 255u);
    const uint32_t p3 = g((w2 >> 8) & 255u, (w2 >> 16) & 255u, w2 >> 24);
    gray[i] = p0 | (p1 << 8) | (p2 << 16) | (p3 << 24);
}

__global__ void to_float_kernel(const uint8_t *__restrict__ g, float *__restrict__ out, size_t n)
{
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n)
        out[i] = (float)g[i] * (1.0f / 255.0f);
}

// ---- separable Gaussian, replicate border, ascending tap order
constexpr int MAX_TAPS = 33;
struct taps_t
{
    int n;
    float k[MAX_TAPS];
};

// ---- fused separable Gaussian + the 3x3-pattern stencil that consumes it.  A 256-thread workgroup owns a
// 64 x 32 output tile: the input tile (+ halo) is staged in LDS once, the row pass and the column pass run
// LDS -> LDS, and the consumer (Scharr flow / gradient magnitude / scale-s derivatives) reads the blurred tile
// from LDS, so one launch moves 4 B/pixel in and 4-8 B/pixel out instead of the 16 + 8..12 of separate passes.
// Every value is the same float expression as the separate passes (ascending tap order, clamped source
// coordinates for the blur, reflected coordinates for the stencil), so the fusion does not change a bit.
constexpr int BT_X = 64, BT_Y = 32; // blur tiles
// cv::solve(A, b, dst, DECOMP_LU) for AKAZE's 2 x 2 system [Dxx Dxy; Dxy Dyy] d = -[Dx Dy] (core/src/lapack.cpp, the 2 x 2 fast
// path for CV_32FC1): Cramer's rule in double; a singular system leaves the offset at (0, 0) - the restatement's subpixel_solve
__device__ __forceinline__ float2 subpixel_solve(float Dxx, float Dxy, float Dyy, float Dx, float Dy)
{
    const float b0 = -Dx, b1 = -Dy;
    double d = (double)Dxx * Dyy - (double)Dxy * Dxy;
    float2 r = make_float2(0.0f, 0.0f);
    if (d != 0.)
    {
        d = 1. / d;
        r.x = (float)(((double)b0 * Dyy - (double)b1 * Dxy) * d);
        r.y = (float)(((double)b1 * Dxx - (double)b0 * Dxy) * d);
    }
    return r;
}

constexpr int DT_Y = 24;             // detection tiles are 64 x 24 (16, 24, 32 rows measured: 43, 39, 40 us per image for the determinant kernels)
enum
{
    BLUR_PLAIN = 0,
    BLUR_FLOW = 1,
    BLUR_MODG = 2,
    BLUR_DERIV = 3,
    BLUR_FLOW_DERIV = 4 // a level's Lsmooth feeds both its conductivity and its detector derivatives: one pass
};

// Tile kernels are launched as a 1-D grid of 8 * ceil(tiles / 8) workgroups per image.  Workgroups are dealt to the 8
// XCDs round-robin (id % 8), each XCD with its own L2; this map hands every XCD one contiguous eighth of the image's
// tiles in row-major order, so the halos neighbouring tiles share are read through the same L2 instead of being
// fetched from HBM once per XCD.  Returns false for the padding workgroups.
__device__ __forceinline__ bool xcd_tile(int tiles_x, int tiles_y, int *tx, int *ty)
{
    const int total = tiles_x * tiles_y, chunk = (total + 7) / 8;
    const int t = (int)(blockIdx.x & 7) * chunk + (int)(blockIdx.x >> 3);
    if ((int)(blockIdx.x >> 3) >= chunk || t >= total)
        return false;
    *ty = t / tiles_x;
    *tx = t - *ty * tiles_x;
    return true;
}

// 3-row pattern (wa * a + wb * b) + wa * c of Scharr-type derivatives, samples fetched through `at(x, y)`
template <typename F>
__device__ __forceinline__ void pattern_xy(F at, int xm, int x, int xp, int ym, int y, int yp, float wa, float wb, float *dx,
                                           float *dy)
{
    const float a = at(xp, ym) - at(xm, ym);
    const float b = at(xp, y) - at(xm, y);
    const float c = at(xp, yp) - at(xm, yp);
    *dx = (wa * a + wb * b) + wa * c;
    const float d = at(xm, yp) - at(xm, ym);
    const float e = at(x, yp) - at(x, ym);
    const float f = at(xp, yp) - at(xp, ym);
    *dy = (wa * d + wb * e) + wa * f;
}

// the same pattern on an LDS tile whose row stride and sample distance are compile-time constants: every access is
// the centre address plus an immediate offset (tiles away from the image border need no reflection)
template <int D, int STRIDE>
__device__ __forceinline__ void pattern_lds(const float *c, float wa, float wb, float *dx, float *dy)
{
    const float a = c[-D * STRIDE + D] - c[-D * STRIDE - D];
    const float b = c[D] - c[-D];
    const float cc = c[D * STRIDE + D] - c[D * STRIDE - D];
    *dx = (wa * a + wb * b) + wa * cc;
    const float d = c[D * STRIDE - D] - c[-D * STRIDE - D];
    const float e = c[D * STRIDE] - c[-D * STRIDE];
    const float f = c[D * STRIDE + D] - c[-D * STRIDE + D];
    *dy = (wa * d + wb * e) + wa * f;
}

struct blur_args
{
    const float *in;
    size_t in_stride;
    float *out0, *out1;
    size_t out_stride;
    int w, h;
    const float *kcontrast; // FLOW
    int n_octave_steps;     // FLOW
    unsigned int *partial_max; // MODG: [image][workgroup] bit patterns of the tile maxima
    float *out2;               // DERIV: not null = the blurred image too (stride out2_stride).  DERIV: out0 = the interleaved (Lx, Ly) float2 plane, out_stride in float2;
                               // FLOW_DERIV: out1 = that plane with out2_stride (float2), out0 = conductivity with out_stride
    size_t out2_stride;
};

template <int MODE, int M /*margin of the blurred tile*/, int R /*tap radius*/>
__global__ __launch_bounds__(256) void blur_fused_kernel(blur_args A, taps_t t)
{
    constexpr int BW = BT_X + 2 * M, BH = BT_Y + 2 * M, IW = BW + 2 * R, IH = BH + 2 * R;
    // LDS rows are padded to multiples of 4 floats so that a thread can take 4 neighbouring outputs of the row pass
    // from two 16-byte reads and store them with one 16-byte write
    constexpr int BWQ = (BW + 3) / 4, BWP = 4 * BWQ, IWP = BWP + 2 * R + ((4 - ((2 * R) & 3)) & 3);
    __shared__ __attribute__((aligned(16))) float tin[IWP * IH];  // input tile; reused for the blurred tile (BW x BH)
    __shared__ __attribute__((aligned(16))) float trow[BWP * IH]; // row pass
    static_assert(IWP * IH >= BW * BH && (IWP & 3) == 0, "blur tile layout");
    const int w = A.w, h = A.h;
    const int tiles_x = (w + BT_X - 1) / BT_X, tiles_y = (h + BT_Y - 1) / BT_Y;
    int tile_x, tile_y;
    if (!xcd_tile(tiles_x, tiles_y, &tile_x, &tile_y))
        return;
    const int x0 = tile_x * BT_X, y0 = tile_y * BT_Y;
    const int bx0 = x0 - M, by0 = y0 - M;
    const float *I = A.in + (size_t)blockIdx.z * A.in_stride;
    // workgroups whose input window lies inside the image fetch it as 16-byte quads from a 4-pixel aligned start (a third
    // of the load instructions); needs rows that start 16-byte aligned, i.e. a width that is a multiple of 4
    const int ix0 = bx0 - R, iy0 = by0 - R;
    const int ax0 = ix0 & ~3;
    const bool quad_window = M > 0 && (w & 3) == 0 && ((A.in_stride & 3) == 0) && ix0 >= 0 && iy0 >= 0 && ix0 + IW <= w && iy0 + IH <= h &&
                             (((uintptr_t)A.in & 15) == 0);
    if (quad_window)
    {
        constexpr int NQ = (IW + 3 + 3) / 4; // quads per row, whatever the misalignment of ix0 (0..3 pixels)
        constexpr int QITERS = (NQ * IH + 255) / 256;
        const int shift = ix0 - ax0;
        float4 q[QITERS];
#pragma unroll
        for (int it = 0; it < QITERS; it++)
        {
            const int idx = threadIdx.x + it * 256;
            const int ly = idx / NQ, j = idx - ly * NQ;
            const bool in = idx < NQ * IH && ax0 + 4 * j < w;
            q[it] = in ? *reinterpret_cast<const float4 *>(I + (size_t)(iy0 + ly) * w + ax0 + 4 * j) : make_float4(0.0f, 0.0f, 0.0f, 0.0f);
        }
#pragma unroll
        for (int it = 0; it < QITERS; it++)
        {
            const int idx = threadIdx.x + it * 256;
            const int ly = idx / NQ, j = idx - ly * NQ;
            if (idx < NQ * IH)
            {
                const int lx = 4 * j - shift;
                float *row = &tin[ly * IWP];
                if (lx >= 0 && lx < IW)
                    row[lx] = q[it].x;
                if (lx + 1 >= 0 && lx + 1 < IW)
                    row[lx + 1] = q[it].y;
                if (lx + 2 >= 0 && lx + 2 < IW)
                    row[lx + 2] = q[it].z;
                if (lx + 3 >= 0 && lx + 3 < IW)
                    row[lx + 3] = q[it].w;
            }
        }
    }
    else
    {
        // every load of the tile is issued before the first LDS store (a rolled loop would wait out one memory
        // latency per iteration)
        constexpr int ITERS = (IW * IH + 255) / 256;
        float v[ITERS];
#pragma unroll
        for (int it = 0; it < ITERS; it++)
        {
            const int idx = threadIdx.x + it * 256;
            const int ly = idx / IW, lx = idx - ly * IW;
            v[it] = (idx < IW * IH) ? I[(size_t)clampi(by0 - R + ly, 0, h - 1) * w + clampi(bx0 - R + lx, 0, w - 1)] : 0.0f;
        }
#pragma unroll
        for (int it = 0; it < ITERS; it++)
        {
            const int idx = threadIdx.x + it * 256;
            if (idx < IW * IH)
                tin[(idx / IW) * IWP + (idx % IW)] = v[it];
        }
    }
    __syncthreads();
    // Row pass: a thread takes 4 neighbouring outputs of one row - its 4 + 2R inputs arrive as 16-byte LDS reads and every
    // tap feeds two packed fp32 multiplies and two packed adds (v_pk_mul_f32 / v_pk_add_f32; a multiply and an add per
    // tap and output as before, in ascending tap order: the same float expression, a third of the LDS instructions).
    typedef float pk2 __attribute__((ext_vector_type(2)));
    for (int idx = threadIdx.x; idx < BWQ * IH; idx += 256)
    {
        const int ly = idx / BWQ, q = idx - ly * BWQ;
        const float *src = &tin[ly * IWP + 4 * q];
        float v[4 + 2 * R + 3];
#pragma unroll
        for (int i = 0; i < (4 + 2 * R + 3) / 4; i++)
        {
            const float4 f = *reinterpret_cast<const float4 *>(src + 4 * i);
            v[4 * i] = f.x, v[4 * i + 1] = f.y, v[4 * i + 2] = f.z, v[4 * i + 3] = f.w;
        }
        pk2 a01 = {0.0f, 0.0f}, a23 = {0.0f, 0.0f};
#pragma unroll
        for (int i = 0; i < 2 * R + 1; i++)
        {
            const pk2 k = {t.k[i], t.k[i]};
            a01 = a01 + k * pk2{v[i], v[i + 1]};
            a23 = a23 + k * pk2{v[i + 2], v[i + 3]};
        }
        *reinterpret_cast<float4 *>(&trow[ly * BWP + 4 * q]) = make_float4(a01.x, a01.y, a23.x, a23.y);
    }
    __syncthreads();
    // Column pass: a thread takes 4 vertically neighbouring outputs of one column: 4 + 2R reads for 4 outputs
    // A wavefront takes the first 64 columns of a group of four rows (lane = column: consecutive banks), the groups dealt
    // round-robin to the four wavefronts; the 2 M columns beyond 64 of all groups go to the last two wavefronts in one more
    // pass.  With lanes running across the end of a 64 + 2 M wide row two row groups shared a wavefront, 4 BWP words apart,
    // and the lanes of the second sat on banks of the first (0.8 conflict cycles per LDS instruction by the counters).
    constexpr int BHQ = (BH + 3) / 4, XC = BW - BT_X, COL_ITERS = (BHQ + 3) / 4 + (XC > 0 ? 1 : 0);
    static_assert(BT_X == 64 && XC * BHQ <= 128, "one lane per column; the extra columns fit two wavefronts");
    for (int pass = 0; pass < COL_ITERS; pass++)
    {
        int qy, lx;
        if (pass < (BHQ + 3) / 4)
        {
            qy = pass * 4 + (int)(threadIdx.x >> 6);
            lx = (int)(threadIdx.x & 63);
            if (qy >= BHQ)
                continue;
        }
        else
        {
            const int e = (int)threadIdx.x - 128;
            if (e < 0 || e >= XC * BHQ)
                continue;
            qy = e / (XC > 0 ? XC : 1);
            lx = BT_X + e - qy * (XC > 0 ? XC : 1);
        }
        const int ly0 = 4 * qy;
        float v[4 + 2 * R];
#pragma unroll
        for (int i = 0; i < 4 + 2 * R; i++)
            v[i] = (ly0 + i < IH) ? trow[(ly0 + i) * BWP + lx] : 0.0f;
        pk2 a01 = {0.0f, 0.0f}, a23 = {0.0f, 0.0f};
#pragma unroll
        for (int i = 0; i < 2 * R + 1; i++)
        {
            const pk2 k = {t.k[i], t.k[i]};
            a01 = a01 + k * pk2{v[i], v[i + 1]};
            a23 = a23 + k * pk2{v[i + 2], v[i + 3]};
        }
        const float acc4[4] = {a01.x, a01.y, a23.x, a23.y};
#pragma unroll
        for (int j = 0; j < 4; j++)
        {
            const int ly = ly0 + j;
            if (ly >= BH)
                continue;
            if (MODE == BLUR_PLAIN)
            {
                const int x = bx0 + lx, y = by0 + ly;
                if (x < w && y < h)
                    A.out0[(size_t)blockIdx.z * A.out_stride + (size_t)y * w + x] = acc4[j];
            }
            else
                tin[ly * BW + lx] = acc4[j];
        }
    }
    if (MODE == BLUR_PLAIN)
        return;
    __syncthreads();
    auto at = [&](int xx, int yy) { return tin[(yy - by0) * BW + (xx - bx0)]; };
    float k = 0.0f, inv = 0.0f, vmax = 0.0f;
    if (MODE == BLUR_FLOW || MODE == BLUR_FLOW_DERIV)
    {
        k = A.kcontrast[blockIdx.z];
        for (int i = 0; i < A.n_octave_steps; i++) // kcontrast *= 0.75 at every new octave, one rounding per step
            k = k * 0.75f;
        inv = 1.0f / (k * k);
    }
    const bool tiny = w <= 2 * M + 2 || h <= 2 * M + 2; // uniform: only then can an index need more than one reflection
    // uniform per workgroup: the tile and its stencil margin lie inside the image, so nothing reflects and every
    // pixel exists - the consumer then reads the blurred tile at fixed offsets from its own position
    const bool inner_tile = M > 0 && bx0 >= 0 && by0 >= 0 && x0 + BT_X + M <= w && y0 + BT_Y + M <= h;
    if (inner_tile)
    {
#pragma unroll 4
        for (int idx = threadIdx.x; idx < BT_X * BT_Y; idx += 256)
        {
            const int ly = idx / BT_X, lx = idx - ly * BT_X;
            const int x = x0 + lx, y = y0 + ly;
            const float *c = &tin[(ly + M) * BW + (lx + M)];
            const size_t o = (size_t)blockIdx.z * A.out_stride + (size_t)y * w + x;
            if (MODE == BLUR_DERIV || MODE == BLUR_FLOW_DERIV)
            {
                const float wgt = 10.0f / 3.0f;
                const float nrm = 1.0f / (2.0f * (float)M * (wgt + 2.0f));
                const float wn = wgt * nrm;
                float dx, dy;
                pattern_lds<(M > 0 ? M : 1), BW>(c, nrm, wn, &dx, &dy);
                // Lx and Ly leave as one interleaved float2 plane: their consumers (determinant, orientation, descriptor)
                // always want both at the same pixel, and the descriptor's gathers are what bounds it
                if (MODE == BLUR_DERIV)
                {
                    reinterpret_cast<float2 *>(A.out0)[o] = make_float2(dx, dy);
                    if (A.out2) // (wave-uniform) the blurred image itself: level 0's base image and its derivatives in one launch
                        A.out2[(size_t)blockIdx.z * A.out2_stride + (size_t)y * w + x] = c[0];
                }
                else
                    reinterpret_cast<float2 *>(A.out1)[(size_t)blockIdx.z * A.out2_stride + (size_t)y * w + x] = make_float2(dx, dy);
            }
            if (MODE != BLUR_DERIV)
            {
                float lx_, ly_;
                pattern_lds<1, BW>(c, 3.0f, 10.0f, &lx_, &ly_);
                if (MODE == BLUR_FLOW || MODE == BLUR_FLOW_DERIV)
                    A.out0[o] = 1.0f / (1.0f + inv * (lx_ * lx_ + ly_ * ly_));
                else
                {
                    const bool interior = x >= 1 && x < w - 1 && y >= 1 && y < h - 1;
                    const float m = interior ? sqrtf(lx_ * lx_ + ly_ * ly_) : 0.0f;
                    A.out0[o] = m;
                    vmax = fmaxf(vmax, m);
                }
            }
        }
    }
    else
#pragma unroll 4
    for (int idx = threadIdx.x; idx < BT_X * BT_Y; idx += 256)
    {
        const int ly = idx / BT_X, lx = idx - ly * BT_X;
        const int x = x0 + lx, y = y0 + ly;
        if (x >= w || y >= h)
            continue;
        const size_t o = (size_t)blockIdx.z * A.out_stride + (size_t)y * w + x;
        if (MODE == BLUR_DERIV || MODE == BLUR_FLOW_DERIV)
        {
            const float wgt = 10.0f / 3.0f;
            const float nrm = 1.0f / (2.0f * (float)M * (wgt + 2.0f));
            const float wn = wgt * nrm;
            float dx, dy;
            if (tiny)
                pattern_xy(at, reflect101(x - M, w), x, reflect101(x + M, w), reflect101(y - M, h), y, reflect101(y + M, h),
                           nrm, wn, &dx, &dy);
            else
                pattern_xy(at, reflect101_once(x - M, w), x, reflect101_once(x + M, w), reflect101_once(y - M, h), y,
                           reflect101_once(y + M, h), nrm, wn, &dx, &dy);

            if (MODE == BLUR_DERIV)
            {
                reinterpret_cast<float2 *>(A.out0)[o] = make_float2(dx, dy);
                if (A.out2)
                    A.out2[(size_t)blockIdx.z * A.out2_stride + (size_t)y * w + x] = at(x, y);
            }
            else
                reinterpret_cast<float2 *>(A.out1)[(size_t)blockIdx.z * A.out2_stride + (size_t)y * w + x] = make_float2(dx, dy);
        }
        if (MODE != BLUR_DERIV)
        {
            float lx_, ly_;
            if (tiny)
                pattern_xy(at, reflect101(x - 1, w), x, reflect101(x + 1, w), reflect101(y - 1, h), y, reflect101(y + 1, h),
                           3.0f, 10.0f, &lx_, &ly_);
            else
                pattern_xy(at, reflect101_once(x - 1, w), x, reflect101_once(x + 1, w), reflect101_once(y - 1, h), y,
                           reflect101_once(y + 1, h), 3.0f, 10.0f, &lx_, &ly_);
            if (MODE == BLUR_FLOW || MODE == BLUR_FLOW_DERIV)
                A.out0[o] = 1.0f / (1.0f + inv * (lx_ * lx_ + ly_ * ly_));
            else
            {
                const bool interior = x >= 1 && x < w - 1 && y >= 1 && y < h - 1;
                const float m = interior ? sqrtf(lx_ * lx_ + ly_ * ly_) : 0.0f;
                A.out0[o] = m;
                vmax = fmaxf(vmax, m);
            }
        }
    }
    if (MODE == BLUR_MODG)
    {
        // tile maximum -> partial_max (non-negative floats order like their bit patterns); reduced per image
        // by hmax_reduce_kernel: no same-address atomics
        __shared__ unsigned int wmax[4];
        unsigned int bits = __float_as_uint(vmax);
        for (int off = 32; off >= 1; off >>= 1)
            bits = max(bits, (unsigned int)__shfl_xor((int)bits, off));
        if ((threadIdx.x & 63) == 0)
            wmax[threadIdx.x >> 6] = bits;
        __syncthreads();
        if (threadIdx.x == 0)
            A.partial_max[(size_t)blockIdx.z * (tiles_x * tiles_y) + tile_y * tiles_x + tile_x] =
                max(max(wmax[0], wmax[1]), max(wmax[2], wmax[3]));
    }
}

__global__ void hmax_reduce_kernel(const unsigned int *__restrict__ partial, int n_partial, unsigned int *__restrict__ hmax_bits)
{
    __shared__ unsigned int wmax[4];
    unsigned int bits = 0;
    for (int i = threadIdx.x; i < n_partial; i += 256)
        bits = max(bits, partial[(size_t)blockIdx.x * n_partial + i]);
    for (int off = 32; off >= 1; off >>= 1)
        bits = max(bits, (unsigned int)__shfl_xor((int)bits, off));
    if ((threadIdx.x & 63) == 0)
        wmax[threadIdx.x >> 6] = bits;
    __syncthreads();
    if (threadIdx.x == 0)
        hmax_bits[blockIdx.x] = max(max(wmax[0], wmax[1]), max(wmax[2], wmax[3]));
}

// image rows per workgroup (8: 4.15 us per image - a workgroup's 304 LDS clears and 304 global atomics for 2 048 pixels; 32: 3.3; 64: 3.35;
// 4 or 16 copies of every bin against same-word LDS atomics: 3.3 / 4.8)
constexpr int HIST_ROWS = 32;
__global__ void hist_kernel(const float *__restrict__ modg, int w, int h, size_t stride,
                            const unsigned int *__restrict__ hmax_bits, int nbins, unsigned int *__restrict__ hist)
{
    // workgroup-private histogram in LDS (each workgroup walks `rows_per_block` image rows), flushed with
    // one global atomic per non-empty bin: integer counts, so the result does not depend on the order
    __shared__ unsigned int lh[304];
    for (int i = threadIdx.x; i <= nbins; i += blockDim.x)
        lh[i] = 0;
    __syncthreads();
    const int x = blockIdx.x * blockDim.x + threadIdx.x;
    const float hmax = __uint_as_float(hmax_bits[blockIdx.z]);
    const int rows_per_block = HIST_ROWS;
    // compute_kcontrast (OpenCV 4.x): every interior pixel, zeros included, goes to bin (int)(m * ((nbins - 1) / hmax))
    const float to_bin = (float)(nbins - 1) / hmax;
    for (int r = 0; r < rows_per_block; r++)
    {
        const int y = blockIdx.y * rows_per_block + r;
        const bool inside = !(x < 1 || x >= w - 1 || y < 1 || y >= h - 1) && hmax != 0.0f;
        if (inside)
            atomicAdd(&lh[(int)(modg[(size_t)blockIdx.z * stride + (size_t)y * w + x] * to_bin)], 1u);
    }
    __syncthreads();
    for (int i = threadIdx.x; i <= nbins; i += blockDim.x)
        if (lh[i])
            atomicAdd(hist + (size_t)blockIdx.z * (nbins + 1) + i, lh[i]);
}

__global__ void kcontrast_kernel(const unsigned int *__restrict__ hist, const unsigned int *__restrict__ hmax_bits,
                                 int nbins, float perc, float *__restrict__ kcontrast, int n_images, int total)
{
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= n_images)
        return;
    // bin 0 is the background; the contrast is hmax * k / nbins at the first k >= 1 whose lower bins 1 .. k - 1 hold the percentile
    const unsigned int *hh = hist + (size_t)b * (nbins + 1);
    const float hmax = __uint_as_float(hmax_bits[b]);
    float kc = 0.03f;
    if (hmax != 0.0f)
    {
        const int nthreshold = (int)((float)(total - (int)hh[0]) * perc);
        int nelements = 0;
        for (int k = 1; k < nbins; k++)
        {
            if (nelements >= nthreshold)
            {
                kc = hmax * (float)k / (float)nbins;
                break;
            }
            nelements += (int)hh[k];
        }
    }
    kcontrast[b] = kc;
}

// ---- diffusion (the PM-G2 conductivity is BLUR_FLOW above)
// Up to FED_FUSE explicit diffusion steps per launch, in registers.  A wavefront owns a 64-column x (NLD_TY + 2K)-row
// strip: each lane keeps its column of L and of the two conductivity sums (c[i] + c[i+1], c[i] + c[i+w]) in
// VGPRs, the horizontal neighbours arrive by DPP wavefront shifts, and the strip is swept top to bottom once per step,
// in place (a row needs the old row below and the already computed flux from the row above).  The valid region
// shrinks by one pixel per step, so 64 - 2K columns x NLD_TY rows are written: 12 B/pixel of HBM traffic per launch
// instead of per step, at ~14 VALU instructions per pixel-step (the LDS-tiled form of this loop was VALU-bound).
// The arithmetic is the one-step form's: flux (c_a + c_b) * (L_b - L_a) evaluated once per pixel pair and used
// with both signs, 0 across the image border.
constexpr int FED_FUSE = 4;
constexpr int NLD_TY = 16; // output rows per wavefront strip (8, 16, 24, 32 measured: 39, 37, 44, 43 us per image)
struct fed_tau_group
{
    float tau[FED_FUSE];
};
// the neighbouring lane's value by a DPP wavefront shift (one vector move; __shfl_down / __shfl_up by one went through the LDS
// crossbar: two ds_bpermute and their waits per pixel-step): lane i takes lane i + 1 (i - 1); the last (first) lane keeps its
// own value, as the shuffles did.  (39.5 -> 37 us per image.  Tried on top and dropped: a second instantiation without
// the border selects for strips inside the image - 18.6 + 18.6 against 17.6 + 18.1 us.)
__device__ __forceinline__ float lane_right(float x)
{
    const int xi = __float_as_int(x);
    return __int_as_float(__builtin_amdgcn_update_dpp(xi, xi, 0x130 /* wave_shl:1 */, 0xf, 0xf, false));
}
__device__ __forceinline__ float lane_left(float x)
{
    const int xi = __float_as_int(x);
    return __int_as_float(__builtin_amdgcn_update_dpp(xi, xi, 0x138 /* wave_shr:1 */, 0xf, 0xf, false));
}

// the same shifts for the strip kernels, whose first and last lane hold stencil margin only: what those lanes receive does
// not matter, so the move needs no copy of the old value in front of it (bound_ctrl: lanes without a source take 0)
__device__ __forceinline__ float strip_right(float x)
{
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), 0x130 /* wave_shl:1 */, 0xf, 0xf, true));
}
__device__ __forceinline__ float strip_left(float x)
{
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), 0x138 /* wave_shr:1 */, 0xf, 0xf, true));
}

template <int K>
__global__ __launch_bounds__(256) void nld_fused_kernel(const float *__restrict__ Lin, const float *__restrict__ cflow,
                                                        float *__restrict__ Lout, int w, int h, size_t l_stride,
                                                        size_t c_stride, size_t out_stride, fed_tau_group T)
{
    constexpr int TY = NLD_TY, RH = TY + 2 * K, OW = 64 - 2 * K;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int X0 = (blockIdx.x * 4 + wv) * OW; // first output column of this wavefront
    if (X0 >= w)
        return;
    const int Y0 = blockIdx.y * TY;
    const int gx = X0 - K + lane;
    const bool col_in = gx >= 0 && gx < w;
    const bool has_right = col_in && gx + 1 < w, has_left = gx > 0 && col_in;
    const float *L = Lin + (size_t)blockIdx.z * l_stride, *c = cflow + (size_t)blockIdx.z * c_stride;
    float Lr[RH], cx[RH], cy[RH];
#pragma unroll
    for (int r = 0; r < RH; r++)
    {
        const int gy = Y0 - K + r;
        const bool in = col_in && gy >= 0 && gy < h;
        Lr[r] = in ? L[(size_t)gy * w + gx] : 0.0f;
        cy[r] = in ? c[(size_t)gy * w + gx] : 0.0f;
    }
#pragma unroll
    for (int r = 0; r < RH; r++)
        cx[r] = cy[r] + lane_right(cy[r]);
#pragma unroll
    for (int r = 0; r + 1 < RH; r++)
        cy[r] = cy[r] + cy[r + 1];
#pragma unroll
    for (int j = 0; j < K; j++)
    {
        const float half = 0.5f * T.tau[j];
        float flux_above = 0.0f;
#pragma unroll
        for (int r = 0; r < RH; r++)
        {
            const int gy = Y0 - K + r;
            const float Lc = Lr[r];
            const float d = lane_right(Lc) - Lc;
            const float xpos = has_right ? cx[r] * d : 0.0f;
            const float xleft = lane_left(xpos);
            const float xneg = has_left ? xleft : 0.0f;
            float ypos = 0.0f;
            if (r + 1 < RH)
                ypos = (gy + 1 < h) ? cy[r] * (Lr[r + 1] - Lc) : 0.0f;
            const float yneg = gy > 0 ? flux_above : 0.0f;
            Lr[r] = Lc + half * (((xpos - xneg) + ypos) - yneg); // (xpos - xneg + ypos - yneg, as the C expression associates)
            flux_above = ypos;
        }
    }
    if (lane >= K && lane < 64 - K && gx < w)
    {
#pragma unroll
        for (int r = K; r < K + TY; r++)
        {
            const int gy = Y0 - K + r;
            if (gy < h)
                Lout[(size_t)blockIdx.z * out_stride + (size_t)gy * w + gx] = Lr[r];
        }
    }
}

__global__ void halfsample_kernel(const float *__restrict__ in, int w, int h, size_t in_stride, float *__restrict__ out,
                                  int ow, int oh, size_t out_stride)
{
    const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y;
    if (x >= ow)
        return;
    const float *I = in + (size_t)blockIdx.z * in_stride;
    const int x0 = min(2 * x, w - 1), x1 = min(2 * x + 1, w - 1), y0 = min(2 * y, h - 1), y1 = min(2 * y + 1, h - 1);
    out[(size_t)blockIdx.z * out_stride + (size_t)y * ow + x] =
        ((I[(size_t)y0 * w + x0] + I[(size_t)y0 * w + x1]) + (I[(size_t)y1 * w + x0] + I[(size_t)y1 * w + x1])) * 0.25f;
}

// The next octave's first image when a dimension of the octave before is odd (1067 -> 533 rows of a 3:2 image): cv::resize
// INTER_AREA no longer sees an integer scale and runs its general area path on BOTH axes - computeResizeAreaTab's overlap weights
// (the ones resize_area_kernel uses for the 8-bit image), per source row buf += S * alpha in table order, then sum = beta * buf
// for the first source row of a destination row and sum += beta * buf for the others.  One thread per destination pixel; the
// table entries of its column and row (at most four each at a scale of two) are recomputed in double as the table builder does.
struct area_taps
{
    int first, n; // source indices first .. first + n - 1
    float a[4];
};
__device__ __forceinline__ area_taps area_taps_of(int d, int ssize, int dsize)
{
    const double scale = 1.0 / ((double)dsize / ssize); // (cv::resize given a size: scale_x = 1. / inv_scale_x, inv_scale_x = dsize / ssize)
    const double f1 = d * scale, f2 = f1 + scale, cell = fmin(scale, ssize - f1);
    int s1 = (int)ceil(f1), s2 = (int)floor(f2);
    s2 = min(s2, ssize - 1);
    s1 = min(s1, s2);
    area_taps t;
    t.n = 0;
    t.first = s1;
    if (s1 - f1 > 1e-3)
    {
        t.first = s1 - 1;
        t.a[t.n++] = (float)((s1 - f1) / cell);
    }
    for (int sx = s1; sx < s2 && t.n < 4; sx++)
        t.a[t.n++] = (float)(1.0 / cell);
    if (f2 - s2 > 1e-3 && t.n < 4)
        t.a[t.n++] = (float)(fmin(fmin(f2 - s2, 1.0), cell) / cell);
    return t;
}
__global__ void halfsample_area_kernel(const float *__restrict__ in, int w, int h, size_t in_stride, float *__restrict__ out,
                                       int ow, int oh, size_t out_stride)
{
    const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y;
    if (x >= ow)
        return;
    const float *I = in + (size_t)blockIdx.z * in_stride;
    const area_taps tx = area_taps_of(x, w, ow), ty = area_taps_of(y, h, oh);
    float sum = 0.0f;
    for (int e = 0; e < ty.n; e++)
    {
        const float *S = I + (size_t)(ty.first + e) * w + tx.first;
        float buf = 0.0f;
        for (int k = 0; k < tx.n; k++)
            buf = buf + S[k] * tx.a[k];
        sum = e == 0 ? ty.a[e] * buf : sum + ty.a[e] * buf;
    }
    out[(size_t)blockIdx.z * out_stride + (size_t)y * ow + x] = sum;
}

__global__ void copy_plane_kernel(const float *__restrict__ in, size_t in_stride, float *__restrict__ out,
                                  size_t out_stride, size_t n)
{
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n)
        out[(size_t)blockIdx.z * out_stride + i] = in[(size_t)blockIdx.z * in_stride + i];
}

// ---- detection
struct cand_t
{
    int level, x, y;
    float response;
    float dx, dy; // sub-pixel offset of the extremum from the quadratic fit of its 3 x 3 determinants (|.| > 1: no keypoint)
};

// Scale-normalised Hessian determinant of a 64 x 24 tile and its strict 3x3 maxima above the threshold in one
// pass: the Lx / Ly tiles (+ halo S + 1) are staged in LDS, the determinant tile (+ halo 1) is built over them,
// and the level's sparse maxima map (response at maxima, 0 elsewhere) is written next to the determinant.
template <int S>
__global__ __launch_bounds__(256) void det_maxima_kernel(const float2 *__restrict__ Lxy, size_t stride,
                                                         float2 *__restrict__ Fit, float *__restrict__ Rmax, int w, int h,
                                                         float thr, unsigned int *__restrict__ tile_counts, int tile_off,
                                                         int n_tiles, float margin, unsigned long long *__restrict__ mask,
                                                         size_t mask_stride)
{
    constexpr int HW = S + 1, RW = BT_X + 2 * HW, RH = DT_Y + 2 * HW, DW = BT_X + 2, DH = DT_Y + 2;
    __shared__ float tx[RW * RH], ty[RW * RH];
    static_assert(DW * DH <= RW * RH, "the determinant tile reuses the Lx tile's storage");
    float *td = tx; // written only after every thread has its determinants in registers (barrier in between)
    const int tiles_x = (w + BT_X - 1) / BT_X, tiles_y = (h + DT_Y - 1) / DT_Y;
    int tile_x, tile_y;
    if (!xcd_tile(tiles_x, tiles_y, &tile_x, &tile_y))
        return;
    const int x0 = tile_x * BT_X, y0 = tile_y * DT_Y;
    const int rx0 = x0 - HW, ry0 = y0 - HW;
    const float2 *XY = Lxy + (size_t)blockIdx.z * stride;
    // The determinant tile (DW = 66 columns) is walked a tile row per wavefront - lane = column 0..63, four rows per pass -
    // and its last two columns in one extra pass: with lanes running across a row wrap (66 is not a multiple of the wave)
    // two rows share a wavefront and their 72-word pitch puts lanes of the second row on banks of the first (the
    // counters showed 0.7 conflict cycles per LDS instruction in this kernel).
    constexpr int ROW_ITERS = (DH + 3) / 4, DITERS = ROW_ITERS + 1;
    static_assert(BT_X == 64 && 2 * DH <= 256, "one lane per column; the two extra columns fit one pass");
    float dreg[DITERS]; // this thread's determinants; 0 = outside the image / the tile
    const int wave_id = threadIdx.x >> 6, lane_id = threadIdx.x & 63;
    auto cell_of = [&](int it, int *ly, int *lx) -> bool {
        if (it < ROW_ITERS)
        {
            *ly = it * 4 + wave_id;
            *lx = lane_id;
            return *ly < DH;
        }
        *ly = (int)threadIdx.x >> 1;
        *lx = 64 + ((int)threadIdx.x & 1);
        return (int)threadIdx.x < 2 * DH;
    };
    // uniform per workgroup: tile + halo inside the image -> no reflection, every pixel exists, fixed LDS offsets
    const bool inner_tile = rx0 >= 0 && ry0 >= 0 && x0 + BT_X + HW <= w && y0 + DT_Y + HW <= h;
    if (inner_tile)
    {
        // a tile row per wavefront and pass: lane = column (the row's first 64 pixels are one 512-byte run), the RW - 64
        // columns beyond ride on the first lanes; no index arithmetic per element, no bounds to test (the flattened walk
        // below spent ~15 vector instructions per element on idx / RW and the four comparisons)
        constexpr int RITERS = (RH + 3) / 4;
        static_assert(RW > 64 && RW - 64 <= 64, "a row is 64 columns and a remainder of at most 64");
        float2 va[RITERS], vb[RITERS];
        const float2 *row0 = XY + (size_t)(ry0 + wave_id) * w + rx0 + lane_id;
#pragma unroll
        for (int it = 0; it < RITERS; it++) // all loads in flight before the first LDS store
        {
            const bool row = it * 4 + wave_id < RH;
            va[it] = row ? row0[(size_t)(4 * it) * w] : make_float2(0.0f, 0.0f);
            vb[it] = row && lane_id < RW - 64 ? row0[(size_t)(4 * it) * w + 64] : make_float2(0.0f, 0.0f);
        }
#pragma unroll
        for (int it = 0; it < RITERS; it++)
            if (it * 4 + wave_id < RH)
            {
                const int o = (it * 4 + wave_id) * RW + lane_id;
                tx[o] = va[it].x;
                ty[o] = va[it].y;
                if (lane_id < RW - 64)
                {
                    tx[o + 64] = vb[it].x;
                    ty[o + 64] = vb[it].y;
                }
            }
    }
    else
    {
        constexpr int ITERS = (RW * RH + 255) / 256; // all loads in flight before the first LDS store
        float vx[ITERS], vy[ITERS];
#pragma unroll
        for (int it = 0; it < ITERS; it++)
        {
            const int idx = threadIdx.x + it * 256;
            const int ly = idx / RW, lx = idx - ly * RW;
            const int gx = rx0 + lx, gy = ry0 + ly;
            const bool in = idx < RW * RH && gx >= 0 && gx < w && gy >= 0 && gy < h;
            const float2 v = in ? XY[(size_t)gy * w + gx] : make_float2(0.0f, 0.0f);
            vx[it] = v.x;
            vy[it] = v.y;
        }
#pragma unroll
        for (int it = 0; it < ITERS; it++)
        {
            const int idx = threadIdx.x + it * 256;
            if (idx < RW * RH)
            {
                tx[idx] = vx[it];
                ty[idx] = vy[it];
            }
        }
    }
    __syncthreads();
    const float wgt = 10.0f / 3.0f;
    const float nrm = 1.0f / (2.0f * (float)S * (wgt + 2.0f));
    const float wn = wgt * nrm;
    const float s4 = (float)(S * S * S * S);
    auto atx = [&](int xx, int yy) { return tx[(yy - ry0) * RW + (xx - rx0)]; };
    auto aty = [&](int xx, int yy) { return ty[(yy - ry0) * RW + (xx - rx0)]; };
    const bool tiny = w <= 2 * S + 2 || h <= 2 * S + 2; // uniform: only then can an index need more than one reflection
    if (inner_tile)
    {
#pragma unroll
        for (int it = 0; it < DITERS; it++)
        {
            int ly, lx;
            dreg[it] = 0.0f;
            if (!cell_of(it, &ly, &lx))
                continue;
            const int ci = (ly + S) * RW + (lx + S); // (x, y) in the Lx / Ly tiles: their origin is (x0 - S - 1, y0 - S - 1)
            float lxx, lxy, tmp, lyy;
            pattern_lds<S, RW>(&tx[ci], nrm, wn, &lxx, &lxy);
            pattern_lds<S, RW>(&ty[ci], nrm, wn, &tmp, &lyy);
            const float d = (lxx * lyy - lxy * lxy) * s4;
            dreg[it] = d;
        }
    }
    else
#pragma unroll
    for (int it = 0; it < DITERS; it++)
    {
        int ly, lx;
        dreg[it] = 0.0f;
        if (!cell_of(it, &ly, &lx))
            continue;
        const int x = x0 - 1 + lx, y = y0 - 1 + ly;
        if (x < 0 || x >= w || y < 0 || y >= h)
            continue;
        int xm, xp, ym, yp;
        if (tiny)
        {
            xm = reflect101(x - S, w), xp = reflect101(x + S, w), ym = reflect101(y - S, h), yp = reflect101(y + S, h);
        }
        else
        {
            xm = reflect101_once(x - S, w), xp = reflect101_once(x + S, w), ym = reflect101_once(y - S, h),
            yp = reflect101_once(y + S, h);
        }
        float lxx, lxy, tmp, lyy;
        pattern_xy(atx, xm, x, xp, ym, y, yp, nrm, wn, &lxx, &lxy);
        pattern_xy(aty, xm, x, xp, ym, y, yp, nrm, wn, &tmp, &lyy);
        const float d = (lxx * lyy - lxy * lxy) * s4;
        dreg[it] = d;
    }
    __syncthreads(); // every determinant is in a register: the Lx tile's storage becomes the determinant tile
#pragma unroll
    for (int it = 0; it < DITERS; it++)
    {
        int ly, lx;
        if (cell_of(it, &ly, &lx))
            td[ly * DW + lx] = dreg[it];
    }
    __syncthreads();
    // one wavefront per tile row: the row's maxima as one 64-bit word of the level's bit mask, the responses only where a
    // bit is set (the rest of Rmax is never read: the list builder and the suppression walk the mask)
    unsigned int found = 0;
    static_assert(BT_X == 64 && DT_Y % 4 == 0, "a wavefront per tile row, four rows per pass");
    // interior pixel whose descriptor window [round(x - margin) - 1, round(x + margin) + 1] stays inside the level
    // image (AKAZE's Find_Scale_Space_Extrema; margin = 10 sqrt(2) * sigma_size): the column's half of the test once per
    // lane (a lane keeps its column through the passes), the row's half once per pass
    const int mx = x0 + (int)(threadIdx.x & 63);
    const bool x_ok = mx >= 1 && mx < w - 1 && (int)rintf((float)mx - margin) - 1 >= 0 && (int)rintf((float)mx + margin) + 1 < w;
#pragma unroll
    for (int it = 0; it < DT_Y / 4; it++)
    {
        const int ly = it * 4 + wave_id, lx = threadIdx.x & 63;
        const int x = mx, y = y0 + ly;
        float out = 0.0f;
        float2 fit = make_float2(0.0f, 0.0f);
        const bool y_ok = y >= 1 && y < h - 1 && (int)rintf((float)y - margin) - 1 >= 0 && (int)rintf((float)y + margin) + 1 < h;
        if (x_ok && y_ok)
        {
            const int ci = (ly + 1) * DW + (lx + 1);
            const float v = td[ci];
            if (v > thr)
            {
                bool mx = true;
#pragma unroll
                for (int dy = -1; dy <= 1; dy++)
#pragma unroll
                    for (int dx = -1; dx <= 1; dx++)
                        if ((dx || dy) && !(v > td[ci + dy * DW + dx]))
                            mx = false;
                if (mx)
                {
                    out = v;
                    // the sub-pixel fit of AKAZE's Do_Subpixel_Refinement on the nine determinants around the extremum,
                    // which are all here: the determinant plane itself is then never written (4 bytes per level pixel
                    // of HBM traffic that only these few thousand neighbourhoods per image were ever read from)
                    const float vxp = td[ci + 1], vxm = td[ci - 1], vyp = td[ci + DW], vym = td[ci - DW];
                    const float Dx = 0.5f * (vxp - vxm), Dy = 0.5f * (vyp - vym);
                    const float Dxx = (vxp + vxm) - 2.0f * v, Dyy = (vyp + vym) - 2.0f * v;
                    const float Dxy = 0.25f * ((td[ci + DW + 1] + td[ci - DW - 1]) - (td[ci + DW - 1] + td[ci - DW + 1]));
                    fit = subpixel_solve(Dxx, Dxy, Dyy, Dx, Dy);
                }
            }
        }
        const unsigned long long m = __ballot(out != 0.0f);
        if (out != 0.0f)
        {
            Rmax[(size_t)blockIdx.z * stride + (size_t)y * w + x] = out;
            Fit[(size_t)blockIdx.z * stride + (size_t)y * w + x] = fit;
        }
        if (lx == 0 && y < h)
        {
            mask[(size_t)blockIdx.z * mask_stride + (size_t)y * tiles_x + tile_x] = m;
            found += (unsigned int)__popcll(m);
        }
    }
    // number of maxima of this tile: the candidate list is laid out tile by tile (scan_tiles_kernel)
    __shared__ unsigned int wsum[4];
    for (int off = 32; off >= 1; off >>= 1)
        found += (unsigned int)__shfl_xor((int)found, off);
    if ((threadIdx.x & 63) == 0)
        wsum[threadIdx.x >> 6] = found;
    __syncthreads();
    if (threadIdx.x == 0)
        tile_counts[(size_t)blockIdx.z * n_tiles + tile_off + tile_y * tiles_x + tile_x] =
            wsum[0] + wsum[1] + wsum[2] + wsum[3];
}

// ---- round 5: the same determinant + maxima pass as a register strip.  The counters had det_maxima_kernel issue-bound
// (145 vector instructions per pixel: sixteen LDS taps, their address arithmetic and the products of every pixel's own
// copy of the Scharr pattern).  The pattern's terms are shared between pixels: with hd(x, y) = f(x + S, y) - f(x - S, y)
// and vd(x, y) = f(x, y + S) - f(x, y - S),
//     d/dx f = (wa hd(x, y - S) + wb hd(x, y)) + wa hd(x, y + S),     d/dy f = (wa vd(x - S, y) + wb vd(x, y)) + wa vd(x + S, y)
// - the same float expressions pattern_lds evaluates, every difference and every product formed once per pixel instead of
// three times.  A wavefront owns a strip of 128 columns (a PAIR per lane: every operation is a packed fp32 instruction)
// and walks it top to bottom with the rows it still needs in registers; what comes from the lanes S columns away goes
// through a row buffer in LDS that belongs to the wavefront (no workgroup barrier anywhere).  ~20 vector instructions
// per pixel.  Nothing reflects: a maximum needs its descriptor window inside the image (margin >= 28 pixels), so no
// determinant within the stencil's reach of the border is ever looked at; loads outside the image are clamped into it.
// The maxima leave as bits of the level's mask words by atomicOr and as per-tile counts by atomicAdd (both zeroed by
// the host; integers, so the order is free): a strip is 116 / 120 columns wide, not a multiple of the 64-pixel words.
constexpr int DS_PAD = 8;  // columns of padding on either side of a wavefront's row buffers (>= S)
// output rows of a strip, rounded up to whole ring turns.  The taller, the smaller the share of the rows above and below that
// only feed the stencils - and the fewer, longer wavefronts a launch has: 120 rows for both kernels measured 257 us per image
// against 239 with these (the last round of a launch runs half empty, the small levels have too few strips); 48 rows for the
// determinant alone 22.8 against 22.1
constexpr int DET_STRIP_ROWS = 32, LEVEL_STRIP_ROWS = 64;
constexpr int DS_NOTES = 64; // noted maxima a wavefront works off together (det_strip_kernel)
constexpr int DS_RING = 1; // the ring of requested rows is DS_RING (2 S + 1) long (2: 10 - 16 rows in flight per lane at 2 - 3
                           // wavefronts per SIMD instead of 4 at 4: 31.1 us per image against 28.5)
template <int S> struct det_strip_geom
{
    static constexpr int HALO = (S + 2) & ~1;   // columns of stencil margin on either side, even: 16-byte aligned pair loads
    static constexpr int OW = 128 - 2 * HALO;   // output columns per strip
    // the rows a strip still needs live in rings of U = 2 S + 1 registers; the row loop is unrolled U times (ring slots
    // are then compile-time constants) and rolled over NB blocks: NB U input rows give NB U - 2 S - 2 rows of maxima
    // (the loop body is RB = 2 U rows, which lets the ring of requested rows be RB long: RB - 2 row loads in flight per lane)
    static constexpr int U = 2 * S + 1, RB = DS_RING * U, NB = (DET_STRIP_ROWS + 2 * S + 2 + RB - 1) / RB, NR = NB * RB;
    static constexpr int H = NR - 2 * S - 2;    // output rows per strip
};

__device__ __forceinline__ float max3f(float a, float b, float c)
{
    float r;
    asm("v_max3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c)); // (no NaN among determinants of finite pixels)
    return r;
}
__device__ __forceinline__ float max2f(float a, float b)
{
    float r;
    asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); // (fmaxf would canonicalise both operands first)
    return r;
}
__device__ __forceinline__ float2 subpixel_fit(float v, float vxm, float vxp, float vym, float vyp, float vmm, float vpm, float vmp, float vpp)
{
    // AKAZE's Do_Subpixel_Refinement on the nine determinants around an extremum (vpm = (x + 1, y - 1), ...)
    const float Dx = 0.5f * (vxp - vxm), Dy = 0.5f * (vyp - vym);
    const float Dxx = (vxp + vxm) - 2.0f * v, Dyy = (vyp + vym) - 2.0f * v;
    const float Dxy = 0.25f * ((vpp + vmm) - (vmp + vpm));
    return subpixel_solve(Dxx, Dxy, Dyy, Dx, Dy);
}

template <int S>
__global__ __launch_bounds__(256) void det_strip_kernel(const float2 *__restrict__ Lxy, size_t stride, float2 *__restrict__ Fit,
                                                        float *__restrict__ Rmax, int w, int h, float thr,
                                                        unsigned int *__restrict__ tile_counts, int tile_off, int n_tiles,
                                                        int4 win /*columns .x .. .y and rows .z .. .w may hold a maximum*/,
                                                        unsigned long long *__restrict__ mask, size_t mask_stride)
{
    typedef float pk2 __attribute__((ext_vector_type(2)));
    typedef det_strip_geom<S> G;
    constexpr int HALO = G::HALO, OW = G::OW, U = G::U, RB = G::RB, NB = G::NB, SH = G::H;
    constexpr int LROW = 128 + 2 * DS_PAD;
    __shared__ __attribute__((aligned(16))) float ex[4][3][LROW];
    const int lane = threadIdx.x & 63, wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int strips_x = (w + OW - 1) / OW, strips_y = (h + SH - 1) / SH;
    const int groups = (strips_x * strips_y + 3) / 4, chunk = (groups + 7) / 8;
    // workgroups go to the XCDs round-robin: every XCD takes a contiguous eighth of the strips (neighbours share halos in its L2)
    const int g = (int)(blockIdx.x & 7) * chunk + (int)(blockIdx.x >> 3);
    if ((int)(blockIdx.x >> 3) >= chunk || g >= groups)
        return;
    const int sid = g * 4 + wv;
    if (sid >= strips_x * strips_y)
        return;
    const int sy = sid / strips_x, sx = sid - sy * strips_x;
    const int X0 = sx * OW, Y0 = sy * SH;
    const int cx = X0 - HALO + 2 * lane;                // this lane's two columns: cx, cx + 1
    const int cxc = min(max(cx, 0), max(w - 2, 0));     // clamped into the image (such columns are never maxima)
    const float2 *XY = Lxy + (size_t)blockIdx.z * stride;
    float *const eX = &ex[wv][0][DS_PAD + 2 * lane], *const eA = &ex[wv][1][DS_PAD + 2 * lane], *const eB = &ex[wv][2][DS_PAD + 2 * lane];
    auto wave_sync = []() {
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront"); // LDS writes of the wave before LDS reads after
        __builtin_amdgcn_wave_barrier();
    };
    const float wgt = 10.0f / 3.0f;
    const float nrm = 1.0f / (2.0f * (float)S * (wgt + 2.0f));
    const float wn = wgt * nrm;
    const float s4 = (float)(S * S * S * S);
    const pk2 wa = {nrm, nrm}, wb = {wn, wn}, s44 = {s4, s4};
    // a column this strip owns, inside the window of interior pixels whose descriptor patch stays inside the level image
    // (Find_Scale_Space_Extrema's margin test is monotone in x and in y: the host evaluates it once per level, det_window)
    auto col_ok = [&](int x) { return x >= X0 && x < X0 + OW && x >= win.x && x <= win.y; };
    const bool xok0 = col_ok(cx), xok1 = col_ok(cx + 1);
    const int tiles_x = (w + BT_X - 1) / BT_X;

    constexpr int PF = RB - 2 < 4 ? RB - 2 : (DS_RING == 1 ? (S == 3 ? 3 : 4) : RB - 2); // rows requested ahead (S = 3: one less keeps it at 128 registers)
    float4 ld[RB];
    auto request = [&](int r, int slot) {
        const int gy = min(max(Y0 - (S + 1) + r, 0), h - 1);
        ld[slot] = *reinterpret_cast<const float4 *>(XY + (size_t)gy * w + cxc);
    };
#pragma unroll
    for (int r = 0; r < PF; r++)
        request(r, r);
    // rings indexed by (input row) mod U; what a slot holds before its first row arrives is never looked at by a row
    // whose maxima count (the warm-up rows fall outside [Y0, Y0 + SH))
    pk2 X[U], Y[U], pa_hd[U], pb_hd[U], D[U], T[U], Sd[U];
    float DL[U], DR[U];
#pragma unroll
    for (int j = 0; j < U; j++)
    {
        X[j] = Y[j] = pa_hd[j] = pb_hd[j] = D[j] = T[j] = Sd[j] = pk2{0.0f, 0.0f};
        DL[j] = DR[j] = 0.0f;
    }
    // noted maxima: {x, y, v, v(x - 1)}, {v(x + 1), v(y - 1), v(y + 1), v(x - 1, y - 1)}, {v(x + 1, y - 1), v(x - 1, y + 1), v(x + 1, y + 1), -}
    __shared__ float4 notes[4][DS_NOTES][3];
    unsigned int n_noted = 0; // (wave-uniform)
    auto flush_notes = [&](unsigned int n) {
        wave_sync();
        if ((unsigned int)lane < n)
        {
            const float4 r0 = notes[wv][lane][0], r1 = notes[wv][lane][1], r2 = notes[wv][lane][2];
            const int x = __float_as_int(r0.x), y = __float_as_int(r0.y);
            const size_t at = (size_t)blockIdx.z * stride + (size_t)y * w + x;
            Rmax[at] = r0.z;
            Fit[at] = subpixel_fit(r0.z, r0.w, r1.x, r1.y, r1.z, r1.w, r2.x, r2.y, r2.z);
            atomicOr(&mask[(size_t)blockIdx.z * mask_stride + (size_t)y * tiles_x + (x >> 6)], 1ull << (x & 63));
            atomicAdd(&tile_counts[(size_t)blockIdx.z * n_tiles + tile_off + (y / DT_Y) * tiles_x + (x >> 6)], 1u);
        }
        wave_sync();
    };
    for (int k = 0; k < NB; k++)
    {
#pragma unroll
        for (int jj = 0; jj < RB; jj++)
        {
            const int r = k * RB + jj;              // input row of the strip; image row Y0 - (S + 1) + r
            constexpr int UU = U;                   // (ring slots: r - 2 S -> j + 1, c = r - S -> j + S + 1, m = c - 1 -> j + S, m - 1 -> j + S - 1)
            const int j = jj % UU;
            const int s_top = (j + 1) % UU, s_c = (j + S + 1) % UU, s_m = (j + S) % UU, s_mm = (j + S - 1) % UU;
            request(r + PF, (jj + PF) % RB);
            X[j] = pk2{ld[jj].x, ld[jj].z};
            Y[j] = pk2{ld[jj].y, ld[jj].w};
            const pk2 vdx = X[j] - X[s_top], vdy = Y[j] - Y[s_top];
            const pk2 pbx = wb * vdx, pby = wb * vdy;
            *reinterpret_cast<pk2 *>(eX) = X[j];
            *reinterpret_cast<pk2 *>(eA) = wa * vdx;
            *reinterpret_cast<pk2 *>(eB) = wa * vdy;
            wave_sync();
            const pk2 XL = {eX[-S], eX[-S + 1]}, XR = {eX[S], eX[S + 1]};
            const pk2 paxL = {eA[-S], eA[-S + 1]}, paxR = {eA[S], eA[S + 1]};
            const pk2 payL = {eB[-S], eB[-S + 1]}, payR = {eB[S], eB[S + 1]};
            const pk2 hd = XR - XL;
            pa_hd[j] = wa * hd;
            pb_hd[j] = wb * hd;
            // row c = r - S: its determinants are complete now
            const pk2 lxx = (pa_hd[s_top] + pb_hd[s_c]) + pa_hd[j];
            const pk2 lxy = (paxL + pbx) + paxR;
            const pk2 lyy = (payL + pby) + payR;
            const pk2 d = (lxx * lyy - lxy * lxy) * s44;
            D[s_c] = d;
            DL[s_c] = strip_left(d.y);  // determinant at column cx - 1
            DR[s_c] = strip_right(d.x); // at column cx + 2
            T[s_c] = pk2{max3f(DL[s_c], d.x, d.y), max3f(d.x, d.y, DR[s_c])};
            Sd[s_c] = pk2{max2f(DL[s_c], d.y), max2f(d.x, DR[s_c])};
            // row m = c - 1: rows m - 1, m, m + 1 are there once r >= 2 S + 2.  A strict maximum of its 3 x 3 neighbourhood
            // is greater than the largest of the eight (straight-line code: the tests are cheaper than branches around them)
            const int y = Y0 - (S + 1) + r - S - 1;
            const bool y_ok = (r >= 2 * S + 2) & (y >= win.z) & (y <= win.w);
            const float v0 = D[s_m].x, v1 = D[s_m].y;
            const bool is0 = y_ok & xok0 & (v0 > thr) & (v0 > max3f(T[s_mm].x, T[s_c].x, Sd[s_m].x));
            const bool is1 = y_ok & xok1 & (v1 > thr) & (v1 > max3f(T[s_mm].y, T[s_c].y, Sd[s_m].y));
            // a maximum is only NOTED here - position, value, the eight determinants around it, into the wavefront's list in LDS -
            // and worked off (sub-pixel fit, two divisions; the response, the mask bit and the tile count in HBM) by all lanes
            // together when the list is full or the strip done: in the row loop it cost every row with a maximum (one in
            // five) some 130 instructions for one or two active lanes
            const unsigned long long m0 = __ballot(is0), m1 = __ballot(is1);
            if (m0 | m1)
            {
                const unsigned int n0 = (unsigned int)__popcll(m0), n1 = (unsigned int)__popcll(m1);
                if (n_noted + n0 + n1 > DS_NOTES)
                {
                    flush_notes(n_noted);
                    n_noted = 0;
                }
                const unsigned long long below = (1ull << lane) - 1ull;
                if (is0)
                {
                    float4 *rec = &notes[wv][n_noted + (unsigned int)__popcll(m0 & below)][0];
                    rec[0] = make_float4(__int_as_float(cx), __int_as_float(y), v0, DL[s_m]);
                    rec[1] = make_float4(v1, D[s_mm].x, D[s_c].x, DL[s_mm]);
                    rec[2] = make_float4(D[s_mm].y, DL[s_c], D[s_c].y, 0.0f);
                }
                if (is1)
                {
                    float4 *rec = &notes[wv][n_noted + n0 + (unsigned int)__popcll(m1 & below)][0];
                    rec[0] = make_float4(__int_as_float(cx + 1), __int_as_float(y), v1, v0);
                    rec[1] = make_float4(DR[s_m], D[s_mm].y, D[s_c].y, D[s_mm].x);
                    rec[2] = make_float4(DR[s_mm], D[s_c].x, DR[s_c], 0.0f);
                }
                n_noted += n0 + n1;
            }
            wave_sync(); // the row buffers have been read by every lane before the next row's stores
        }
    }
    if (n_noted)
        flush_notes(n_noted);
}

// ---- round 5: a level's passes in ONE launch, in the same register-strip form: Lsmooth (Gaussian(1) of the image the
// level starts from) -> conductivity and scale-s derivatives (blur_fused_kernel<FLOW_DERIV>), and behind them the level's
// first K <= 4 explicit diffusion steps (nld_fused_kernel<K>) - for the levels of the first octave all of them, so that the
// conductivity plane is neither written nor read and the image is read once: 16 bytes per pixel and level instead of 28.
// A wavefront owns 128 columns (a pair per lane) and streams its rows top to bottom; every stage keeps the few rows
// it still needs in registers and runs a fixed number of rows behind the loads:
//     input row r -> row pass (neighbour pairs by DPP wavefront shifts) -> column pass: Lsmooth row c = r - 2
//     -> conductivity of row f = r - 3 (Scharr pattern at distance 1: neighbours by DPP; reflected taps along the image
//        border are selects, so the plane is exact everywhere - the diffusion spreads it)
//     -> (Lx, Ly) of row r - 2 - S (pattern at distance S: the lanes S columns away through the wavefront's own LDS rows,
//        one wavefront-level synchronisation per row, two alternating sets of rows)
//     -> diffusion step q = 1..K on row f - q: each step lags the one before by a row and keeps one row of its input and
//        the flux through its upper edge (the in-place sweep of nld_fused_kernel, turned into a pipeline)
// Every value is the float expression of the tile / sweep kernels (ascending taps from 0; the pattern's (wa a + wb b) + wa c
// with every difference and product formed once; the flux (c_a + c_b)(L_b - L_a) used with both signs, 0 across the
// border).  Replicated borders of the Gaussian: clamped loads.  The reflected taps of the scale-s pattern within S pixels of
// the border: other ring slots for the rows (wave-uniform selects), per-lane LDS offsets for the columns of a border strip.
// Measured and dropped: the determinant + maxima stage of det_strip_kernel behind the derivatives in this launch (156 - 208
// registers, two wavefronts per SIMD: 70 us per image for the fused launch against 63 for the two).
template <int S, int K> struct level_strip_geom
{
    static constexpr int U = 2 * S + 1;
    static constexpr int RD = S + 2, RN = K > 0 ? K + 3 : 3;
    static constexpr int REACH = RD > RN ? RD : RN; // input rows / columns beyond an output, and rows an output lags its last input
    static constexpr int HALO = (REACH + 1) & ~1;
    static constexpr int OW = 128 - 2 * HALO;
    // (an even number of ring turns: U is odd, so the strip's H = NB U - 2 REACH output rows are even then and the half-sampled
    // rows of the epilogue - pairs of output rows - never straddle two strips)
    static constexpr int NB0 = (LEVEL_STRIP_ROWS + 2 * REACH + U - 1) / U, NB = NB0 + (NB0 & 1), NR = NB * U, H = NR - 2 * REACH;
    static_assert((H & 1) == 0, "a strip owns whole pairs of rows");
};
struct level_strip_args
{
    const float *in;
    size_t in_stride;
    float *flow;
    size_t flow_stride;
    float2 *Lxy;
    size_t lxy_stride;
    float *Lout; // the image after the K steps
    size_t lout_stride;
    unsigned int *hmax_bits; // MODG: per image, the bit pattern of the largest gradient magnitude (zeroed by the host)
    float *half_out; // not null: the next octave's first image too - halfsample_kernel's 2 x 2 means of Lout - (w / 2) x (h / 2), w and h even
    size_t half_stride;
    int w, h;
    const float *kcontrast;
    int n_octave_steps;
    float k[5]; // Gaussian(1) taps
    fed_tau_group T;
};

// 1 / d for 1 <= d < 2^96, correctly rounded: the compiler's own division sequence (v_rcp_f32, Newton step, quotient,
// two residual corrections) without the scaling and fix-up instructions that only act outside that range
__device__ __forceinline__ float recip_ge1(float d)
{
    float r = __builtin_amdgcn_rcpf(d);
    const float e = __builtin_fmaf(-d, r, 1.0f);
    r = __builtin_fmaf(e, r, r);
    float q = r; // 1.0f * r
    const float e2 = __builtin_fmaf(-d, q, 1.0f);
    q = __builtin_fmaf(e2, r, q);
    const float e3 = __builtin_fmaf(-d, q, 1.0f);
    return __builtin_fmaf(e3, r, q);
}

template <int S, int K, bool SF /*the conductivity plane is stored: later launches of the level need it*/,
          bool MODG = false /*the contrast factor's pass instead: |gradient| of the Gaussian(1) image (0 along the border) into the
                              flow plane and its maximum per image; nothing else*/>
__global__ __launch_bounds__(256) void level_strip_kernel(level_strip_args A)
{
    typedef float pk2 __attribute__((ext_vector_type(2)));
    typedef level_strip_geom<S, K> G;
    constexpr int HALO = G::HALO, OW = G::OW, U = G::U, NB = G::NB, SH = G::H, REACH = G::REACH;
    constexpr int LROW = 128 + 2 * DS_PAD, SET = 2 * LROW;
    __shared__ __attribute__((aligned(16))) float ex[4][2][SET];
    const int lane = threadIdx.x & 63, wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int w = A.w, h = A.h;
    const int strips_x = (w + OW - 1) / OW, strips_y = (h + SH - 1) / SH;
    const int groups = (strips_x * strips_y + 3) / 4, chunk = (groups + 7) / 8;
    const int g = (int)(blockIdx.x & 7) * chunk + (int)(blockIdx.x >> 3);
    if ((int)(blockIdx.x >> 3) >= chunk || g >= groups)
        return;
    const int sid = g * 4 + wv;
    if (sid >= strips_x * strips_y)
        return;
    const int sy = sid / strips_x, sx = sid - sy * strips_x;
    const int X0 = sx * OW, Y0 = sy * SH;
    const int cx = X0 - HALO + 2 * lane;            // this lane's two columns: cx, cx + 1 (cx and w are even)
    const int cxc = min(max(cx, 0), w - 2);
    const bool own_cols = cx >= X0 && cx < X0 + OW && cx < w;
    const float *I = A.in + (size_t)blockIdx.z * A.in_stride;
    float *const e0 = &ex[wv][0][DS_PAD + 2 * lane];
    float kc = MODG ? 1.0f : A.kcontrast[blockIdx.z];
    float grad_max = 0.0f; // (MODG: over the pixels this lane owns)
    for (int i = 0; i < A.n_octave_steps; i++) // kcontrast *= 0.75 at every new octave, one rounding per step
        kc = kc * 0.75f;
    const float inv1 = 1.0f / (kc * kc);
    const float wgt = 10.0f / 3.0f;
    const float nrm1 = 1.0f / (2.0f * (float)S * (wgt + 2.0f));
    const float wn1 = wgt * nrm1;
    const pk2 wa = {nrm1, nrm1}, wb = {wn1, wn1}, inv = {inv1, inv1};
    const pk2 k0 = {A.k[0], A.k[0]}, k1 = {A.k[1], A.k[1]}, k2 = {A.k[2], A.k[2]}, k3 = {A.k[3], A.k[3]}, k4 = {A.k[4], A.k[4]};
    const pk2 c3 = {3.0f, 3.0f}, c10 = {10.0f, 10.0f}, zero = {0.0f, 0.0f}, one = {1.0f, 1.0f};
    // rows of the planes this strip writes: a wave-uniform base plus this lane's byte offset
    char *const flow_b = reinterpret_cast<char *>(A.flow + (size_t)blockIdx.z * A.flow_stride);
    char *const lxy_b = reinterpret_cast<char *>(A.Lxy + (size_t)blockIdx.z * A.lxy_stride);
    char *const lout_b = reinterpret_cast<char *>(A.Lout + (size_t)blockIdx.z * A.lout_stride);
    const unsigned int col4 = (unsigned int)cxc * 4u, col8 = (unsigned int)cxc * 8u;
    const int row_end = min(Y0 + SH, h); // rows [Y0, row_end) are this strip's

    // rows requested ahead.  The diffusion needs the raw rows r - 3 and r - 4 again: where the ring is long enough for both
    // (S = 4: 4 rows ahead + 5 behind = 9 slots) they simply stay in the ring of loaded rows (10 registers less: 3 wavefronts
    // per SIMD instead of 2 for S = 4, K = 4)
    constexpr bool RAW_IN_LD = K > 0 && 4 + 5 <= U;
    constexpr int PF = RAW_IN_LD ? 4 : (U - 1 < 6 ? U - 1 : 6);
    // rings indexed by (row) mod U
    float2 ld[U];
    pk2 gr[U], Ls[U], p3[U], p10[U], paS[U], pbS[U], Lraw[U], C[U], CX[U], CY[U];
    pk2 Lsave[K + 1], Fsave[K + 1];
    float half_even = 0.0f; // (half-sampling epilogue: the even row's pair sum)
    auto request = [&](int r, int slot) {
        const int gy = min(max(Y0 - REACH + r, 0), h - 1);
        ld[slot] = *reinterpret_cast<const float2 *>(reinterpret_cast<const char *>(I + (size_t)gy * w) + col4);
    };
#pragma unroll
    for (int r = 0; r < PF; r++)
        request(r, r);
#pragma unroll
    for (int j = 0; j < U; j++)
        gr[j] = Ls[j] = p3[j] = p10[j] = paS[j] = pbS[j] = Lraw[j] = C[j] = CX[j] = CY[j] = zero;
#pragma unroll
    for (int q = 0; q <= K; q++)
        Lsave[q] = Fsave[q] = zero;

    // One input row.  EDGE: the strip comes within reach of the image border - replicated columns for the Gaussian,
    // reflected taps for the two patterns, no flux across the border; the strips inside the image (three quarters of a
    // 1600 x 1200 level) run the same rows without any of those selects.
    auto row = [&](auto edge_tag, const int r, const int j) __attribute__((always_inline)) {
        constexpr bool EDGE = decltype(edge_tag)::value;
        constexpr auto sl = [](int x) constexpr { return ((x % (2 * S + 1)) + (2 * S + 1)) % (2 * S + 1); };
        auto wave_sync = []() {
            __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
            __builtin_amdgcn_wave_barrier();
        };
        const int gy = Y0 - REACH + r; // image row of this input row
        float *const eb = e0 + (r & 1) * SET;
        float *const eL = eb, *const eV = eb + LROW;
        request(r + PF, sl(j + PF));
        // ---- row pass
        pk2 own = {ld[j].x, ld[j].y};
        if (EDGE) // replicate border: both columns of a lane outside the image take the border pixel
            own = cx < 0 ? pk2{own.x, own.x} : (cx >= w ? pk2{own.y, own.y} : own);
        if (RAW_IN_LD)
            ld[j] = make_float2(own.x, own.y);
        else
            Lraw[j] = own;
        auto raw_row = [&](int slot) { return RAW_IN_LD ? pk2{ld[slot].x, ld[slot].y} : Lraw[slot]; };
        const pk2 Lm = {strip_left(own.x), strip_left(own.y)}, Lp = {strip_right(own.x), strip_right(own.y)};
        pk2 t = zero + k0 * Lm;
        t = t + k1 * pk2{Lm.y, own.x};
        t = t + k2 * own;
        t = t + k3 * pk2{own.y, Lp.x};
        t = t + k4 * Lp;
        gr[j] = t;
        // ---- column pass: Lsmooth of row c = r - 2
        pk2 a = zero + k0 * gr[sl(j - 4)];
        a = a + k1 * gr[sl(j - 3)];
        a = a + k2 * gr[sl(j - 2)];
        a = a + k3 * gr[sl(j - 1)];
        a = a + k4 * gr[j];
        const int s_c = sl(j - 2);
        // ---- conductivity of row f = c - 1 (image row yf); reflected taps: row / column -1 is 1, h is h - 2
        const int yf = gy - 3, s_f = sl(j - 3);
        const bool first_col = EDGE && cx == 0, last_col = EDGE && cx == w - 2;
        const bool f_top = EDGE && yf == 0, f_bot = EDGE && yf == h - 1;
        pk2 cf;
        {
            // (the wavefront shifts run with every lane active - a lane masked off would hand its neighbour nothing -
            // and the border lanes choose afterwards)
            const float lft_n = strip_left(a.y), rgt_n = strip_right(a.x);
            const float lft = first_col ? a.y : lft_n, rgt = last_col ? a.x : rgt_n;
            const pk2 hd1 = pk2{a.y, rgt} - pk2{lft, a.x};
            p3[s_c] = c3 * hd1;
            p10[s_c] = c10 * hd1;
            const pk2 vd1 = (f_top || f_bot) ? zero : a - Ls[sl(s_c - 2)];
            const pk2 q3 = c3 * vd1, q10 = c10 * vd1;
            const float q3l_n = strip_left(q3.y), q3r_n = strip_right(q3.x);
            const float q3l = first_col ? q3.y : q3l_n, q3r = last_col ? q3.x : q3r_n;
            const pk2 ptop = f_top ? p3[s_c] : p3[sl(s_c - 2)], pbot = f_bot ? p3[sl(s_c - 2)] : p3[s_c];
            const pk2 lx = (ptop + p10[sl(s_c - 1)]) + pbot;
            const pk2 ly = (pk2{q3l, q3.x} + q10) + pk2{q3.y, q3r};
            if (MODG)
            {
                // blur_fused_kernel<BLUR_MODG>'s value: sqrtf(lx^2 + ly^2), 0 on the image's outermost pixels
                const pk2 sum = lx * lx + ly * ly;
                const bool row_in = yf >= 1 && yf < h - 1;
                cf = pk2{(row_in && cx >= 1) ? sqrtf(sum.x) : 0.0f, (row_in && cx + 1 < w - 1) ? sqrtf(sum.y) : 0.0f};
                if (own_cols && yf >= Y0 && yf < row_end)
                    grad_max = fmaxf(grad_max, fmaxf(cf.x, cf.y));
            }
            else
            {
                const pk2 den = one + inv * (lx * lx + ly * ly);
                cf = pk2{recip_ge1(den.x), recip_ge1(den.y)};
            }
            if (SF && own_cols && yf >= Y0 && yf < row_end)
                *reinterpret_cast<float2 *>(flow_b + (size_t)yf * w * 4 + col4) = make_float2(cf.x, cf.y);
        }
        if (MODG)
        {
            Ls[s_c] = a; // (the next rows' vertical differences; the stages below are not part of this pass)
            return;
        }
        // ---- what the lanes S columns away need of this row: Lsmooth (row c) and wa vd (row d = c - S).  The pattern's
        // rows d - S and d + S reflect at the image border (BORDER_REFLECT_101): row -t is row t, i.e. the ring slot 2 t
        // rows back when d = t < S; row h - 1 + e is row h - 1 - e, the slot 2 e rows back.
        const int yd = gy - 2 - S;           // image row of d
        const int e_bot = gy - 2 - (h - 1);  // > 0: row c lies e_bot rows below the image
        pk2 LsT = Ls[sl(s_c + 1)], LsB = a;  // rows c - 2 S and c
        if (EDGE && yd < S)
        {
#pragma unroll
            for (int tt = 0; tt < S; tt++)
                if (yd == tt)
                    LsT = tt == 0 ? a : Ls[sl(s_c - 2 * tt)];
        }
        if (EDGE && e_bot > 0)
        {
#pragma unroll
            for (int ee = 1; ee <= S; ee++)
                if (e_bot == ee)
                    LsB = Ls[sl(s_c - 2 * ee)];
        }
        const pk2 vdS = LsB - LsT;
        Ls[s_c] = a;
        const pk2 pbv = wb * vdS;
        *reinterpret_cast<pk2 *>(eL) = a;
        *reinterpret_cast<pk2 *>(eV) = wa * vdS;
        wave_sync();
        pk2 aL, aR, vL, vR;
        if (!EDGE)
        {
            aL = pk2{eL[-S], eL[-S + 1]}, aR = pk2{eL[S], eL[S + 1]};
            vL = pk2{eV[-S], eV[-S + 1]}, vR = pk2{eV[S], eV[S + 1]};
        }
        else
        {
            // columns reflect too: every lane reads where its own taps land (relative to column cx)
            const int cxi = min(max(cx, 0), w - 2);
            const int oLx = reflect101_once(cxi - S, w) - cx, oLy = reflect101_once(cxi + 1 - S, w) - cx;
            const int oRx = reflect101_once(cxi + S, w) - cx, oRy = reflect101_once(cxi + 1 + S, w) - cx;
            aL = pk2{eL[oLx], eL[oLy]}, aR = pk2{eL[oRx], eL[oRy]};
            vL = pk2{eV[oLx], eV[oLy]}, vR = pk2{eV[oRx], eV[oRy]};
        }
        // ---- diffusion steps (independent of the exchange: they fill its latency)
        if (K > 0)
        {
            C[s_f] = cf;
            const float crx = strip_right(cf.x);
            CX[s_f] = cf + pk2{cf.y, crx};         // c(x) + c(x + 1)
            CY[sl(s_f - 1)] = C[sl(s_f - 1)] + cf; // c(y) + c(y + 1) of row f - 1
            pk2 Ln = raw_row(s_f);                 // the previous step's row below the one a step works on
            // a right neighbour of column cx + 1 / a left neighbour of column cx exists
            const bool hr1 = !EDGE || cx + 2 < w, hl0 = !EDGE || cx > 0;
#pragma unroll
            for (int q = 1; q <= K; q++)
            {
                const int s_y = sl(s_f - q), yq = yf - q; // step q works on image row yq
                const pk2 Lc = q == 1 ? raw_row(s_y) : Lsave[q - 1];
                const float right = strip_right(Lc.x);
                const pk2 d = pk2{Lc.y, right} - Lc;
                pk2 xpos = CX[s_y] * d;
                xpos.y = hr1 ? xpos.y : 0.0f;
                const float xl = strip_left(xpos.y);
                const pk2 xneg = {hl0 ? xl : 0.0f, xpos.x};
                const pk2 ypos = (!EDGE || yq + 1 < h) ? CY[s_y] * (Ln - Lc) : zero;
                const pk2 yneg = (!EDGE || yq > 0) ? Fsave[q] : zero;
                const float half = 0.5f * A.T.tau[q - 1];
                const pk2 Lq = Lc + pk2{half, half} * (((xpos - xneg) + ypos) - yneg); // (xpos - xneg + ypos - yneg, left to right)
                Fsave[q] = ypos;
                if (q > 1)
                    Lsave[q - 1] = Ln;
                Ln = Lq;
                if (q == K && own_cols && yq >= Y0 && yq < row_end)
                {
                    *reinterpret_cast<float2 *>(lout_b + (size_t)yq * w * 4 + col4) = make_float2(Lq.x, Lq.y);
                    if (A.half_out) // (wave-uniform) the lane's pair of columns and this pair of rows are one half-sampled pixel
                    {
                        const float pair_sum = Lq.x + Lq.y;
                        if (yq & 1)
                            A.half_out[(size_t)blockIdx.z * A.half_stride + (size_t)(yq >> 1) * (w >> 1) + (cx >> 1)] = (half_even + pair_sum) * 0.25f;
                        else
                            half_even = pair_sum;
                    }
                }
            }
        }
        // ---- (Lx, Ly) of row d = c - S
        const pk2 hdS = aR - aL;
        paS[s_c] = wa * hdS;
        pbS[s_c] = wb * hdS;
        pk2 paT = paS[sl(s_c + 1)], paB = paS[s_c];
        if (EDGE && yd < S)
        {
#pragma unroll
            for (int tt = 0; tt < S; tt++)
                if (yd == tt)
                    paT = paS[sl(s_c - 2 * tt)];
        }
        if (EDGE && e_bot > 0)
        {
#pragma unroll
            for (int ee = 1; ee <= S; ee++)
                if (e_bot == ee)
                    paB = paS[sl(s_c - 2 * ee)];
        }
        const pk2 Lx = (paT + pbS[sl(s_c - S)]) + paB;
        const pk2 Ly = (vL + pbv) + vR;
        if (own_cols && yd >= Y0 && yd < row_end)
            *reinterpret_cast<float4 *>(lxy_b + (size_t)yd * w * 8 + col8) = make_float4(Lx.x, Ly.x, Lx.y, Ly.y);
    };
    // does any row or column this strip touches come within reach of the border?  (wave-uniform)
    const bool edge = Y0 - REACH - 4 < S || Y0 + SH + REACH + 4 > h - 1 - S || X0 - HALO - 2 < S || X0 - HALO + 130 > w - 1 - S;
    if (edge)
    {
        for (int k = 0; k < NB; k++)
        {
#pragma unroll
            for (int j = 0; j < U; j++)
                row(std::true_type{}, k * U + j, j);
        }
    }
    else
    {
        for (int k = 0; k < NB; k++)
        {
#pragma unroll
            for (int j = 0; j < U; j++)
                row(std::false_type{}, k * U + j, j);
        }
    }
    if (MODG)
    {
        // the image's maximum: non-negative floats order like their bit patterns; a wavefront's maximum, then one atomic per strip
        unsigned int bits = __float_as_uint(grad_max);
        for (int off = 32; off >= 1; off >>= 1)
            bits = max(bits, (unsigned int)__shfl_xor((int)bits, off));
        if (lane == 0 && bits)
            atomicMax(&A.hmax_bits[blockIdx.z], bits);
    }
}

// Measured and dropped: the diffusion pipeline of level_strip_kernel on its own for the FED groups a level kernel leaves (K <= 8
// steps per launch instead of nld_fused_kernel's 4: 31 -> 19 launches per chunk).  Those groups belong to the upper octaves,
// whose launches have too few strips to fill the SIMDs: 19.1 us per image against the tile kernel's 11.0.

struct levels_dev
{
    int n;
    level_info l[16];
};

// Candidate list of an image = the non-zero entries of its maxima maps, laid out tile by tile in a space-filling
// order (tile_seq: level by level, Morton order of the 64 x 24 detection tiles inside a level).  Neighbouring list entries
// are neighbouring pixels, which is what keeps the window scans of the suppression and the patch gathers of the
// descriptor inside the L2: with an arbitrary order the descriptor kernel alone pulled ~0.9 GB per image through
// the fabric, 14x the size of the pyramid it samples.  No global atomics: the offsets come from a prefix sum of the
// per-tile counts det_maxima_kernel leaves behind.
__global__ __launch_bounds__(256) void scan_tiles_kernel(const unsigned int *__restrict__ tile_counts,
                                                         const unsigned int *__restrict__ tile_seq, int n_tiles,
                                                         unsigned int *__restrict__ tile_base, unsigned int *__restrict__ n_cands)
{
    __shared__ unsigned int wsum[4], carry;
    const unsigned int b = blockIdx.x;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    if (threadIdx.x == 0)
        carry = 0;
    __syncthreads();
    for (int start = 0; start < n_tiles; start += 256)
    {
        const int i = start + threadIdx.x;
        const unsigned int tile = i < n_tiles ? tile_seq[i] : 0;
        const unsigned int v = i < n_tiles ? tile_counts[(size_t)b * n_tiles + tile] : 0;
        unsigned int incl = v; // inclusive scan inside the wavefront
        for (int off = 1; off < 64; off <<= 1)
        {
            const unsigned int t = (unsigned int)__shfl_up((int)incl, off);
            if (lane >= off)
                incl += t;
        }
        if (lane == 63)
            wsum[wv] = incl;
        __syncthreads();
        unsigned int base = carry;
        for (int j = 0; j < wv; j++)
            base += wsum[j];
        if (i < n_tiles)
            tile_base[(size_t)b * n_tiles + tile] = base + incl - v;
        __syncthreads();
        if (threadIdx.x == 0)
            carry += wsum[0] + wsum[1] + wsum[2] + wsum[3];
        __syncthreads();
    }
    if (threadIdx.x == 0)
        n_cands[b] = carry;
}

// One wavefront per tile: lane r holds the mask word of tile row r, a shuffle prefix sum of the popcounts gives each
// row its slots, and only the responses of set bits are read.  List order inside a tile: row-major.
constexpr int COLLECT_TPW = 4; // tiles per wavefront of collect_tiles_kernel
__global__ __launch_bounds__(256) void collect_tiles_kernel(const float *__restrict__ Rmax, const float2 *__restrict__ Fit,
                                                            size_t img_stride,
                                                            const unsigned long long *__restrict__ mask, size_t mask_stride,
                                                            levels_dev L, const unsigned int *__restrict__ tile_base, int n_tiles,
                                                            cand_t *__restrict__ cands, unsigned int max_cands,
                                                            const unsigned int *__restrict__ tile_counts)
{
    // (a wavefront takes COLLECT_TPW tiles, two at a time - lanes 0 .. 23 and 24 .. 47 hold the rows of one tile each -: with a
    // wavefront per tile the launch was 680 k wavefronts of a few instructions per chunk, 4.9 us per image of dispatching)
    static_assert(2 * DT_Y <= 64, "two tiles' rows per wavefront");
    const unsigned int b = blockIdx.z;
    const int lane = threadIdx.x & 63, half = lane >= DT_Y ? 1 : 0, row = lane - half * DT_Y;
    const int first = (blockIdx.x * 4 + (int)(threadIdx.x >> 6)) * COLLECT_TPW;
    for (int pair = 0; pair < COLLECT_TPW; pair += 2)
    {
        if (first + pair >= n_tiles) // (wave-uniform)
            break;
        const int tile = first + pair + half;
        bool live = lane < 2 * DT_Y && tile < n_tiles;
        // (a tile without maxima - half of them on the coarse levels - is done before its mask rows are asked for)
        live = live && tile_counts[(size_t)b * n_tiles + tile] != 0;
        if (__ballot(live) == 0)
            continue;
        int level = 0;
        while (level + 1 < L.n && tile >= L.l[level + 1].tile_off)
            level++;
        const level_info l = L.l[level];
        const int t = tile - l.tile_off;
        const int ty = t / l.tiles_x, tx = t - ty * l.tiles_x;
        const float *R = Rmax + (size_t)b * img_stride + l.off;
        const float2 *F = Fit + (size_t)b * img_stride + l.off;
        const int y = ty * DT_Y + row;
        unsigned long long m = 0;
        if (live && y < l.h)
            m = mask[(size_t)b * mask_stride + (size_t)l.mask_off + (size_t)y * l.tiles_x + tx];
        const unsigned int cnt = (unsigned int)__popcll(m);
        unsigned int incl = cnt;
        for (int off = 1; off < 64; off <<= 1)
        {
            const unsigned int v = (unsigned int)__shfl_up((int)incl, off);
            if (lane >= off)
                incl += v;
        }
        // (the second tile's rows start counting after the first tile's last row)
        const unsigned int first_total = (unsigned int)__builtin_amdgcn_readlane((int)incl, DT_Y - 1);
        unsigned int slot = (live ? tile_base[(size_t)b * n_tiles + tile] : 0u) + incl - cnt - (half ? first_total : 0u);
        while (m)
        {
            const int bit = __ffsll((long long)m) - 1;
            m &= m - 1;
            const int x = tx * BT_X + bit;
            if (slot < max_cands)
            {
                const float2 f = F[(size_t)y * l.w + x];
                cands[(size_t)b * max_cands + slot] = cand_t{level, x, y, R[(size_t)y * l.w + x], f.x, f.y};
            }
            slot++;
        }
    }
}

// Workgroup -> work item map.  Workgroups are dealt to the 8 XCDs round-robin (id % 8); mode 2 hands each XCD
// whole groups of 64 consecutive items in turn, so that list neighbours - spatial neighbours - share an L2 while
// the XCDs stay balanced (a static eighth per XCD, mode 1, left XCDs idle: the share of suppressed candidates
// varies along the list).  Mode 0 is the identity.  Returns false when there is no item for this workgroup.
__device__ __forceinline__ bool xcd_contiguous(unsigned int block, unsigned int n_items, unsigned int *item, int mode)
{
    if (mode == 0)
    {
        *item = block;
        return block < n_items;
    }
    const unsigned int x = block & 7, j = block >> 3;
    if (mode == 1)
    {
        const unsigned int chunk = (n_items + 7) / 8;
        *item = x * chunk + j;
        return j < chunk && *item < n_items;
    }
    if (mode == 2)
    {
        *item = ((j >> 6) * 8 + x) * 64 + (j & 63);
        return *item < n_items;
    }
    *item = ((j >> 8) * 8 + x) * 256 + (j & 255); // mode 3: groups of 256
    return *item < n_items;
}

// ---- AKAZEFeatures::Find_Scale_Space_Extrema's suppression (OpenCV 4.x: three passes over per-level keypoint masks; the CPU
// restatement's suppress_masks_4x quotes the rule).  Pass 1, inside a level in raster order:
// a maximum looks for the FIRST keypoint already set within sigma_size (window [y - r, y + r) x [x - r, x + r) in raster order,
// Euclidean test) - none: it is set; it is stronger: that one is cleared and it is set; otherwise it is dropped.  Pass 2, levels
// upwards: a set keypoint clears the first set keypoint of the level BELOW within sigma_size * octave step of its projection
// if it is stronger.  Pass 3, levels downwards: the same against the level ABOVE (radius: that level's sigma_size).  The
// outcome depends on the order of the turns, so the order is reproduced - in rounds instead of in sequence (DESIGN.md section
// 4.1 has the argument; the CPU restatement's suppress_masks_4x_in_rounds is this schedule and agrees with
// the sequential form on every candidate of the census images):
//   * pass 1: a maximum takes its turn once every maximum in front of it in raster order within 2 sigma_size - 1 (Chebyshev)
//     has had its own: two maxima further apart read and write disjoint windows;
//   * passes 2 and 3 clear keypoints of the OTHER level only, so the levels of a pass do not depend on each other; a keypoint
//     whose window in the other level is empty when the pass starts has no turn at all (suppress_window_kernel), and inside a
//     level a keypoint waits for the keypoints with a turn in front of it whose windows in the other level can overlap its own
//     (2 sigma_size - 1 / 2 sigma_size' * octave step - 1).
// Per pass: round 0 as one launch over all candidates (suppress_round0_kernel: whether a point can go at once is a question to
// masks the launch does not write, so test and turn are one step), then a workgroup per (image, level) on the list round 0
// left (suppress_rounds_kernel) - the candidate list is level by level (tile_seq), so a level's candidates are one range of it.
// A round there = every point still waiting tests the `pending` mask of its box, a barrier, the ready ones take their turns
// (atomics on the mask words), a barrier.  Nothing a workgroup reads is written by another one during a launch (pass 1 stays
// inside the level; in passes 2 and 3 a level's workgroup is the only one to clear bits of the level below / above, and who
// has a turn comes from `found`, not from the masks), so workgroup-scope ordering - the barriers - is all it takes: an
// agent-scope fence per round (L2 write-back + invalidate on this part) made the three launches 9 ms per 100 images.  Rounds
// are latency-bound (~8 us each): 2 - 14 per level on a 1600 x 1200 image.
constexpr int SUP_THREADS = 256;
constexpr int SUP_LDS = 2048; // list entries a workgroup keeps in LDS (two lists); a longer list continues in HBM
constexpr unsigned int SUP_READY = 0x80000000u; // list entry: x | y << 16 | this flag

__device__ __forceinline__ unsigned long long sup_window(int lo, int hi) // bits lo .. hi of a word, 0 <= lo <= hi <= 63
{
    return (~0ull << lo) & (~0ull >> (63 - hi));
}
// The suppression's masks hold 8 x 8 pixels per 64-bit word - bit (y & 7) * 8 + (x & 7) of word (y >> 3) * sup_tx + (x >> 3) -: a
// box of 15 x 8 pixels is 3 x 2 words and a window of 16 x 16 at most 3 x 3, where the maxima masks' layout (64 x 1 pixels
// per word) takes a word or two per ROW, 16 - 32 loads per point, each lane of a wavefront in a cache line of its own.
__device__ __forceinline__ size_t sup_word(const level_info &l, int x, int y)
{
    return (size_t)(y >> 3) * l.sup_tx + (x >> 3);
}
__device__ __forceinline__ unsigned long long sup_bit(int x, int y)
{
    return 1ull << (((y & 7) << 3) | (x & 7));
}
// columns lo .. hi of every row of a word / rows lo .. hi of a word (0 .. 7; nothing when hi < lo)
__device__ __forceinline__ unsigned long long sup_cols(int lo, int hi)
{
    return hi < lo ? 0ull : (unsigned long long)(((2u << hi) - 1u) & ~((1u << lo) - 1u)) * 0x0101010101010101ull;
}
__device__ __forceinline__ unsigned long long sup_rows(int lo, int hi)
{
    return hi < lo ? 0ull : sup_window(8 * lo, 8 * hi + 7);
}
// a point of this pass in front of (x, y) in raster order inside the box of half-width r that has not had its turn.  r <= 7
// (every level of the default scale space: sigma_size is 2, 3 or 4): 3 x 2 words requested without a branch - a word
// outside the box is read again and masked out
__device__ __forceinline__ bool sup_pending(const unsigned long long *Pm, const level_info &l, int x, int y, int r, bool generic)
{
    const int x0 = max(x - r, 0), x1 = min(x + r, l.w - 1), y0 = max(y - r, 0);
    const int txa = x0 >> 3, txb = x1 >> 3, tya = y0 >> 3, tyb = y >> 3;
    unsigned long long any = 0;
    if (r <= 7 && !generic)
    {
        // rows y0 .. y - 1 of the upper word row, rows 0 .. y - 1 of the lower one when there are two, and in the word row
        // of y the columns in front of x
        const unsigned long long rows_a = sup_rows(y0 - 8 * tya, min(y - 1 - 8 * tya, 7)),
                                 rows_b = tyb != tya ? sup_rows(0, y - 1 - 8 * tyb) : 0ull, row_y = sup_rows(y & 7, y & 7);
#pragma unroll
        for (int j = 0; j < 3; j++)
        {
            const int tx = min(txa + j, txb);
            const bool inside = txa + j <= txb;
            const unsigned long long wa = Pm[(size_t)tya * l.sup_tx + tx], wb = Pm[(size_t)tyb * l.sup_tx + tx];
            const unsigned long long cols = sup_cols(max(x0 - 8 * tx, 0), min(x1 - 8 * tx, 7)),
                                     front = sup_cols(max(x0 - 8 * tx, 0), min(x - 1 - 8 * tx, 7));
            any |= inside ? (wa & cols & rows_a) | (wb & ((cols & rows_b) | (front & row_y))) : 0ull;
        }
    }
    else
        for (int ty = tya; ty <= tyb; ty++)
            for (int tx = txa; tx <= txb; tx++)
            {
                const unsigned long long cols = sup_cols(max(x0 - 8 * tx, 0), min(x1 - 8 * tx, 7)),
                                         front = sup_cols(max(x0 - 8 * tx, 0), min(x - 1 - 8 * tx, 7));
                any |= Pm[(size_t)ty * l.sup_tx + tx] & ((cols & sup_rows(max(y0 - 8 * ty, 0), min(y - 1 - 8 * ty, 7))) |
                                                         (ty == tyb ? front & sup_rows(y & 7, y & 7) : 0ull));
            }
    return any != 0;
}
// find_neighbor_point: the first set bit, in raster order, of [y - r, y + r) x [x - r, x + r) within r (Euclidean) of (x, y).
// r <= 8 (sigma_size times an octave step of at most 2): at most 3 x 3 words, requested together
__device__ __forceinline__ bool sup_first_set(const unsigned long long *W, const level_info &l, int x, int y, int r, int *fx, int *fy,
                                              bool generic)
{
    const int x0 = max(x - r, 0), x1 = min(x + r, l.w) - 1, y0 = max(y - r, 0), y1 = min(y + r, l.h) - 1;
    if (x1 < x0 || y1 < y0)
        return false;
    const int txa = x0 >> 3, txb = x1 >> 3, tya = y0 >> 3, tyb = y1 >> 3;
    if (r <= 8 && !generic)
    {
        unsigned long long m[3][3], any = 0;
#pragma unroll
        for (int i = 0; i < 3; i++)
#pragma unroll
            for (int j = 0; j < 3; j++)
            {
                const int ty = min(tya + i, tyb), tx = min(txa + j, txb);
                const unsigned long long v = W[(size_t)ty * l.sup_tx + tx];
                m[i][j] = tya + i <= tyb && txa + j <= txb
                              ? v & sup_cols(max(x0 - 8 * tx, 0), min(x1 - 8 * tx, 7)) & sup_rows(max(y0 - 8 * ty, 0), min(y1 - 8 * ty, 7))
                              : 0ull;
                any |= m[i][j];
            }
        if (any == 0)
            return false;
#pragma unroll
        for (int i = 0; i < 3; i++)
        {
            if ((m[i][0] | m[i][1] | m[i][2]) == 0)
                continue;
            for (int q = 0; q < 8; q++) // the word row's pixel rows top down, 24 columns each
            {
                unsigned int row = (unsigned int)((m[i][0] >> (8 * q)) & 0xffull) | ((unsigned int)((m[i][1] >> (8 * q)) & 0xffull) << 8) |
                                   ((unsigned int)((m[i][2] >> (8 * q)) & 0xffull) << 16);
                const int yy = 8 * (tya + i) + q;
                while (row)
                {
                    const int xx = 8 * txa + __ffs((int)row) - 1;
                    row &= row - 1;
                    if ((xx - x) * (xx - x) + (yy - y) * (yy - y) <= r * r)
                    {
                        *fx = xx;
                        *fy = yy;
                        return true;
                    }
                }
            }
        }
        return false;
    }
    for (int yy = y0; yy <= y1; yy++)
        for (int xx = x0; xx <= x1; xx++)
            if ((W[sup_word(l, xx, yy)] & sup_bit(xx, yy)) && (xx - x) * (xx - x) + (yy - y) * (yy - y) <= r * r)
            {
                *fx = xx;
                *fy = yy;
                return true;
            }
    return false;
}
// the maxima masks (64 x 1 pixels per word) in the suppression's layout: one thread per 64 x 8 pixels - eight row words in, a
// byte transpose, up to eight 8 x 8 words out (a thread per 8 x 8 word asked for every row word eight times: 0.19 ms per 100 images)
__global__ __launch_bounds__(256) void suppress_tiles_kernel(const unsigned long long *__restrict__ mask, size_t mask_stride,
                                                             unsigned long long *__restrict__ out, size_t sup_stride, levels_dev L)
{
    const int rel = blockIdx.x * 256 + threadIdx.x;
    const unsigned int b = blockIdx.z;
    const level_info l = L.l[blockIdx.y];
    if (rel >= l.tiles_x * ((l.h + 7) / 8))
        return;
    const int ty = rel / l.tiles_x, wx = rel - ty * l.tiles_x;
    unsigned long long in[8];
#pragma unroll
    for (int q = 0; q < 8; q++)
        in[q] = 8 * ty + q < l.h ? mask[(size_t)b * mask_stride + (size_t)l.mask_off + (size_t)(8 * ty + q) * l.tiles_x + wx] : 0ull;
#pragma unroll
    for (int j = 0; j < 8; j++)
    {
        if (8 * wx + j >= l.sup_tx)
            break;
        unsigned long long o = 0;
#pragma unroll
        for (int q = 0; q < 8; q++)
            o |= ((in[q] >> (8 * j)) & 0xffull) << (8 * q);
        out[(size_t)b * sup_stride + (size_t)l.sup_off + (size_t)ty * l.sup_tx + 8 * wx + j] = o;
    }
}
// first list entry of every level of every image (the candidate list is level by level, tile_seq; it is complete: n <= max_cands
// was checked) - looked up once per chunk: as two dependent loads in front of every wavefront's list append it was the
// longest chain of the round-0 launches
__global__ __launch_bounds__(64) void suppress_level_first_kernel(const unsigned int *__restrict__ n_cands, unsigned int max_cands, levels_dev L,
                                                                  const unsigned int *__restrict__ tile_base,
                                                                  const unsigned int *__restrict__ tile_seq, int n_tiles,
                                                                  unsigned int *__restrict__ level_first)
{
    const unsigned int b = blockIdx.x;
    const int level = threadIdx.x;
    if (level > L.n)
        return;
    level_first[(size_t)b * (L.n + 1) + level] =
        level == L.n ? min(n_cands[b], max_cands) : tile_base[(size_t)b * n_tiles + tile_seq[L.l[level].tile_off]];
}

// Passes 2 and 3, before round 0: which keypoints have anything to do.  A keypoint whose window in the other level holds no
// keypoint when the pass starts never finds one (the pass only clears) and clears nothing: it takes no turn and nobody waits
// for it.  That is five of six keypoints - and what keeps the dependence graph of the second octave (a maximum per 55 pixels,
// boxes of 11 x 6) under its percolation threshold: 7 - 9 rounds instead of 29 - 35.  `own` are the keypoints the previous
// pass left: nothing writes them during this launch, and the launches after it ask `found` and `has`, not the masks, who has
// a turn (the pass clears keypoints of the levels next to the one whose turns it takes); found[k] = x | y << 16 | SUP_READY of the first keypoint in the window, 0 without one; `has` gets the bits of
// the keypoints that found one.
constexpr unsigned int SUP_FOUND = 0x80000000u;
template <int PASS>
__global__ __launch_bounds__(256) void suppress_window_kernel(const cand_t *__restrict__ cands, const unsigned int *__restrict__ n_cands,
                                                              unsigned int max_cands, const unsigned long long *__restrict__ own,
                                                              unsigned long long *__restrict__ has, size_t mask_stride, levels_dev L,
                                                              unsigned int *__restrict__ found, int hooks)
{
    const unsigned int b = blockIdx.z, k = blockIdx.x * 256 + threadIdx.x, n = min(n_cands[b], max_cands);
    if (k >= n)
        return;
    const cand_t c = cands[(size_t)b * max_cands + k];
    unsigned int f = 0;
    if (!((PASS == 2 && c.level == 0) || (PASS == 3 && c.level == L.n - 1))) // (no level below / above: no turns)
    {
        const level_info l = L.l[c.level];
        const size_t word = (size_t)b * mask_stride + (size_t)l.sup_off + sup_word(l, c.x, c.y);
        const unsigned long long bit = sup_bit(c.x, c.y);
        if (own[word] & bit)
        {
            const level_info lo = L.l[PASS == 2 ? c.level - 1 : c.level + 1];
            const int diff = PASS == 2 ? 1 << (l.octave - lo.octave) : 1, shift = PASS == 3 ? lo.octave - l.octave : 0;
            const int r = PASS == 2 ? l.sigma_size * diff : lo.sigma_size;
            const int px = PASS == 2 ? c.x * diff : c.x >> shift, py = PASS == 2 ? c.y * diff : c.y >> shift;
            int fx = 0, fy = 0;
            if (sup_first_set(own + (size_t)b * mask_stride + lo.sup_off, lo, px, py, r, &fx, &fy, (hooks & 1) != 0))
            {
                f = (unsigned int)fx | ((unsigned int)fy << 16) | SUP_FOUND;
                atomicOr(&has[word], bit);
            }
        }
    }
    found[(size_t)b * max_cands + k] = f;
}

// Round 0 of a pass, one THREAD per candidate over the whole chunk: whether a point can take its turn at once depends on the
// static masks only (`own`: pass 1 - the maxima, passes 2 / 3 - the keypoints suppress_window_kernel found something for).  85 -
// 99 % of the points can, and since which ones is known without looking at anything the launch writes, test and turn are
// one step here (no barrier): a ready point's window holds nothing another ready point writes.  In pass 1 such a turn is
// "set the bit" (a keypoint already set inside its window would be a point in front of it that has not had its turn); in
// passes 2 and 3 the first keypoint in the window is still the one suppress_window_kernel found.  The others are marked in
// `pend` and appended to their level's list for the rounds kernel.
template <int PASS>
__global__ __launch_bounds__(256) void suppress_round0_kernel(const cand_t *__restrict__ cands, const unsigned int *__restrict__ n_cands,
                                                              unsigned int max_cands, const float *__restrict__ Rmax, size_t img_stride,
                                                              const unsigned long long *__restrict__ own, unsigned long long *pend,
                                                              unsigned long long *kmask, size_t mask_stride, levels_dev L,
                                                              const unsigned int *__restrict__ level_first,
                                                              unsigned int *__restrict__ turns, unsigned int *__restrict__ waiting,
                                                              const unsigned int *__restrict__ found, int hooks)
{
    // (a list of the keypoints pass 1 left instead of a thread per candidate in passes 2 / 3 - a third of the wavefronts - was
    // measured: the list's one counter per image cost 0.97 ms per 100 images, and the launches on the list were no faster; moving
    // a workgroup's points with a turn to its first lanes, so that wavefronts without one leave: 0.23 -> 0.19 ms in pass 3, nothing
    // in pass 2, where a third of the candidates have one - not kept; the launches by mask WORD instead of by candidate - a thread
    // per 8 x 8 word walking its points, whole words stored, no candidate records read, no clears: 24.9 us per image against 14.9,
    // a wavefront then takes as many dependent steps as its fullest word has points; two candidates per thread, their chains side
    // by side: 14.9 -> 14.5 us per image)
    const unsigned int b = blockIdx.z, k = blockIdx.x * 256 + threadIdx.x, n = min(n_cands[b], max_cands);
    cand_t c = cand_t{0, 0, 0, 0.f, 0.f, 0.f};
    bool has_turn = k < n;
    unsigned int f = 0;
    if (has_turn)
    {
        if (PASS != 1)
        {
            f = found[(size_t)b * max_cands + k];
            has_turn = f != 0;
        }
        if (has_turn)
            c = cands[(size_t)b * max_cands + k];
    }
    // (list neighbours are of one level except where two levels' ranges meet: the level's constants by scalar loads when the
    // wavefront agrees on the level, per lane - a dozen vector loads each - otherwise)
    const int lv0 = __builtin_amdgcn_readfirstlane(c.level);
    const bool one_level = __all(!has_turn || c.level == lv0);
    bool waits = false;
    if (has_turn)
    {
        const level_info l = one_level ? L.l[lv0] : L.l[c.level];
        const size_t word = (size_t)b * mask_stride + (size_t)l.sup_off + sup_word(l, c.x, c.y);
        const unsigned long long bit = sup_bit(c.x, c.y);
        const int other = PASS == 1 ? 0 : (PASS == 2 ? -1 : 1);
        const level_info lo = one_level ? L.l[lv0 + other] : L.l[c.level + other];
        const int shift = PASS == 3 ? lo.octave - l.octave : 0;
        const int box = PASS == 3 ? 2 * lo.sigma_size * (1 << shift) - 1 : 2 * l.sigma_size - 1;
        waits = sup_pending(own + (size_t)b * mask_stride + l.sup_off, l, c.x, c.y, box, (hooks & 1) != 0);
        if (waits)
            atomicOr(&pend[word], bit);
        else if (PASS == 1)
            atomicOr(&kmask[word], bit);
        else
        {
            const int fx = (int)(f & 0xffffu), fy = (int)((f & ~SUP_FOUND) >> 16);
            if (c.response > Rmax[(size_t)b * img_stride + lo.off + (size_t)fy * lo.w + fx])
                atomicAnd(&kmask[(size_t)b * mask_stride + lo.sup_off + sup_word(lo, fx, fy)], ~sup_bit(fx, fy));
        }
    }
    // the waiting points go to their level's list (its range of `turns`: a level's candidates are one range of the candidate
    // list), one counter update per wavefront and level - the lanes of a wavefront are list neighbours, of one level or two
    const int lane = threadIdx.x & 63;
    unsigned long long todo = __ballot(waits);
    while (todo)
    {
        const int leader = __ffsll((long long)todo) - 1;
        const int lv = __builtin_amdgcn_readlane(c.level, leader);
        const unsigned long long same = __ballot(waits && c.level == lv);
        unsigned int base = 0;
        if (lane == leader)
            base = atomicAdd(&waiting[(size_t)b * L.n + lv], (unsigned int)__popcll(same));
        base = (unsigned int)__builtin_amdgcn_readlane((int)base, leader);
        if (waits && c.level == lv)
            turns[(size_t)b * 2 * max_cands + level_first[(size_t)b * (L.n + 1) + lv] + base +
                  (unsigned int)__popcll(same & ((1ull << lane) - 1ull))] = (unsigned int)c.x | ((unsigned int)c.y << 16);
        todo &= ~same;
    }
}

// The rounds after round 0: a workgroup per (image, level) on the list round 0 left - short (a tenth to a third of the level's
// points, shrinking fast), so small workgroups: with 1 024 threads each the launch held every wave slot of the device through its
// barriers and the other launch sequences' bandwidth-bound kernels beside it lost a seventh of their rate.
template <int PASS>
__global__ __launch_bounds__(SUP_THREADS) void suppress_rounds_kernel(const unsigned int *__restrict__ n_cands, unsigned int max_cands,
                                                                      const float *__restrict__ Rmax, size_t img_stride,
                                                                      unsigned long long *pend, unsigned long long *kmask, size_t mask_stride,
                                                                      levels_dev L, const unsigned int *__restrict__ level_first,
                                                                      unsigned int *turns, unsigned int *waiting, unsigned int *__restrict__ stats,
                                                                      int hooks)
{
    // pend: the points round 0 left waiting (all zero again when the launch ends); kmask: the keypoints
    // (image fastest: workgroups go to the 8 XCDs round-robin, and with the level fastest the four levels of the first octave -
    // 2/3 of all points - met on four XCDs)
    const unsigned int b = blockIdx.x, tid = threadIdx.x;
    const int level = (int)blockIdx.y + (PASS == 2 ? 1 : 0); // (pass 2: no level below level 0; pass 3: none above the last)
    const unsigned int m0 = waiting[(size_t)b * L.n + level];
    if (m0 == 0)
        return;
    const long long t_begin = stats ? (long long)wall_clock64() : 0;
    const unsigned int first = level_first[(size_t)b * (L.n + 1) + level];
    const level_info l = L.l[level], lo = L.l[PASS == 1 ? level : (PASS == 2 ? level - 1 : level + 1)];
    unsigned long long *Pm = pend + (size_t)b * mask_stride + l.sup_off, *W = kmask + (size_t)b * mask_stride + l.sup_off,
                       *Wo = kmask + (size_t)b * mask_stride + lo.sup_off;
    const float *R = Rmax + (size_t)b * img_stride + l.off, *Ro = Rmax + (size_t)b * img_stride + lo.off;
    // the windows of two points can overlap when the points are at most `box` apart; the window in the other level
    const int diff = PASS == 2 ? 1 << (l.octave - lo.octave) : 1, shift = PASS == 3 ? lo.octave - l.octave : 0;
    const int box = PASS == 3 ? 2 * lo.sigma_size * (1 << shift) - 1 : 2 * l.sigma_size - 1;
    const int r = PASS == 1 ? l.sigma_size : (PASS == 2 ? l.sigma_size * diff : lo.sigma_size);
    // the two lists of points waiting for their turn: the first SUP_LDS entries in LDS, the rest in the level's range of `turns`
    __shared__ unsigned int s_list[2][SUP_LDS];
    __shared__ unsigned int s_n[2];
    unsigned int *spill0 = turns + (size_t)b * 2 * max_cands + first, *spill1 = spill0 + max_cands;
    const unsigned int in_lds = (hooks & 2) ? 64u : (unsigned int)SUP_LDS; // (test hook sup_small_lists: the HBM part of the lists)
    auto get = [&](int which, unsigned int idx) { return idx < in_lds ? s_list[which][idx] : (which ? spill1 : spill0)[idx]; };
    auto put = [&](int which, unsigned int idx, unsigned int v) {
        if (idx < in_lds)
            s_list[which][idx] = v;
        else
            (which ? spill1 : spill0)[idx] = v;
    };
    for (unsigned int idx = tid; idx < min(m0, in_lds); idx += SUP_THREADS)
        s_list[0][idx] = spill0[idx];
    if (tid == 0)
        s_n[0] = m0;
    __syncthreads();
    int cur = 0;
    unsigned int rounds = 0;
    for (;; rounds++)
    {
        const unsigned int m = s_n[cur];
        if (m == 0)
            break;
        if (tid == 0)
            s_n[cur ^ 1] = 0;
        for (unsigned int idx = tid; idx < m; idx += SUP_THREADS) // who is ready
        {
            const unsigned int e = get(cur, idx);
            if (!sup_pending(Pm, l, (int)(e & 0xffffu), (int)(e >> 16), box, (hooks & 1) != 0))
                put(cur, idx, e | SUP_READY);
        }
        __syncthreads();
        for (unsigned int idx = tid; idx < m; idx += SUP_THREADS) // the turns
        {
            const unsigned int e = get(cur, idx);
            if (!(e & SUP_READY))
            {
                put(cur ^ 1, atomicAdd(&s_n[cur ^ 1], 1u), e);
                continue;
            }
            const int x = (int)(e & 0xffffu), y = (int)((e & ~SUP_READY) >> 16);
            const size_t word = sup_word(l, x, y);
            const unsigned long long bit = sup_bit(x, y);
            const float response = R[(size_t)y * l.w + x];
            int fx = 0, fy = 0;
            if (PASS == 1)
            {
                bool keep = true;
                if (sup_first_set(W, l, x, y, r, &fx, &fy, (hooks & 1) != 0))
                {
                    if (response > R[(size_t)fy * l.w + fx])
                        atomicAnd(&W[sup_word(l, fx, fy)], ~sup_bit(fx, fy));
                    else
                        keep = false;
                }
                if (keep)
                    atomicOr(&W[word], bit);
            }
            else
            {
                const int px = PASS == 2 ? x * diff : x >> shift, py = PASS == 2 ? y * diff : y >> shift;
                if (sup_first_set(Wo, lo, px, py, r, &fx, &fy, (hooks & 1) != 0) && response > Ro[(size_t)fy * lo.w + fx])
                    atomicAnd(&Wo[sup_word(lo, fx, fy)], ~sup_bit(fx, fy));
            }
            atomicAnd(&Pm[word], ~bit);
        }
        __syncthreads();
        cur ^= 1;
    }
    if (tid == 0)
    {
        waiting[(size_t)b * L.n + level] = 0; // (for the next pass)
        if (stats) // (OCHIP_VERBOSE=extract) rounds, 100 MHz ticks and points of this (pass, image, level)
        {
            unsigned int *o = stats + (((size_t)(PASS - 1) * gridDim.x + b) * L.n + level) * 3;
            o[0] = rounds, o[1] = (unsigned int)((long long)wall_clock64() - t_begin), o[2] = m0;
        }
    }
}

// the list's flags once the three passes are through
__global__ __launch_bounds__(256) void suppress_dead_kernel(const cand_t *__restrict__ cands, const unsigned int *__restrict__ n_cands,
                                                            unsigned int max_cands, const unsigned long long *__restrict__ kmask,
                                                            size_t mask_stride, levels_dev L, unsigned char *__restrict__ dead)
{
    const unsigned int b = blockIdx.z, k = blockIdx.x * 256 + threadIdx.x;
    if (k >= min(n_cands[b], max_cands))
        return;
    const cand_t c = cands[(size_t)b * max_cands + k];
    const level_info l = L.l[c.level];
    dead[(size_t)b * max_cands + k] = kmask[(size_t)b * mask_stride + (size_t)l.sup_off + sup_word(l, c.x, c.y)] & sup_bit(c.x, c.y) ? 0 : 1;
}

// ---- the float functions of the orientation and the descriptor, as the CPU restatement has them (oracle D2): cv::fastAtan2 in
// degrees, hal::fastAtan32f's radians = degrees * (float)(CV_PI / 180), glibc's sinf / cosf (double kernels, reduction by pi / 2)
__device__ __forceinline__ float fast_atan2_deg(float y, float x)
{
    const float scale = (float)(180.0 / 3.1415926535897932384626433832795);
    const float p1 = 0.9997878412794807f * scale, p3 = -0.3258083974640975f * scale, p5 = 0.1555786518463281f * scale,
                p7 = -0.04432655554792128f * scale;
    const float ax = fabsf(x), ay = fabsf(y);
    float a, c, c2;
    if (ax >= ay)
    {
        c = ay / (ax + 2.220446e-16f);
        c2 = c * c;
        a = (((p7 * c2 + p5) * c2 + p3) * c2 + p1) * c;
    }
    else
    {
        c = ax / (ay + 2.220446e-16f);
        c2 = c * c;
        a = 90.f - (((p7 * c2 + p5) * c2 + p3) * c2 + p1) * c;
    }
    if (x < 0)
        a = 180.f - a;
    if (y < 0)
        a = 360.f - a;
    return a;
}
#define OCHIP_DEG2RAD_F ((float)(3.1415926535897932384626433832795 / 180.0))
__device__ __forceinline__ float fast_atan2(float y, float x)
{
    return fast_atan2_deg(y, x) * OCHIP_DEG2RAD_F;
}

// glibc's sinf / cosf for |a| < 120 (sysdeps/ieee754/flt-32/s_sinf.c, s_cosf.c, sincosf.h; the restatement's libm_sinf /
// libm_cosf, which tests/test_oracle_akaze_properties.py pins to libm over every float of [0, 6.3])
__device__ __forceinline__ float libm_sinf_poly(double x, double x2, bool negated, int n)
{
    const double sgn = negated ? -1.0 : 1.0;
    if ((n & 1) == 0)
    {
        const double x3 = x * x2;
        const double s1 = 0x1.1107605230bc4p-7 + x2 * -0x1.994eb3774cf24p-13;
        const double x7 = x3 * x2;
        const double s = x + x3 * -0x1.555545995a603p-3;
        return (float)(s + x7 * s1);
    }
    const double x4 = x2 * x2;
    const double c2 = sgn * -0x1.6c087e89a359dp-10 + x2 * (sgn * 0x1.99343027bf8c3p-16);
    const double c1 = sgn * -0x1.ffffffd0c621cp-2 + x2 * (sgn * 0x1.55553e1068f19p-5);
    const double x6 = x4 * x2;
    const double c = sgn * 0x1p0 + x2 * c1;
    return (float)(c + x6 * c2);
}
__device__ __forceinline__ void libm_sincosf(float a, float *s, float *c)
{
    const unsigned int top = (__float_as_uint(a) >> 20) & 0x7ffu;
    double x = a;
    if (top < ((0x3f490fdbu >> 20) & 0x7ffu)) // |a| < pi / 4 (abstop12 of 0x1.921FB6p-1f)
    {
        const double x2 = x * x;
        if (top < ((0x39800000u >> 20) & 0x7ffu)) // |a| < 2^-12
        {
            *s = a;
            *c = 1.0f;
            return;
        }
        *s = libm_sinf_poly(x, x2, false, 0);
        *c = libm_sinf_poly(x, x2, false, 1);
        return;
    }
    const double r = x * 0x1.45F306DC9C883p+23;
    const int n = ((int)r + 0x800000) >> 24;
    x = x - n * 0x1.921FB54442D18p0;
    const double sign[4] = {1.0, -1.0, -1.0, 1.0};
    const double sg = sign[n & 3];
    const bool negated = (n & 2) != 0;
    *s = libm_sinf_poly(x * sg, x * x, negated, n);
    *c = libm_sinf_poly(x * sg, x * x, negated, n ^ 1);
}

// The candidates that survived the suppression, in list (tile) order: the descriptor kernel runs over these only, so that
// the four wavefronts of one of its workgroups are four live spatial neighbours.  One workgroup per image, ballot prefix.
// (these one-workgroup-per-image scans: 1 024 threads, eight items per thread and trip, one barrier per trip - the wavefronts'
// sums alternate between two LDS rows and every thread keeps the running total itself.  With 256 threads, one item per
// thread and three barriers per trip live_list_kernel took 105 trips for an image's 27 k candidates: 2.0 us per image)
constexpr int SCAN_THREADS = 1024, SCAN_WAVES = SCAN_THREADS / 64, SCAN_PER = 8;
// exclusive prefix of `total` over the workgroup's threads (+ what `carry` holds), `carry` advanced by the workgroup's sum
__device__ __forceinline__ unsigned int scan_workgroup(unsigned int total, unsigned int (&wsum)[2][SCAN_WAVES], int trip, unsigned int &carry)
{
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    unsigned int incl = total; // inclusive scan of the threads' totals inside the wavefront
    for (int off = 1; off < 64; off <<= 1)
    {
        const unsigned int t = (unsigned int)__shfl_up((int)incl, off);
        if (lane >= off)
            incl += t;
    }
    if (lane == 63)
        wsum[trip & 1][wv] = incl;
    __syncthreads();
    unsigned int before = 0, all = 0;
#pragma unroll
    for (int j = 0; j < SCAN_WAVES; j++)
    {
        const unsigned int v = wsum[trip & 1][j];
        before += j < wv ? v : 0u;
        all += v;
    }
    const unsigned int excl = carry + before + incl - total;
    carry += all;
    return excl;
}

__global__ __launch_bounds__(SCAN_THREADS) void live_list_kernel(const unsigned char *__restrict__ dead, const unsigned int *__restrict__ n_cands,
                                                                 unsigned int max_cands, unsigned int *__restrict__ live,
                                                                 unsigned int *__restrict__ n_live)
{
    __shared__ unsigned int wsum[2][SCAN_WAVES];
    const unsigned int b = blockIdx.x, n = min(n_cands[b], max_cands);
    unsigned int carry = 0;
    int trip = 0;
    for (unsigned int start = 0; start < n; start += SCAN_THREADS * SCAN_PER, trip++)
    {
        const unsigned int first = start + threadIdx.x * SCAN_PER;
        unsigned int alive = 0, total = 0; // bit i: candidate first + i survives
#pragma unroll
        for (int i = 0; i < SCAN_PER; i++)
        {
            const bool v = first + i < n && dead[(size_t)b * max_cands + first + i] == 0;
            alive |= v ? 1u << i : 0u;
            total += v ? 1u : 0u;
        }
        unsigned int pos = scan_workgroup(total, wsum, trip, carry);
#pragma unroll
        for (int i = 0; i < SCAN_PER; i++)
            if (alive & (1u << i))
                live[(size_t)b * max_cands + pos++] = first + i;
    }
    if (threadIdx.x == 0)
        n_live[b] = carry;
}

typedef float pkf2 __attribute__((ext_vector_type(2))); // two fp32 lanes of one packed VALU instruction


// ---- round 4: the descriptor kernel again.  The counters of round 3 had the CU's LDS pipe as busy as
// its vector issue; with both relieved (below) what bounds the kernel is the rate at which a CU's L1 looks up the cache
// lines of the gathers: ~800 line accesses per keypoint, about one per cycle.  Three changes, every sum still in the
// restatement's sequential order:
//  * orientation windows (round 6: OpenCV 4.x's form).  The 109 samples are sorted into 42 angle slices - a
//    slice's members as a 128-bit set in LDS (atomic OR), its start by a lane scan of the sets' sizes, a sample's place from
//    the members with a higher number -, and lane w sums window w: one run of the sorted, cyclic list, eight LDS reads ahead of
//    their adds.  (Rounds 4 - 5 had the 3.x form: a table of the 84 window ends gave every sample a 42-bit membership mask and
//    the window loop set EXEC to it, 109 masked adds per keypoint.)
//  * cell sums: every (cell, channel) sum is a chain of its own on its own lane (4-byte LDS reads at immediate offsets,
//    one add per step): 13 lanes' worth of cells x 3 channels = 39 lanes, the 3 x 3 grid's cells chained in pairs and the
//    4 x 4 grid's in fours so that every lane walks ~100 samples - half the adds, a quarter of the LDS cycles.
//  * gathers in image order.  Which lane fetches which lattice point is free (the LDS image is indexed by the point): the
//    points are dealt to the lanes sorted by image row, then column, for the keypoint's orientation (32 classes, host
//    table), so a gather instruction covers a band of a few image rows whose neighbouring lanes share cache lines; the
//    orientation samples likewise go row by row.
// geometry of a keypoint in its level image, shared by the three kernels (same expressions, same roundings)
struct kp_geom
{
    float kx, ky, size, xf, yf;
    int s;
};
__device__ __forceinline__ kp_geom keypoint_geometry(const cand_t &c, const level_info &l, float derivative_factor)
{
    kp_geom g;
    // ratio is a power of two: x / ratio == x * (1 / ratio) bit for bit (no result here comes near the denormals),
    // and 1 / ratio is an exponent field - three IEEE division sequences less per keypoint and kernel
    const float ratio = (float)(1 << l.octave), inv_ratio = __uint_as_float((unsigned int)(127 - l.octave) << 23);
    g.kx = ((float)c.x + c.dx) * ratio + 0.5f * (ratio - 1.0f);
    g.ky = ((float)c.y + c.dy) * ratio + 0.5f * (ratio - 1.0f);
    g.size = 2.0f * (l.esigma * derivative_factor);
    g.xf = g.kx * inv_ratio;
    g.yf = g.ky * inv_ratio;
    g.s = (int)rintf(0.5f * g.size * inv_ratio);
    return g;
}

struct gather_tab
{
    // orientation samples in image order: entry L = q | (i + 6) << 8 | (j + 6) << 16 of the L-th sample sorted by row j, then
    // column i (q = its place in the restatement's i-outer, j-inner order); 0xFFFFFFFF beyond the 109 samples
    unsigned int ori[128];
    float ori_g[128]; // its Gaussian weight
    // lattice points in image order for 32 orientation classes: 4 (a + 10) | 4 (bb + 10) << 8 | byte offset of the point's
    // record in the LDS image << 16; 0xFFFFFFFF = no point
    unsigned int lat[32][448];
    // M-LDB comparison list: bit -> byte offsets of its two cell means in the LDS array of means, a | b << 16 (entries
    // beyond 485: 0, a value against itself)
    unsigned int bit_ofs[512];
    // what lane i sums in the cell phase (describe3_kernel, "cell sums"); 3 << 11 = idle
    unsigned int chain[64];
};

// Round 5: a seventh of the vector instructions went (the lattice's coordinates from a 2 x 21-entry table in LDS instead of
// sixteen instructions per point; the upper ten window bits of three samples per v_readlane) and the kernel's time did not
// move (63.9 -> 64.3 us per image): with every lane of a gather its own cache-line lookup - 441 x 2 + 109 per keypoint, one
// per cycle and CU - the L1's tag pipe is what the kernel waits for, 40 us per image of lookups at the measured clock before
// any miss.  One lookup per point would need (Lx, Ly, Lt) interleaved in one plane: +8 .. 16 bytes per pixel for the
// HBM-bound level and determinant kernels, which costs what it saves.  The leaner issue stays (it leaves slots to the
// other launch sequences' kernels).
// One 64-thread wavefront per surviving candidate (four consecutive ones per workgroup, wave-level barriers only):
// sub-pixel position, dominant orientation, 486-bit M-LDB.
constexpr int DESC_WPB = 4; // keypoints (wavefronts) per workgroup (8: 66.6 us per image against 63.3 - sharing a CU among more list neighbours does not raise the L1 hit rate)
__global__ __launch_bounds__(64 * DESC_WPB) void describe3_kernel(const cand_t *__restrict__ cands, const unsigned int *__restrict__ n_live,
                                                       unsigned int max_cands, const unsigned int *__restrict__ live,
                                                       const float *__restrict__ Lt, const float2 *__restrict__ Lxy,
                                                       size_t img_stride, levels_dev L, float derivative_factor,
                                                       const gather_tab *__restrict__ gtab,
                                                       float *__restrict__ kp_out /*[b][max][6]*/,
                                                       unsigned long long *__restrict__ desc_out /*[b][max][8]*/,
                                                       unsigned char *__restrict__ valid_out, int remap,
                                                       unsigned long long *__restrict__ vmask, size_t mask_stride)
{
    __shared__ float vals_all[DESC_WPB][30][3]; // cell sums, then cell means; row 29 takes the store of a chain that has no cell left
    __shared__ float smp_all[DESC_WPB][441 * 3 + 1];
    __shared__ float lat_all[DESC_WPB][2][32]; // ((k - 10) cos) s and ((k - 10) sin) s, k = 0 .. 20: the lattice's coordinates
    const int lane = threadIdx.x & 63, wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    float(&vals)[30][3] = vals_all[wv];
    float *const smp = smp_all[wv];
    // the orientation's data are dead before the lattice is stored and share its LDS
    float2 *const osmp = reinterpret_cast<float2 *>(smp);                          // [109] weighted (Lx, Ly), sorted by angle slice
    unsigned long long *const oset = reinterpret_cast<unsigned long long *>(smp) + 112; // [42][2] a slice's samples as a set of their numbers
    int *const oslice = reinterpret_cast<int *>(smp) + 2 * 112 + 4 * 42;                // [43] samples in the slices below
    auto wave_sync = []() {
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront"); // LDS writes of the wave before LDS reads after
        __builtin_amdgcn_wave_barrier();
    };
    const unsigned int b = blockIdx.z;
    const unsigned int n = n_live[b];
    unsigned int kb;
    if (!xcd_contiguous(blockIdx.x, (n + DESC_WPB - 1) / DESC_WPB, &kb, remap))
        return;
    const unsigned int kl = kb * DESC_WPB + wv;
    if (kl >= n)
        return;
    const size_t slot = (size_t)b * max_cands + live[(size_t)b * max_cands + kl];
    // everything whose address is known up front is requested here, together
    const cand_t c = cands[slot];
    const unsigned int oe0 = gtab->ori[lane], oe1 = gtab->ori[lane + 64];
    const float og0 = gtab->ori_g[lane], og1 = gtab->ori_g[lane + 64];
    const unsigned int chain = gtab->chain[lane];
    unsigned int tbits[8];
#pragma unroll
    for (int wd = 0; wd < 8; wd++)
        tbits[wd] = gtab->bit_ofs[wd * 64 + lane];
    const level_info l = L.l[c.level];
    const int w = l.w, h = l.h;
    if (!(fabsf(c.dx) <= 1.0f && fabsf(c.dy) <= 1.0f))
    {
        if (lane == 0)
            valid_out[slot] = 0;
        return;
    }
    const kp_geom g = keypoint_geometry(c, l, derivative_factor);
    const float xf = g.xf, yf = g.yf;
    const float *pLt = Lt + (size_t)b * img_stride + l.off;
    const float2 *pLxy = Lxy + (size_t)b * img_stride + l.off;

    // ---- dominant orientation (OpenCV 4.x's Compute_Main_Orientation): 109 samples of the radius-6 disc around the ROUNDED
    // position, two per lane, fetched in image order; a counting sort of their angles into 42 slices - inside a slice the LATER
    // sample first (quantized_counting_sort fills each slice from its end) -; 42 windows of 7 slices, each summed in sorted order
    {
        const bool second = oe1 != 0xFFFFFFFFu; // (lanes 0..44)
        const int q0 = (int)(oe0 & 255u), i0 = (int)((oe0 >> 8) & 255u) - 6, j0 = (int)((oe0 >> 16) & 255u) - 6;
        const int q1 = (int)(oe1 & 255u), i1 = (int)((oe1 >> 8) & 255u) - 6, j1 = (int)((oe1 >> 16) & 255u) - 6;
        const int x0 = (int)rintf(xf), y0 = (int)rintf(yf);
        const int iy0 = clampi(y0 + j0 * g.s, 0, h - 1), ix0 = clampi(x0 + i0 * g.s, 0, w - 1);
        const int iy1 = clampi(y0 + j1 * g.s, 0, h - 1), ix1 = clampi(x0 + i1 * g.s, 0, w - 1);
        const float2 g0 = *reinterpret_cast<const float2 *>(reinterpret_cast<const char *>(pLxy) + (unsigned int)(iy0 * w + ix0) * 8u);
        const float2 g1 = second ? *reinterpret_cast<const float2 *>(reinterpret_cast<const char *>(pLxy) + (unsigned int)(iy1 * w + ix1) * 8u)
                                 : make_float2(0.0f, 0.0f);
        // (the slices' member sets are cleared while the loads are in flight)
        if (lane < 42)
            oset[2 * lane] = 0ull, oset[2 * lane + 1] = 0ull;
        wave_sync();
        const float rx0 = og0 * g0.x, ry0 = og0 * g0.y, rx1 = og1 * g1.x, ry1 = og1 * g1.y;
        const float ang_step = (float)(2.0 * 3.14159265358979323846 / 42);
        int k0 = (int)(fast_atan2(ry0, rx0) / ang_step), k1 = (int)(fast_atan2(ry1, rx1) / ang_step);
        k0 = (k0 < 0 || k0 >= 42) ? 0 : k0;
        k1 = (k1 < 0 || k1 >= 42) ? 0 : k1;
        // a slice's members as a 128-bit set of sample numbers (LDS atomics: whatever order they arrive in, the set is the same)
        atomicOr(&oset[2 * k0 + (q0 >> 6)], 1ull << (q0 & 63));
        if (second)
            atomicOr(&oset[2 * k1 + (q1 >> 6)], 1ull << (q1 & 63));
        wave_sync();
        // slice[k] = samples in the slices below k (lanes 0 .. 42: an exclusive scan of the sets' sizes)
        const int cnt = lane < 42 ? __popcll(oset[2 * lane]) + __popcll(oset[2 * lane + 1]) : 0;
        int incl = cnt;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1)
        {
            const int t = __shfl_up(incl, off);
            if (lane >= off)
                incl += t;
        }
        if (lane < 43)
            oslice[lane] = incl - cnt;
        wave_sync();
        // a sample's place: its slice's start + the members of the slice with a HIGHER number
        auto place = [&](int k, int q) {
            const unsigned long long mx = oset[2 * k], my = oset[2 * k + 1];
            const unsigned long long above = q < 64 ? (q == 63 ? 0ull : mx >> (q + 1)) : 0ull;
            const unsigned long long above_hi = q < 64 ? my : (q == 127 ? 0ull : my >> (q - 63));
            return oslice[k] + __popcll(above) + __popcll(above_hi);
        };
        osmp[place(k0, q0)] = make_float2(rx0, ry0);
        if (second)
            osmp[place(k1, q1)] = make_float2(rx1, ry1);
    }
    wave_sync();
    float angle;
    {
        // lane sn < 42 sums window sn: the sorted samples from slice[sn] on, slice[min(sn + 7, 42)] - slice[sn] of them and,
        // for the last six windows, slice[sn + 7 - 42] more from the start of the list - one run of the cyclic list
        const int sn = min(lane, 41), last = min(sn + 7, 42), remain = sn + 7 - 42;
        const int start = oslice[sn];
        const int len = lane < 42 ? oslice[last] - start + (remain > 0 ? oslice[remain] : 0) : 0;
        int longest = len;
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1)
            longest = max(longest, __shfl_xor(longest, off));
        longest = __builtin_amdgcn_readfirstlane(longest);
        pkf2 sum = {0.0f, 0.0f}; // (sumX, sumY) of the lane's window
        for (int t0 = 0; t0 < longest; t0 += 8)
        {
            pkf2 xy[8]; // (read eight ahead of their adds)
#pragma unroll
            for (int u = 0; u < 8; u++)
            {
                int i = start + t0 + u;
                i = i >= 109 ? i - 109 : i;
                const float2 sm = osmp[min(i, 108)];
                xy[u] = pkf2{sm.x, sm.y};
            }
#pragma unroll
            for (int u = 0; u < 8; u++)
                if (t0 + u < len)
                    sum += xy[u];
        }
        float wmag = -1.0f, wangle = 0.0f;
        if (lane < 42)
        {
            wmag = sum.x * sum.x + sum.y * sum.y;
            wangle = fast_atan2_deg(sum.y, sum.x); // (KeyPoint::angle: degrees)
        }
        // the first window with the largest norm (the loop's strict `norm > maxNorm` from window 0 on): a butterfly maximum, a
        // ballot of the lanes that hold it, the first of them
        float mx = wmag;
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1)
        {
            const float o = __shfl_xor(mx, off);
            asm("v_max_f32 %0, %0, %1" : "+v"(mx) : "v"(o)); // (no NaN here: fmaxf would canonicalise both operands first)
        }
        const int widx = __builtin_ctzll(__ballot(wmag == mx));
        const float best_angle = __uint_as_float((unsigned int)__builtin_amdgcn_readlane((int)__float_as_uint(wangle), widx));
        angle = best_angle * OCHIP_DEG2RAD_F; // what the descriptor rotates by (and the interface reports)
    }
    float si, co;
    libm_sincosf(angle, &si, &co);
    const float fs = (float)g.s;
    bool all_inside = true;
    // ---- every grid (2x2, 3x3, 4x4) samples the same rotated 21 x 21 lattice: gather it once with all lanes (7 rounds of
    // 64 points in the image order of the keypoint's orientation class: first every address, then every load - 14 in
    // flight per lane -, then the rotations and the LDS stores at the points' own places).  A lattice point lies within
    // 10 s (|cos| + |sin|) < 14.2 s pixels of (xf, yf) in x and in y; the detector only keeps extrema whose window is
    // inside the level image (Find_Scale_Space_Extrema's margin), so the whole-patch test below holds for every keypoint
    // and the per-point tests of the restatement are only compiled for the case that it does not.
    {
        const int cls = __builtin_amdgcn_readfirstlane(min(31, (int)(angle * 5.0929581789406507f))); // 32 / (2 pi)
        const unsigned int *order = gtab->lat[cls];
        unsigned int pe[7];
#pragma unroll
        for (int t = 0; t < 7; t++)
            pe[t] = order[lane + 64 * t];
        // a lattice point (a, bb) sits at yf + ((bb s) cos + (a s) sin), xf + ((-bb s) sin + (a s) cos) - the subset sampler's
        // expression (MLDB_Descriptor_Subset_Invoker: l * scale * co with an integer scale): the four products take 21 values each
        // per keypoint ((-bb s) sin = -((bb s) sin)) - formed once here by 21 lanes, read by every point from LDS at the byte
        // offsets the table carries, instead of 16 vector instructions per point
        if (lane < 21)
        {
            const float fk = (float)((lane - 10) * g.s);
            lat_all[wv][0][lane] = fk * co;
            lat_all[wv][1][lane] = fk * si;
        }
        wave_sync(); // the orientation's LDS has been read by every lane
        const char *const lat_b = reinterpret_cast<const char *>(&lat_all[wv][0][0]);
        auto lattice_at = [&](unsigned int e, float *sy, float *sx) {
            const unsigned int oa = e & 0x7Cu, ob = (e >> 8) & 0x7Cu; // 4 (a + 10), 4 (bb + 10)
            const float ca = *reinterpret_cast<const float *>(lat_b + oa), sa = *reinterpret_cast<const float *>(lat_b + 128 + oa);
            const float cb = *reinterpret_cast<const float *>(lat_b + ob), sb = *reinterpret_cast<const float *>(lat_b + 128 + ob);
            *sy = yf + (cb + sa);
            *sx = xf + (-sb + ca);
        };
        const float reach = 14.2f * fs + 2.0f;
        const bool patch_inside = xf - reach >= 0.0f && yf - reach >= 0.0f && xf + reach <= (float)(w - 1) && yf + reach <= (float)(h - 1);
        float ri[7], rx[7], ry[7];
        char *const smp_b = reinterpret_cast<char *>(smp);
        if (patch_inside) // (wave-uniform)
        {
#pragma unroll
            for (int t = 0; t < 7; t++)
            {
                float sy, sx;
                lattice_at(pe[t], &sy, &sx);
                const int y1 = (int)rintf(sy), x1 = (int)rintf(sx);
                // 32-bit byte offsets from a wave-uniform base (a level plane is far below 2^29 pixels): the loads take
                // the base from SGPRs and the offset from one VGPR, no 64-bit address arithmetic per lane
                const unsigned int o = pe[t] != 0xFFFFFFFFu ? (unsigned int)(y1 * w + x1) : 0u;
                // (loads that bypass the L1 - nt / sc1 - were measured: 94 us per image against 63; the kernel sits at the rate the
                // L2 delivers lines to the L1s, ~17 TB/s chip-wide, and what hits in the L1 is what keeps it there)
                ri[t] = *reinterpret_cast<const float *>(reinterpret_cast<const char *>(pLt) + o * 4u);
                const float2 gr = *reinterpret_cast<const float2 *>(reinterpret_cast<const char *>(pLxy) + o * 8u);
                rx[t] = gr.x;
                ry[t] = gr.y;
            }
#pragma unroll
            for (int t = 0; t < 7; t++)
                if (pe[t] != 0xFFFFFFFFu)
                {
                    float *rec = reinterpret_cast<float *>(smp_b + (pe[t] >> 16));
                    rec[0] = ri[t];
                    rec[1] = -rx[t] * si + ry[t] * co;
                    rec[2] = rx[t] * co + ry[t] * si;
                }
        }
        else
        {
            bool inside[7];
#pragma unroll
            for (int t = 0; t < 7; t++)
            {
                float sy, sx;
                lattice_at(pe[t], &sy, &sx);
                const int y1 = (int)rintf(sy), x1 = (int)rintf(sx);
                inside[t] = pe[t] != 0xFFFFFFFFu && !(x1 < 0 || y1 < 0 || x1 >= w || y1 >= h);
                const unsigned int o = inside[t] ? (unsigned int)(y1 * w + x1) : 0u;
                ri[t] = *reinterpret_cast<const float *>(reinterpret_cast<const char *>(pLt) + o * 4u);
                const float2 gr = *reinterpret_cast<const float2 *>(reinterpret_cast<const char *>(pLxy) + o * 8u);
                rx[t] = gr.x;
                ry[t] = gr.y;
            }
#pragma unroll
            for (int t = 0; t < 7; t++)
                if (pe[t] != 0xFFFFFFFFu)
                {
                    const float rry = rx[t] * co + ry[t] * si, rrx = -rx[t] * si + ry[t] * co;
                    // samples outside the image are stored as +0: x + 0 == x (only a -0 sum would turn +0, which no
                    // `>` comparison of the descriptor can see), so the skip of the restatement needs no branch in the sums
                    float *rec = reinterpret_cast<float *>(smp_b + (pe[t] >> 16));
                    rec[0] = inside[t] ? ri[t] : 0.0f;
                    rec[1] = inside[t] ? rrx : 0.0f;
                    rec[2] = inside[t] ? rry : 0.0f;
                    all_inside = all_inside && inside[t];
                }
        }
    }
    all_inside = __all(all_inside) != 0;
    wave_sync();
    // ---- cell sums.  chain word (host table): lattice point of the first sample | channel << 9 | class << 11 | first cell <<
    // 13 | "the pair's second cell follows the first in memory" << 18.  Classes: 0 = a 10 x 10 cell of the 2 x 2 grid,
    // 1 = two 7 x 7 cells of the 3 x 3 grid one after the other (the ninth alone), 2 = four 5 x 5 cells of the 4 x 4 grid,
    // 3 = idle lane.  All lanes execute the same 100 reads at immediate offsets 12 t from their own pointer; where a
    // lane's walk does not continue 12 bytes on (its cell's row ends, its next cell begins) the pointer takes the
    // difference: one add of a per-lane constant at the steps where some class turns, 32 in all.
    {
        const int cls = (int)((chain >> 11) & 3u), ch = (int)((chain >> 9) & 3u);
        if (cls != 3)
        {
            const char *pb = reinterpret_cast<const char *>(smp + 3 * (chain & 511u) + ch);
            float *sp = &vals[(chain >> 13) & 31u][ch];
            const bool big = cls == 0, med = cls == 1, sml = cls == 2;
            const int r5row = sml ? 192 : 0, r5cell = sml ? -1008 : 0;
            const int r10row = sml ? 192 : (big ? 132 : 0), r10cell = sml ? -1008 : (big ? 132 : 0);
            const int r7row = med ? 168 : 0, r7cell = (med && !((chain >> 18) & 1u)) ? -1512 : 0;
            // a chain whose cells are used up stores into the spare row
            const int sp_step = (med && ((chain >> 13) & 31u) == 12u) ? (29 - 12) * 3 : 3;
            float acc = 0.0f;
#pragma unroll
            for (int t = 0; t < 100; t++)
            {
                acc = acc + *reinterpret_cast<const float *>(pb + 12 * t);
                const bool end5 = t % 25 == 24, end7 = t == 48 || t == 97, end10 = t == 99;
                if (end5 || end10)
                {
                    if (sml || (end10 && big))
                    {
                        *sp = acc;
                        sp += sp_step;
                        acc = 0.0f;
                    }
                }
                if (end7)
                {
                    if (med)
                    {
                        *sp = acc;
                        sp += sp_step;
                        acc = 0.0f;
                    }
                }
                if (t % 5 == 4 && t != 99)
                    pb += (t % 10 == 4) ? ((t == 24 || t == 74) ? r5cell : r5row) : (t == 49 ? r10cell : r10row);
                if (t == 48)
                    pb += r7cell;
                else if (t < 97 && (t % 49) % 7 == 6)
                    pb += r7row;
            }
        }
    }
    wave_sync();
    // (the subset path compares the cells' SUMS: no means)
    for (int wd = 0; wd < 8; wd++)
    {
        // (entries beyond bit 485 compare a value with itself)
        const unsigned int e = tbits[wd];
        const char *vb = reinterpret_cast<const char *>(&vals[0][0]);
        const bool on = *reinterpret_cast<const float *>(vb + (e & 0xFFFFu)) > *reinterpret_cast<const float *>(vb + (e >> 16));
        const unsigned long long word = __ballot(on);
        if (lane == 0)
            desc_out[slot * 8 + wd] = word;
    }
    if (lane == 0)
    {
        float *o = kp_out + slot * 6;
        o[0] = g.kx;
        o[1] = g.ky;
        o[2] = g.size;
        o[3] = angle;
        o[4] = c.response;
        o[5] = (float)c.level;
        valid_out[slot] = 1;
        // the keypoint's bit in the image's (level, y, x)-ordered mask: its rank there is its place in the output
        atomicOr(&vmask[(size_t)b * mask_stride + (size_t)l.mask_off + (size_t)c.y * l.tiles_x + (c.x >> 6)],
                 1ull << (c.x & 63));
    }
}

// The keypoints leave the device in AKAZE's detection order (level, then row, then column of the extremum - the order
// of Find_Scale_Space_Extrema's loops), not in the tile order of the candidate list: the reference sorts them by
// response with an unstable std::sort, whose result depends on the order it starts from whenever two responses are
// equal (918 of the 1 000 rendered C3 views have such ties).  The maxima-mask layout [level][row][64-pixel word] IS
// that order, so a keypoint's output position is the number of valid bits before its own: rank_scan_kernel turns the
// valid mask into per-word exclusive counts (one workgroup per image, 8 words per thread and pass), and
// compact_ordered_kernel places every valid slot.  Entries beyond max_kp are dropped; counts[b] is the number found
// and the host reports the overflow.
__global__ __launch_bounds__(SCAN_THREADS) void rank_scan_kernel(const unsigned long long *__restrict__ vmask, size_t mask_stride,
                                                                 unsigned int *__restrict__ wbase, unsigned int *__restrict__ counts)
{
    constexpr int PER = SCAN_PER;
    __shared__ unsigned int wsum[2][SCAN_WAVES];
    const unsigned int b = blockIdx.x;
    const unsigned long long *M = vmask + (size_t)b * mask_stride;
    unsigned int *Wb = wbase + (size_t)b * mask_stride;
    unsigned int carry = 0;
    int trip = 0;
    for (size_t start = 0; start < mask_stride; start += (size_t)SCAN_THREADS * PER, trip++)
    {
        const size_t first = start + (size_t)threadIdx.x * PER;
        unsigned int pc[PER], total = 0;
#pragma unroll
        for (int i = 0; i < PER; i++)
        {
            pc[i] = first + i < mask_stride ? (unsigned int)__popcll(M[first + i]) : 0u;
            total += pc[i];
        }
        unsigned int base = scan_workgroup(total, wsum, trip, carry);
#pragma unroll
        for (int i = 0; i < PER; i++)
        {
            if (first + i < mask_stride)
                Wb[first + i] = base;
            base += pc[i];
        }
    }
    if (threadIdx.x == 0)
        counts[b] = carry;
}

__global__ __launch_bounds__(256) void compact_ordered_kernel(const unsigned char *__restrict__ valid, const cand_t *__restrict__ cands,
                                                              const unsigned int *__restrict__ n_cands, unsigned int max_cands,
                                                              const unsigned long long *__restrict__ vmask,
                                                              const unsigned int *__restrict__ wbase, size_t mask_stride,
                                                              levels_dev L, const float *__restrict__ kp,
                                                              const unsigned long long *__restrict__ desc, float *__restrict__ kp_out,
                                                              unsigned long long *__restrict__ desc_out, unsigned int max_kp)
{
    const unsigned int b = blockIdx.z, k = blockIdx.x * 256 + threadIdx.x;
    if (k >= min(n_cands[b], max_cands))
        return;
    const size_t slot = (size_t)b * max_cands + k;
    if (!valid[slot])
        return;
    const cand_t c = cands[slot];
    const level_info l = L.l[c.level];
    const size_t word = (size_t)b * mask_stride + (size_t)l.mask_off + (size_t)c.y * l.tiles_x + (c.x >> 6);
    const unsigned int pos = wbase[word] + (unsigned int)__popcll(vmask[word] & ((1ull << (c.x & 63)) - 1ull));
    if (pos < max_kp)
    {
        const size_t o = (size_t)b * max_kp + pos;
#pragma unroll
        for (int i = 0; i < 6; i++)
            kp_out[o * 6 + i] = kp[slot * 6 + i];
#pragma unroll
        for (int i = 0; i < 8; i++)
            desc_out[o * 8 + i] = desc[slot * 8 + i];
    }
}

// ---------------------------------------------------------------------------------------- host side
std::vector<float> gaussian_taps(float sigma)
{
    int ksize = (int)std::ceil(2.0f * (1.0f + (sigma - 0.8f) / 0.3f));
    if ((ksize % 2) == 0)
        ksize += 1;
    if (ksize < 1)
        ksize = 1;
    std::vector<double> k(ksize);
    double sum = 0;
    const int r = ksize / 2;
    for (int i = 0; i < ksize; i++)
    {
        const double x = i - r;
        k[i] = std::exp(-0.5 * x * x / ((double)sigma * sigma));
        sum += k[i];
    }
    std::vector<float> out(ksize);
    for (int i = 0; i < ksize; i++)
        out[i] = (float)(k[i] / sum);
    return out;
}

bool is_prime(int n)
{
    if (n < 2)
        return false;
    for (int d = 2; d * d <= n; d++)
        if (n % d == 0)
            return false;
    return true;
}

std::vector<float> fed_taus(float T, float tau_max) // FED cycle for stopping time T, kappa-cycle reordering
{
    const double t = (double)T;
    const int n = (int)(std::ceil(std::sqrt(3.0 * t / tau_max + 0.25) - 0.5 - 1.0e-8) + 0.5);
    std::vector<float> tau;
    if (n <= 0)
        return tau;
    const double scale = 3.0 * t / (tau_max * (double)(n * (n + 1)));
    const double c = 1.0 / (4.0 * n + 2.0), d = scale * tau_max / 2.0;
    std::vector<double> tauh(n);
    for (int k = 0; k < n; k++)
    {
        const double hh = std::cos(M_PI * (2.0 * k + 1.0) * c);
        tauh[k] = d / (hh * hh);
    }
    tau.resize(n);
    const int kappa = n / 2;
    int prime = n + 1;
    while (!is_prime(prime))
        prime++;
    for (int k = 0, l = 0; l < n; ++k, ++l)
    {
        int index = 0;
        while ((index = ((k + 1) * kappa) % prime - 1) >= n)
            k++;
        tau[l] = (float)tauh[index];
    }
    return tau;
}

struct area_tab
{
    std::vector<int> off, si;
    std::vector<float> alpha;
};
// cv::resize INTER_AREA decimation table (computeResizeAreaTab).  `scale` is cv::resize's own scale_x = 1. / inv_scale_x - with
// inv_scale_x the fx it was called with (extract_features passes the FLOAT 1600 / max side as a double: 1 / 0.4000000059604645 is
// not 2.5, and the taps' weights differ in their last bits) or dsize / ssize when it was given a size
area_tab area_table(int ssize, int dsize, double scale)
{
    area_tab t;
    for (int dx = 0; dx < dsize; dx++)
    {
        t.off.push_back((int)t.si.size());
        const double fsx1 = dx * scale, fsx2 = fsx1 + scale;
        const double cell = std::min(scale, ssize - fsx1);
        int sx1 = (int)std::ceil(fsx1), sx2 = (int)std::floor(fsx2);
        sx2 = std::min(sx2, ssize - 1);
        sx1 = std::min(sx1, sx2);
        if (sx1 - fsx1 > 1e-3)
        {
            t.si.push_back(sx1 - 1);
            t.alpha.push_back((float)((sx1 - fsx1) / cell));
        }
        for (int sx = sx1; sx < sx2; sx++)
        {
            t.si.push_back(sx);
            t.alpha.push_back((float)(1.0 / cell));
        }
        if (fsx2 - sx2 > 1e-3)
        {
            t.si.push_back(sx2);
            t.alpha.push_back((float)(std::min(std::min(fsx2 - sx2, 1.0), cell) / cell));
        }
    }
    t.off.push_back((int)t.si.size());
    return t;
}

template <typename T>
int up(ochip_ctx *ctx, std::vector<std::pair<void *, size_t>> &allocs, T **dst, const T *src, size_t n)
{
    size_t got = 0;
    void *d = ochip_pool_get(ctx, (n ? n : 1) * sizeof(T), &got);
    if (!d)
        return ochip_fail(ctx, OCHIP_ENOMEM, "device allocation of %zu bytes failed in akaze", n * sizeof(T));
    allocs.emplace_back(d, got);
    // on the context's own stream (not the device's default stream, where the uploads of all contexts would queue up
    // behind each other); the wait keeps the caller's buffer semantics of a synchronous copy
    if (src && n)
    {
        if (hipMemcpyAsync(d, src, n * sizeof(T), hipMemcpyHostToDevice, ctx->stream) != hipSuccess ||
            ochip_stream_wait(ctx, ctx->stream) != hipSuccess)
            return ochip_fail(ctx, OCHIP_EHIP, "hipMemcpy failed in akaze");
    }
    *dst = (T *)d;
    return OCHIP_OK;
}

// ---- synthetic views (test / benchmark DATA, not part of the hot path): a jittered ground lattice of
// Gaussian blobs on the plane z = a x + b y, seen through a pinhole camera.  Every view of one seed shows
// the same ground, so features extracted from different views really correspond.
__device__ __forceinline__ uint32_t hash32(uint32_t x)
{
    x ^= x >> 16;
    x *= 0x7feb352dU;
    x ^= x >> 15;
    x *= 0x846ca68bU;
    x ^= x >> 16;
    return x;
}
__device__ __forceinline__ float hashf(uint32_t x) // [0, 1)
{
    return (float)(hash32(x) >> 8) * (1.0f / 16777216.0f);
}
struct synth_cam
{
    float pos[3], q[4]; // camera -> world quaternion x y z w
};
__global__ void render_views_kernel(uint8_t *__restrict__ out, int w, int h, const synth_cam *__restrict__ cams, float f,
                                    float ppx, float ppy, float pa, float pb, float x0, float y0, float spacing,
                                    uint32_t seed)
{
    const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y;
    if (x >= w)
        return;
    const synth_cam c = cams[blockIdx.z];
    // world ray of the pixel centre (Eigen _transformVector of the unit camera ray)
    const float ux = ((float)x - ppx) / f, uy = ((float)y - ppy) / f;
    const float qx = c.q[0], qy = c.q[1], qz = c.q[2], qw = c.q[3];
    float uvx = qy * 1.0f - qz * uy, uvy = qz * ux - qx * 1.0f, uvz = qx * uy - qy * ux;
    uvx += uvx;
    uvy += uvy;
    uvz += uvz;
    const float dx = ux + qw * uvx + (qy * uvz - qz * uvy), dy = uy + qw * uvy + (qz * uvx - qx * uvz),
                dz = 1.0f + qw * uvz + (qx * uvy - qy * uvx);
    const float t = (pa * c.pos[0] + pb * c.pos[1] - c.pos[2]) / (dz - pa * dx - pb * dy);
    const float gx = c.pos[0] + t * dx, gy = c.pos[1] + t * dy;
    const int ci = (int)floorf((gx - x0) / spacing), cj = (int)floorf((gy - y0) / spacing);
    float v = 0.5f;
    for (int di = -1; di <= 1; di++)
        for (int dj = -1; dj <= 1; dj++)
        {
            const uint32_t id = (uint32_t)(ci + di) * 73856093u ^ (uint32_t)(cj + dj) * 19349663u ^ seed;
            const float bx = x0 + ((float)(ci + di) + 0.5f + 0.6f * (hashf(id) - 0.5f)) * spacing;
            const float by = y0 + ((float)(cj + dj) + 0.5f + 0.6f * (hashf(id + 1) - 0.5f)) * spacing;
            const float amp = (0.25f + 0.5f * hashf(id + 2)) * ((hash32(id + 3) & 1) ? 1.0f : -1.0f);
            const float sg = spacing * (0.10f + 0.14f * hashf(id + 4));
            const float r2 = (gx - bx) * (gx - bx) + (gy - by) * (gy - by);
            v += amp * __expf(-r2 / (2.0f * sg * sg));
        }
    const float o = fminf(255.0f, fmaxf(0.0f, v * 255.0f));
    const uint8_t g = (uint8_t)o;
    uint8_t *p = out + ((size_t)blockIdx.z * w * h + (size_t)y * w + x) * 3;
    p[0] = g;
    p[1] = g;
    p[2] = g;
}

int akaze_run(ochip_ctx *ctx, const uint8_t *images_bgr, bool on_device, uint32_t n_images, int width, int height,
              uint32_t max_kp, float *kp6, uint64_t *desc, uint32_t *counts, int *work_wh,
              const ochip_feature_lists *lists = nullptr, double nms_radius = 8.0);

} // namespace

extern "C"
{

int ochip_akaze_batch(ochip_ctx *ctx, const uint8_t *images_bgr, uint32_t n_images, int width, int height,
                      uint32_t max_kp, float *kp6, uint64_t *desc, uint32_t *counts, int *work_wh)
{
    return akaze_run(ctx, images_bgr, false, n_images, width, height, max_kp, kp6, desc, counts, work_wh);
}

int ochip_akaze_batch_dev(ochip_ctx *ctx, const uint8_t *images_bgr_dev, uint32_t n_images, int width, int height,
                          uint32_t max_kp, float *kp6, uint64_t *desc, uint32_t *counts, int *work_wh)
{
    return akaze_run(ctx, images_bgr_dev, true, n_images, width, height, max_kp, kp6, desc, counts, work_wh);
}

int ochip_akaze_features(ochip_ctx *ctx, const uint8_t *images_bgr, uint32_t n_images, int width, int height, uint32_t max_kp,
                         double nms_radius, uint32_t *counts, const ochip_feature_lists *lists, int *work_wh)
{
    if (!lists || !lists->records || !lists->response || !lists->slot || !lists->num_sparse || !lists->conflict)
        return OCHIP_EINVAL;
    return akaze_run(ctx, images_bgr, false, n_images, width, height, max_kp, nullptr, nullptr, counts, work_wh, lists, nms_radius);
}

int ochip_akaze_features_dev(ochip_ctx *ctx, const uint8_t *images_bgr_dev, uint32_t n_images, int width, int height, uint32_t max_kp,
                             double nms_radius, uint32_t *counts, const ochip_feature_lists *lists, int *work_wh)
{
    if (!lists || !lists->records || !lists->response || !lists->slot || !lists->num_sparse || !lists->conflict)
        return OCHIP_EINVAL;
    return akaze_run(ctx, images_bgr_dev, true, n_images, width, height, max_kp, nullptr, nullptr, counts, work_wh, lists, nms_radius);
}

int ochip_synth_views_alloc(ochip_ctx *ctx, uint32_t n_images, int width, int height, uint8_t **images_dev)
{
    if (!ctx || !images_dev)
        return OCHIP_EINVAL;
    OCHIP_HIP(ctx, hipSetDevice(ctx->device));
    *images_dev = nullptr;
    if (hipMalloc((void **)images_dev, (size_t)(n_images ? n_images : 1) * width * height * 3) != hipSuccess)
        return ochip_fail(ctx, OCHIP_ENOMEM, "hipMalloc(%zu) for synthetic views failed", (size_t)n_images * width * height * 3);
    return OCHIP_OK;
}

int ochip_synth_views_read(ochip_ctx *ctx, const uint8_t *images_dev, uint32_t index, int width, int height,
                           uint8_t *host_out)
{
    if (!ctx || !images_dev || !host_out)
        return OCHIP_EINVAL;
    OCHIP_HIP(ctx, hipSetDevice(ctx->device));
    const size_t bytes = (size_t)width * height * 3;
    OCHIP_HIP(ctx, hipMemcpy(host_out, images_dev + (size_t)index * bytes, bytes, hipMemcpyDeviceToHost));
    return OCHIP_OK;
}

void ochip_synth_views_free(ochip_ctx *ctx, uint8_t *images_dev)
{
    if (ctx && images_dev)
    {
        (void)hipSetDevice(ctx->device);
        (void)hipFree(images_dev);
    }
}

// cams: n_images x {pos3, quat4 (x y z w)} doubles; model3 = {f, ppx, ppy}; plane2 = {a, b} of z = a x + b y;
// lattice3 = {x0, y0, spacing} of the blob lattice.  Writes n_images x height x width x 3 bytes at
// images_dev + first_image * height * width * 3.
int ochip_synth_render_views(ochip_ctx *ctx, uint8_t *images_dev, uint32_t first_image, uint32_t n_images, int width,
                             int height, const double *cams, const double *model3, const double *plane2,
                             const double *lattice3, uint32_t seed)
{
    if (!ctx || !images_dev || !cams || !model3 || !plane2 || !lattice3)
        return OCHIP_EINVAL;
    if (n_images == 0)
        return OCHIP_OK;
    OCHIP_HIP(ctx, hipSetDevice(ctx->device));
    std::vector<synth_cam> hc(n_images);
    for (uint32_t i = 0; i < n_images; i++)
    {
        for (int k = 0; k < 3; k++)
            hc[i].pos[k] = (float)cams[7 * i + k];
        for (int k = 0; k < 4; k++)
            hc[i].q[k] = (float)cams[7 * i + 3 + k];
    }
    synth_cam *dc = nullptr;
    if (hipMalloc((void **)&dc, n_images * sizeof(synth_cam)) != hipSuccess)
        return ochip_fail(ctx, OCHIP_ENOMEM, "hipMalloc failed");
    hipError_t e = hipMemcpy(dc, hc.data(), n_images * sizeof(synth_cam), hipMemcpyHostToDevice);
    if (e == hipSuccess)
    {
        hipLaunchKernelGGL(render_views_kernel, dim3((width + 255) / 256, height, n_images), dim3(256), 0, ctx->stream,
                           images_dev + (size_t)first_image * width * height * 3, width, height, (const synth_cam *)dc,
                           (float)model3[0], (float)model3[1], (float)model3[2], (float)plane2[0], (float)plane2[1],
                           (float)lattice3[0], (float)lattice3[1], (float)lattice3[2], seed);
        e = ochip_stream_wait(ctx, ctx->stream);
    }
    (void)hipFree(dc);
    if (e != hipSuccess)
        return ochip_fail(ctx, OCHIP_EHIP, "ochip_synth_render_views: %s", hipGetErrorString(e));
    return OCHIP_OK;
}

} // extern "C"

namespace
{

// gather_tab: which lane fetches which sample (any assignment is correct - the LDS images are indexed by the sample; this
// one makes the lanes of a gather instruction neighbours in the image, so that they share cache lines).
const gather_tab &host_gather_tab()
{
    static const gather_tab G = []() {
        gather_tab T{};
        // the orientation samples' weights: OpenCV's literal table gauss25 (Sample_Derivative_Response_Radius6; the Gaussian of
        // sigma 2.5 printed to eight decimals with pi = 3.14159), weight = gauss25[|i|][|j|]
        static const float gauss25[7][7] = {{0.02546481f, 0.02350698f, 0.01849125f, 0.01239505f, 0.00708017f, 0.00344629f, 0.00142946f},
                                            {0.02350698f, 0.02169968f, 0.01706957f, 0.01144208f, 0.00653582f, 0.00318132f, 0.00131956f},
                                            {0.01849125f, 0.01706957f, 0.01342740f, 0.00900066f, 0.00514126f, 0.00250252f, 0.00103800f},
                                            {0.01239505f, 0.01144208f, 0.00900066f, 0.00603332f, 0.00344629f, 0.00167749f, 0.00069579f},
                                            {0.00708017f, 0.00653582f, 0.00514126f, 0.00344629f, 0.00196855f, 0.00095820f, 0.00039744f},
                                            {0.00344629f, 0.00318132f, 0.00250252f, 0.00167749f, 0.00095820f, 0.00046640f, 0.00019346f},
                                            {0.00142946f, 0.00131956f, 0.00103800f, 0.00069579f, 0.00039744f, 0.00019346f, 0.00008024f}};
        std::vector<float> gw(169);
        for (int i = -6; i <= 6; i++)
            for (int j = -6; j <= 6; j++)
                gw[(i + 6) * 13 + (j + 6)] = gauss25[std::abs(i)][std::abs(j)];
        // orientation samples: the restatement walks i (x) outer, j (y) inner; image order is j outer, i inner
        struct os
        {
            int q, i, j;
        };
        std::vector<os> S;
        int q = 0;
        for (int i = -6; i <= 6; i++)
            for (int j = -6; j <= 6; j++)
                if (i * i + j * j < 36)
                    S.push_back({q++, i, j});
        std::sort(S.begin(), S.end(), [](const os &a, const os &b) { return a.j != b.j ? a.j < b.j : a.i < b.i; });
        for (int L = 0; L < 128; L++)
        {
            T.ori[L] = 0xFFFFFFFFu;
            T.ori_g[L] = 0.0f;
            if (L < (int)S.size())
            {
                T.ori[L] = (unsigned int)S[L].q | ((unsigned int)(S[L].i + 6) << 8) | ((unsigned int)(S[L].j + 6) << 16);
                T.ori_g[L] = gw[(size_t)(S[L].i + 6) * 13 + (S[L].j + 6)];
            }
        }
        // lattice points (a, bb) land at (x, y) = xf + s (-bb sin + a cos), yf + s (bb cos + a sin): sorted by image row
        // (rounded for a typical spacing of 3 pixels), then by column, for the middle of each of 32 orientation classes
        for (int c = 0; c < 32; c++)
        {
            const double th = (c + 0.5) * (2.0 * M_PI / 32.0), si = std::sin(th), co = std::cos(th);
            struct lp
            {
                long row;
                double x;
                int a, bb;
            };
            std::vector<lp> P;
            for (int a = -10; a <= 10; a++)
                for (int bb = -10; bb <= 10; bb++)
                    P.push_back({std::lround(3.0 * (bb * co + a * si)), -bb * si + a * co, a, bb});
            std::sort(P.begin(), P.end(), [](const lp &u, const lp &v) { return u.row != v.row ? u.row < v.row : u.x < v.x; });
            for (int k = 0; k < 448; k++)
                T.lat[c][k] = k < (int)P.size() ? (unsigned int)(4 * (P[k].a + 10)) | ((unsigned int)(4 * (P[k].bb + 10)) << 8) |
                                                      ((unsigned int)(((P[k].a + 10) * 21 + (P[k].bb + 10)) * 12) << 16)
                                                : 0xFFFFFFFFu;
        }
        // the comparison list in the order the descriptor's bits are written.  The reference asks for descriptor_size = 486, and
        // any size other than 0 takes OpenCV's subset path: generateDescriptorSubsample (AKAZEFeatures.cpp) draws the 162 cell
        // pairs in the order of cv::RNG(1024) (state = (uint64)(unsigned)state * 4164903690U + (unsigned)(state >> 32); rng(N) =
        // next() % N; the first six picks forced to rows 0 .. 5 after the draw; the row picked is overwritten by the last live
        // one), bit 3 i + c = pick i in channel c of (Lt, rx co + ry si, -rx si + ry co).  Here the cells are numbered 0..3 |
        // 4..12 | 13..28 with the a (= k) range as the slow index, a sum at vals[cell][channel], channels (Lt, -rx si + ry co,
        // rx co + ry si): OpenCV's cell j of grid g is (a range j % g, bb range j / g), its channels 1 and 2 are ours 2 and 1.
        {
            int full[162][3], c = 0; // grid, first cell, second cell (OpenCV's numbering)
            for (int i = 0; i < 3; i++)
                for (int j = 0; j < (i + 2) * (i + 2); j++)
                    for (int k = j + 1; k < (i + 2) * (i + 2); k++, c++)
                        full[c][0] = i, full[c][1] = j, full[c][2] = k;
            const int base[3] = {0, 4, 13}, channel_of[3] = {0, 2, 1};
            auto cell_of = [&](int grid, int j) { return base[grid] + (j % (grid + 2)) * (grid + 2) + j / (grid + 2); };
            uint64_t state = 1024;
            for (int i = 0; i < 162; i++)
            {
                state = (uint64_t)(uint32_t)state * 4164903690u + (uint32_t)(state >> 32);
                int k = (int)((uint32_t)state % (uint32_t)(162 - i));
                if (i < 6)
                    k = i;
                for (int ch = 0; ch < 3; ch++)
                    T.bit_ofs[3 * i + ch] = (unsigned int)((cell_of(full[k][0], full[k][1]) * 3 + channel_of[ch]) * 4) |
                                            ((unsigned int)((cell_of(full[k][0], full[k][2]) * 3 + channel_of[ch]) * 4) << 16);
                for (int q = 0; q < 3; q++)
                    full[k][q] = full[162 - i - 1][q];
            }
        }
        // the cell chains: lane 3 c + channel sums channel `channel` of chain c - the four 10 x 10 cells, the 7 x 7 cells in
        // pairs (the ninth alone), the 5 x 5 cells in fours.  Word: lattice point of the first sample | channel << 9 |
        // class << 11 | first cell << 13 | "the pair's second cell follows the first in memory" << 18
        for (int i = 0; i < 64; i++)
            T.chain[i] = 3u << 11;
        struct ch_t
        {
            int cls, first_cell, i, j, follows;
        };
        std::vector<ch_t> chains;
        for (int k = 0; k < 4; k++)
            chains.push_back({0, k, -10 + (k / 2) * 10, -10 + (k % 2) * 10, 0});
        for (int k = 0; k < 9; k += 2) // cell 2 ends at lattice point 146 and cell 3 begins at 147
            chains.push_back({1, 4 + k, -10 + (k / 3) * 7, -10 + (k % 3) * 7, k == 2 ? 1 : 0});
        for (int k = 0; k < 16; k += 4)
            chains.push_back({2, 13 + k, -10 + (k / 4) * 5, -10, 0});
        for (size_t cidx = 0; cidx < chains.size(); cidx++)
            for (int ch = 0; ch < 3; ch++)
            {
                const ch_t &cc = chains[cidx];
                const unsigned int p0 = (unsigned int)((cc.i + 10) * 21 + (cc.j + 10));
                T.chain[3 * cidx + ch] = p0 | ((unsigned int)ch << 9) | ((unsigned int)cc.cls << 11) |
                                         ((unsigned int)cc.first_cell << 13) | ((unsigned int)cc.follows << 18);
            }
        return T;
    }();
    return G;
}

// images: n_images x h x w x 3 BGR bytes (host, or device when on_device).  Working size: the INTER_AREA
// downscale to max side 1600 of extract_features.cpp:26-27.  Outputs (host): per image up to max_kp keypoints,
// in unspecified order: kp6 = {x, y, size, angle(rad), response, level} in working-image pixels, desc = 8 x u64,
// counts[i] = number written for image i (<= max_kp).  work_wh receives the working width/height.
int akaze_run(ochip_ctx *ctx, const uint8_t *images_bgr, bool on_device, uint32_t n_images, int width, int height,
              uint32_t max_kp, float *kp6, uint64_t *desc, uint32_t *counts, int *work_wh, const ochip_feature_lists *lists,
              double nms_radius)
{
    if (!ctx || !counts || (n_images && (!images_bgr || (!lists && (!kp6 || !desc)))))
        return OCHIP_EINVAL;
    if (width <= 0 || height <= 0)
        return ochip_fail(ctx, OCHIP_EINVAL, "bad image size %d x %d", width, height);
    OCHIP_HIP(ctx, hipSetDevice(ctx->device));
    hipStream_t st = ctx->stream;
    const double scale = std::min(1.f, float(1600) / (float)std::max(width, height));
    const int W = (int)std::lrint(width * scale), H = (int)std::lrint(height * scale);
    if (work_wh)
    {
        work_wh[0] = W;
        work_wh[1] = H;
    }
    for (uint32_t i = 0; i < n_images; i++)
        counts[i] = 0;
    if (n_images == 0)
        return OCHIP_OK;

    // ---- level table (AKAZEFeatures::Allocate_Memory_Evolution)
    const int omax = 4, nsub = 4;
    const float soffset = 1.6f, dfactor = 1.5f, dthreshold = 0.00005f;
    levels_dev LV{};
    std::vector<std::vector<float>> tsteps;
    size_t img_stride = 0;
    int n_tiles = 0;
    size_t mask_stride = 0; // words of one image's maxima bit masks
    size_t sup_stride = 0;  // words of one image's masks of the suppression (8 x 8 pixels per word)
    {
        std::vector<float> etime;
        for (int i = 0; i < omax; i++)
        {
            const float rfactor = 1.0f / (float)(1 << i);
            const int lw = (int)(W * rfactor), lh = (int)(H * rfactor);
            if ((lw < 80 || lh < 40) && i != 0) // (Allocate_Memory_Evolution: "smallest possible octave" - 80 wide, 40 high)
                break;
            for (int j = 0; j < nsub; j++)
            {
                level_info &l = LV.l[LV.n++];
                l.octave = i;
                l.w = lw;
                l.h = lh;
                l.esigma = soffset * std::pow(2.0f, (float)j / (float)nsub + (float)i);
                l.sigma_size = (int)std::lrintf(l.esigma * dfactor / (float)(1 << i));
                l.off = img_stride;
                img_stride += (size_t)lw * lh;
                l.tiles_x = (lw + BT_X - 1) / BT_X;
                l.tile_off = n_tiles;
                n_tiles += l.tiles_x * ((lh + DT_Y - 1) / DT_Y);
                l.mask_off = (int)mask_stride;
                mask_stride += (size_t)l.tiles_x * lh;
                l.sup_off = (int)sup_stride;
                l.sup_tx = (lw + 7) / 8;
                sup_stride += (size_t)l.sup_tx * ((lh + 7) / 8);
                etime.push_back(0.5f * (l.esigma * l.esigma));
            }
        }
        tsteps.resize(LV.n);
        for (int i = 1; i < LV.n; i++)
            tsteps[i] = fed_taus(etime[i] - etime[i - 1], 0.25f);
    }
    const size_t plane0 = (size_t)W * H;
    const uint32_t B = n_images;
    const uint32_t max_cands = std::max<uint32_t>(max_kp * 4, 1u << 16);

    std::vector<std::pair<void *, size_t>> allocs;
    auto cleanup = [&]() {
        (void)ochip_stream_wait(ctx, st);
        for (auto &a : allocs)
            ochip_pool_put(ctx, a.first, a.second);
    };
    int rc = OCHIP_OK;
#define AK(call)                                                                                                       \
    do                                                                                                                 \
    {                                                                                                                  \
        if (rc == OCHIP_OK)                                                                                            \
            rc = (call);                                                                                               \
    } while (0)
    uint8_t *d_bgr = nullptr, *d_gray = nullptr;
    float *d_img = nullptr, *d_flow = nullptr, *d_ping = nullptr;
    float2 *d_Lxy = nullptr; // (Lx, Ly) interleaved, indexed like the other pyramids
    float2 *d_Fit = nullptr; // (dx, dy) of the sub-pixel fit at the maxima (sparse, like d_Rmax)
    float *d_Lt = nullptr, *d_Rmax = nullptr, *d_kc = nullptr,
          *d_kp = nullptr;
    unsigned int *d_hmax = nullptr, *d_hist = nullptr, *d_ncand = nullptr, *d_pmax = nullptr;
    cand_t *d_cands = nullptr;
    unsigned char *d_dead = nullptr, *d_valid = nullptr;
    unsigned long long *d_desc = nullptr, *d_descc = nullptr;
    float *d_kpc = nullptr;
    unsigned int *d_counts = nullptr, *d_tile_counts = nullptr, *d_tile_base = nullptr, *d_tile_seq = nullptr;
    // maxima / valid keypoints; the suppression's masks (8 x 8 pixels per word): keypoints, points waiting for a turn, points
    // with a turn (pass 1: the maxima in this layout)
    unsigned long long *d_mask = nullptr, *d_vmask = nullptr, *d_kmask = nullptr, *d_pmask = nullptr, *d_rmask = nullptr;
    unsigned int *d_found = nullptr;                       // the suppression's passes 2 / 3: a keypoint's first keypoint of the other level
    unsigned int *d_level_first = nullptr;                 // first candidate of every level of every image (+ the list's length)
    unsigned int *d_turns = nullptr, *d_waiting = nullptr; // its two lists of points waiting for their turn, their lengths per (image, level)
    unsigned int *d_wbase = nullptr, *d_live = nullptr, *d_nlive = nullptr;
    gather_tab *d_gtab = nullptr;
    const size_t src_px = (size_t)width * height;
    // 1-D tile grids padded to a multiple of 8 workgroups (xcd_tile)
    auto tiles = [&](int w, int h) { return dim3(8 * ((((w + BT_X - 1) / BT_X) * ((h + BT_Y - 1) / BT_Y) + 7) / 8), 1, B); };
    auto det_tiles = [&](int w, int h) { return dim3(8 * ((((w + BT_X - 1) / BT_X) * ((h + DT_Y - 1) / DT_Y) + 7) / 8), 1, B); };
    const dim3 tiles0 = tiles(W, H);
    const int n_tiles0 = ((W + BT_X - 1) / BT_X) * ((H + BT_Y - 1) / BT_Y);
    if (on_device)
        d_bgr = const_cast<uint8_t *>(images_bgr);
    else
        AK(up(ctx, allocs, &d_bgr, images_bgr, (size_t)B * src_px * 3));
    AK(up<uint8_t>(ctx, allocs, &d_gray, nullptr, (size_t)B * src_px + 4));
    AK(up<float>(ctx, allocs, &d_img, nullptr, (size_t)B * plane0));
    AK(up<float>(ctx, allocs, &d_flow, nullptr, (size_t)B * plane0));
    AK(up<float>(ctx, allocs, &d_ping, nullptr, (size_t)B * plane0));
    AK(up<unsigned int>(ctx, allocs, &d_pmax, nullptr, (size_t)B * n_tiles0));
    AK(up<float>(ctx, allocs, &d_Lt, nullptr, (size_t)B * img_stride));
    AK(up<float2>(ctx, allocs, &d_Lxy, nullptr, (size_t)B * img_stride));
    AK(up<float2>(ctx, allocs, &d_Fit, nullptr, (size_t)B * img_stride));
    AK(up<float>(ctx, allocs, &d_Rmax, nullptr, (size_t)B * img_stride));
    AK(up<float>(ctx, allocs, &d_kc, nullptr, B));
    AK(up<unsigned int>(ctx, allocs, &d_hmax, nullptr, B));
    AK(up<unsigned int>(ctx, allocs, &d_hist, nullptr, (size_t)B * 301));
    AK(up<unsigned int>(ctx, allocs, &d_ncand, nullptr, B));
    AK(up<cand_t>(ctx, allocs, &d_cands, nullptr, (size_t)B * max_cands));
    AK(up<unsigned char>(ctx, allocs, &d_dead, nullptr, (size_t)B * max_cands));
    AK(up<unsigned char>(ctx, allocs, &d_valid, nullptr, (size_t)B * max_cands));
    AK(up<float>(ctx, allocs, &d_kp, nullptr, (size_t)B * max_cands * 6));
    AK(up<unsigned long long>(ctx, allocs, &d_desc, nullptr, (size_t)B * max_cands * 8));
    AK(up<float>(ctx, allocs, &d_kpc, nullptr, (size_t)B * max_kp * 6));
    AK(up<unsigned long long>(ctx, allocs, &d_descc, nullptr, (size_t)B * max_kp * 8));
    AK(up<unsigned int>(ctx, allocs, &d_counts, nullptr, B));
    AK(up<unsigned int>(ctx, allocs, &d_tile_counts, nullptr, (size_t)B * n_tiles));
    AK(up<unsigned int>(ctx, allocs, &d_tile_base, nullptr, (size_t)B * n_tiles));
    AK(up<unsigned long long>(ctx, allocs, &d_mask, nullptr, (size_t)B * mask_stride));
    AK(up<unsigned long long>(ctx, allocs, &d_vmask, nullptr, (size_t)B * mask_stride));
    AK(up<unsigned long long>(ctx, allocs, &d_kmask, nullptr, (size_t)B * sup_stride));
    AK(up<unsigned long long>(ctx, allocs, &d_pmask, nullptr, (size_t)B * sup_stride));
    AK(up<unsigned long long>(ctx, allocs, &d_rmask, nullptr, (size_t)B * sup_stride));
    AK(up<unsigned int>(ctx, allocs, &d_turns, nullptr, (size_t)B * 2 * max_cands));
    AK(up<unsigned int>(ctx, allocs, &d_waiting, nullptr, (size_t)B * LV.n));
    AK(up<unsigned int>(ctx, allocs, &d_level_first, nullptr, (size_t)B * (LV.n + 1)));
    AK(up<unsigned int>(ctx, allocs, &d_found, nullptr, (size_t)B * max_cands));
    AK(up<unsigned int>(ctx, allocs, &d_wbase, nullptr, (size_t)B * mask_stride));
    AK(up<unsigned int>(ctx, allocs, &d_live, nullptr, (size_t)B * max_cands));
    AK(up<unsigned int>(ctx, allocs, &d_nlive, nullptr, B));
    {
        // processing order of the detection tiles: level by level, Morton order inside a level
        std::vector<std::pair<uint64_t, unsigned int>> keyed;
        keyed.reserve(n_tiles);
        auto spread = [](uint32_t v) {
            uint64_t x = v;
            x = (x | (x << 16)) & 0x0000FFFF0000FFFFull;
            x = (x | (x << 8)) & 0x00FF00FF00FF00FFull;
            x = (x | (x << 4)) & 0x0F0F0F0F0F0F0F0Full;
            x = (x | (x << 2)) & 0x3333333333333333ull;
            x = (x | (x << 1)) & 0x5555555555555555ull;
            return x;
        };
        for (int i = 0; i < LV.n; i++)
        {
            const level_info &l = LV.l[i];
            const int ty_n = (l.h + DT_Y - 1) / DT_Y;
            for (int ty = 0; ty < ty_n; ty++)
                for (int tx = 0; tx < l.tiles_x; tx++)
                    keyed.emplace_back(((uint64_t)i << 48) | spread((uint32_t)tx) | (spread((uint32_t)ty) << 1),
                                       (unsigned int)(l.tile_off + ty * l.tiles_x + tx));
        }
        std::sort(keyed.begin(), keyed.end());
        std::vector<unsigned int> seq(keyed.size());
        for (size_t i = 0; i < keyed.size(); i++)
            seq[i] = keyed[i].second;
        AK(up(ctx, allocs, &d_tile_seq, seq.data(), seq.size()));
    }
    {
        // the descriptor kernel's tables (built once per process): gather orders, cell chains, bit list
        AK(up(ctx, allocs, &d_gtab, &host_gather_tab(), 1));
    }
    if (rc != OCHIP_OK)
    {
        cleanup();
        return rc;
    }
    auto grid2 = [&](int w, int h) { return dim3((w + 255) / 256, h, B); };
    auto taps_of = [](const std::vector<float> &k) {
        taps_t t{};
        t.n = (int)k.size();
        for (int i = 0; i < t.n && i < MAX_TAPS; i++)
            t.k[i] = k[i];
        return t;
    };
    const taps_t g1 = taps_of(gaussian_taps(1.0f)), g0 = taps_of(gaussian_taps(soffset));
    if (g1.n != 5 || g0.n != 9)
    {
        cleanup();
        return ochip_fail(ctx, OCHIP_EINVAL, "akaze: unexpected Gaussian kernel sizes %d / %d", g1.n, g0.n);
    }
    hipEvent_t e0, e1;
    ochip_prof_begin(ctx, OCHIP_K_AKAZE, &e0, &e1);

    // ---- grey, downscale, float
    const size_t n_px = (size_t)B * src_px;
    const bool resized = !(W == width && H == height);
    area_tab tx, ty;
    bool staged = false; // can every resize workgroup stage its source window in LDS? (taps per column, window size)
    if (resized)
    {
        tx = area_table(width, W, 1.0 / scale);
        ty = area_table(height, H, 1.0 / scale);
        staged = width % 4 == 0;
        for (int x0 = 0; x0 < W && staged; x0 += 256)
        {
            const int x1 = std::min(x0 + 256, W);
            const int c0 = tx.si[tx.off[x0]] & ~3, c1 = tx.si[tx.off[x1] - 1];
            staged = c1 - c0 + 1 <= RS_PITCH - 3;
            for (int x = x0; x < x1 && staged; x++)
                staged = tx.off[x + 1] - tx.off[x] <= RESIZE_MAX_TAPS;
        }
        for (int y0 = 0; y0 < H && staged; y0 += RESIZE_ROWS)
        {
            const int y1 = std::min(y0 + RESIZE_ROWS, H);
            staged = ty.si[ty.off[y1] - 1] - ty.si[ty.off[y0]] + 1 <= RS_ROWS;
        }
    }
    const bool fused_grey = staged && ((uintptr_t)d_bgr & 3) == 0; // the resize converts BGR itself
    if (!fused_grey)
    {
        if (((uintptr_t)d_bgr & 3) == 0)
            hipLaunchKernelGGL(gray4_kernel, dim3((unsigned)(((n_px + 3) / 4 + 255) / 256)), dim3(256), 0, st,
                               (const uint32_t *)d_bgr, (uint32_t *)d_gray, n_px / 4);
        if (((uintptr_t)d_bgr & 3) != 0 || (n_px & 3))
        {
            // unaligned source or a tail of < 4 pixels: byte-wise
            const size_t first = ((uintptr_t)d_bgr & 3) ? 0 : (n_px & ~(size_t)3);
            hipLaunchKernelGGL(gray_kernel, dim3((unsigned)((n_px - first + 255) / 256)), dim3(256), 0, st, d_bgr + 3 * first,
                               d_gray + first, n_px - first);
        }
    }
    if (W == width && H == height)
        hipLaunchKernelGGL(to_float_kernel, dim3((unsigned)((B * plane0 + 255) / 256)), dim3(256), 0, st, d_gray, d_img,
                           B * plane0);
    else
    {
        int *xo, *xs, *yo, *ys;
        float *xa, *ya;
        AK(up(ctx, allocs, &xo, tx.off.data(), tx.off.size()));
        AK(up(ctx, allocs, &xs, tx.si.data(), tx.si.size()));
        AK(up(ctx, allocs, &xa, tx.alpha.data(), tx.alpha.size()));
        AK(up(ctx, allocs, &yo, ty.off.data(), ty.off.size()));
        AK(up(ctx, allocs, &ys, ty.si.data(), ty.si.size()));
        AK(up(ctx, allocs, &ya, ty.alpha.data(), ty.alpha.size()));
        if (rc != OCHIP_OK)
        {
            cleanup();
            return rc;
        }
        const dim3 rgrid((W + 255) / 256, (H + RESIZE_ROWS - 1) / RESIZE_ROWS, B);
        int most_taps = 0;
        for (int x = 0; x < W; x++)
            most_taps = std::max(most_taps, tx.off[x + 1] - tx.off[x]);
        // a scale of exactly 2 (a 3200-pixel side): cv::resize's integer path - the vector body of ResizeAreaFast rounds
        // (sum + 2) >> 2, sixteen columns at a time (the 16-bit lanes of an AVX2 build; the scalar tail rounds to even like
        // the general path).  Scales of 4 and 8 give the same bytes either way (every operation is exact).
        const int half_up_cols = scale == 0.5 ? W - W % 16 : 0;
        if (fused_grey && most_taps <= 4)
            hipLaunchKernelGGL((resize_area_lds_kernel<true, 4>), rgrid, dim3(256), 0, st, (const uint8_t *)d_bgr, width, height, d_img,
                               W, H, xo, xs, xa, yo, ys, ya, half_up_cols);
        else if (fused_grey)
            hipLaunchKernelGGL((resize_area_lds_kernel<true, RESIZE_MAX_TAPS>), rgrid, dim3(256), 0, st, (const uint8_t *)d_bgr, width, height, d_img,
                               W, H, xo, xs, xa, yo, ys, ya, half_up_cols);
        else if (staged)
            hipLaunchKernelGGL((resize_area_lds_kernel<false, RESIZE_MAX_TAPS>), rgrid, dim3(256), 0, st, d_gray, width, height, d_img, W, H, xo, xs,
                               xa, yo, ys, ya, half_up_cols);
        else
            hipLaunchKernelGGL(resize_area_kernel, rgrid, dim3(256), 0, st, d_gray, width, height, d_img, W, H, xo, xs, xa, yo, ys,
                               ya, half_up_cols);
    }

    // ---- contrast factor: Gaussian(1) + gradient magnitude + per-tile maxima in one pass, then the histogram
    OCHIP_HIP(ctx, hipMemsetAsync(d_hist, 0, (size_t)B * 301 * 4, st));
    // (as a register strip - the level kernel's Gaussian and distance-1 pattern with the magnitude where the conductivity is - when
    // the launch has the strips to fill the device, like the levels below; else the tile kernel and the reduction of its tile maxima)
    const bool modg_tiles = ochip_test_hook("tile_levels"), modg_force = ochip_test_hook("strip_levels");
    const bool modg_strip = !modg_tiles && (W & 1) == 0 && (plane0 & 1) == 0 && W >= 64 && H >= 64 && ((uintptr_t)d_img & 15) == 0 &&
                            ((uintptr_t)d_flow & 15) == 0 && (modg_force || (size_t)W * H * B >= ((size_t)32 << 20));
    if (modg_strip)
    {
        OCHIP_HIP(ctx, hipMemsetAsync(d_hmax, 0, (size_t)B * 4, st));
        level_strip_args sa{};
        sa.in = d_img, sa.in_stride = plane0;
        sa.flow = d_flow, sa.flow_stride = plane0;
        sa.w = W, sa.h = H;
        sa.hmax_bits = d_hmax;
        for (int q = 0; q < 5; q++)
            sa.k[q] = g1.k[q];
        typedef level_strip_geom<2, 0> MG;
        const int strips = ((W + MG::OW - 1) / MG::OW) * ((H + MG::H - 1) / MG::H);
        hipLaunchKernelGGL((level_strip_kernel<2, 0, true, true>), dim3(8 * (((strips + 3) / 4 + 7) / 8), 1, B), dim3(256), 0, st, sa);
    }
    else
    {
        blur_args a{d_img, plane0, d_flow, nullptr, plane0, W, H, nullptr, 0, d_pmax, nullptr, 0};
        hipLaunchKernelGGL((blur_fused_kernel<BLUR_MODG, 1, 2>), tiles0, dim3(256), 0, st, a, g1);
        hipLaunchKernelGGL(hmax_reduce_kernel, dim3(B), dim3(256), 0, st, (const unsigned int *)d_pmax, n_tiles0, d_hmax);
    }
    hipLaunchKernelGGL(hist_kernel, dim3((W + 255) / 256, (H + HIST_ROWS - 1) / HIST_ROWS, B), dim3(256), 0, st, (const float *)d_flow, W, H,
                       plane0, d_hmax, 300, d_hist);
    hipLaunchKernelGGL(kcontrast_kernel, dim3((B + 63) / 64), dim3(64), 0, st, d_hist, d_hmax, 300, 0.7f, d_kc, (int)B,
                       (int)((W - 2) * (H - 2))); // (the interior pixels)

    // ---- nonlinear scale space
    // level 0: the base image (Gaussian(soffset) of the resized view) and - its Lsmooth being its Lt - the detector's scale-s
    // derivatives of it, one launch (two before: the derivative pass read the base image again)
    bool level0_derivatives_done = false;
    {
        const level_info &l0 = LV.l[0];
        blur_args a{d_img, plane0, (float *)(d_Lxy + l0.off), nullptr, img_stride, W, H, nullptr, 0, nullptr, d_Lt + l0.off, img_stride};
        level0_derivatives_done = true;
        if (l0.sigma_size == 2)
            hipLaunchKernelGGL((blur_fused_kernel<BLUR_DERIV, 2, 4>), tiles0, dim3(256), 0, st, a, g0);
        else if (l0.sigma_size == 3)
            hipLaunchKernelGGL((blur_fused_kernel<BLUR_DERIV, 3, 4>), tiles0, dim3(256), 0, st, a, g0);
        else if (l0.sigma_size == 4)
            hipLaunchKernelGGL((blur_fused_kernel<BLUR_DERIV, 4, 4>), tiles0, dim3(256), 0, st, a, g0);
        else
        {
            level0_derivatives_done = false;
            blur_args p{d_img, plane0, d_Lt + l0.off, nullptr, img_stride, W, H, nullptr, 0, nullptr, nullptr, 0};
            hipLaunchKernelGGL((blur_fused_kernel<BLUR_PLAIN, 0, 4>), tiles0, dim3(256), 0, st, p, g0);
        }
    }
    // the register-strip kernels load pairs of pixels with 8 / 16-byte loads: even widths and plane offsets (every level
    // of an image whose working width is a multiple of 8; the tile kernels take the rest, and OCHIP_TEST_HOOKS=tile_det
    // / tile_levels all of it)
    // Which form pays depends on how many strips a launch has: a strip is one wavefront walking 40 - 90 rows, and a level of
    // 400 x 300 pixels x 100 images is 2 400 of them on 1 024 SIMDs (level kernel: 150 us per launch against the tile kernels'
    // 75; 200 x 150: 150 against 30) - the strips take the levels of >= STRIP_MIN_PIXELS pixels per launch (the first two
    // octaves of a chunk of the bench), the tiles the rest; OCHIP_TEST_HOOKS=strip_levels / strip_det: strips wherever they can
    const bool strip_hook = !ochip_test_hook("tile_det"), level_hook = !ochip_test_hook("tile_levels");
    const bool force_level_strips = ochip_test_hook("strip_levels"), force_det_strips = ochip_test_hook("strip_det");
    // (OCHIP_STRIP_MIN_PIXELS: the tests lower the threshold so that a small batch takes the mixed route of a bench chunk -
    // strips on its large levels, tiles on the small ones - and is compared with the restatement)
    const char *strip_env = std::getenv("OCHIP_STRIP_MIN_PIXELS");
    const size_t STRIP_MIN_PIXELS = strip_env && *strip_env ? (size_t)std::strtoull(strip_env, nullptr, 10) : (size_t)32 << 20;
    auto strips_pay = [&](const level_info &l, bool forced) { return forced || (size_t)l.w * l.h * B >= STRIP_MIN_PIXELS; };
    bool det_strips = strip_hook && (img_stride & 1) == 0 && (plane0 & 1) == 0 && ((uintptr_t)d_Lxy & 15) == 0 && ((uintptr_t)d_Lt & 15) == 0 &&
                      ((uintptr_t)d_flow & 15) == 0 && ((uintptr_t)d_ping & 15) == 0;
    for (int i = 0; i < LV.n; i++)
        det_strips = det_strips && (LV.l[i].w & 1) == 0 && (LV.l[i].off & 1) == 0 && LV.l[i].w >= 8 && LV.l[i].h >= 8;
    const bool level_strips = det_strips && level_hook;
    OCHIP_HIP(ctx, hipMemsetAsync(d_ncand, 0, B * 4, st));
    if (det_strips)
    {
        // the strips add their maxima to zeroed masks and tile counts (the tile form writes every word itself)
        OCHIP_HIP(ctx, hipMemsetAsync(d_mask, 0, (size_t)B * mask_stride * 8, st));
        OCHIP_HIP(ctx, hipMemsetAsync(d_tile_counts, 0, (size_t)B * n_tiles * 4, st));
    }
    // the columns / rows whose descriptor window [round(x - margin) - 1, round(x + margin) + 1] stays inside the level
    // (the predicate det_maxima_kernel evaluates per pixel: an interval in x and in y)
    auto det_window = [&](const level_info &l) {
        const float margin = (10.0f * std::sqrt(2.0f)) * (float)l.sigma_size; // descriptor window half width, M-LDB
        int4 win = make_int4(1, 0, 1, 0);
        auto span = [&](int n, int *lo, int *hi) {
            bool any = false;
            for (int v = 1; v < n - 1; v++)
                if ((int)std::rint((float)v - margin) - 1 >= 0 && (int)std::rint((float)v + margin) + 1 < n)
                {
                    if (!any)
                        *lo = v;
                    *hi = v;
                    any = true;
                }
        };
        span(l.w, &win.x, &win.y);
        span(l.h, &win.z, &win.w);
        return win;
    };
    int octave_steps = 0;
    int half_sampled_level = -1; // the level whose first image a level kernel has already written (half-sampling epilogue)
    for (int i = 1; i < LV.n; i++)
    {
        const level_info &l = LV.l[i], &p = LV.l[i - 1];
        float *cur = d_Lt + l.off;
        const size_t np = (size_t)l.w * l.h;
        const size_t n_steps = tsteps[i].size();
        // the level's FED steps in balanced groups of <= FED_FUSE, a launch each; the first group of a strip level (2 .. 4 steps)
        // rides in the level's launch
        const size_t tile_groups = (n_steps + FED_FUSE - 1) / FED_FUSE;
        const size_t first_size = tile_groups ? n_steps / tile_groups + (n_steps % tile_groups ? 1 : 0) : 0;
        const bool strip_level = level_strips && strips_pay(l, force_level_strips) && l.sigma_size >= 2 && l.sigma_size <= 4 && l.w >= 64 &&
                                 l.h >= 64 && (tile_groups == 0 || (first_size >= 2 && first_size <= 4));
        std::vector<size_t> group_sizes;
        for (size_t g = 0; g < tile_groups; g++)
            group_sizes.push_back(n_steps / tile_groups + (g < n_steps % tile_groups ? 1 : 0));
        const size_t n_groups = group_sizes.size();
        // the level starts from the previous level's image (half-sampled at a new octave); the FED steps ping-pong
        // between the level plane and a scratch plane, arranged so that the last step lands in the level plane
        const float *src = d_Lt + p.off;
        size_t src_stride = img_stride;
        if (l.octave > p.octave)
        {
            float *dst = (n_groups % 2 == 0) ? cur : d_ping;
            const size_t dst_stride = (n_groups % 2 == 0) ? img_stride : plane0;
            if (p.w != 2 * l.w || p.h != 2 * l.h) // (an odd dimension: OpenCV's general area path, see halfsample_area_kernel)
                hipLaunchKernelGGL(halfsample_area_kernel, grid2(l.w, l.h), dim3(256), 0, st, (const float *)(d_Lt + p.off), p.w, p.h,
                                   img_stride, dst, l.w, l.h, dst_stride);
            else if (half_sampled_level != i) // (else: the last level kernel of the octave before wrote it with its own rows)
                hipLaunchKernelGGL(halfsample_kernel, grid2(l.w, l.h), dim3(256), 0, st, (const float *)(d_Lt + p.off), p.w, p.h,
                                   img_stride, dst, l.w, l.h, dst_stride);
            src = dst;
            src_stride = dst_stride;
            octave_steps++;
        }
        auto group_size = [&](size_t g) { return group_sizes[g]; };
        size_t first_group = 0; // launches of the diffusion kernels start at this group
        if (strip_level)
        {
            // ONE launch: Lsmooth -> conductivity + (Lx, Ly), and the level's first group of diffusion steps behind them (the
            // whole cycle for the levels of the first octave: then the conductivity plane is not even stored)
            const int K = n_groups ? (int)group_size(0) : 0;
            const bool to_cur = n_groups == 0 || ((n_groups - 1) % 2) == 0;
            float *dst = to_cur ? cur : d_ping;
            const size_t dst_stride = to_cur ? img_stride : plane0;
            level_strip_args sa{};
            sa.in = src, sa.in_stride = src_stride;
            sa.flow = d_flow, sa.flow_stride = plane0;
            sa.Lxy = d_Lxy + l.off, sa.lxy_stride = img_stride;
            sa.Lout = dst, sa.lout_stride = dst_stride;
            sa.w = l.w, sa.h = l.h;
            sa.kcontrast = d_kc, sa.n_octave_steps = octave_steps;
            for (int q = 0; q < 5; q++)
                sa.k[q] = g1.k[q];
            for (int q = 0; q < K; q++)
                sa.T.tau[q] = tsteps[i][q];
            // the octave's last level with its whole FED cycle in this launch: the next octave's first image - 2 x 2 means of this
            // level's final rows, a lane's pair of columns x a pair of rows - leaves with them (halfsample_kernel read the
            // level again: 2.5 us per image at the first octave boundary)
            sa.half_out = nullptr;
            sa.half_stride = 0;
            if (i + 1 < LV.n && LV.l[i + 1].octave > l.octave && n_groups == 1 && K > 0 && l.w == 2 * LV.l[i + 1].w && l.h == 2 * LV.l[i + 1].h)
            {
                const size_t next_groups = (tsteps[i + 1].size() + FED_FUSE - 1) / FED_FUSE;
                const bool next_to_cur = next_groups % 2 == 0;
                sa.half_out = next_to_cur ? d_Lt + LV.l[i + 1].off : d_ping;
                sa.half_stride = next_to_cur ? img_stride : plane0;
                half_sampled_level = i + 1;
            }
            const bool store_flow = n_groups > 1 || std::getenv("OCHIP_DUMP_PLANES") != nullptr;
            auto sgrid = [&](int ow, int sh) {
                const int strips = ((l.w + ow - 1) / ow) * ((l.h + sh - 1) / sh);
                return dim3(8 * (((strips + 3) / 4 + 7) / 8), 1, B);
            };
#define OCHIP_LEVEL_STRIP(SS, KK, FF)                                                                                                     \
    hipLaunchKernelGGL((level_strip_kernel<SS, KK, FF>), sgrid(level_strip_geom<SS, KK>::OW, level_strip_geom<SS, KK>::H), dim3(256), 0, st, sa)
#define OCHIP_LEVEL_STRIP_K(SS)                                                                                                           \
    do                                                                                                                                    \
    {                                                                                                                                     \
        if (K == 0)                                                                                                                       \
            OCHIP_LEVEL_STRIP(SS, 0, true);                                                                                               \
        else if (K == 2 && store_flow)                                                                                                    \
            OCHIP_LEVEL_STRIP(SS, 2, true);                                                                                               \
        else if (K == 2)                                                                                                                  \
            OCHIP_LEVEL_STRIP(SS, 2, false);                                                                                              \
        else if (K == 3 && store_flow)                                                                                                    \
            OCHIP_LEVEL_STRIP(SS, 3, true);                                                                                               \
        else if (K == 3)                                                                                                                  \
            OCHIP_LEVEL_STRIP(SS, 3, false);                                                                                              \
        else if (store_flow)                                                                                                              \
            OCHIP_LEVEL_STRIP(SS, 4, true);                                                                                               \
        else                                                                                                                              \
            OCHIP_LEVEL_STRIP(SS, 4, false);                                                                                              \
    } while (0)
            if (l.sigma_size == 2)
                OCHIP_LEVEL_STRIP_K(2);
            else if (l.sigma_size == 3)
                OCHIP_LEVEL_STRIP_K(3);
            else
                OCHIP_LEVEL_STRIP_K(4);
#undef OCHIP_LEVEL_STRIP_K
#undef OCHIP_LEVEL_STRIP
            if (n_groups)
            {
                src = dst;
                src_stride = dst_stride;
                first_group = 1;
            }
        }
        else
        {
            // Lsmooth of the level (Gaussian(1) of the image it starts from) -> conductivity AND the detector's
            // scale-s derivatives Lx, Ly (AKAZE computes both on evolution[i].Lsmooth), one pass
            blur_args a{src, src_stride, d_flow, (float *)(d_Lxy + l.off), plane0, l.w, l.h, d_kc, octave_steps, nullptr, nullptr, img_stride};
            if (l.sigma_size == 2)
                hipLaunchKernelGGL((blur_fused_kernel<BLUR_FLOW_DERIV, 2, 2>), tiles(l.w, l.h), dim3(256), 0, st, a, g1);
            else if (l.sigma_size == 3)
                hipLaunchKernelGGL((blur_fused_kernel<BLUR_FLOW_DERIV, 3, 2>), tiles(l.w, l.h), dim3(256), 0, st, a, g1);
            else if (l.sigma_size == 4)
                hipLaunchKernelGGL((blur_fused_kernel<BLUR_FLOW_DERIV, 4, 2>), tiles(l.w, l.h), dim3(256), 0, st, a, g1);
            else
            {
                cleanup();
                return ochip_fail(ctx, OCHIP_EINVAL, "akaze: derivative scale %d outside 2..4", l.sigma_size);
            }
        }
        if (const char *dump = std::getenv("OCHIP_DUMP_PLANES"))
            if (i == 1)
            {
                // debugging aid: level 1's conductivity plane of the first image
                std::vector<float> hf(np);
                OCHIP_HIP(ctx, ochip_stream_wait(ctx, st));
                OCHIP_HIP(ctx, hipMemcpy(hf.data(), d_flow, np * 4, hipMemcpyDeviceToHost));
                if (FILE *f = std::fopen((std::string(dump) + "_flow1.f32").c_str(), "wb"))
                {
                    std::fwrite(hf.data(), 4, hf.size(), f);
                    std::fclose(f);
                }
            }
        if (n_steps == 0)
            hipLaunchKernelGGL(copy_plane_kernel, dim3((unsigned)((np + 255) / 256), 1, B), dim3(256), 0, st, src, src_stride,
                               cur, img_stride, np);
        for (size_t g = 0, k = 0; g < n_groups; g++)
        {
            const size_t gsz = group_size(g);
            fed_tau_group T{};
            for (size_t q = 0; q < gsz; q++)
                T.tau[q] = tsteps[i][k + q];
            k += gsz;
            if (g < first_group)
                continue;
            const bool to_cur = ((n_groups - 1 - g) % 2) == 0;
            float *dst = to_cur ? cur : d_ping;
            const size_t dst_stride = to_cur ? img_stride : plane0;
            const int ow = 64 - 2 * (int)gsz; // output columns per wavefront
            const dim3 gr((l.w + 4 * ow - 1) / (4 * ow), (l.h + NLD_TY - 1) / NLD_TY, B);
            if (gsz == 1)
                hipLaunchKernelGGL((nld_fused_kernel<1>), gr, dim3(256), 0, st, src, (const float *)d_flow, dst, l.w, l.h,
                                   src_stride, plane0, dst_stride, T);
            else if (gsz == 2)
                hipLaunchKernelGGL((nld_fused_kernel<2>), gr, dim3(256), 0, st, src, (const float *)d_flow, dst, l.w, l.h,
                                   src_stride, plane0, dst_stride, T);
            else if (gsz == 3)
                hipLaunchKernelGGL((nld_fused_kernel<3>), gr, dim3(256), 0, st, src, (const float *)d_flow, dst, l.w, l.h,
                                   src_stride, plane0, dst_stride, T);
            else
                hipLaunchKernelGGL((nld_fused_kernel<4>), gr, dim3(256), 0, st, src, (const float *)d_flow, dst, l.w, l.h,
                                   src_stride, plane0, dst_stride, T);
            src = dst;
            src_stride = dst_stride;
        }
    }
    // ---- derivatives of level 0, determinant, maxima
    for (int i = 0; i < LV.n; i++)
    {
        const level_info &l = LV.l[i];
        if (i == 0 && !level0_derivatives_done)
        {
            // level 0's Lsmooth is its Lt: derivatives without a further blur (a one-tap identity kernel keeps the
            // same code path; 0 + 1 * x is exact)
            taps_t one{};
            one.n = 1;
            one.k[0] = 1.0f;
            blur_args a{d_Lt + l.off, img_stride, (float *)(d_Lxy + l.off), nullptr, img_stride, l.w, l.h, nullptr, 0, nullptr, nullptr, 0};
            if (l.sigma_size == 2)
                hipLaunchKernelGGL((blur_fused_kernel<BLUR_DERIV, 2, 0>), tiles(l.w, l.h), dim3(256), 0, st, a, one);
            else if (l.sigma_size == 3)
                hipLaunchKernelGGL((blur_fused_kernel<BLUR_DERIV, 3, 0>), tiles(l.w, l.h), dim3(256), 0, st, a, one);
            else if (l.sigma_size == 4)
                hipLaunchKernelGGL((blur_fused_kernel<BLUR_DERIV, 4, 0>), tiles(l.w, l.h), dim3(256), 0, st, a, one);
            else
            {
                cleanup();
                return ochip_fail(ctx, OCHIP_EINVAL, "akaze: derivative scale %d outside 2..4", l.sigma_size);
            }
        }
        {
            const float2 *lxy = d_Lxy + l.off;
            float2 *ld = d_Fit + l.off;
            float *rm = d_Rmax + l.off;
            const float margin = (10.0f * std::sqrt(2.0f)) * (float)l.sigma_size; // descriptor window half width, M-LDB
            const int4 win = det_window(l);
            auto strip_grid = [&](int ow, int sh) {
                const int strips = ((l.w + ow - 1) / ow) * ((l.h + sh - 1) / sh);
                return dim3(8 * (((strips + 3) / 4 + 7) / 8), 1, B);
            };
            const bool det_strip = det_strips && strips_pay(l, force_det_strips);
            if (det_strip && l.sigma_size == 2)
                hipLaunchKernelGGL((det_strip_kernel<2>), strip_grid(det_strip_geom<2>::OW, det_strip_geom<2>::H), dim3(256), 0, st, lxy, img_stride, ld, rm, l.w,
                                   l.h, dthreshold, d_tile_counts, l.tile_off, n_tiles, win, d_mask + l.mask_off, mask_stride);
            else if (det_strip && l.sigma_size == 3)
                hipLaunchKernelGGL((det_strip_kernel<3>), strip_grid(det_strip_geom<3>::OW, det_strip_geom<3>::H), dim3(256), 0, st, lxy, img_stride, ld, rm, l.w,
                                   l.h, dthreshold, d_tile_counts, l.tile_off, n_tiles, win, d_mask + l.mask_off, mask_stride);
            else if (det_strip)
                hipLaunchKernelGGL((det_strip_kernel<4>), strip_grid(det_strip_geom<4>::OW, det_strip_geom<4>::H), dim3(256), 0, st, lxy, img_stride, ld, rm, l.w,
                                   l.h, dthreshold, d_tile_counts, l.tile_off, n_tiles, win, d_mask + l.mask_off, mask_stride);
            else if (l.sigma_size == 2)
                hipLaunchKernelGGL((det_maxima_kernel<2>), det_tiles(l.w, l.h), dim3(256), 0, st, lxy, img_stride, ld, rm, l.w,
                                   l.h, dthreshold, d_tile_counts, l.tile_off, n_tiles, margin, d_mask + l.mask_off, mask_stride);
            else if (l.sigma_size == 3)
                hipLaunchKernelGGL((det_maxima_kernel<3>), det_tiles(l.w, l.h), dim3(256), 0, st, lxy, img_stride, ld, rm, l.w,
                                   l.h, dthreshold, d_tile_counts, l.tile_off, n_tiles, margin, d_mask + l.mask_off, mask_stride);
            else
                hipLaunchKernelGGL((det_maxima_kernel<4>), det_tiles(l.w, l.h), dim3(256), 0, st, lxy, img_stride, ld, rm, l.w,
                                   l.h, dthreshold, d_tile_counts, l.tile_off, n_tiles, margin, d_mask + l.mask_off, mask_stride);
        }
    }
    if (const char *dump = std::getenv("OCHIP_DUMP_PLANES"))
    {
        // debugging aid: the first image's Lt and (Lx, Ly) pyramids as raw float files <prefix>_lt.f32 / _lxy.f32
        std::vector<float> hl(img_stride), hx(2 * img_stride);
        OCHIP_HIP(ctx, ochip_stream_wait(ctx, st));
        OCHIP_HIP(ctx, hipMemcpy(hl.data(), d_Lt, img_stride * 4, hipMemcpyDeviceToHost));
        OCHIP_HIP(ctx, hipMemcpy(hx.data(), d_Lxy, img_stride * 8, hipMemcpyDeviceToHost));
        for (int which = 0; which < 2; which++)
            if (FILE *f = std::fopen((std::string(dump) + (which ? "_lxy.f32" : "_lt.f32")).c_str(), "wb"))
            {
                std::fwrite(which ? hx.data() : hl.data(), 4, which ? hx.size() : hl.size(), f);
                std::fclose(f);
            }
    }
    hipLaunchKernelGGL(scan_tiles_kernel, dim3(B), dim3(256), 0, st, (const unsigned int *)d_tile_counts,
                       (const unsigned int *)d_tile_seq, n_tiles, d_tile_base, d_ncand);
    hipLaunchKernelGGL(collect_tiles_kernel, dim3((n_tiles + 4 * COLLECT_TPW - 1) / (4 * COLLECT_TPW), 1, B), dim3(256), 0, st, (const float *)d_Rmax,
                       (const float2 *)d_Fit, img_stride,
                       (const unsigned long long *)d_mask, mask_stride, LV, (const unsigned int *)d_tile_base, n_tiles, d_cands,
                       max_cands, (const unsigned int *)d_tile_counts);
    std::vector<unsigned int> ncand(B);
    OCHIP_HIP(ctx, hipMemcpyAsync(ncand.data(), d_ncand, B * 4, hipMemcpyDeviceToHost, st));
    OCHIP_HIP(ctx, ochip_stream_wait(ctx, st));
    unsigned int max_n = 0;
    for (uint32_t b = 0; b < B; b++)
    {
        if (ncand[b] > max_cands)
        {
            cleanup();
            return ochip_fail(ctx, OCHIP_ENOMEM, "image %u has %u extrema candidates; raise max_kp (candidate capacity %u)", b,
                              ncand[b], max_cands);
        }
        max_n = std::max(max_n, ncand[b]);
    }
    const int xcd_remap = 2; // groups of 64 list neighbours per XCD (see xcd_contiguous; 0, 1 and 3 measured slower)
    if (max_n > 0)
    {
        OCHIP_HIP(ctx, hipMemsetAsync(d_vmask, 0, (size_t)B * mask_stride * 8, st));
        const dim3 per_cand((max_n + 255) / 256, 1, B);
        // test hooks: the suppression's code for radii beyond the default scale space's / its lists beyond their LDS part
        const int sup_hooks = (ochip_test_hook("sup_generic") ? 1 : 0) | (ochip_test_hook("sup_small_lists") ? 2 : 0);
        unsigned int *d_sup_stats = nullptr;
        if (ochip_verbose("extract"))
        {
            AK(up<unsigned int>(ctx, allocs, &d_sup_stats, nullptr, (size_t)3 * B * LV.n * 3));
            OCHIP_HIP(ctx, hipMemsetAsync(d_sup_stats, 0, (size_t)3 * B * LV.n * 3 * 4, st));
        }
        OCHIP_HIP(ctx, hipMemsetAsync(d_kmask, 0, (size_t)B * sup_stride * 8, st));
        OCHIP_HIP(ctx, hipMemsetAsync(d_pmask, 0, (size_t)B * sup_stride * 8, st));
        OCHIP_HIP(ctx, hipMemsetAsync(d_waiting, 0, (size_t)B * LV.n * 4, st));
        hipLaunchKernelGGL(suppress_level_first_kernel, dim3(B), dim3(64), 0, st, (const unsigned int *)d_ncand, max_cands, LV,
                           (const unsigned int *)d_tile_base, (const unsigned int *)d_tile_seq, n_tiles, d_level_first);
        // (64 x 8-pixel groups: the grid covers the largest level, a level per blockIdx.y)
        hipLaunchKernelGGL(suppress_tiles_kernel, dim3((unsigned int)((LV.l[0].tiles_x * ((LV.l[0].h + 7) / 8) + 255) / 256), LV.n, B),
                           dim3(256), 0, st, (const unsigned long long *)d_mask, mask_stride, d_rmask, sup_stride, LV);
#define SUP_PASS(PASS, LEVELS)                                                                                                                 \
    hipLaunchKernelGGL(suppress_round0_kernel<PASS>, per_cand, dim3(256), 0, st, (const cand_t *)d_cands, (const unsigned int *)d_ncand,       \
                       max_cands, (const float *)d_Rmax, img_stride, (const unsigned long long *)d_rmask, d_pmask, d_kmask, sup_stride, LV,    \
                       (const unsigned int *)d_level_first, d_turns, d_waiting, (const unsigned int *)d_found, sup_hooks);                     \
    hipLaunchKernelGGL(suppress_rounds_kernel<PASS>, dim3(B, (LEVELS)), dim3(SUP_THREADS), 0, st, (const unsigned int *)d_ncand, max_cands,    \
                       (const float *)d_Rmax, img_stride, d_pmask, d_kmask, sup_stride, LV, (const unsigned int *)d_level_first, d_turns,      \
                       d_waiting, d_sup_stats, sup_hooks)
#define SUP_WINDOWS(PASS)                                                                                                                      \
    OCHIP_HIP(ctx, hipMemsetAsync(d_rmask, 0, (size_t)B * sup_stride * 8, st));                                                                \
    hipLaunchKernelGGL(suppress_window_kernel<PASS>, per_cand, dim3(256), 0, st, (const cand_t *)d_cands, (const unsigned int *)d_ncand,       \
                       max_cands, (const unsigned long long *)d_kmask, d_rmask, sup_stride, LV, d_found, sup_hooks)
        SUP_PASS(1, LV.n);
        if (LV.n > 1)
        {
            SUP_WINDOWS(2);
            SUP_PASS(2, LV.n - 1);
            SUP_WINDOWS(3);
            SUP_PASS(3, LV.n - 1);
        }
#undef SUP_WINDOWS
        hipLaunchKernelGGL(suppress_dead_kernel, per_cand, dim3(256), 0, st, (const cand_t *)d_cands, (const unsigned int *)d_ncand,
                           max_cands, (const unsigned long long *)d_kmask, sup_stride, LV, d_dead);
        if (d_sup_stats)
        {
            std::vector<unsigned int> hs((size_t)3 * B * LV.n * 3);
            OCHIP_HIP(ctx, hipMemcpyAsync(hs.data(), d_sup_stats, hs.size() * 4, hipMemcpyDeviceToHost, st));
            OCHIP_HIP(ctx, ochip_stream_wait(ctx, st));
            for (int pass = 0; pass < 3; pass++)
            {
                std::string line;
                for (int i = 0; i < LV.n; i++)
                {
                    unsigned int rounds = 0, ticks = 0, pts = 0;
                    for (uint32_t b = 0; b < B; b++)
                    {
                        const unsigned int *o = &hs[(((size_t)pass * B + b) * LV.n + i) * 3];
                        rounds = std::max(rounds, o[0]), ticks = std::max(ticks, o[1]), pts = std::max(pts, o[2]);
                    }
                    char buf[64];
                    std::snprintf(buf, sizeof buf, " %u/%.0fus/%u", rounds, ticks * 0.01, pts);
                    line += buf;
                }
                std::fprintf(stderr, "[ochip extract] suppression pass %d, per level max over %u images of rounds/time/points:%s\n", pass + 1, B,
                             line.c_str());
            }
        }
        // the survivors in list order, then the descriptor over those only (slots it never visits stay invalid)
        hipLaunchKernelGGL(live_list_kernel, dim3(B), dim3(SCAN_THREADS), 0, st, (const unsigned char *)d_dead, (const unsigned int *)d_ncand,
                           max_cands, d_live, d_nlive);
        OCHIP_HIP(ctx, hipMemsetAsync(d_valid, 0, (size_t)B * max_cands, st));
        // the grid covers the longest list of survivors, not of candidates (a third to a half of the workgroups would
        // find nothing to do, and an empty workgroup still takes a slot with its 22 KB of LDS for a microsecond): one more
        // 400-byte read-back per chunk, hidden under the other launch sequences in flight
        std::vector<unsigned int> nlive(B);
        OCHIP_HIP(ctx, hipMemcpyAsync(nlive.data(), d_nlive, B * 4, hipMemcpyDeviceToHost, st));
        OCHIP_HIP(ctx, ochip_stream_wait(ctx, st));
        const unsigned int max_live = *std::max_element(nlive.begin(), nlive.end());
        // (capping the workgroups a CU holds with unused dynamic LDS, to leave wave slots to the other sequences' HBM-bound
        // kernels beside this L2-bound one, was measured: 3 120 images/s with none, 3 100 / 2 940 / 2 830 at 4 / 3 / 2 per CU)
        if (max_live > 0)
        {
            hipLaunchKernelGGL(describe3_kernel, dim3(512 * (((max_live + DESC_WPB - 1) / DESC_WPB + 511) / 512), 1, B), dim3(64 * DESC_WPB), 0, st,
                               (const cand_t *)d_cands, (const unsigned int *)d_nlive, max_cands, (const unsigned int *)d_live,
                               (const float *)d_Lt, (const float2 *)d_Lxy, img_stride, LV, dfactor,
                               (const gather_tab *)d_gtab, d_kp, d_desc, d_valid, xcd_remap, d_vmask,
                               mask_stride);
        }
    }
    if (max_n > 0)
    {
        hipLaunchKernelGGL(rank_scan_kernel, dim3(B), dim3(SCAN_THREADS), 0, st, (const unsigned long long *)d_vmask, mask_stride, d_wbase,
                           d_counts);
        hipLaunchKernelGGL(compact_ordered_kernel, dim3((max_n + 255) / 256, 1, B), dim3(256), 0, st,
                           (const unsigned char *)d_valid, (const cand_t *)d_cands, (const unsigned int *)d_ncand, max_cands,
                           (const unsigned long long *)d_vmask, (const unsigned int *)d_wbase, mask_stride, LV,
                           (const float *)d_kp, (const unsigned long long *)d_desc, d_kpc, d_descc, max_kp);
    }
    else
        OCHIP_HIP(ctx, hipMemsetAsync(d_counts, 0, B * 4, st));
    ochip_prof_end(ctx, OCHIP_K_AKAZE, e0, e1);
    OCHIP_HIP(ctx, hipGetLastError());
    // ---- results: the per-image counts first, then exactly the compacted keypoints and descriptors, straight
    // into the caller's arrays (page-locked ones from ochip_host_alloc make these copies run at link speed)
    OCHIP_HIP(ctx, hipMemcpyAsync(counts, d_counts, B * 4, hipMemcpyDeviceToHost, st));
    OCHIP_HIP(ctx, ochip_stream_wait(ctx, st));
    uint32_t most = 0;
    for (uint32_t b = 0; b < B && rc == OCHIP_OK; b++)
    {
        if (counts[b] > max_kp)
        {
            rc = ochip_fail(ctx, OCHIP_ENOMEM, "image %u has %u keypoints, more than max_kp = %u", b, counts[b], max_kp);
            break;
        }
        most = std::max(most, counts[b]);
    }
    if (rc == OCHIP_OK && lists)
        // the tail of extract_features prepared on the device (features.hip): records, suppression flags, what the
        // host's sort needs - instead of the raw keypoint arrays
        rc = ochip::feature_lists_enqueue(ctx, &allocs, B, max_kp, d_kpc, d_descc, d_counts, most, W, H, scale, nms_radius, lists);
    else if (rc == OCHIP_OK && most > 0)
    {
        // one strided copy per array instead of one per image and array (2 x 100 launches per chunk): every image's row is
        // copied up to the longest list of the chunk - a few per cent more bytes, the counts say where each list ends
        OCHIP_HIP(ctx, hipMemcpy2DAsync(kp6, (size_t)max_kp * 24, d_kpc, (size_t)max_kp * 24, (size_t)most * 24, B, hipMemcpyDeviceToHost, st));
        OCHIP_HIP(ctx, hipMemcpy2DAsync(desc, (size_t)max_kp * 64, d_descc, (size_t)max_kp * 64, (size_t)most * 64, B, hipMemcpyDeviceToHost, st));
    }
    if (rc != OCHIP_OK)
        for (uint32_t b = 0; b < B; b++)
            counts[b] = 0;
    cleanup();
    return rc;
#undef AK
}

} // namespace

