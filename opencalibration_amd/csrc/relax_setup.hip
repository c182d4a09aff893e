// libochip.so — set-up of the ground-plane relax on the device (gfx950): gridFilterMatchesPerImage and the 2-ray block
// assembly of RelaxProblem::setupGroundPlaneProblem.
//
// Replaces, for every edge of the problem, src/relax/relax_problem.cpp:234-309 (score each inlier match - ray
// intersection, angle, descriptor and RANSAC terms -, keep the best one per cell of each image's (fraction)^2 grid:
// GridFilter::addMeasurement, include/opencalibration/relax/grid_filter.hpp:33-51) and :388-560 with fixed intrinsics
// (a kept match becomes a residual block if its two rays' closest approach lies over the plane's border triangle).
// On the host this was half of the CPU time of a step (0.5 of 1.0 CPU-seconds per 1 000 images: ~10 fp64 divisions and
// square roots per match, 5.4 M matches); here it is one wavefront per edge and a few microseconds.
//
// Arithmetic: the expressions of csrc/host/relax_util.hpp in the same order, fp64, no contraction, so scores compare
// exactly as on the host and in the restatement.  Ties: GridFilter keeps the FIRST measurement of a cell in descending
// score order, and the reference sorts with an unstable std::sort - when the best score of a cell is shared by two
// matches the outcome is libstdc++'s permutation.  Such an edge (and one with a match outside the cell table) is only
// FLAGGED here; the caller decides it with the host code and overrides the edge's flags (none in the synthetic surveys).
#include "ctx.hpp"
#include "undistort.hpp"
#include "relax_setup_geom.hpp"

#include <vector>

namespace
{

struct setup_dev
{
    const ochip_plane_edge *edges;
    uint32_t n_edges;
    const ochip_plane_inlier *inliers;
    const double *cam_pos, *cam_q; // [n_cams][3], [n_cams][4]
    const double *models;          // [n_models][10]: f, ppx, ppy, k1, k2, k3, p1, p2, columns, rows
    double tri[6];                 // the plane's border triangle, corners in the searcher's (fixed-up) order
    double res;                    // grid fraction
    double *score;                 // [n_inliers]
    unsigned char *keep;           // [n_inliers] bit 0: on the source image's whitelist, bit 1: on the destination's
    unsigned char *inexact;        // [n_edges]
    unsigned int *count;           // [n_edges + 1] blocks per edge, then their exclusive prefix
    uint32_t *blk_a, *blk_b;
    double *blk_rays;
};

// gridFilterMatchesPerImage for one edge per wavefront
__global__ __launch_bounds__(256) void plane_filter_kernel(setup_dev S)
{
    __shared__ unsigned long long best_s[4][G_MAX * G_MAX], best_d[4][G_MAX * G_MAX];
    __shared__ unsigned int cnt_s[4][G_MAX * G_MAX], cnt_d[4][G_MAX * G_MAX];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const uint32_t e = blockIdx.x * 4 + wv;
    if (e >= S.n_edges)
        return;
    const ochip_plane_edge ed = S.edges[e];
    for (int c = lane; c < G_MAX * G_MAX; c += 64)
    {
        best_s[wv][c] = 0ull;
        best_d[wv][c] = 0ull;
        cnt_s[wv][c] = 0u;
        cnt_d[wv][c] = 0u;
    }
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
    __builtin_amdgcn_wave_barrier();
    double Rs[3][3], Rd[3][3], ms[8], md[8];
    to_matrix(S.cam_q + 4 * (size_t)ed.cam_a, Rs);
    to_matrix(S.cam_q + 4 * (size_t)ed.cam_b, Rd);
    const double *pa = S.cam_pos + 3 * (size_t)ed.cam_a, *pb = S.cam_pos + 3 * (size_t)ed.cam_b;
    const v3 so{pa[0], pa[1], pa[2]}, d_o{pb[0], pb[1], pb[2]};
    const double *Ma = S.models + 10 * (size_t)ed.model_a, *Mb = S.models + 10 * (size_t)ed.model_b;
    for (int k = 0; k < 8; k++)
    {
        ms[k] = Ma[k];
        md[k] = Mb[k];
    }
    const double cols_s = Ma[8], rows_s = Ma[9], cols_d = Mb[8], rows_d = Mb[9];
    const bool homography = (ed.flags & 1u) != 0;
    const ochip_plane_inlier *in = S.inliers + ed.inlier_offset;
    bool bad = false;
    // pass 1: scores and the cells' best score
    auto score_of = [&](const ochip_plane_inlier &m) -> double {
        double r1[3], r2[3];
        ochip_ud::image_to_3d(m.px1, ms, r1);
        ochip_ud::image_to_3d(m.px2, md, r2);
        return plane_match_score(r1, r2, Rs, Rd, so, d_o, m, ms, md, ed.H, homography);
    };
    auto cells_of = [&](const ochip_plane_inlier &m, int *cs, int *cd) -> bool {
        return plane_match_cells(m, cols_s, rows_s, cols_d, rows_d, S.res, cs, cd);
    };
    double *score = S.score + ed.inlier_offset;
    for (uint32_t i = lane; i < ed.n_inliers; i += 64)
    {
        const ochip_plane_inlier m = in[i];
        const double s = score_of(m);
        score[i] = s;
        if (!(s > 0))
            continue;
        int cs, cd;
        if (!cells_of(m, &cs, &cd))
        {
            bad = true;
            continue;
        }
        // positive doubles order like their bit patterns
        atomicMax(&best_s[wv][cs], (unsigned long long)__double_as_longlong(s));
        atomicMax(&best_d[wv][cd], (unsigned long long)__double_as_longlong(s));
    }
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
    __builtin_amdgcn_wave_barrier();
    // pass 2: the matches that hold their cell's best score; two of them in one cell is a tie for the host
    for (uint32_t i = lane; i < ed.n_inliers; i += 64)
    {
        const ochip_plane_inlier m = in[i];
        const double s = score[i]; // this lane's own store
        unsigned char k = 0;
        int cs, cd;
        if (s > 0 && cells_of(m, &cs, &cd))
        {
            const unsigned long long bits = (unsigned long long)__double_as_longlong(s);
            if (best_s[wv][cs] == bits)
            {
                k |= 1;
                atomicAdd(&cnt_s[wv][cs], 1u);
            }
            if (best_d[wv][cd] == bits)
            {
                k |= 2;
                atomicAdd(&cnt_d[wv][cd], 1u);
            }
        }
        S.keep[ed.inlier_offset + i] = k;
    }
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
    __builtin_amdgcn_wave_barrier();
    for (int c = lane; c < G_MAX * G_MAX; c += 64)
        bad = bad || cnt_s[wv][c] > 1u || cnt_d[wv][c] > 1u;
    const bool any_bad = __any(bad) != 0;
    if (lane == 0)
        S.inexact[e] = any_bad ? 1 : 0;
}

// addRayTriangleMeasurementCost with fixed intrinsics for one edge per wavefront: EMIT = false counts the blocks, true
// writes them at the edge's offset in match order
template <bool EMIT> __global__ __launch_bounds__(256) void plane_blocks_kernel(setup_dev S)
{
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const uint32_t e = blockIdx.x * 4 + wv;
    if (e >= S.n_edges)
        return;
    const ochip_plane_edge ed = S.edges[e];
    double ms[8], md[8], qa[4], qb[4];
    const double *Ma = S.models + 10 * (size_t)ed.model_a, *Mb = S.models + 10 * (size_t)ed.model_b;
    for (int k = 0; k < 8; k++)
    {
        ms[k] = Ma[k];
        md[k] = Mb[k];
    }
    for (int k = 0; k < 4; k++)
    {
        qa[k] = S.cam_q[4 * (size_t)ed.cam_a + k];
        qb[k] = S.cam_q[4 * (size_t)ed.cam_b + k];
    }
    const double *pa = S.cam_pos + 3 * (size_t)ed.cam_a, *pb = S.cam_pos + 3 * (size_t)ed.cam_b;
    const v3 so{pa[0], pa[1], pa[2]}, d_o{pb[0], pb[1], pb[2]};
    const ochip_plane_inlier *in = S.inliers + ed.inlier_offset;
    unsigned int base = EMIT ? S.count[e] : 0u, total = 0;
    const unsigned long long below = (1ull << lane) - 1ull;
    for (uint32_t i0 = 0; i0 < ed.n_inliers; i0 += 64)
    {
        const uint32_t i = i0 + lane;
        bool ok = false;
        double r1[3], r2[3];
        if (i < ed.n_inliers && S.keep[ed.inlier_offset + i] != 0)
        {
            const ochip_plane_inlier m = in[i];
            ochip_ud::image_to_3d(m.px1, ms, r1);
            ochip_ud::image_to_3d(m.px2, md, r2);
            ok = plane_block_inside(r1, r2, qa, qb, so, d_o, S.tri);
        }
        const unsigned long long mask = __ballot(ok);
        if (EMIT && ok)
        {
            const size_t at = (size_t)base + total + (unsigned int)__popcll(mask & below);
            S.blk_a[at] = ed.cam_a;
            S.blk_b[at] = ed.cam_b;
            double *o = S.blk_rays + 6 * at;
            o[0] = r1[0], o[1] = r1[1], o[2] = r1[2], o[3] = r2[0], o[4] = r2[1], o[5] = r2[2];
        }
        total += (unsigned int)__popcll(mask);
    }
    if (!EMIT && lane == 0)
        S.count[e] = total;
}

// exclusive prefix of the edges' block counts (count[n_edges] = the total), one workgroup
__global__ __launch_bounds__(1024) void plane_scan_kernel(unsigned int *count, uint32_t n)
{
    __shared__ unsigned int part[1024];
    const uint32_t per = (n + 1023) / 1024, lo = threadIdx.x * per, hi = min(lo + per, n);
    unsigned int s = 0;
    for (uint32_t i = lo; i < hi; i++)
        s += count[i];
    part[threadIdx.x] = s;
    __syncthreads();
    for (int off = 1; off < 1024; off <<= 1)
    {
        const unsigned int o = threadIdx.x >= (unsigned)off ? part[threadIdx.x - off] : 0u;
        __syncthreads();
        part[threadIdx.x] += o;
        __syncthreads();
    }
    unsigned int run = part[threadIdx.x] - s;
    for (uint32_t i = lo; i < hi; i++)
    {
        const unsigned int c = count[i];
        count[i] = run;
        run += c;
    }
    if (threadIdx.x == 1023)
        count[n] = part[1023];
}

} // namespace

struct ochip_plane_setup
{
    ochip_ctx *ctx = nullptr;
    setup_dev dev{};
    uint64_t n_inliers = 0, total = 0;
    bool emitted = false;
    std::vector<std::pair<void *, size_t>> allocs;
};

extern "C"
{

void ochip_plane_setup_destroy(ochip_plane_setup *s)
{
    if (!s)
        return;
    (void)ochip_stream_wait(s->ctx, s->ctx->stream);
    for (auto &a : s->allocs)
        ochip_pool_put(s->ctx, a.first, a.second);
    delete s;
}

int ochip_plane_setup_create(ochip_ctx *ctx, const ochip_plane_edge *edges, uint32_t n_edges, const ochip_plane_inlier *inliers,
                             uint64_t n_inliers, const double *cam_pos, const double *cam_q, uint32_t n_cams,
                             const double *models10, uint32_t n_models, const double *triangle_xy6, double grid_fraction,
                             uint8_t *keep_out, uint8_t *inexact_out, ochip_plane_setup **out)
{
    if (!ctx || !out)
        return OCHIP_EINVAL;
    *out = nullptr;
    if ((n_edges && (!edges || !inexact_out)) || (n_inliers && (!inliers || !keep_out)) || (n_cams && (!cam_pos || !cam_q)) ||
        (n_models && !models10) || !triangle_xy6 || !(grid_fraction > 0))
        return ochip_fail(ctx, OCHIP_EINVAL, "ochip_plane_setup_create: bad argument");
    for (uint32_t e = 0; e < n_edges; e++)
        if (edges[e].cam_a >= n_cams || edges[e].cam_b >= n_cams || edges[e].model_a >= n_models || edges[e].model_b >= n_models ||
            edges[e].inlier_offset + edges[e].n_inliers > n_inliers)
            return ochip_fail(ctx, OCHIP_EINVAL, "ochip_plane_setup_create: edge %u is out of range", e);
    OCHIP_HIP(ctx, hipSetDevice(ctx->device));
    hipStream_t st = ctx->stream;
    auto *s = new ochip_plane_setup();
    s->ctx = ctx;
    s->n_inliers = n_inliers;
    int rc = OCHIP_OK;
    auto up = [&](const void *src, size_t bytes) -> void * {
        size_t got = 0;
        void *p = ochip_pool_get(ctx, bytes ? bytes : 16, &got);
        if (!p)
        {
            rc = ochip_fail(ctx, OCHIP_ENOMEM, "ochip_plane_setup_create: device allocation of %zu bytes failed", bytes);
            return nullptr;
        }
        s->allocs.emplace_back(p, got);
        if (src && bytes && hipMemcpyAsync(p, src, bytes, hipMemcpyHostToDevice, st) != hipSuccess)
            rc = ochip_fail(ctx, OCHIP_EHIP, "ochip_plane_setup_create: upload failed");
        return p;
    };
    setup_dev &D = s->dev;
    D.n_edges = n_edges;
    D.edges = (const ochip_plane_edge *)up(edges, (size_t)n_edges * sizeof(ochip_plane_edge));
    D.inliers = (const ochip_plane_inlier *)up(inliers, (size_t)n_inliers * sizeof(ochip_plane_inlier));
    D.cam_pos = (const double *)up(cam_pos, (size_t)n_cams * 24);
    D.cam_q = (const double *)up(cam_q, (size_t)n_cams * 32);
    D.models = (const double *)up(models10, (size_t)n_models * 80);
    D.score = (double *)up(nullptr, (size_t)n_inliers * 8);
    D.keep = (unsigned char *)up(nullptr, (size_t)n_inliers);
    D.inexact = (unsigned char *)up(nullptr, (size_t)n_edges);
    D.count = (unsigned int *)up(nullptr, ((size_t)n_edges + 1) * 4);
    D.blk_a = (uint32_t *)up(nullptr, (size_t)n_inliers * 4);
    D.blk_b = (uint32_t *)up(nullptr, (size_t)n_inliers * 4);
    D.blk_rays = (double *)up(nullptr, (size_t)n_inliers * 48);
    for (int i = 0; i < 6; i++)
        D.tri[i] = triangle_xy6[i];
    D.res = grid_fraction;
    if (rc != OCHIP_OK)
    {
        ochip_plane_setup_destroy(s);
        return rc;
    }
    const bool table_fits = 1.0 / grid_fraction < (double)(G_MAX - 1);
    if (n_edges && table_fits)
        hipLaunchKernelGGL(plane_filter_kernel, dim3((n_edges + 3) / 4), dim3(256), 0, st, D);
    hipError_t e = hipGetLastError();
    if (e == hipSuccess && n_inliers && table_fits)
        e = hipMemcpyAsync(keep_out, D.keep, (size_t)n_inliers, hipMemcpyDeviceToHost, st);
    if (e == hipSuccess && n_edges && table_fits)
        e = hipMemcpyAsync(inexact_out, D.inexact, (size_t)n_edges, hipMemcpyDeviceToHost, st);
    if (e == hipSuccess)
        e = ochip_stream_wait(ctx, st); // (also: the pageable sources of the uploads have been consumed)
    if (e != hipSuccess)
    {
        ochip_plane_setup_destroy(s);
        return ochip_fail(ctx, OCHIP_EHIP, "ochip_plane_setup_create: %s", hipGetErrorString(e));
    }
    if (!table_fits) // a grid finer than the per-wave tables: every edge is the caller's
    {
        for (uint32_t k = 0; k < n_edges; k++)
            inexact_out[k] = 1;
        for (uint64_t k = 0; k < n_inliers; k++)
            keep_out[k] = 0;
    }
    *out = s;
    return OCHIP_OK;
}

int ochip_plane_setup_override(ochip_plane_setup *s, uint64_t first_inlier, uint64_t n, const uint8_t *keep)
{
    if (!s || (n && !keep) || first_inlier + n > s->n_inliers)
        return OCHIP_EINVAL;
    ochip_ctx *ctx = s->ctx;
    OCHIP_HIP(ctx, hipSetDevice(ctx->device));
    s->emitted = false;
    if (n)
    {
        OCHIP_HIP(ctx, hipMemcpyAsync(s->dev.keep + first_inlier, keep, (size_t)n, hipMemcpyHostToDevice, ctx->stream));
        OCHIP_HIP(ctx, ochip_stream_wait(ctx, ctx->stream));
    }
    return OCHIP_OK;
}

int ochip_plane_setup_blocks(ochip_plane_setup *s, uint32_t *blk_a, uint32_t *blk_b, double *blk_rays, uint64_t capacity,
                             uint64_t *n_blocks)
{
    if (!s || !n_blocks)
        return OCHIP_EINVAL;
    ochip_ctx *ctx = s->ctx;
    OCHIP_HIP(ctx, hipSetDevice(ctx->device));
    hipStream_t st = ctx->stream;
    const setup_dev &D = s->dev;
    *n_blocks = 0;
    if (D.n_edges == 0)
        return OCHIP_OK;
    if (!s->emitted)
    {
        hipLaunchKernelGGL(plane_blocks_kernel<false>, dim3((D.n_edges + 3) / 4), dim3(256), 0, st, D);
        hipLaunchKernelGGL(plane_scan_kernel, dim3(1), dim3(1024), 0, st, D.count, D.n_edges);
        hipLaunchKernelGGL(plane_blocks_kernel<true>, dim3((D.n_edges + 3) / 4), dim3(256), 0, st, D);
        OCHIP_HIP(ctx, hipGetLastError());
        unsigned int total_dev = 0;
        OCHIP_HIP(ctx, hipMemcpyAsync(&total_dev, D.count + D.n_edges, 4, hipMemcpyDeviceToHost, st));
        OCHIP_HIP(ctx, ochip_stream_wait(ctx, st));
        s->total = total_dev;
        s->emitted = true;
    }
    const uint64_t total = s->total;
    *n_blocks = total;
    if (!blk_a && !blk_b && !blk_rays) // the count only
        return OCHIP_OK;
    if (total > capacity)
        return ochip_fail(ctx, OCHIP_ENOMEM, "ochip_plane_setup_blocks: %llu blocks, room for %llu", (unsigned long long)total,
                          (unsigned long long)capacity);
    if (total)
    {
        if (!blk_a || !blk_b || !blk_rays)
            return ochip_fail(ctx, OCHIP_EINVAL, "ochip_plane_setup_blocks: NULL output");
        OCHIP_HIP(ctx, hipMemcpyAsync(blk_a, D.blk_a, (size_t)total * 4, hipMemcpyDeviceToHost, st));
        OCHIP_HIP(ctx, hipMemcpyAsync(blk_b, D.blk_b, (size_t)total * 4, hipMemcpyDeviceToHost, st));
        OCHIP_HIP(ctx, hipMemcpyAsync(blk_rays, D.blk_rays, (size_t)total * 48, hipMemcpyDeviceToHost, st));
        OCHIP_HIP(ctx, ochip_stream_wait(ctx, st));
    }
    return OCHIP_OK;
}

} // extern "C"
