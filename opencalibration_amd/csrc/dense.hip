// libochip.so — the descriptor search of dense guided matching (reference: densifyMesh, src/dense/dense_stereo.cpp:
// 245-283): for a query (a dense feature of one image, a pixel predicted in another image) the dense features of that
// image inside the 150 px disc around the prediction are compared by Hamming distance; the nearest, the second nearest
// and the number of features in the disc come back.  The ratio / absolute threshold decision stays on the host in fp64
// exactly as the reference writes it (match.hip does the same for the sparse matcher).
//
// Layout in HBM (ochip_dense_index): every image's dense features sorted by the cell of a uniform grid over the image
// (cell edge > search radius, so a disc touches at most 3 x 3 cells and a cell row's three cells are one contiguous run):
// descriptors [total][8] u64, locations [total] double2, per image the cell start table.  One wavefront per query: the
// lanes stride over the (at most three) runs, test the squared pixel distance in fp64 (`<`, as jk::KDTree's ball
// query, KDTree.h:398), XOR + popcount the 64-byte descriptor against the query's (wave-uniform, scalar registers), keep
// (best, second, index, count) per lane and merge across the wave with a tie-aware butterfly.  Bound: L2 -> L1 bytes of
// the candidates' descriptors (64 B per candidate in the disc, 16 B per candidate in the 3 x 3 cells); queries that
// follow each other hit the same cells (the host emits them along the source image's Hilbert walk).
#include "ctx.hpp"

#include <new>

namespace
{

struct dense_image_meta
{
    uint64_t feat_base, cell_base;
    int32_t ncx, ncy;
    double ox, oy;
};

constexpr uint32_t NONE_COUNT = 0xFFFFu;

struct lane_state
{
    uint32_t best, second, idx, count;
};

__device__ __forceinline__ lane_state merge(const lane_state &a, const lane_state &b)
{
    lane_state r;
    r.count = a.count + b.count;
    if (a.best < b.best)
    {
        r.best = a.best;
        r.idx = a.idx;
        r.second = min(a.second, b.best);
    }
    else if (b.best < a.best)
    {
        r.best = b.best;
        r.idx = b.idx;
        r.second = min(b.second, a.best);
    }
    else // two candidates tie for the best distance: second best == best, the ratio test fails on the host
    {
        r.best = a.best;
        r.idx = min(a.idx, b.idx);
        r.second = a.best;
    }
    return r;
}

__global__ __launch_bounds__(256) void dense_match_kernel(const dense_image_meta *__restrict__ meta, const uint64_t *__restrict__ desc,
                                                          const double2 *__restrict__ loc, const uint32_t *__restrict__ cell_start,
                                                          const ochip_dense_query *__restrict__ queries, uint64_t n_queries,
                                                          double radius_sq, double cell_size, ochip_dense_result *__restrict__ out)
{
    const uint64_t qi = (uint64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (qi >= n_queries)
        return;
    const int lane = threadIdx.x & 63;
    const ochip_dense_query q = queries[qi];
    const dense_image_meta m = meta[q.cand_image];
    uint64_t qd[8];
    {
        const uint64_t *p = desc + 8 * (size_t)q.src_feature;
#pragma unroll
        for (int w = 0; w < 8; w++)
            qd[w] = p[w];
    }
    const int cx = (int)floor((q.px - m.ox) / cell_size), cy = (int)floor((q.py - m.oy) / cell_size);
    lane_state s{NONE_COUNT, NONE_COUNT, 0xFFFFFFFFu, 0u};
    const int c0 = max(cx - 1, 0), c1 = min(cx + 1, m.ncx - 1);
    if (c0 <= c1)
        for (int ry = max(cy - 1, 0); ry <= min(cy + 1, m.ncy - 1); ry++)
        {
            const uint32_t *cs = cell_start + m.cell_base + (size_t)ry * m.ncx;
            const uint32_t begin = cs[c0], end = cs[c1 + 1];
            for (uint32_t k = begin + lane; k < end; k += 64)
            {
                const double2 p = loc[m.feat_base + k];
                const double dx = p.x - q.px, dy = p.y - q.py;
                if (!(dx * dx + dy * dy < radius_sq))
                    continue;
                const uint64_t *cd = desc + 8 * (m.feat_base + k);
                uint32_t d = 0;
#pragma unroll
                for (int w = 0; w < 8; w++)
                    d += (uint32_t)__popcll(cd[w] ^ qd[w]);
                s.count++;
                if (d < s.second) // the reference's sequential rule (:262-276)
                {
                    if (d < s.best)
                    {
                        s.second = s.best;
                        s.best = d;
                        s.idx = k;
                    }
                    else
                        s.second = d;
                }
            }
        }
    for (int off = 32; off >= 1; off >>= 1)
    {
        lane_state o;
        o.best = __shfl_xor(s.best, off);
        o.second = __shfl_xor(s.second, off);
        o.idx = __shfl_xor(s.idx, off);
        o.count = __shfl_xor(s.count, off);
        s = merge(s, o);
    }
    if (lane == 0)
    {
        ochip_dense_result r;
        r.best_feature = s.idx;
        r.best_count = (uint16_t)s.best;
        r.second_count = (uint16_t)s.second;
        r.nearby = s.count;
        out[qi] = r;
    }
}

} // namespace

struct ochip_dense_index
{
    ochip_ctx *ctx = nullptr;
    uint32_t n_images = 0;
    uint64_t total_features = 0;
    double cell_size = 0;
    std::vector<std::pair<void *, size_t>> blocks;
    dense_image_meta *meta = nullptr;
    uint64_t *desc = nullptr;
    double2 *loc = nullptr;
    uint32_t *cell_start = nullptr;
};

extern "C"
{

int ochip_dense_index_create(ochip_ctx *ctx, uint32_t n_images, const uint64_t *feat_off, const uint64_t *desc8, const double *loc2,
                             const uint64_t *cell_off, const uint32_t *cell_start, const int32_t *grid2, const double *origin2,
                             double cell_size, ochip_dense_index **out)
{
    if (!ctx || !out || !feat_off || !cell_off || !grid2 || !origin2 || !(cell_size > 0))
        return ctx ? ochip_fail(ctx, OCHIP_EINVAL, "ochip_dense_index_create: bad argument") : OCHIP_EINVAL;
    OCHIP_HIP(ctx, hipSetDevice(ctx->device));
    ochip_dense_index *ix = new (std::nothrow) ochip_dense_index();
    if (!ix)
        return ochip_fail(ctx, OCHIP_ENOMEM, "out of host memory");
    ix->ctx = ctx;
    ix->n_images = n_images;
    ix->total_features = feat_off[n_images];
    ix->cell_size = cell_size;
    std::vector<dense_image_meta> meta(n_images ? n_images : 1);
    for (uint32_t i = 0; i < n_images; i++)
    {
        meta[i].feat_base = feat_off[i];
        meta[i].cell_base = cell_off[i];
        meta[i].ncx = grid2[2 * i];
        meta[i].ncy = grid2[2 * i + 1];
        meta[i].ox = origin2[2 * i];
        meta[i].oy = origin2[2 * i + 1];
        if ((uint64_t)meta[i].ncx * meta[i].ncy + 1 != cell_off[i + 1] - cell_off[i])
        {
            delete ix;
            return ochip_fail(ctx, OCHIP_EINVAL, "ochip_dense_index_create: image %u has %d x %d cells but %llu table entries", i,
                              meta[i].ncx, meta[i].ncy, (unsigned long long)(cell_off[i + 1] - cell_off[i]));
        }
    }
    auto upload = [&](void **dst, const void *src, size_t bytes) {
        size_t got = 0;
        void *d = ochip_pool_get(ctx, bytes ? bytes : 16, &got);
        if (!d)
            return ochip_fail(ctx, OCHIP_ENOMEM, "device allocation of %zu bytes failed for the dense index", bytes);
        ix->blocks.emplace_back(d, got);
        if (bytes && hipMemcpyAsync(d, src, bytes, hipMemcpyHostToDevice, ctx->stream) != hipSuccess)
            return ochip_fail(ctx, OCHIP_EHIP, "hipMemcpyAsync failed for the dense index");
        *dst = d;
        return (int)OCHIP_OK;
    };
    int rc = upload((void **)&ix->meta, meta.data(), meta.size() * sizeof(dense_image_meta));
    if (rc == OCHIP_OK)
        rc = upload((void **)&ix->desc, desc8, (size_t)ix->total_features * 64);
    if (rc == OCHIP_OK)
        rc = upload((void **)&ix->loc, loc2, (size_t)ix->total_features * 16);
    if (rc == OCHIP_OK)
        rc = upload((void **)&ix->cell_start, cell_start, (size_t)cell_off[n_images] * 4);
    if (rc == OCHIP_OK && ochip_stream_wait(ctx, ctx->stream) != hipSuccess)
        rc = ochip_fail(ctx, OCHIP_EHIP, "stream wait failed for the dense index");
    if (rc != OCHIP_OK)
    {
        for (auto &b : ix->blocks)
            ochip_pool_put(ctx, b.first, b.second);
        delete ix;
        return rc;
    }
    *out = ix;
    return OCHIP_OK;
}

void ochip_dense_index_destroy(ochip_dense_index *ix)
{
    if (!ix)
        return;
    for (auto &b : ix->blocks)
        ochip_pool_put(ix->ctx, b.first, b.second);
    delete ix;
}

int ochip_dense_match(ochip_dense_index *ix, const ochip_dense_query *queries, uint64_t n_queries, double radius, ochip_dense_result *out)
{
    if (!ix || (n_queries && (!queries || !out)))
        return OCHIP_EINVAL;
    ochip_ctx *ctx = ix->ctx;
    if (!(radius > 0) || !(radius < ix->cell_size))
        return ochip_fail(ctx, OCHIP_EINVAL, "ochip_dense_match: the search radius %g must be below the index's cell size %g", radius,
                          ix->cell_size);
    if (n_queries == 0)
        return OCHIP_OK;
    OCHIP_HIP(ctx, hipSetDevice(ctx->device));
    for (uint64_t i = 0; i < n_queries; i++)
        if (queries[i].cand_image >= ix->n_images || queries[i].src_feature >= ix->total_features)
            return ochip_fail(ctx, OCHIP_EINVAL, "ochip_dense_match: query %llu is out of range", (unsigned long long)i);
    // chunks of at most 2^24 queries: bounded scratch, and the copy of chunk k + 1 overlaps nothing worth a second stream
    const uint64_t CHUNK = 1ull << 24;
    size_t got_q = 0, got_r = 0;
    const uint64_t cap = n_queries < CHUNK ? n_queries : CHUNK;
    ochip_dense_query *dq = (ochip_dense_query *)ochip_pool_get(ctx, cap * sizeof(ochip_dense_query), &got_q);
    ochip_dense_result *dr = (ochip_dense_result *)ochip_pool_get(ctx, cap * sizeof(ochip_dense_result), &got_r);
    int rc = OCHIP_OK;
    if (!dq || !dr)
        rc = ochip_fail(ctx, OCHIP_ENOMEM, "device allocation failed for %llu dense queries", (unsigned long long)cap);
    for (uint64_t at = 0; rc == OCHIP_OK && at < n_queries; at += CHUNK)
    {
        const uint64_t n = n_queries - at < CHUNK ? n_queries - at : CHUNK;
        if (hipMemcpyAsync(dq, queries + at, n * sizeof(ochip_dense_query), hipMemcpyHostToDevice, ctx->stream) != hipSuccess)
        {
            rc = ochip_fail(ctx, OCHIP_EHIP, "hipMemcpyAsync of the dense queries failed");
            break;
        }
        hipEvent_t e0, e1;
        ochip_prof_begin(ctx, OCHIP_K_DENSE, &e0, &e1);
        hipLaunchKernelGGL(dense_match_kernel, dim3((unsigned)((n + 3) / 4)), dim3(256), 0, ctx->stream, ix->meta, ix->desc, ix->loc,
                           ix->cell_start, dq, n, radius * radius, ix->cell_size, dr);
        ochip_prof_end(ctx, OCHIP_K_DENSE, e0, e1);
        if (hipGetLastError() != hipSuccess ||
            hipMemcpyAsync(out + at, dr, n * sizeof(ochip_dense_result), hipMemcpyDeviceToHost, ctx->stream) != hipSuccess ||
            ochip_stream_wait(ctx, ctx->stream) != hipSuccess)
            rc = ochip_fail(ctx, OCHIP_EHIP, "dense match launch failed: %s", hipGetErrorString(hipGetLastError()));
    }
    if (dq)
        ochip_pool_put(ctx, dq, got_q);
    if (dr)
        ochip_pool_put(ctx, dr, got_r);
    return rc;
}

} // extern "C"
