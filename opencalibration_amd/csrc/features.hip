// libochip.so — extract_features' tail on the device (gfx950).
//
// src/extract/extract_features.cpp:38-87 turns the keypoints cv::AKAZE returned into the image's feature list: rescale to
// original-image pixels, std::sort by descending response, a greedy 8 px non-maximum suppression in that order through a
// KD-tree, output [sparse..., dense...].  All of it runs here, per chunk of images:
//   * the ORDER is libstdc++'s unstable std::sort's, reproduced move for move by std_sort.hip on (response, detection
//     index) records - it decides the result wherever two responses are equal, which happens in almost every image;
//   * the greedy suppression, as a fixed point: a feature is sparse when every stronger feature within the radius is dense
//     and dense as soon as one of them is sparse (the greedy pass's answer, reached in as many rounds as the longest chain
//     of undecided neighbours) - same fp64 distance test as the reference's KD-tree query
//     (nn[0].distance * scale^2 > radius^2, extract_features.cpp:72);
//   * the image's whole output list [sparse..., dense...] as 88-byte feature_2d records (location = pt / scale in fp64,
//     strength, 486-bit descriptor): the host copies it in one piece.
// The one thing not restated on the device is the heap sort libstdc++ falls back to at introsort's depth limit: an image
// whose responses drive it there (none has) is flagged, and the host then sorts and suppresses that image itself from the
// records (slot[s]: where detection index s went), as it did for every image before round 3.
#include "ctx.hpp"

#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <vector>

using namespace ochip;

namespace
{

constexpr float CELL = 8.5f; // grid cell in working-image pixels: > radius, so 3 x 3 cells hold every neighbour
enum : unsigned char
{
    UNDECIDED = 0,
    SPARSE = 1,
    DENSE = 2
};

// internal arrays: rows of S entries per image (S >= the longest list of the chunk); host arrays: rows of max_kp
struct feat_dev
{
    unsigned int B, S, max_kp;
    const float *kp6;                 // [B][max_kp][6], detection order
    const unsigned long long *desc;   // [B][max_kp][8]
    const unsigned int *counts;       // [B]
    unsigned long long *recs;         // [B][S] response key << 32 | detection index; sorted in place (std_sort.hip)
    unsigned int *seg_begin, *seg_end; // [B] the images' segments of recs
    float *resp;                      // [B][S] responses in detection order
    double2 *loc;                     // [B][S] pt / scale of the r-th strongest
    unsigned int *cell;               // [B][S] its grid cell
    int gw, gh;
    unsigned int *cell_start, *cell_fill; // [B][gw * gh + 1], [B][gw * gh]
    unsigned int *items;              // [B][S] features by cell
    unsigned char *state;             // [B][S]
    unsigned int *slot_of_rank;       // [B][S] output slot of the r-th strongest
    unsigned int *slot;               // [B][S] output slot of detection index s (the seed: its sparse slot, 0)
    unsigned int *n_sparse;           // [B]
    unsigned char *conflict;          // [B]
    unsigned char *records;           // [B][S + 1][88]
    double scale, scale2, radius2;
};

__device__ __forceinline__ unsigned int feat_count(const feat_dev &F, unsigned int b)
{
    return min(F.counts[b], F.max_kp);
}
__device__ __forceinline__ bool feat_within(const feat_dev &F, const double2 &a, const double2 &c)
{
    const double dx = a.x - c.x, dy = a.y - c.y;
    double d = 0;
    d += dx * dx;
    d += dy * dy;
    return !(d * F.scale2 > F.radius2);
}

__global__ void feat_keys_kernel(feat_dev F)
{
    const unsigned int b = blockIdx.z, i = blockIdx.x * blockDim.x + threadIdx.x;
    const unsigned int n = feat_count(F, b);
    if (i == 0)
    {
        F.seg_begin[b] = b * F.S;
        F.seg_end[b] = b * F.S + n;
    }
    if (i >= n)
        return;
    const size_t o = (size_t)b * F.S + i;
    const float r = F.kp6[((size_t)b * F.max_kp + i) * 6 + 4];
    // responses are positive floats (determinants above the detector threshold): they order like their bit patterns;
    // anything else (never produced by the detector) still gets a total order
    const unsigned int u = __float_as_uint(r);
    const unsigned int key = (u & 0x80000000u) ? ~u : (u | 0x80000000u);
    F.resp[o] = r;
    F.recs[o] = ((unsigned long long)key << 32) | i;
}

// per feature in strength order: location, grid cell, the cell's population
__global__ void feat_cells_kernel(feat_dev F)
{
    const unsigned int b = blockIdx.z, r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= feat_count(F, b))
        return;
    const size_t o = (size_t)b * F.S + r;
    const size_t src = (size_t)b * F.max_kp + (unsigned int)F.recs[o];
    const float x = F.kp6[src * 6], y = F.kp6[src * 6 + 1];
    F.loc[o] = make_double2((double)x / F.scale, (double)y / F.scale); // keypoints[i].pt / scale, extract_features.cpp:44-45
    const int cx = min(max((int)(x / CELL), 0), F.gw - 1), cy = min(max((int)(y / CELL), 0), F.gh - 1);
    const unsigned int c = (unsigned int)(cy * F.gw + cx);
    F.cell[o] = c;
    atomicAdd(&F.cell_fill[(size_t)b * F.gw * F.gh + c], 1u);
}

// exclusive scan of an image's cell populations; the fill cursors are zeroed for the next kernel.  One workgroup per image.
constexpr int SCAN_THREADS = 256; // (a 1 024-thread workgroup waits for sixteen free wave slots on one CU: under the other
                                  // launch sequences' kernels that took up to 12 ms)
__global__ __launch_bounds__(SCAN_THREADS) void feat_scan_kernel(feat_dev F)
{
    __shared__ unsigned int s_scan[SCAN_THREADS];
    const unsigned int b = blockIdx.x, t = threadIdx.x;
    const int n_cells = F.gw * F.gh;
    unsigned int *start = F.cell_start + (size_t)b * (n_cells + 1), *fill = F.cell_fill + (size_t)b * n_cells;
    const int per = (n_cells + SCAN_THREADS - 1) / SCAN_THREADS;
    const int c0 = min((int)t * per, n_cells), c1 = min(c0 + per, n_cells);
    unsigned int sum = 0;
    for (int c = c0; c < c1; c++)
        sum += fill[c];
    s_scan[t] = sum;
    __syncthreads();
    for (int off = 1; off < SCAN_THREADS; off <<= 1)
    {
        const unsigned int v = t >= (unsigned int)off ? s_scan[t - off] : 0u;
        __syncthreads();
        s_scan[t] += v;
        __syncthreads();
    }
    unsigned int run = s_scan[t] - sum;
    for (int c = c0; c < c1; c++)
    {
        const unsigned int k = fill[c];
        start[c] = run;
        fill[c] = 0;
        run += k;
    }
    if (t == SCAN_THREADS - 1)
        start[n_cells] = s_scan[SCAN_THREADS - 1];
}

__global__ void feat_fill_kernel(feat_dev F)
{
    const unsigned int b = blockIdx.z, r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= feat_count(F, b))
        return;
    const size_t o = (size_t)b * F.S + r;
    const int n_cells = F.gw * F.gh;
    const unsigned int c = F.cell[o];
    const unsigned int at = F.cell_start[(size_t)b * (n_cells + 1) + c] + atomicAdd(&F.cell_fill[(size_t)b * n_cells + c], 1u);
    F.items[(size_t)b * F.S + at] = r;
    F.state[o] = r == 0 ? SPARSE : UNDECIDED; // the strongest feature seeds the sparse list (extract_features.cpp:60-62)
}

// the suppression's rule for feature r given the states of the stronger features near it (states only ever go from
// UNDECIDED to their final value, so reading them while other threads decide is harmless)
template <typename StateOf>
__device__ __forceinline__ unsigned char feat_decide(const feat_dev &F, unsigned int b, unsigned int r, StateOf state_of)
{
    const size_t base = (size_t)b * F.S;
    const int n_cells = F.gw * F.gh;
    const unsigned int *start = F.cell_start + (size_t)b * (n_cells + 1);
    const unsigned int *items = F.items + base;
    const double2 me = F.loc[base + r];
    const int c = (int)F.cell[base + r], cx = c % F.gw, cy = c / F.gw;
    bool open_near = false;
    for (int yy = max(cy - 1, 0); yy <= min(cy + 1, F.gh - 1); yy++)
    {
        // three neighbouring cells of a row are one run of the item list
        const unsigned int i0 = start[yy * F.gw + max(cx - 1, 0)], i1 = start[yy * F.gw + min(cx + 1, F.gw - 1) + 1];
        for (unsigned int i = i0; i < i1; i++)
        {
            const unsigned int q = items[i];
            if (q >= r)
                continue;
            const unsigned char sq = state_of(q);
            if (sq == DENSE)
                continue;
            if (!feat_within(F, me, F.loc[base + q]))
                continue;
            if (sq == SPARSE)
                return DENSE;
            open_near = true;
        }
    }
    return open_near ? UNDECIDED : SPARSE;
}

// one round over every feature of every image (the bulk is decided after a few of these)
__global__ void feat_round_kernel(feat_dev F)
{
    const unsigned int b = blockIdx.z, r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= feat_count(F, b))
        return;
    unsigned char *st = F.state + (size_t)b * F.S;
    if (st[r] != UNDECIDED)
        return;
    const unsigned char s = feat_decide(F, b, r, [&](unsigned int q) { return st[q]; });
    if (s != UNDECIDED)
        st[r] = s;
}

// One workgroup per image: the rounds that are left (in LDS), the order-dependence check, the output slots.
constexpr int FIN_THREADS = 256;
__global__ __launch_bounds__(FIN_THREADS) void feat_finish_kernel(feat_dev F)
{
    extern __shared__ unsigned char state[]; // [S]
    __shared__ unsigned int s_scan[FIN_THREADS];
    __shared__ int s_again;
    const unsigned int b = blockIdx.x, t = threadIdx.x;
    const unsigned int n = feat_count(F, b);
    const size_t base = (size_t)b * F.S;
    if (t == 0)
        s_again = 0;
    for (unsigned int r = t; r < n; r += FIN_THREADS)
        state[r] = F.state[base + r];
    __syncthreads();
    for (;;)
    {
        bool any_left = false;
        for (unsigned int r = t; r < n; r += FIN_THREADS)
        {
            if (state[r] != UNDECIDED)
                continue;
            const unsigned char s = feat_decide(F, b, r, [&](unsigned int q) { return state[q]; });
            if (s != UNDECIDED)
                state[r] = s;
            else
                any_left = true;
        }
        if (any_left)
            s_again = 1;
        __syncthreads();
        const int again = s_again;
        __syncthreads();
        if (t == 0)
            s_again = 0;
        if (!again)
            break;
        __syncthreads();
    }
    // ---- output slots: sparse features in strength order, then the dense ones, headed by the seed (visited again by the
    //      reference's loop, extract_features.cpp:64-66).  Exclusive scan of the sparse flags, a contiguous run per thread.
    const unsigned int per = (n + FIN_THREADS - 1) / FIN_THREADS;
    const unsigned int r0 = min(t * per, n), r1 = min(r0 + per, n);
    unsigned int sum = 0;
    for (unsigned int r = r0; r < r1; r++)
        sum += state[r] == SPARSE ? 1u : 0u;
    s_scan[t] = sum;
    __syncthreads();
    for (int off = 1; off < FIN_THREADS; off <<= 1)
    {
        const unsigned int v = t >= (unsigned int)off ? s_scan[t - off] : 0u;
        __syncthreads();
        s_scan[t] += v;
        __syncthreads();
    }
    const unsigned int total_sparse = s_scan[FIN_THREADS - 1];
    unsigned int before = s_scan[t] - sum; // sparse features among the ranks below r0
    for (unsigned int r = r0; r < r1; r++)
    {
        const bool sp = state[r] == SPARSE;
        // a dense feature's slot: after the sparse list and the seed's second entry, in rank order among the dense ones
        F.slot_of_rank[base + r] = sp ? before : total_sparse + 1u + (r - before);
        before += sp ? 1u : 0u;
    }
    __syncthreads();
    if (t == 0)
        F.n_sparse[b] = n ? total_sparse : 0u;
}

// the records at their slots (the seed twice: slot 0 and slot n_sparse), slot[s] for the host's re-seating
__global__ void feat_records_kernel(feat_dev F)
{
    const unsigned int b = blockIdx.z, r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= feat_count(F, b))
        return;
    const size_t o = (size_t)b * F.S + r;
    const unsigned int s = (unsigned int)F.recs[o];
    const size_t src = (size_t)b * F.max_kp + s;
    unsigned long long rec[11];
    const double2 l = F.loc[o];
    rec[0] = (unsigned long long)__double_as_longlong(l.x);
    rec[1] = (unsigned long long)__double_as_longlong(l.y);
    rec[2] = (unsigned long long)__float_as_uint(F.kp6[src * 6 + 4]); // strength, then 4 bytes of padding
#pragma unroll
    for (int w = 0; w < 8; w++)
        rec[3 + w] = F.desc[src * 8 + w];
    const unsigned int at = F.slot_of_rank[o];
    unsigned long long *out = reinterpret_cast<unsigned long long *>(F.records + ((size_t)b * (F.S + 1) + at) * 88);
#pragma unroll
    for (int w = 0; w < 11; w++)
        out[w] = rec[w];
    if (r == 0)
    {
        unsigned long long *again = reinterpret_cast<unsigned long long *>(F.records + ((size_t)b * (F.S + 1) + F.n_sparse[b]) * 88);
#pragma unroll
        for (int w = 0; w < 11; w++)
            again[w] = rec[w];
    }
    F.slot[(size_t)b * F.S + s] = at;
}

} // namespace

namespace ochip
{

// d_kp6 / d_desc / d_counts: the compacted keypoints of B images (stride max_kp) in detection order, on the device.
// Enqueues everything on the context's stream and the copies into `out` (host arrays, page-locked ones copy at link
// speed); the caller waits for the stream.  Device blocks are recorded in `allocs` (returned to the pool by the caller
// after that wait).  `most`: the longest list of the chunk (rows are copied up to it).
int feature_lists_enqueue(ochip_ctx *ctx, std::vector<std::pair<void *, size_t>> *allocs, uint32_t B, uint32_t max_kp,
                          const float *d_kp6, const unsigned long long *d_desc, const unsigned int *d_counts, uint32_t most,
                          int work_w, int work_h, double scale, double nms_radius, const ochip_feature_lists *out)
{
    hipStream_t st = ctx->stream;
    if (B == 0)
        return OCHIP_OK;
    if (most == 0)
    {
        std::memset(out->conflict, 0, B);
        std::memset(out->num_sparse, 0, (size_t)B * 4);
        return OCHIP_OK;
    }
    if (most > 60000)
        return ochip_fail(ctx, OCHIP_EINVAL, "feature lists: %u keypoints in one image exceed the suppression kernel's 60 000", most);
    feat_dev F{};
    F.B = B;
    F.S = (most + 63) / 64 * 64;
    F.max_kp = max_kp;
    F.kp6 = d_kp6;
    F.desc = d_desc;
    F.counts = d_counts;
    F.gw = (int)((float)work_w / CELL) + 2;
    F.gh = (int)((float)work_h / CELL) + 2;
    F.scale = scale;
    F.scale2 = scale * scale;
    F.radius2 = nms_radius * nms_radius;
    const size_t N = (size_t)B * F.S, n_cells = (size_t)F.gw * F.gh;
    auto dev = [&](size_t bytes) -> void * {
        size_t got = 0;
        void *p = ochip_pool_get(ctx, std::max<size_t>(bytes, 16), &got);
        if (p)
            allocs->emplace_back(p, got);
        return p;
    };
    F.recs = (unsigned long long *)dev(N * 8);
    F.seg_begin = (unsigned int *)dev((size_t)B * 4);
    F.seg_end = (unsigned int *)dev((size_t)B * 4);
    F.resp = (float *)dev(N * 4);
    F.loc = (double2 *)dev(N * 16);
    F.cell = (unsigned int *)dev(N * 4);
    F.cell_start = (unsigned int *)dev((size_t)B * (n_cells + 1) * 4);
    F.cell_fill = (unsigned int *)dev((size_t)B * n_cells * 4);
    F.items = (unsigned int *)dev(N * 4);
    F.state = (unsigned char *)dev(N);
    F.slot_of_rank = (unsigned int *)dev(N * 4);
    F.slot = (unsigned int *)dev(N * 4);
    F.n_sparse = (unsigned int *)dev((size_t)B * 4);
    F.conflict = (unsigned char *)dev(B);
    F.records = (unsigned char *)dev((size_t)B * (F.S + 1) * 88);
    if (!F.recs || !F.seg_begin || !F.seg_end || !F.resp || !F.loc || !F.cell || !F.cell_start || !F.cell_fill || !F.items || !F.state ||
        !F.slot_of_rank || !F.slot || !F.n_sparse || !F.conflict || !F.records)
        return ochip_fail(ctx, OCHIP_ENOMEM, "feature lists: device allocation failed");
    const dim3 wide((F.S + 255) / 256, 1, B);
    hipLaunchKernelGGL(feat_keys_kernel, wide, dim3(256), 0, st, F);
    // the strength order: std::sort by descending response from detection order, as the reference's (std_sort.hip);
    // conflict[b] = that image ran into introsort's depth limit and is the host's
    {
        const int src = std_sort_enqueue(ctx, allocs, F.recs, N, F.seg_begin, F.seg_end, B, most, F.conflict);
        if (src != OCHIP_OK)
            return src;
    }
    OCHIP_HIP(ctx, hipMemsetAsync(F.cell_fill, 0, (size_t)B * n_cells * 4, st));
    hipLaunchKernelGGL(feat_cells_kernel, wide, dim3(256), 0, st, F);
    hipLaunchKernelGGL(feat_scan_kernel, dim3(B), dim3(SCAN_THREADS), 0, st, F);
    hipLaunchKernelGGL(feat_fill_kernel, wide, dim3(256), 0, st, F);
    static const int rounds = getenv("OCHIP_FEATURE_ROUNDS") ? std::max(0, atoi(getenv("OCHIP_FEATURE_ROUNDS"))) : 6;
    for (int k = 0; k < rounds; k++)
        hipLaunchKernelGGL(feat_round_kernel, wide, dim3(256), 0, st, F);
    hipLaunchKernelGGL(feat_finish_kernel, dim3(B), dim3(FIN_THREADS), (size_t)F.S, st, F);
    hipLaunchKernelGGL(feat_records_kernel, wide, dim3(256), 0, st, F);
    OCHIP_HIP(ctx, hipGetLastError());
    OCHIP_HIP(ctx, hipMemcpy2DAsync(out->records, ((size_t)max_kp + 1) * 88, F.records, ((size_t)F.S + 1) * 88, ((size_t)most + 1) * 88, B,
                                    hipMemcpyDeviceToHost, st));
    OCHIP_HIP(ctx, hipMemcpy2DAsync(out->response, (size_t)max_kp * 4, F.resp, (size_t)F.S * 4, (size_t)most * 4, B, hipMemcpyDeviceToHost, st));
    OCHIP_HIP(ctx, hipMemcpy2DAsync(out->slot, (size_t)max_kp * 4, F.slot, (size_t)F.S * 4, (size_t)most * 4, B, hipMemcpyDeviceToHost, st));
    OCHIP_HIP(ctx, hipMemcpyAsync(out->num_sparse, F.n_sparse, (size_t)B * 4, hipMemcpyDeviceToHost, st));
    OCHIP_HIP(ctx, hipMemcpyAsync(out->conflict, F.conflict, B, hipMemcpyDeviceToHost, st));
    return OCHIP_OK;
}

} // namespace ochip

extern "C" int ochip_feature_lists_from_keypoints(ochip_ctx *ctx, const float *kp6, const uint64_t *desc, const uint32_t *counts,
                                                  uint32_t n_images, uint32_t max_kp, int work_w, int work_h, double scale,
                                                  double nms_radius, const ochip_feature_lists *out)
{
    if (!ctx || !out || !counts || (n_images && (!kp6 || !desc)) || !out->records || !out->response || !out->slot || !out->num_sparse ||
        !out->conflict)
        return OCHIP_EINVAL;
    OCHIP_HIP(ctx, hipSetDevice(ctx->device));
    hipStream_t st = ctx->stream;
    std::vector<std::pair<void *, size_t>> allocs;
    auto cleanup = [&]() {
        (void)ochip_stream_wait(ctx, st);
        for (auto &a : allocs)
            ochip_pool_put(ctx, a.first, a.second);
    };
    uint32_t most = 0;
    for (uint32_t b = 0; b < n_images; b++)
    {
        if (counts[b] > max_kp)
            return ochip_fail(ctx, OCHIP_EINVAL, "image %u: %u keypoints, more than max_kp = %u", b, counts[b], max_kp);
        most = std::max(most, counts[b]);
    }
    const size_t N = (size_t)n_images * max_kp;
    size_t g0 = 0, g1 = 0, g2 = 0;
    float *d_kp = (float *)ochip_pool_get(ctx, std::max<size_t>(N * 24, 16), &g0);
    unsigned long long *d_desc = (unsigned long long *)ochip_pool_get(ctx, std::max<size_t>(N * 64, 16), &g1);
    unsigned int *d_counts = (unsigned int *)ochip_pool_get(ctx, std::max<size_t>((size_t)n_images * 4, 16), &g2);
    if (d_kp)
        allocs.emplace_back(d_kp, g0);
    if (d_desc)
        allocs.emplace_back(d_desc, g1);
    if (d_counts)
        allocs.emplace_back(d_counts, g2);
    int rc = OCHIP_OK;
    if (!d_kp || !d_desc || !d_counts)
        rc = ochip_fail(ctx, OCHIP_ENOMEM, "feature lists: device allocation failed");
    if (rc == OCHIP_OK && N &&
        (hipMemcpyAsync(d_kp, kp6, N * 24, hipMemcpyHostToDevice, st) != hipSuccess ||
         hipMemcpyAsync(d_desc, desc, N * 64, hipMemcpyHostToDevice, st) != hipSuccess ||
         hipMemcpyAsync(d_counts, counts, (size_t)n_images * 4, hipMemcpyHostToDevice, st) != hipSuccess))
        rc = ochip_fail(ctx, OCHIP_EHIP, "feature lists: upload failed");
    if (rc == OCHIP_OK)
        rc = feature_lists_enqueue(ctx, &allocs, n_images, max_kp, d_kp, d_desc, d_counts, most, work_w, work_h, scale, nms_radius, out);
    cleanup();
    return rc;
}
