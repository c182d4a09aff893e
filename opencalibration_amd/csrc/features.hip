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
// The heap sort libstdc++ falls back to at introsort's depth limit is restated on the device as well (std_sort.hip, round 5):
// conflict[b] is only set when one of the sort's work queues overflows (their capacities rule it out), and the host then
// sorts and suppresses that image itself from the records (slot[s]: where detection index s went), as it did for every
// image before round 3.
#include "ctx.hpp"

#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <vector>

using namespace ochip;

namespace
{

enum : unsigned char
{
    UNDECIDED = 0,
    SPARSE = 1,
    DENSE = 2
};

// A greedy suppression problem per image: n[b] points in strength order (loc[b * S + r] = the r-th strongest), a point is
// kept when no kept stronger point lies within the radius: !(d * scale2 > radius2) means "within", d = dx^2 + dy^2 summed as
// the reference's KD-tree does.  The points are binned into square cells of `cell` (>= the radius in loc's units, so 3 x 3
// cells hold every neighbour).
struct nms_dev
{
    unsigned int B, S;
    const unsigned int *n;   // [B] points per image
    const double2 *loc;      // [B][S]
    double cell, scale2, radius2;
    int gw, gh;
    unsigned int *cell_id;               // [B][S]
    unsigned int *cell_start, *cell_fill; // [B][gw * gh + 1], [B][gw * gh]
    unsigned int *items;                 // [B][S] points by cell
    double2 *item_loc;                   // [B][S] their locations, in the same order (a cell row's candidates are one run)
    unsigned char *state;                // [B][S]
};

// internal arrays: rows of S entries per image (S >= the longest list of the chunk); host arrays: rows of max_kp
struct feat_dev
{
    unsigned int B, S, max_kp;
    const float *kp6;                 // [B][max_kp][6], detection order
    const unsigned long long *desc;   // [B][max_kp][8]
    const unsigned int *counts;       // [B]
    unsigned int *n;                  // [B] min(counts, max_kp)
    unsigned long long *recs;         // [B][S] response key << 32 | detection index; sorted in place (std_sort.hip)
    unsigned int *seg_begin, *seg_end; // [B] the images' segments of recs
    float *resp;                      // [B][S] responses in detection order
    double2 *loc;                     // [B][S] pt / scale of the r-th strongest
    unsigned int *slot_of_rank;       // [B][S] output slot of the r-th strongest
    unsigned int *slot;               // [B][S] output slot of detection index s (the seed: its sparse slot, 0)
    unsigned int *n_sparse;           // [B]
    unsigned char *records;           // [B][S + 1][88]
    double scale;
    // the 40 px subset of the sparse features (spatially_subsample_feature_indices, src/match/match_features.cpp:8-52)
    unsigned long long *sub_recs;     // [B][S] strength key << 32 | sparse slot; sorted in place
    unsigned int *sub_begin, *sub_end; // [B]
    double2 *sub_loc;                 // [B][S] location of the r-th strongest sparse feature
    unsigned int *subset;             // [B][S] accepted sparse slots in strength order
    unsigned int *n_subset;           // [B]
};

__device__ __forceinline__ bool nms_within(const nms_dev &N, const double2 &a, const double2 &c)
{
    const double dx = a.x - c.x, dy = a.y - c.y;
    double d = 0;
    d += dx * dx;
    d += dy * dy;
    return !(d * N.scale2 > N.radius2);
}

__global__ void feat_keys_kernel(feat_dev F)
{
    const unsigned int b = blockIdx.z, i = blockIdx.x * blockDim.x + threadIdx.x;
    const unsigned int n = min(F.counts[b], F.max_kp);
    if (i == 0)
    {
        F.seg_begin[b] = b * F.S;
        F.seg_end[b] = b * F.S + n;
        F.n[b] = n;
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

// locations in strength order: keypoints[i].pt / scale, extract_features.cpp:44-45
__global__ void feat_loc_kernel(feat_dev F)
{
    const unsigned int b = blockIdx.z, r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= F.n[b])
        return;
    const size_t o = (size_t)b * F.S + r;
    const size_t src = (size_t)b * F.max_kp + (unsigned int)F.recs[o];
    F.loc[o] = make_double2((double)F.kp6[src * 6] / F.scale, (double)F.kp6[src * 6 + 1] / F.scale);
}

// ---- the suppression ------------------------------------------------------------------------------------------------
__global__ void nms_cells_kernel(nms_dev N)
{
    const unsigned int b = blockIdx.z, r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= N.n[b])
        return;
    const size_t o = (size_t)b * N.S + r;
    const double2 p = N.loc[o];
    const int cx = min(max((int)(p.x / N.cell), 0), N.gw - 1), cy = min(max((int)(p.y / N.cell), 0), N.gh - 1);
    const unsigned int c = (unsigned int)(cy * N.gw + cx);
    N.cell_id[o] = c;
    atomicAdd(&N.cell_fill[(size_t)b * N.gw * N.gh + c], 1u);
}

// exclusive scan of an image's cell populations; the fill cursors are zeroed for the next kernel.  One workgroup per image.
constexpr int SCAN_THREADS = 256; // (a 1 024-thread workgroup waits for sixteen free wave slots on one CU: under the other
                                  // launch sequences' kernels that took up to 12 ms)
__global__ __launch_bounds__(SCAN_THREADS) void nms_scan_kernel(nms_dev N)
{
    __shared__ unsigned int s_scan[SCAN_THREADS];
    const unsigned int b = blockIdx.x, t = threadIdx.x;
    const int n_cells = N.gw * N.gh;
    unsigned int *start = N.cell_start + (size_t)b * (n_cells + 1), *fill = N.cell_fill + (size_t)b * n_cells;
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

__global__ void nms_fill_kernel(nms_dev N)
{
    const unsigned int b = blockIdx.z, r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= N.n[b])
        return;
    const size_t o = (size_t)b * N.S + r;
    const int n_cells = N.gw * N.gh;
    const unsigned int c = N.cell_id[o];
    const unsigned int at = N.cell_start[(size_t)b * (n_cells + 1) + c] + atomicAdd(&N.cell_fill[(size_t)b * n_cells + c], 1u);
    N.items[(size_t)b * N.S + at] = r;
    N.item_loc[(size_t)b * N.S + at] = N.loc[o];
    N.state[o] = UNDECIDED;
}

// the suppression's rule for point r given the states of the stronger points near it (states only ever go from
// UNDECIDED to their final value, so reading them while other threads decide is harmless).  The strongest point has no
// stronger neighbour and is kept: the seed of the reference's lists.
template <typename StateOf>
__device__ __forceinline__ unsigned char nms_decide(const nms_dev &N, unsigned int b, unsigned int r, StateOf state_of)
{
    const size_t base = (size_t)b * N.S;
    const int n_cells = N.gw * N.gh;
    const unsigned int *start = N.cell_start + (size_t)b * (n_cells + 1);
    const unsigned int *items = N.items + base;
    const double2 *item_loc = N.item_loc + base;
    const double2 me = N.loc[base + r];
    const int c = (int)N.cell_id[base + r], cx = c % N.gw, cy = c / N.gw;
    bool open_near = false;
    for (int yy = max(cy - 1, 0); yy <= min(cy + 1, N.gh - 1); yy++)
    {
        // three neighbouring cells of a row are one run of the item list
        const unsigned int i0 = start[yy * N.gw + max(cx - 1, 0)], i1 = start[yy * N.gw + min(cx + 1, N.gw - 1) + 1];
        for (unsigned int i = i0; i < i1; i++)
        {
            // (rank and location stream in with the run; the state - a dependent load - only for a stronger point in reach)
            const unsigned int q = items[i];
            if (q >= r)
                continue;
            if (!nms_within(N, me, item_loc[i]))
                continue;
            const unsigned char sq = state_of(q);
            if (sq == SPARSE)
                return DENSE;
            if (sq == UNDECIDED)
                open_near = true;
        }
    }
    return open_near ? UNDECIDED : SPARSE;
}

// one round over every point of every image (the bulk is decided after a few of these)
__global__ void nms_round_kernel(nms_dev N)
{
    const unsigned int b = blockIdx.z, r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= N.n[b])
        return;
    unsigned char *st = N.state + (size_t)b * N.S;
    if (st[r] != UNDECIDED)
        return;
    const unsigned char s = nms_decide(N, b, r, [&](unsigned int q) { return st[q]; });
    if (s != UNDECIDED)
        st[r] = s;
}

// One workgroup per image: the rounds that are left (in LDS), then what the caller makes of the kept / suppressed points:
// EXTRACT: the output slots of extract_features' list [sparse..., dense...] (the seed twice); SUBSET: the kept points'
// payloads in order (the 40 px subset).
constexpr int FIN_THREADS = 256;
template <bool SUBSET> __global__ __launch_bounds__(FIN_THREADS) void nms_finish_kernel(nms_dev N, feat_dev F)
{
    extern __shared__ unsigned char state[]; // [S]
    __shared__ unsigned int s_scan[FIN_THREADS];
    __shared__ int s_again;
    const unsigned int b = blockIdx.x, t = threadIdx.x;
    const unsigned int n = N.n[b];
    const size_t base = (size_t)b * N.S;
    if (t == 0)
        s_again = 0;
    for (unsigned int r = t; r < n; r += FIN_THREADS)
        state[r] = N.state[base + r];
    __syncthreads();
    for (;;)
    {
        bool any_left = false;
        for (unsigned int r = t; r < n; r += FIN_THREADS)
        {
            if (state[r] != UNDECIDED)
                continue;
            const unsigned char s = nms_decide(N, b, r, [&](unsigned int q) { return state[q]; });
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
    // exclusive scan of the kept flags, a contiguous run of ranks per thread
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
    const unsigned int total_kept = s_scan[FIN_THREADS - 1];
    unsigned int before = s_scan[t] - sum; // kept points among the ranks below r0
    for (unsigned int r = r0; r < r1; r++)
    {
        const bool kept = state[r] == SPARSE;
        if (SUBSET)
        {
            if (kept)
                F.subset[base + before] = (unsigned int)F.sub_recs[base + r]; // indices.push_back(idx), match_features.cpp:49
        }
        else
            // sparse features in strength order, then the dense ones behind the seed's second entry (the reference's loop
            // visits its seed again, extract_features.cpp:64-66)
            F.slot_of_rank[base + r] = kept ? before : total_kept + 1u + (r - before);
        before += kept ? 1u : 0u;
    }
    __syncthreads();
    if (t == 0)
        (SUBSET ? F.n_subset : F.n_sparse)[b] = n ? total_kept : 0u;
}

// the records at their slots (the seed twice: slot 0 and slot n_sparse), slot[s] for the host; and the sparse features'
// strength keys for the 40 px subset's sort: spatially_subsample_feature_indices std::sorts the sparse list's indices by
// strength from their order in the list (match_features.cpp:38-42)
__global__ void feat_records_kernel(feat_dev F)
{
    const unsigned int b = blockIdx.z, r = blockIdx.x * blockDim.x + threadIdx.x;
    const unsigned int n = F.n[b];
    if (r == 0)
    {
        F.sub_begin[b] = b * F.S;
        F.sub_end[b] = b * F.S + (n ? F.n_sparse[b] : 0u);
    }
    if (r >= n)
        return;
    const size_t o = (size_t)b * F.S + r;
    const unsigned long long sorted = F.recs[o];
    const unsigned int s = (unsigned int)sorted;
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
    const unsigned int ns = F.n_sparse[b];
    unsigned long long *out = reinterpret_cast<unsigned long long *>(F.records + ((size_t)b * (F.S + 1) + at) * 88);
#pragma unroll
    for (int w = 0; w < 11; w++)
        out[w] = rec[w];
    if (r == 0)
    {
        unsigned long long *again = reinterpret_cast<unsigned long long *>(F.records + ((size_t)b * (F.S + 1) + ns) * 88);
#pragma unroll
        for (int w = 0; w < 11; w++)
            again[w] = rec[w];
    }
    F.slot[(size_t)b * F.S + s] = at;
    if (at < ns) // a sparse feature: record (strength key, its index in the feature list) at its place in the list
        F.sub_recs[(size_t)b * F.S + at] = (sorted & 0xFFFFFFFF00000000ull) | at;
}

// The sparse list is in strength order already (it is the strength-sorted list minus the suppressed features), so the
// subset's std::sort can only change it where two neighbours have EQUAL strength (what an unstable sort does to equal keys
// is the one thing that depends on the order it is given).  An image without such a pair has its segment emptied here:
// the sort then leaves its records alone, which is their sorted order - the only one distinct keys have.
__global__ __launch_bounds__(256) void feat_subset_ties_kernel(feat_dev F)
{
    const unsigned int b = blockIdx.x;
    const unsigned int n = F.n[b] ? F.n_sparse[b] : 0u;
    __shared__ int tied;
    if (threadIdx.x == 0)
        tied = 0;
    __syncthreads();
    const unsigned long long *r = F.sub_recs + (size_t)b * F.S;
    int mine = 0;
    for (unsigned int i = threadIdx.x; i + 1 < n; i += 256)
        mine |= (r[i] >> 32) == (r[i + 1] >> 32);
    if (mine)
        tied = 1; // (benign race: every writer stores 1)
    __syncthreads();
    if (threadIdx.x == 0 && !tied)
        F.sub_end[b] = F.sub_begin[b];
}

// location of the r-th strongest sparse feature (after the subset's sort)
__global__ void feat_subset_loc_kernel(feat_dev F)
{
    const unsigned int b = blockIdx.z, r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= F.n_sparse[b] || F.n[b] == 0)
        return;
    const size_t o = (size_t)b * F.S + r;
    const unsigned int at = (unsigned int)F.sub_recs[o];
    const double *rec = reinterpret_cast<const double *>(F.records + ((size_t)b * (F.S + 1) + at) * 88);
    F.sub_loc[o] = make_double2(rec[0], rec[1]);
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
        if (out->num_subset)
            std::memset(out->num_subset, 0, (size_t)B * 4);
        if (out->subset_conflict)
            std::memset(out->subset_conflict, 0, B);
        return OCHIP_OK;
    }
    if (most > 60000)
        return ochip_fail(ctx, OCHIP_EINVAL, "feature lists: %u keypoints in one image exceed the suppression kernel's 60 000", most);
    const bool want_subset = out->subset && out->num_subset && out->subset_conflict && out->subset_spacing > 0;
    feat_dev F{};
    F.B = B;
    F.S = (most + 63) / 64 * 64;
    F.max_kp = max_kp;
    F.kp6 = d_kp6;
    F.desc = d_desc;
    F.counts = d_counts;
    F.scale = scale;
    const size_t N = (size_t)B * F.S;
    auto dev = [&](size_t bytes) -> void * {
        size_t got = 0;
        void *p = ochip_pool_get(ctx, std::max<size_t>(bytes, 16), &got);
        if (p)
            allocs->emplace_back(p, got);
        return p;
    };
    // a suppression problem over the working image: cells a little larger than the radius, in the locations' units
    // (original-image pixels = working pixels / scale)
    bool alloc_ok = true;
    auto make_nms = [&](const unsigned int *n, const double2 *loc, double radius_in_loc_units, double scale2, double radius2) {
        nms_dev M{};
        M.B = B;
        M.S = F.S;
        M.n = n;
        M.loc = loc;
        M.cell = radius_in_loc_units * 1.0625;
        M.scale2 = scale2;
        M.radius2 = radius2;
        M.gw = (int)((double)work_w / scale / M.cell) + 2;
        M.gh = (int)((double)work_h / scale / M.cell) + 2;
        const size_t n_cells = (size_t)M.gw * M.gh;
        M.cell_id = (unsigned int *)dev(N * 4);
        M.cell_start = (unsigned int *)dev((size_t)B * (n_cells + 1) * 4);
        M.cell_fill = (unsigned int *)dev((size_t)B * n_cells * 4);
        M.items = (unsigned int *)dev(N * 4);
        M.item_loc = (double2 *)dev(N * 16);
        M.state = (unsigned char *)dev(N);
        alloc_ok = alloc_ok && M.cell_id && M.cell_start && M.cell_fill && M.items && M.item_loc && M.state;
        return M;
    };
    F.n = (unsigned int *)dev((size_t)B * 4);
    F.recs = (unsigned long long *)dev(N * 8);
    F.seg_begin = (unsigned int *)dev((size_t)B * 4);
    F.seg_end = (unsigned int *)dev((size_t)B * 4);
    F.resp = (float *)dev(N * 4);
    F.loc = (double2 *)dev(N * 16);
    F.slot_of_rank = (unsigned int *)dev(N * 4);
    F.slot = (unsigned int *)dev(N * 4);
    F.n_sparse = (unsigned int *)dev((size_t)B * 4);
    F.records = (unsigned char *)dev((size_t)B * (F.S + 1) * 88);
    F.sub_recs = (unsigned long long *)dev(N * 8);
    F.sub_begin = (unsigned int *)dev((size_t)B * 4);
    F.sub_end = (unsigned int *)dev((size_t)B * 4);
    F.sub_loc = (double2 *)dev(N * 16);
    F.subset = (unsigned int *)dev(N * 4);
    F.n_subset = (unsigned int *)dev((size_t)B * 4);
    unsigned char *conflict = (unsigned char *)dev(B), *conflict2 = (unsigned char *)dev(B);
    if (!F.n || !F.recs || !F.seg_begin || !F.seg_end || !F.resp || !F.loc || !F.slot_of_rank || !F.slot || !F.n_sparse || !F.records ||
        !F.sub_recs || !F.sub_begin || !F.sub_end || !F.sub_loc || !F.subset || !F.n_subset || !conflict || !conflict2)
        return ochip_fail(ctx, OCHIP_ENOMEM, "feature lists: device allocation failed");
    // the reference's test is nn.distance * scale^2 > radius^2 on locations in original pixels (extract_features.cpp:72)
    nms_dev M8 = make_nms(F.n, F.loc, nms_radius / scale, scale * scale, nms_radius * nms_radius);
    // and nn.distance > spacing^2 for the subset (match_features.cpp:31), over the sparse features
    nms_dev M40{};
    if (want_subset)
        M40 = make_nms(F.n_sparse, F.sub_loc, out->subset_spacing, 1.0, out->subset_spacing * out->subset_spacing);
    if (!alloc_ok)
        return ochip_fail(ctx, OCHIP_ENOMEM, "feature lists: device allocation failed");
    const dim3 wide((F.S + 255) / 256, 1, B);
    constexpr int rounds = 6; // (4 measured in round 5: the two launches saved cost the per-image finish more than they took;
                              // per image and suppression the rounds take 2.0, 0.9, 0.5 and then 0.18 us each: a later round with
                              // 16 points per thread - one 16-byte load of their states - was slower, 11.4 us against 8.0)
    auto suppress = [&](const nms_dev &M, bool subset) -> int {
        OCHIP_HIP(ctx, hipMemsetAsync(M.cell_fill, 0, (size_t)B * M.gw * M.gh * 4, st));
        hipLaunchKernelGGL(nms_cells_kernel, wide, dim3(256), 0, st, M);
        hipLaunchKernelGGL(nms_scan_kernel, dim3(B), dim3(SCAN_THREADS), 0, st, M);
        hipLaunchKernelGGL(nms_fill_kernel, wide, dim3(256), 0, st, M);
        // (measured and dropped in round 5: the rounds as a loop inside one launch, neighbours' states read through the L2 -
        // 15.8 us per image against 8.1, and the per-image finish 6 us slower: without the launch boundary between rounds a
        // thread mostly re-reads states that have not changed yet)
        for (int k = 0; k < rounds; k++)
            hipLaunchKernelGGL(nms_round_kernel, wide, dim3(256), 0, st, M);
        if (subset)
            hipLaunchKernelGGL(nms_finish_kernel<true>, dim3(B), dim3(FIN_THREADS), (size_t)F.S, st, M, F);
        else
            hipLaunchKernelGGL(nms_finish_kernel<false>, dim3(B), dim3(FIN_THREADS), (size_t)F.S, st, M, F);
        return OCHIP_OK;
    };
    hipLaunchKernelGGL(feat_keys_kernel, wide, dim3(256), 0, st, F);
    // the strength order: std::sort by descending response from detection order, as the reference's (std_sort.hip);
    // conflict[b] = that image ran into introsort's depth limit and is the host's
    {
        const int src = std_sort_enqueue(ctx, allocs, F.recs, N, F.seg_begin, F.seg_end, B, most, conflict);
        if (src != OCHIP_OK)
            return src;
    }
    hipLaunchKernelGGL(feat_loc_kernel, wide, dim3(256), 0, st, F);
    {
        const int rc = suppress(M8, false);
        if (rc != OCHIP_OK)
            return rc;
    }
    hipLaunchKernelGGL(feat_records_kernel, wide, dim3(256), 0, st, F);
    if (want_subset)
    {
        // spatially_subsample_feature_indices over the sparse list: indices std::sorted by strength from list order, then
        // the greedy 40 px pass in that order
        hipLaunchKernelGGL(feat_subset_ties_kernel, dim3(B), dim3(256), 0, st, F);
        const int src = std_sort_enqueue(ctx, allocs, F.sub_recs, N, F.sub_begin, F.sub_end, B, most, conflict2);
        if (src != OCHIP_OK)
            return src;
        hipLaunchKernelGGL(feat_subset_loc_kernel, wide, dim3(256), 0, st, F);
        const int rc = suppress(M40, true);
        if (rc != OCHIP_OK)
            return rc;
    }
    OCHIP_HIP(ctx, hipGetLastError());
    OCHIP_HIP(ctx, hipMemcpy2DAsync(out->records, ((size_t)max_kp + 1) * 88, F.records, ((size_t)F.S + 1) * 88, ((size_t)most + 1) * 88, B,
                                    hipMemcpyDeviceToHost, st));
    OCHIP_HIP(ctx, hipMemcpy2DAsync(out->response, (size_t)max_kp * 4, F.resp, (size_t)F.S * 4, (size_t)most * 4, B, hipMemcpyDeviceToHost, st));
    OCHIP_HIP(ctx, hipMemcpy2DAsync(out->slot, (size_t)max_kp * 4, F.slot, (size_t)F.S * 4, (size_t)most * 4, B, hipMemcpyDeviceToHost, st));
    OCHIP_HIP(ctx, hipMemcpyAsync(out->num_sparse, F.n_sparse, (size_t)B * 4, hipMemcpyDeviceToHost, st));
    if (want_subset)
    {
        const size_t width = std::min<size_t>(most, OCHIP_SUBSET_CAP);
        OCHIP_HIP(ctx, hipMemcpy2DAsync(out->subset, (size_t)OCHIP_SUBSET_CAP * 4, F.subset, (size_t)F.S * 4, width * 4, B,
                                        hipMemcpyDeviceToHost, st));
        OCHIP_HIP(ctx, hipMemcpyAsync(out->num_subset, F.n_subset, (size_t)B * 4, hipMemcpyDeviceToHost, st));
        OCHIP_HIP(ctx, hipMemcpyAsync(out->subset_conflict, conflict2, B, hipMemcpyDeviceToHost, st));
    }
    OCHIP_HIP(ctx, hipMemcpyAsync(out->conflict, conflict, B, hipMemcpyDeviceToHost, st));
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
