// libochip.so (internal) — libstdc++'s std::sort on the device, permutation for permutation (gfx950).
//
// Two places of the path order things with an UNSTABLE std::sort whose outcome among equal keys is part of the result:
// the keypoints by response (src/extract/extract_features.cpp:55-56) and the matches by descriptor distance, then the
// PROSAC order by quality (src/match/match_features.cpp:100-101, src/model_inliers/ransac.cpp:83-90).  Equal keys are the
// rule there (a few equal floats among 20 k responses in almost every image; Hamming counts tie all the time), so matching
// the reference means reproducing what GCC's introsort does to the order it is given: bits/stl_algo.h - __introsort_loop
// with the median of three moved to the front, __unguarded_partition, depth limit 2 lg n, threshold 16,
// __final_insertion_sort.  That algorithm is a tree of partitions over disjoint ranges; its moves only depend on the
// comparator's answers, and a partition's moves can be stated from the range's content BEFORE it:
//   * the left scan stops at the elements with !comp(x, pivot), the right scan at those with !comp(pivot, x); the k-th
//     stop from the left is swapped with the k-th stop from the right for as long as the former lies left of the latter
//     (a swapped element is never looked at again by the scan that passed it), say K times;
//   * the cut is the (K + 1)-th left stop or the K-th right stop's position, whichever comes first (a scan that runs
//     into the other side's swapped elements stops at the first of them), or the first left stop when K = 0;
// so a partition is two prefix counts, a count of the pairs in order, K independent swaps.  The recursion's two ranges are
// independent: the long ranges of one depth are partitioned by one launch (a workgroup or a wavefront per range), level
// after level, until a range has at most 1 024 elements; such a range is finished by ONE wavefront in LDS (the same
// partition, the rest of its recursion from a small stack - no more launches, no more HBM round trips).  Ranges of at most
// 16 elements go to the final insertion sort, which never moves an element out of its range (everything left of a range
// is not after it in the order) - one lane per range.  The heap sort libstdc++ falls back to at the depth limit
// (std::__partial_sort: organ pipes and median-of-three killers get there; responses and Hamming counts do not) is restated
// too, move for move, and runs sequentially on one lane (round 5; until then such a segment was flagged for the host).
// fallback[] is now only set when a queue overflows.
// scripts/check_parallel_std_sort.py holds the formulation against std::sort on the CPU, tests/test_gpu_std_sort.py the
// kernels.
//
// Records are 64-bit: key in the high half, payload in the low half; comp(a, b) = key(a) > key(b) (descending; an
// ascending sort complements its keys).
#include "ctx.hpp"

#include <algorithm>
#include <vector>

using namespace ochip;

namespace
{

typedef unsigned long long u64;
constexpr unsigned int THRESHOLD = 16;  // _S_threshold
constexpr unsigned int LOCAL = 1024;    // ranges up to this length are finished by one wavefront in LDS
constexpr unsigned int BIG = 2048;      // ranges longer than this take a whole workgroup, (LOCAL, BIG] a wavefront, per level
constexpr int GROUP = 1024;             // threads of the workgroup that partitions a long range (256 until round 4: the first levels of a
                                        // chunk are one range per image, a pass over 20 k records is a chain of memory round trips per thread)

struct range_t
{
    unsigned int first, last, depth, seg;
};

struct sort_dev
{
    u64 *A;
    unsigned int *listL, *listR; // scratch: stop positions of the range [lo, hi) at [lo, ...)
    range_t *queue[3];           // ranges of the current / next level (rotating: the third one's counter is being reset)
    range_t *big[3];             // same, for the ranges a whole workgroup takes
    range_t *final_ranges;       // ranges of 2..16 elements
    range_t *local;              // ranges of 17..LOCAL elements
    range_t *heap;               // ranges longer than LOCAL that reached the depth limit (counts[10])
    unsigned int *counts;        // [0..2] queue sizes, [3..5] big sizes, [6] final, [7] local, [8] error, [10] heap
    unsigned int cap_queue, cap_final, cap_level; // of local, final_ranges, the level queues
    unsigned char *fallback;     // [n_segs]
    unsigned int *error;         // a queue overflowed (cannot happen with the capacities below; checked anyway)
};

__device__ __forceinline__ bool comp(u64 a, u64 b)
{
    return (unsigned int)(a >> 32) > (unsigned int)(b >> 32);
}

__device__ __forceinline__ void push_range(const sort_dev &S, int next, unsigned int first, unsigned int last, unsigned int depth,
                                           unsigned int seg, bool big_allowed)
{
    const unsigned int len = last - first;
    if (len < 2)
        return;
    if (len <= THRESHOLD)
    {
        const unsigned int at = atomicAdd(&S.counts[6], 1u);
        if (at < S.cap_final)
            S.final_ranges[at] = range_t{first, last, depth, seg};
        else
            *S.error = 1;
        return;
    }
    if (len <= LOCAL)
    {
        const unsigned int at = atomicAdd(&S.counts[7], 1u);
        if (at < S.cap_queue)
            S.local[at] = range_t{first, last, depth, seg};
        else
            *S.error = 1;
        return;
    }
    const bool to_big = big_allowed && len > BIG;
    const unsigned int at = atomicAdd(&S.counts[(to_big ? 3 : 0) + next], 1u);
    if (at < S.cap_level)
        (to_big ? S.big[next] : S.queue[next])[at] = range_t{first, last, depth, seg};
    else
        *S.error = 1;
}

__global__ void sort_init_kernel(sort_dev S, const unsigned int *__restrict__ seg_begin, const unsigned int *__restrict__ seg_end,
                                 unsigned int n_segs)
{
    const unsigned int s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= n_segs)
        return;
    S.fallback[s] = 0;
    const unsigned int first = seg_begin[s], last = seg_end[s];
    if (last <= first)
        return;
    unsigned int lg = 0;
    for (unsigned int n = last - first; n > 1; n >>= 1)
        lg++;
    push_range(S, 0, first, last, 2 * lg, s, true);
}

// inclusive prefix sum over the 64 lanes with DPP moves (row shifts inside the rows of 16, then the broadcasts of lanes 15
// and 31): six vector instructions where six __shfl_up went through the LDS crossbar one after the other
template <int CTRL, int ROW_MASK> __device__ __forceinline__ unsigned int dpp_add_from(unsigned int x)
{
    // lanes outside ROW_MASK, and lanes whose source lies outside their row, take 0
    return x + (unsigned int)__builtin_amdgcn_update_dpp(0, (int)x, CTRL, ROW_MASK, 0xf, true);
}
__device__ __forceinline__ unsigned int wave_inclusive_scan(unsigned int v)
{
    v = dpp_add_from<0x111, 0xf>(v); // row_shr:1
    v = dpp_add_from<0x112, 0xf>(v); // row_shr:2
    v = dpp_add_from<0x114, 0xf>(v); // row_shr:4
    v = dpp_add_from<0x118, 0xf>(v); // row_shr:8
    v = dpp_add_from<0x142, 0xa>(v); // row_bcast:15 into rows 1 and 3
    v = dpp_add_from<0x143, 0xc>(v); // row_bcast:31 into rows 2 and 3
    return v;
}

// ---- the depth limit: std::__partial_sort(first, last, last) = __make_heap + __sort_heap of libstdc++'s bits/stl_heap.h
// (__adjust_heap, __push_heap, __pop_heap restated move for move; comp as above).  Sequential, one lane: a range gets here
// on adversarial inputs only (organ pipes, median-of-three killers), never on responses or Hamming counts; what matters is
// that such a segment comes out as std::sort leaves it without leaving the device.  A: the range's first element
// (LDS or HBM).
template <typename Ptr> __device__ void heap_adjust(Ptr A, unsigned int hole, unsigned int len, u64 value)
{
    const unsigned int top = hole;
    unsigned int child = hole;
    while (child < (len - 1) / 2)
    {
        child = 2 * (child + 1);
        if (comp(A[child], A[child - 1]))
            child--;
        A[hole] = A[child];
        hole = child;
    }
    if ((len & 1) == 0 && child == (len - 2) / 2)
    {
        child = 2 * (child + 1);
        A[hole] = A[child - 1];
        hole = child - 1;
    }
    // __push_heap(first, hole, top, value)
    unsigned int parent = (hole - 1) / 2;
    while (hole > top && comp(A[parent], value))
    {
        A[hole] = A[parent];
        hole = parent;
        parent = (hole - 1) / 2;
    }
    A[hole] = value;
}
template <typename Ptr> __device__ void heap_sort(Ptr A, unsigned int len)
{
    if (len < 2)
        return;
    for (unsigned int parent = (len - 2) / 2;; parent--) // __make_heap
    {
        heap_adjust(A, parent, len, A[parent]);
        if (parent == 0)
            break;
    }
    for (unsigned int last = len; last > 1;) // __sort_heap: __pop_heap(first, last - 1, last - 1)
    {
        --last;
        const u64 value = A[last];
        A[last] = A[0];
        heap_adjust(A, 0u, last, value);
    }
}

// ---- group primitives: G threads that partition one range together (a wavefront or a workgroup)
template <int G> struct group;
template <> struct group<64>
{
    static __device__ __forceinline__ int tid()
    {
        return threadIdx.x & 63;
    }
    static __device__ __forceinline__ void sync()
    {
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
        __builtin_amdgcn_wave_barrier();
    }
    // exclusive prefix of v over the group's threads, *total = the sum
    static __device__ __forceinline__ unsigned int scan(unsigned int v, unsigned int *total, unsigned int *)
    {
        const unsigned int incl = wave_inclusive_scan(v);
        *total = (unsigned int)__builtin_amdgcn_readlane((int)incl, 63);
        return incl - v;
    }
};
template <> struct group<GROUP>
{
    static __device__ __forceinline__ int tid()
    {
        return threadIdx.x;
    }
    static __device__ __forceinline__ void sync()
    {
        __syncthreads();
    }
    // a wavefront scan per wavefront, the wavefronts' totals scanned by the first one: two barriers (a Hillis-Steele scan
    // over the workgroup in LDS took twenty)
    static __device__ __forceinline__ unsigned int scan(unsigned int v, unsigned int *total, unsigned int *lds /*[GROUP]*/)
    {
        static_assert(GROUP / 64 <= 16, "the wavefronts' totals fit one row of a wavefront scan");
        const int t = threadIdx.x, wv = t >> 6, lane = t & 63;
        const unsigned int incl = wave_inclusive_scan(v);
        __syncthreads(); // (lds may still be read as the previous scan's result)
        if (lane == 63)
            lds[wv] = incl;
        __syncthreads();
        if (wv == 0)
        {
            const unsigned int w = lane < GROUP / 64 ? lds[lane] : 0u;
            const unsigned int wi = wave_inclusive_scan(w);
            if (lane < GROUP / 64)
                lds[32 + lane] = wi - w; // exclusive prefix of the wavefronts' totals
            if (lane == GROUP / 64 - 1)
                lds[64] = wi;
        }
        __syncthreads();
        *total = lds[64];
        return lds[32 + wv] + incl - v;
    }
};

// one step of __introsort_loop on range r: median of three to the front, __unguarded_partition, the two ranges it leaves
// (Sink: where the two ranges a partition leaves go - the level queues in HBM, or a workgroup's own queues in LDS)
struct level_sink
{
    const sort_dev &S;
    int next;
    bool big_allowed;
    __device__ __forceinline__ void operator()(unsigned int first, unsigned int last, unsigned int depth, unsigned int seg) const
    {
        push_range(S, next, first, last, depth, seg, big_allowed);
    }
};
template <int G, typename Sink> __device__ void partition_range(const sort_dev &S, const range_t r, const Sink &sink, unsigned int *lds)
{
    u64 *A = S.A;
    const unsigned int t = (unsigned int)group<G>::tid();
    const unsigned int first = r.first, last = r.last;
    if (r.depth == 0)
    {
        if (t == 0) // libstdc++ heap-sorts this range: sort_heap_kernel does, behind the levels
        {
            const unsigned int at = atomicAdd(&S.counts[10], 1u);
            if (at < S.cap_level)
                S.heap[at] = r;
            else
                *S.error = 1;
        }
        return;
    }
    if (t == 0)
    {
        // __move_median_to_first(first, first + 1, mid, last - 1)
        const unsigned int a = first + 1, b = first + (last - first) / 2, c = last - 1;
        const u64 va = A[a], vb = A[b], vc = A[c];
        unsigned int pick;
        if (comp(va, vb))
            pick = comp(vb, vc) ? b : (comp(va, vc) ? c : a);
        else
            pick = comp(va, vc) ? a : (comp(vb, vc) ? c : b);
        const u64 vf = A[first], vp = A[pick];
        A[first] = vp;
        A[pick] = vf;
    }
    group<G>::sync();
    const u64 pivot = A[first];
    const unsigned int lo = first + 1, hi = last, m = hi - lo;
    // the stops of the two scans, in position order: every wavefront of the group takes a contiguous part of the range in
    // steps of 64 consecutive elements (coalesced), ballots give the order inside a step
    constexpr unsigned int W = G / 64;
    const unsigned int wave = t >> 6, lane = t & 63;
    const unsigned int part = ((m + W - 1) / W + 63) / 64 * 64;
    const unsigned int p0 = lo + min(wave * part, m), p1 = min(p0 + part, hi);
    const unsigned long long below = (1ull << lane) - 1ull;
    unsigned int cl = 0, cr = 0; // (wave-uniform)
    constexpr int UN = 4; // steps whose loads are in flight together (the passes are bound by their round trips)
    for (unsigned int base = p0; base < p1; base += 64 * UN)
    {
        u64 v[UN];
        bool valid[UN];
#pragma unroll
        for (int u = 0; u < UN; u++)
        {
            const unsigned int i = base + 64 * u + lane;
            valid[u] = i < p1;
            v[u] = valid[u] ? A[i] : 0ull;
        }
#pragma unroll
        for (int u = 0; u < UN; u++)
        {
            cl += (unsigned int)__popcll(__ballot(valid[u] && !comp(v[u], pivot))); // where "while (comp(*first, pivot)) ++first" stops
            cr += (unsigned int)__popcll(__ballot(valid[u] && !comp(pivot, v[u]))); // where "while (comp(pivot, *last)) --last" stops
        }
    }
    unsigned int nL = cl, nR = cr, ol = 0, orr = 0;
    if (W > 1)
    {
        __syncthreads();
        if (lane == 0)
        {
            lds[wave] = cl;
            lds[W + wave] = cr;
        }
        __syncthreads();
        nL = nR = 0;
        for (unsigned int w = 0; w < W; w++)
        {
            if (w < wave)
            {
                ol += lds[w];
                orr += lds[W + w];
            }
            nL += lds[w];
            nR += lds[W + w];
        }
        __syncthreads();
    }
    unsigned int *LL = S.listL + lo, *LR = S.listR + lo;
    for (unsigned int base = p0; base < p1; base += 64 * UN)
    {
        u64 v[UN];
        bool valid[UN];
#pragma unroll
        for (int u = 0; u < UN; u++)
        {
            const unsigned int i = base + 64 * u + lane;
            valid[u] = i < p1;
            v[u] = valid[u] ? A[i] : 0ull;
        }
#pragma unroll
        for (int u = 0; u < UN; u++)
        {
            const unsigned int i = base + 64 * u + lane;
            const bool sl = valid[u] && !comp(v[u], pivot), sr = valid[u] && !comp(pivot, v[u]);
            const unsigned long long ml = __ballot(sl), mr = __ballot(sr);
            if (sl)
                LL[ol + (unsigned int)__popcll(ml & below)] = i;
            if (sr)
                LR[orr + (unsigned int)__popcll(mr & below)] = i;
            ol += (unsigned int)__popcll(ml);
            orr += (unsigned int)__popcll(mr);
        }
    }
    group<G>::sync();
    // the k-th left stop is swapped with the k-th right stop (from the right) while it lies left of it
    const unsigned int kmax = min(nL, nR);
    unsigned int mine = 0;
    for (unsigned int k = t; k < kmax; k += G)
        mine += LL[k] < LR[nR - 1 - k] ? 1u : 0u;
    unsigned int K;
    (void)group<G>::scan(mine, &K, lds);
    for (unsigned int k = t; k < K; k += G)
    {
        const unsigned int i = LL[k], j = LR[nR - 1 - k];
        const u64 vi = A[i], vj = A[j];
        A[i] = vj;
        A[j] = vi;
    }
    if (t == 0)
    {
        unsigned int cut;
        if (K == 0)
            cut = LL[0];
        else
        {
            cut = LR[nR - K];
            if (K < nL)
                cut = min(cut, LL[K]);
        }
        // __introsort_loop(cut, last, depth_limit) and the loop's next round on [first, cut), both with the decremented limit
        sink(cut, last, r.depth - 1, r.seg);
        sink(first, cut, r.depth - 1, r.seg);
    }
}

// one level: the long ranges by the whole workgroup, one after the other, then the others a wavefront each.  The counter
// of the queues consumed by the level before (refilled by the level after) is reset on the way.
__global__ __launch_bounds__(GROUP) void sort_level_kernel(sort_dev S, int cur, int next, int stale)
{
    __shared__ unsigned int lds[GROUP];
    if (blockIdx.x == 0 && threadIdx.x == 0)
    {
        S.counts[stale] = 0;
        S.counts[3 + stale] = 0;
    }
    const unsigned int nb = min(S.counts[3 + cur], S.cap_level);
    const level_sink sink{S, next, true};
    for (unsigned int idx = blockIdx.x; idx < nb; idx += gridDim.x)
        partition_range<GROUP>(S, S.big[cur][idx], sink, lds);
    const unsigned int n = min(S.counts[cur], S.cap_level);
    for (unsigned int idx = blockIdx.x * (GROUP / 64) + (threadIdx.x >> 6); idx < n; idx += gridDim.x * (GROUP / 64))
        partition_range<64>(S, S.queue[cur][idx], sink, nullptr);
}

// ---- round 5: ONE launch for all levels above LOCAL, a workgroup per SEGMENT.  The ranges of one segment never meet those
// of another, so a workgroup can walk its own segment's partition tree with workgroup barriers only: the long ranges
// (> BIG) one after the other with all 16 wavefronts, the ranges in (LOCAL, BIG] a wavefront each, round after round,
// from two small queues in LDS; what falls to LOCAL or below goes to the same queues in HBM as before (sort_local_kernel,
// sort_final_kernel).  The launch-per-level form walked 2 lg n + 1 = 29 levels for every call - most of them empty - and
// the 100 segments of a chunk are only busy for the first eight: 58 launches per chunk for the two sorts of
// extract_features, now 2.  Segments longer than SEG_MAX (queues sized for it) and calls with few long segments keep
// the level launches.
constexpr unsigned int SEG_MAX = 1u << 17, SEG_BIGQ = 64, SEG_MIDQ = 192;
struct segment_queues
{
    range_t big[2][SEG_BIGQ], mid[2][SEG_MIDQ];
    unsigned int n_big[2], n_mid[2];
    unsigned int overflow;
};
struct segment_sink
{
    const sort_dev &S;
    segment_queues *Q;
    int next;
    __device__ __forceinline__ void operator()(unsigned int first, unsigned int last, unsigned int depth, unsigned int seg) const
    {
        const unsigned int len = last - first;
        if (len <= LOCAL)
        {
            push_range(S, 0, first, last, depth, seg, true); // (<= LOCAL: the final / local queues in HBM)
            return;
        }
        const bool to_big = len > BIG;
        const unsigned int at = atomicAdd(to_big ? &Q->n_big[next] : &Q->n_mid[next], 1u);
        if (at < (to_big ? SEG_BIGQ : SEG_MIDQ))
            (to_big ? Q->big[next] : Q->mid[next])[at] = range_t{first, last, depth, seg};
        else
            Q->overflow = 1;
    }
};
__global__ __launch_bounds__(GROUP) void sort_segments_kernel(sort_dev S, const unsigned int *__restrict__ seg_begin,
                                                             const unsigned int *__restrict__ seg_end)
{
    __shared__ unsigned int lds[GROUP];
    __shared__ segment_queues Q;
    const unsigned int seg = blockIdx.x;
    const unsigned int first = seg_begin[seg], last = seg_end[seg];
    if (threadIdx.x == 0)
    {
        S.fallback[seg] = 0;
        Q.n_big[0] = Q.n_big[1] = Q.n_mid[0] = Q.n_mid[1] = 0;
        Q.overflow = 0;
    }
    __syncthreads();
    if (last <= first)
        return;
    if (threadIdx.x == 0)
    {
        unsigned int lg = 0;
        for (unsigned int n = last - first; n > 1; n >>= 1)
            lg++;
        segment_sink{S, &Q, 0}(first, last, 2 * lg, seg);
    }
    __syncthreads();
    for (int cur = 0;; cur ^= 1)
    {
        const int next = cur ^ 1;
        const unsigned int nb = min(Q.n_big[cur], SEG_BIGQ), nm = min(Q.n_mid[cur], SEG_MIDQ);
        if (nb == 0 && nm == 0)
            break;
        const segment_sink sink{S, &Q, next};
        for (unsigned int idx = 0; idx < nb; idx++)
        {
            partition_range<GROUP>(S, Q.big[cur][idx], sink, lds);
            __syncthreads();
        }
        for (unsigned int idx = threadIdx.x >> 6; idx < nm; idx += GROUP / 64)
            partition_range<64>(S, Q.mid[cur][idx], sink, nullptr);
        __syncthreads();
        if (threadIdx.x == 0)
            Q.n_big[cur] = Q.n_mid[cur] = 0;
        __syncthreads();
    }
    if (threadIdx.x == 0 && Q.overflow)
        S.fallback[seg] = 1; // (a queue of this workgroup overflowed: the caller sorts the segment on the host)
}

// ranges still waiting in the level queues after the last level launched: the host's
__global__ void sort_flag_left_kernel(sort_dev S, int cur)
{
    const unsigned int nq = min(S.counts[cur], S.cap_level), nb = min(S.counts[3 + cur], S.cap_level);
    for (unsigned int i = blockIdx.x * blockDim.x + threadIdx.x; i < nq + nb; i += gridDim.x * blockDim.x)
        S.fallback[(i < nq ? S.queue[cur][i] : S.big[cur][i - nq]).seg] = 1;
}

// A range of at most LOCAL elements, finished by one wavefront in LDS: the rest of its __introsort_loop recursion from a
// stack (the order in which ranges are taken does not matter, they are disjoint), then the final insertion sort of its
// ranges of at most 16, a lane per range.
__global__ __launch_bounds__(256) void sort_local_kernel(sort_dev S)
{
    __shared__ u64 data_all[4][LOCAL];
    __shared__ unsigned short LL_all[4][LOCAL], LR_all[4][LOCAL];
    __shared__ unsigned int stack_all[4][64][2], fin_all[4][LOCAL / 2];
    const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
    u64 *D = data_all[wv];
    unsigned short *LL = LL_all[wv], *LR = LR_all[wv];
    unsigned int(*stack)[2] = stack_all[wv];
    unsigned int *fin = fin_all[wv];
    const unsigned int n_ranges = min(S.counts[7], S.cap_queue);
    for (unsigned int idx = blockIdx.x * 4 + wv; idx < n_ranges; idx += gridDim.x * 4)
    {
        const range_t r = S.local[idx];
        const unsigned int len = r.last - r.first;
        for (unsigned int i = lane; i < len; i += 64)
            D[i] = S.A[r.first + i];
        if (lane == 0)
        {
            stack[0][0] = len << 16; // first | last << 16
            stack[0][1] = r.depth;
        }
        group<64>::sync();
        int sp = 1;           // (uniform)
        unsigned int n_fin = 0;
        while (sp > 0)
        {
            sp--;
            const unsigned int fl = stack[sp][0], depth = stack[sp][1];
            const unsigned int first = fl & 0xFFFFu, last = fl >> 16;
            group<64>::sync(); // (the entry has been read by every lane before lane 0 reuses its slot)
            if (last - first <= THRESHOLD)
            {
                if (last - first >= 2)
                {
                    if (lane == 0)
                        fin[n_fin] = fl;
                    n_fin++;
                }
                continue;
            }
            if (depth == 0)
            {
                if (lane == 0) // libstdc++ heap-sorts this range (the depth limit: adversarial inputs)
                    heap_sort(D + first, last - first);
                group<64>::sync();
                continue;
            }
            if (lane == 0)
            {
                const unsigned int a = first + 1, b = first + (last - first) / 2, c = last - 1;
                const u64 va = D[a], vb = D[b], vc = D[c];
                unsigned int pick;
                if (comp(va, vb))
                    pick = comp(vb, vc) ? b : (comp(va, vc) ? c : a);
                else
                    pick = comp(va, vc) ? a : (comp(vb, vc) ? c : b);
                const u64 vf = D[first], vp = D[pick];
                D[first] = vp;
                D[pick] = vf;
            }
            group<64>::sync();
            const u64 pivot = D[first];
            const unsigned int lo = first + 1, hi = last;
            // the stops of the two scans in position order: 64 consecutive elements per step, ballots for the order
            const unsigned long long below = (1ull << lane) - 1ull;
            unsigned int nL = 0, nR = 0;
            for (unsigned int base = lo; base < hi; base += 64)
            {
                const unsigned int i = base + lane;
                const bool valid = i < hi;
                const u64 v = valid ? D[i] : 0ull;
                const bool sl = valid && !comp(v, pivot), sr = valid && !comp(pivot, v);
                const unsigned long long ml = __ballot(sl), mr = __ballot(sr);
                if (sl)
                    LL[nL + (unsigned int)__popcll(ml & below)] = (unsigned short)i;
                if (sr)
                    LR[nR + (unsigned int)__popcll(mr & below)] = (unsigned short)i;
                nL += (unsigned int)__popcll(ml);
                nR += (unsigned int)__popcll(mr);
            }
            group<64>::sync();
            const unsigned int kmax = min(nL, nR);
            unsigned int mine = 0;
            for (unsigned int k = lane; k < kmax; k += 64)
                mine += LL[k] < LR[nR - 1 - k] ? 1u : 0u;
            unsigned int K;
            (void)group<64>::scan(mine, &K, nullptr);
            for (unsigned int k = lane; k < K; k += 64)
            {
                const unsigned int i = LL[k], j = LR[nR - 1 - k];
                const u64 vi = D[i], vj = D[j];
                D[i] = vj;
                D[j] = vi;
            }
            unsigned int cut;
            if (K == 0)
                cut = LL[0];
            else
            {
                cut = LR[nR - K];
                if (K < nL)
                    cut = min(cut, (unsigned int)LL[K]);
            }
            if (lane == 0)
            {
                stack[sp][0] = cut | (last << 16);
                stack[sp][1] = depth - 1;
                stack[sp + 1][0] = first | (cut << 16);
                stack[sp + 1][1] = depth - 1;
            }
            sp += 2;
            group<64>::sync();
        }
        group<64>::sync();
        // __final_insertion_sort, range by range
        for (unsigned int e = lane; e < n_fin; e += 64)
        {
            const unsigned int first = fin[e] & 0xFFFFu, last = fin[e] >> 16;
            for (unsigned int i = first + 1; i < last; i++)
            {
                const u64 v = D[i];
                unsigned int j = i;
                while (j > first && comp(v, D[j - 1]))
                {
                    D[j] = D[j - 1];
                    j--;
                }
                D[j] = v;
            }
        }
        group<64>::sync();
        for (unsigned int i = lane; i < len; i += 64)
            S.A[r.first + i] = D[i];
        group<64>::sync();
    }
}

// the ranges longer than LOCAL that reached the depth limit, a wavefront each: through LDS when they fit (HEAP_LDS records),
// in place otherwise
constexpr unsigned int HEAP_LDS = 8192;
__global__ __launch_bounds__(64) void sort_heap_kernel(sort_dev S)
{
    __shared__ u64 D[HEAP_LDS];
    const unsigned int n = min(S.counts[10], S.cap_level);
    const int lane = threadIdx.x;
    for (unsigned int idx = blockIdx.x; idx < n; idx += gridDim.x)
    {
        const range_t r = S.heap[idx];
        const unsigned int len = r.last - r.first;
        if (len <= HEAP_LDS)
        {
            for (unsigned int i = lane; i < len; i += 64)
                D[i] = S.A[r.first + i];
            group<64>::sync();
            if (lane == 0)
                heap_sort(D, len);
            group<64>::sync();
            for (unsigned int i = lane; i < len; i += 64)
                S.A[r.first + i] = D[i];
            group<64>::sync();
        }
        else if (lane == 0)
            heap_sort(S.A + r.first, len);
    }
}

// a queue overflowed (the capacities rule it out): nothing of this call is to be trusted, every segment goes to the host
__global__ void sort_error_kernel(sort_dev S, unsigned int n_segs)
{
    if (*S.error == 0)
        return;
    for (unsigned int s = blockIdx.x * blockDim.x + threadIdx.x; s < n_segs; s += gridDim.x * blockDim.x)
        S.fallback[s] = 1;
}

// __final_insertion_sort restricted to a range of at most 16 elements: every element moves left past the elements it is
// strictly before
__global__ void sort_final_kernel(sort_dev S)
{
    const unsigned int n = min(S.counts[6], S.cap_final);
    const unsigned int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= n)
        return;
    const range_t r = S.final_ranges[idx];
    u64 *A = S.A;
    for (unsigned int i = r.first + 1; i < r.last; i++)
    {
        const u64 v = A[i];
        unsigned int j = i;
        while (j > r.first && comp(v, A[j - 1]))
        {
            A[j] = A[j - 1];
            j--;
        }
        A[j] = v;
    }
}

} // namespace

namespace ochip
{

// Sorts the segments [seg_begin[s], seg_end[s]) of recs (device arrays) in place as std::sort with comp(a, b) =
// key(a) > key(b) would; fallback[s] != 0: the segment needs libstdc++'s heap sort and was left partly sorted (its records
// are a permutation of the input).  Enqueued on the context's stream.
int std_sort_enqueue(ochip_ctx *ctx, std::vector<std::pair<void *, size_t>> *allocs, unsigned long long *recs, size_t total_len,
                     const unsigned int *seg_begin, const unsigned int *seg_end, uint32_t n_segs, uint32_t max_len,
                     unsigned char *fallback)
{
    hipStream_t st = ctx->stream;
    if (n_segs == 0 || total_len == 0)
        return OCHIP_OK;
    auto dev = [&](size_t bytes) -> void * {
        size_t got = 0;
        void *p = ochip_pool_get(ctx, std::max<size_t>(bytes, 16), &got);
        if (p)
            allocs->emplace_back(p, got);
        return p;
    };
    sort_dev S{};
    S.A = recs;
    S.cap_queue = (unsigned int)(total_len / (THRESHOLD + 1) + n_segs + 1);
    S.cap_final = (unsigned int)(total_len / 2 + n_segs + 1);
    S.listL = (unsigned int *)dev(total_len * 4);
    S.listR = (unsigned int *)dev(total_len * 4);
    // (queues of the levels in HBM only hold ranges longer than LOCAL)
    const size_t cap_level = total_len / LOCAL + n_segs + 1;
    S.cap_level = (unsigned int)cap_level;
    for (int i = 0; i < 3; i++)
    {
        S.queue[i] = (range_t *)dev(cap_level * sizeof(range_t));
        S.big[i] = (range_t *)dev(cap_level * sizeof(range_t));
    }
    S.heap = (range_t *)dev(cap_level * sizeof(range_t));
    S.final_ranges = (range_t *)dev((size_t)S.cap_final * sizeof(range_t));
    S.local = (range_t *)dev((size_t)S.cap_queue * sizeof(range_t));
    S.counts = (unsigned int *)dev(16 * 4);
    S.fallback = fallback;
    if (!S.heap || !S.listL || !S.listR || !S.queue[0] || !S.queue[1] || !S.queue[2] || !S.big[0] || !S.big[1] || !S.big[2] || !S.final_ranges || !S.local || !S.counts)
        return ochip_fail(ctx, OCHIP_ENOMEM, "std_sort: device allocation failed");
    S.error = S.counts + 8;
    OCHIP_HIP(ctx, hipMemsetAsync(S.counts, 0, 16 * 4, st));
    // a workgroup per segment walks the levels above LOCAL by itself when there are enough segments to fill the device that
    // way (the chunks of extract_features: 100 segments of ~20 k records); few long segments keep a launch per level, which
    // spreads one level's ranges over all workgroups
    // (from 8 segments on: a chunk of 25 images uploaded from host memory sorted with a launch per level - 29 levels, 2.1 ms -
    // where its 25 workgroups walk their trees in 0.4 ms, round 5's threshold of 32 kept the from-host path on the slow route)
    const bool per_level_hook = ochip_test_hook("sort_per_level");
    const bool by_segment = !per_level_hook && n_segs >= 8 && max_len <= SEG_MAX;
    if (by_segment)
    {
        if (max_len > LOCAL)
            hipLaunchKernelGGL(sort_segments_kernel, dim3(n_segs), dim3(GROUP), 0, st, S, seg_begin, seg_end);
        else
            hipLaunchKernelGGL(sort_init_kernel, dim3((n_segs + 255) / 256), dim3(256), 0, st, S, seg_begin, seg_end, n_segs);
    }
    else
        hipLaunchKernelGGL(sort_init_kernel, dim3((n_segs + 255) / 256), dim3(256), 0, st, S, seg_begin, seg_end, n_segs);
    // levels in HBM: while a range can still be longer than LOCAL - up to the depth limit (a range that reaches it is
    // flagged by the level that takes it; the deeper levels of well-split segments find their queues empty)
    int levels = 0;
    if (max_len > LOCAL && !by_segment)
    {
        unsigned int lg = 0;
        for (uint32_t n = max_len; n > 1; n >>= 1)
            lg++;
        levels = 2 * (int)lg + 1;
    }
    // (measured and dropped in round 5: all levels in ONE launch of 64 co-resident workgroups with an agent-scope barrier
    // between levels and an exit at the first empty level - 11.2 us per image against 8.9 for the launches: the first
    // levels are a hundred 20 k-record ranges that want a workgroup each, and a grid large enough for that cannot be
    // promised co-resident beside the other sequences' kernels)
    for (int level = 0; level < levels; level++)
    {
        const double by_depth = (double)n_segs * (double)(1u << std::min(level, 24));
        const unsigned int most = (unsigned int)std::min<double>((double)cap_level, by_depth);
        hipLaunchKernelGGL(sort_level_kernel, dim3(std::min(std::max(most, 1u), 2048u)), dim3(GROUP), 0, st, S, level % 3, (level + 1) % 3,
                           (level + 2) % 3);
    }
    if (levels)
        hipLaunchKernelGGL(sort_flag_left_kernel, dim3(64), dim3(256), 0, st, S, levels % 3);
    if (max_len > LOCAL)
        hipLaunchKernelGGL(sort_heap_kernel, dim3(64), dim3(64), 0, st, S);
    {
        const unsigned int most_local = (unsigned int)(total_len / (THRESHOLD + 1) + n_segs);
        hipLaunchKernelGGL(sort_local_kernel, dim3(std::min((most_local + 3) / 4, 2048u)), dim3(256), 0, st, S);
    }
    hipLaunchKernelGGL(sort_final_kernel, dim3((S.cap_final + 255) / 256), dim3(256), 0, st, S);
    hipLaunchKernelGGL(sort_error_kernel, dim3(16), dim3(256), 0, st, S, n_segs);
    OCHIP_HIP(ctx, hipGetLastError());
    return OCHIP_OK;
}

} // namespace ochip

// keys / payloads of n_segs segments (host arrays; segment s = [offsets[s], offsets[s + 1])) sorted as
// std::sort(comp = key(a) > key(b)) leaves them: for the tests.  fallback_out[s] != 0: not sorted (depth limit).
extern "C" int ochip_debug_std_sort(ochip_ctx *ctx, const uint32_t *keys, const uint32_t *payload, const uint32_t *offsets, uint32_t n_segs,
                                    uint32_t *keys_out, uint32_t *payload_out, uint8_t *fallback_out)
{
    if (!ctx || !offsets || !keys_out || !payload_out || !fallback_out)
        return OCHIP_EINVAL;
    OCHIP_HIP(ctx, hipSetDevice(ctx->device));
    hipStream_t st = ctx->stream;
    const size_t total = offsets[n_segs];
    std::vector<unsigned long long> recs(std::max<size_t>(total, 1));
    for (size_t i = 0; i < total; i++)
        recs[i] = ((unsigned long long)keys[i] << 32) | payload[i];
    std::vector<unsigned int> sb(n_segs), se(n_segs);
    uint32_t max_len = 0;
    for (uint32_t s = 0; s < n_segs; s++)
    {
        sb[s] = offsets[s];
        se[s] = offsets[s + 1];
        max_len = std::max(max_len, se[s] - sb[s]);
    }
    std::vector<std::pair<void *, size_t>> allocs;
    auto cleanup = [&]() {
        (void)ochip_stream_wait(ctx, st);
        for (auto &a : allocs)
            ochip_pool_put(ctx, a.first, a.second);
    };
    auto dev = [&](size_t bytes) -> void * {
        size_t got = 0;
        void *p = ochip_pool_get(ctx, std::max<size_t>(bytes, 16), &got);
        if (p)
            allocs.emplace_back(p, got);
        return p;
    };
    unsigned long long *d_recs = (unsigned long long *)dev(recs.size() * 8);
    unsigned int *d_sb = (unsigned int *)dev((size_t)n_segs * 4), *d_se = (unsigned int *)dev((size_t)n_segs * 4);
    unsigned char *d_fb = (unsigned char *)dev(std::max<uint32_t>(n_segs, 1));
    int rc = OCHIP_OK;
    if (!d_recs || !d_sb || !d_se || !d_fb)
        rc = ochip_fail(ctx, OCHIP_ENOMEM, "std_sort: device allocation failed");
    if (rc == OCHIP_OK && n_segs &&
        (hipMemcpyAsync(d_recs, recs.data(), recs.size() * 8, hipMemcpyHostToDevice, st) != hipSuccess ||
         hipMemcpyAsync(d_sb, sb.data(), (size_t)n_segs * 4, hipMemcpyHostToDevice, st) != hipSuccess ||
         hipMemcpyAsync(d_se, se.data(), (size_t)n_segs * 4, hipMemcpyHostToDevice, st) != hipSuccess))
        rc = ochip_fail(ctx, OCHIP_EHIP, "std_sort: upload failed");
    if (rc == OCHIP_OK)
        rc = std_sort_enqueue(ctx, &allocs, d_recs, total, d_sb, d_se, n_segs, max_len, d_fb);
    if (rc == OCHIP_OK && n_segs &&
        (hipMemcpyAsync(recs.data(), d_recs, recs.size() * 8, hipMemcpyDeviceToHost, st) != hipSuccess ||
         hipMemcpyAsync(fallback_out, d_fb, n_segs, hipMemcpyDeviceToHost, st) != hipSuccess))
        rc = ochip_fail(ctx, OCHIP_EHIP, "std_sort: download failed");
    cleanup();
    if (rc == OCHIP_OK)
        for (size_t i = 0; i < total; i++)
        {
            keys_out[i] = (uint32_t)(recs[i] >> 32);
            payload_out[i] = (uint32_t)recs[i];
        }
    return rc;
}
