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
#include "env.hpp"
#include "undistort.hpp"

#include <algorithm>
#include <new>
#include <vector>

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

// The search of one query by one wavefront: every lane ends with the same (best, second, index, count).
// Round 5.  The counters had a query at 7 us of which 73 % waiting for memory (SQ_WAIT_ANY / SQ_WAVE_CYCLES): a chain of
// dependent round trips - the query's slot record, its image's record, per cell row the two run bounds, then the row's
// locations, then the descriptors of the candidates inside the disc.  Now
//  * a query is PLANNED by one lane - image record, cell, the (up to three) runs' bounds: dense_plan - so that the eight
//    queries of a wavefront are planned side by side by eight lanes (dense_link_kernel): three round trips per wavefront
//    instead of per query;
//  * the runs are walked as one index space with the locations of DS_ROUNDS x 64 candidates in flight;
//  * the candidates inside the disc - a handful of the ~160 scanned - are handed a lane each through a wavefront-private LDS
//    list in ascending index, the order of the reference's sequential rule (:262-276); what that rule leaves (the smallest
//    distance at its first candidate, the smallest of the others) is two wavefront minima (work_off).
constexpr int DS_ROUNDS = 4; // rounds of 64 candidates whose locations are requested together (2: 76 registers instead of 88, six waves per SIMD instead of five, 2.77 ms per launch against 2.81 - the kernel does not wait for occupancy)
__device__ __forceinline__ double bcast64(double v, int src)
{
    const unsigned long long b = (unsigned long long)__double_as_longlong(v);
    const uint32_t lo = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)b, src), hi = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(b >> 32), src);
    return __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
}
struct dense_plan_t
{
    uint64_t feat_base;
    uint32_t rb[3], rn[3]; // the runs of cell rows cy - 1 .. cy + 1: first candidate, length (a row outside the grid: empty)
};
__device__ __forceinline__ dense_plan_t dense_plan(const dense_image_meta &m, const uint32_t *__restrict__ cell_start, double qx, double qy,
                                                   double cell_size)
{
    dense_plan_t P;
    P.feat_base = m.feat_base;
    const int cx = (int)floor((qx - m.ox) / cell_size), cy = (int)floor((qy - m.oy) / cell_size);
    const int c0 = max(cx - 1, 0), c1 = min(cx + 1, m.ncx - 1);
#pragma unroll
    for (int r = 0; r < 3; r++)
    {
        const int ry = cy - 1 + r;
        const bool row_ok = c0 <= c1 && ry >= 0 && ry < m.ncy;
        const uint32_t *cs = cell_start + m.cell_base + (size_t)(row_ok ? ry : 0) * m.ncx;
        const uint32_t begin = row_ok ? cs[c0] : 0u, end = row_ok ? cs[c1 + 1] : 0u;
        P.rb[r] = begin;
        P.rn[r] = end - begin;
    }
    return P;
}
// the plan of the lane `src`, in every lane
__device__ __forceinline__ dense_plan_t dense_plan_from(const dense_plan_t &P, int src)
{
    auto get = [&](uint32_t v) { return (uint32_t)__builtin_amdgcn_readlane((int)v, src); };
    dense_plan_t Q;
    Q.feat_base = ((uint64_t)get((uint32_t)(P.feat_base >> 32)) << 32) | get((uint32_t)P.feat_base);
#pragma unroll
    for (int r = 0; r < 3; r++)
    {
        Q.rb[r] = get(P.rb[r]);
        Q.rn[r] = get(P.rn[r]);
    }
    return Q;
}
// the smallest value of a wavefront, in every lane (as a scalar): four DPP stages inside the rows of 16 lanes, the rows' results read out
template <int CTRL> __device__ __forceinline__ uint32_t dense_dpp(uint32_t x)
{
    return (uint32_t)__builtin_amdgcn_update_dpp((int)x, (int)x, CTRL, 0xf, 0xf, false);
}
__device__ __forceinline__ uint32_t wave_min_u32(uint32_t v)
{
    v = min(v, dense_dpp<0xB1>(v));  // quad_perm [1, 0, 3, 2]
    v = min(v, dense_dpp<0x4E>(v));  // quad_perm [2, 3, 0, 1]
    v = min(v, dense_dpp<0x124>(v)); // row_ror:4
    v = min(v, dense_dpp<0x128>(v)); // row_ror:8
    const uint32_t r0 = (uint32_t)__builtin_amdgcn_readlane((int)v, 0), r1 = (uint32_t)__builtin_amdgcn_readlane((int)v, 16),
                   r2 = (uint32_t)__builtin_amdgcn_readlane((int)v, 32), r3 = (uint32_t)__builtin_amdgcn_readlane((int)v, 48);
    return min(min(r0, r1), min(r2, r3));
}
// P, qx, qy, src_feature: wave-uniform
__device__ __forceinline__ lane_state dense_scan(const dense_plan_t &P, const uint64_t *__restrict__ desc, const double2 *__restrict__ loc,
                                                 uint32_t src_feature, double qx, double qy, double radius_sq, int lane,
                                                 uint32_t *inside_list /*LDS, 64 words per wavefront*/)
{
    uint64_t qd[8];
    {
        const uint64_t *p = desc + 8 * (size_t)src_feature;
#pragma unroll
        for (int w = 0; w < 8; w++)
            qd[w] = p[w];
    }
    lane_state s{NONE_COUNT, NONE_COUNT, 0xFFFFFFFFu, 0u};
    const uint32_t n0 = P.rn[0], n01 = n0 + P.rn[1], total = n01 + P.rn[2];
    auto candidate = [&](uint32_t f) { return f < n0 ? P.rb[0] + f : (f < n01 ? P.rb[1] + (f - n0) : P.rb[2] + (f - n01)); };
    auto wave_sync = []() {
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
        __builtin_amdgcn_wave_barrier();
    };
    auto work_off = [&](uint32_t n) {
        wave_sync();
        // lane j takes the j-th listed candidate (ascending index: the order the reference meets them in, :262-276).  Its rule
        // - a distance below the second best replaces it, one below the best moves the best down - leaves: best = the smallest
        // distance, at its FIRST candidate; second = the smallest among all the others.  Two wavefront minima over
        // (distance << 6 | list position) instead of a walk through the list (a dependent LDS read per candidate:
        // 1.3 us of a 5 us search at ~32 candidates)
        uint32_t key = 0xFFFFFFFFu, k = 0;
        if ((uint32_t)lane < n)
        {
            k = inside_list[lane];
            const uint64_t *cd = desc + 8 * (P.feat_base + k);
            uint32_t d = 0;
#pragma unroll
            for (int w = 0; w < 8; w++)
                d += (uint32_t)__popcll(cd[w] ^ qd[w]);
            key = d << 6 | (uint32_t)lane;
        }
        const uint32_t first = wave_min_u32(key);
        const int at = (int)(first & 63u);
        const uint32_t b2 = first >> 6, k2 = (uint32_t)__builtin_amdgcn_readlane((int)k, at);
        const uint32_t others = wave_min_u32(lane == at ? 0xFFFFFFFFu : key);
        const uint32_t s2 = n >= 2 ? others >> 6 : NONE_COUNT;
        // behind the candidates worked off before (ties stay with the earlier ones)
        if (b2 < s.best)
        {
            s.second = min(s.best, s2);
            s.best = b2;
            s.idx = k2;
        }
        else
            s.second = min(s.second, b2);
        s.count += n;
        wave_sync(); // (the list is rewritten by the next candidates)
    };
    uint32_t listed = 0; // candidates inside the disc waiting in the list (wave-uniform)
    for (uint32_t f0 = 0; f0 < total; f0 += 64 * DS_ROUNDS)
    {
        double2 p[DS_ROUNDS];
        uint32_t k[DS_ROUNDS];
#pragma unroll
        for (int u = 0; u < DS_ROUNDS; u++)
        {
            const uint32_t f = f0 + 64 * u + lane;
            k[u] = candidate(f < total ? f : total - 1);
            p[u] = loc[P.feat_base + k[u]];
        }
#pragma unroll
        for (int u = 0; u < DS_ROUNDS; u++)
        {
            const uint32_t f = f0 + 64 * u + lane;
            const double dx = p[u].x - qx, dy = p[u].y - qy;
            const bool in = f < total && dx * dx + dy * dy < radius_sq;
            const unsigned long long mask = __ballot(in);
            if (mask == 0)
                continue;
            const uint32_t n_in = (uint32_t)__popcll(mask);
            if (listed + n_in > 64) // the list is full: its candidates take their lanes now
            {
                work_off(listed);
                listed = 0;
            }
            if (in)
                inside_list[listed + (uint32_t)__popcll(mask & ((1ull << lane) - 1ull))] = k[u];
            listed += n_in;
        }
    }
    if (listed)
        work_off(listed);
    return s;
}
__device__ __forceinline__ lane_state dense_search(const dense_image_meta &m, const uint64_t *__restrict__ desc, const double2 *__restrict__ loc,
                                                   const uint32_t *__restrict__ cell_start, uint32_t src_feature, double qx, double qy,
                                                   double radius_sq, double cell_size, int lane, uint32_t *inside_list)
{
    return dense_scan(dense_plan(m, cell_start, qx, qy, cell_size), desc, loc, src_feature, qx, qy, radius_sq, lane, inside_list);
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
    __shared__ __attribute__((aligned(8))) uint32_t inside_lists[4][64];
    const lane_state s = dense_search(meta[q.cand_image], desc, loc, cell_start, q.src_feature, q.px, q.py, radius_sq, cell_size, lane,
                                      inside_lists[threadIdx.x >> 6]);
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

// ---- densifyMesh after the mesh intersections, on the device (ochip_dense_link) ------------------------------------
// Per dense feature with a hit point: the 11 nearest cameras (camera_tree.search(point, max, k + 1), dense_stereo.cpp:
// 212-216: squared distance, then index), the point projected into each but the source image (image_from_3d, :225-236),
// one query slot per candidate inside its image; per slot: the disc search above, the accept rule of :278-283 in fp64 as
// the reference writes it, and the union of the two measurements (:285-297) in a lock-free union-find whose partition does
// not depend on the order of the unions (the larger root goes under the smaller).
constexpr int DENSE_K = 11; // MAX_CANDIDATE_IMAGES + 1 (:53, :213)
constexpr int DENSE_COUNTERS = 1024; // words per statistics counter (queries, matches): summed by the host
struct dense_cam
{
    double pos[3], q_inv[4], model[10]; // f ppx ppy k1 k2 k3 p1 p2 cols rows
};

struct dv3
{
    double x, y, z;
};
__device__ __forceinline__ dv3 dcross(const dv3 &a, const dv3 &b)
{
    return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x};
}
// Eigen QuaternionBase::_transformVector (host: relax_util.hpp rotate)
__device__ __forceinline__ dv3 drotate(const double *q, const dv3 &v)
{
    const dv3 qv{q[0], q[1], q[2]};
    dv3 uv = dcross(qv, v);
    uv = dv3{uv.x + uv.x, uv.y + uv.y, uv.z + uv.z};
    const dv3 c = dcross(qv, uv);
    return {(v.x + uv.x * q[3]) + c.x, (v.y + uv.y * q[3]) + c.y, (v.z + uv.z * q[3]) + c.z};
}
// image_from_3d(point, model, position, orientation): host dense_stereo.cpp project() + invert_distortion.cpp image_from_3d
__device__ __forceinline__ void dense_project(const dv3 &point, const dense_cam &c, double px[2])
{
    const dv3 rel{point.x - c.pos[0], point.y - c.pos[1], point.z - c.pos[2]};
    const dv3 r = drotate(c.q_inv, rel);
    const double z = r.z < 1e-3 ? 1e-3 : r.z;
    const double p[2] = {r.x / z, r.y / z};
    double r2[3];
    r2[0] = p[0] * p[0] + p[1] * p[1];
    r2[1] = r2[0] * r2[0];
    r2[2] = r2[1] * r2[0];
    const double *k = c.model + 3, *t = c.model + 6;
    const double radial = k[0] * r2[0] + k[1] * r2[1] + k[2] * r2[2];
    const double prod = p[0] * p[1];
    for (int i = 0; i < 2; i++)
    {
        const double d = (1.0 + radial) * p[i] + 2.0 * prod * t[i] + t[1 - i] * (r2[0] + 2.0 * p[i] * p[i]);
        px[i] = d * c.model[0] + c.model[1 + i];
    }
}

// one thread per dense feature of the batch's images (blockIdx.y = image of the batch).
// The 11 nearest cameras are the 11 smallest (distance, index) pairs whatever order the cameras are met in, and the order
// decides the cost: a camera that does not beat the lane's 11th pair costs a subtraction, three products and a comparison, one
// that does is an 11-stage insertion - which a wavefront pays as soon as one of its 64 lanes inserts.  Met in index order the
// lanes of a wavefront kept inserting all along the list (~50 each at different places: nearly every trip); the cameras are
// met outwards from the source image's own index instead (src, src + 1, src - 1, ...: a flight's neighbours in time are mostly
// its neighbours in space, and nothing depends on it when they are not), so the 11th pair is tight after the first few dozen.
// The positions come from LDS (all of them, staged once per workgroup; PRED_LDS_CAMS at most - larger surveys read the
// camera records through the scalar cache in the same order), four per trip so that their reads are in flight together.
constexpr uint32_t PRED_LDS_CAMS = 2048;
template <bool STAGED>
__global__ __launch_bounds__(256) void dense_predict_kernel(const dense_image_meta *__restrict__ meta, const dense_cam *__restrict__ cams,
                                                            uint32_t n_images, uint32_t first_image, const double *__restrict__ hits,
                                                            uint64_t batch_feat_base, uint32_t *__restrict__ cand_img,
                                                            double2 *__restrict__ cand_px, unsigned long long *__restrict__ n_queries)
{
    extern __shared__ double cam_pos_lds[]; // [n_images][3] when STAGED
    if (STAGED)
    {
        for (uint32_t i = threadIdx.x; i < n_images; i += 256)
        {
            cam_pos_lds[3 * i] = cams[i].pos[0];
            cam_pos_lds[3 * i + 1] = cams[i].pos[1];
            cam_pos_lds[3 * i + 2] = cams[i].pos[2];
        }
        __syncthreads();
    }
    const uint32_t src = first_image + blockIdx.y;
    const uint64_t f0 = meta[src].feat_base, n = meta[src + 1].feat_base - f0; // (meta has n_images + 1 entries)
    const uint64_t k = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    unsigned int emitted = 0;
    if (k < n)
    {
        const uint64_t f = f0 + k, slot0 = (f - batch_feat_base) * DENSE_K;
        const dv3 hit{hits[3 * f], hits[3 * f + 1], hits[3 * f + 2]};
        int used = 0;
        if (!isnan(hit.x))
        {
            double bd[DENSE_K];
            uint32_t bc[DENSE_K];
#pragma unroll
            for (int s = 0; s < DENSE_K; s++)
            {
                bd[s] = INFINITY;
                bc[s] = 0xFFFFFFFFu;
            }
            // the t-th camera outwards from src (wave-uniform): src, src + 1, src - 1, ... while both sides last, then the rest
            const uint32_t below = src, above = n_images - 1 - src, both = 2 * min(below, above);
            auto camera_at = [&](uint32_t t) { return t <= both ? ((t & 1u) ? src + (t + 1) / 2 : src - t / 2) : (below > above ? n_images - 1 - t : t); };
            constexpr int PER_TRIP = 4;
            for (uint32_t t0 = 0; t0 < n_images; t0 += PER_TRIP)
            {
                uint32_t cs[PER_TRIP];
                double px[PER_TRIP], py[PER_TRIP], pz[PER_TRIP];
#pragma unroll
                for (int u = 0; u < PER_TRIP; u++)
                {
                    cs[u] = camera_at(min(t0 + u, n_images - 1)); // (the last trip's spare places: read, not used)
                    if (STAGED)
                        px[u] = cam_pos_lds[3 * cs[u]], py[u] = cam_pos_lds[3 * cs[u] + 1], pz[u] = cam_pos_lds[3 * cs[u] + 2];
                    else
                        px[u] = cams[cs[u]].pos[0], py[u] = cams[cs[u]].pos[1], pz[u] = cams[cs[u]].pos[2];
                }
#pragma unroll
                for (int u = 0; u < PER_TRIP; u++)
                {
                    if (t0 + u >= n_images) // (wave-uniform)
                        break;
                    const double dx = px[u] - hit.x, dy = py[u] - hit.y, dz = pz[u] - hit.z;
                    double d = dx * dx + dy * dy + dz * dz;
                    uint32_t id = cs[u];
                    if (!(d < bd[DENSE_K - 1] || (d == bd[DENSE_K - 1] && id < bc[DENSE_K - 1])))
                        continue;
#pragma unroll
                    for (int s = 0; s < DENSE_K; s++) // sorted insertion: the displaced entries move up, the last one falls off
                    {
                        const bool lt = d < bd[s] || (d == bd[s] && id < bc[s]);
                        const double td = lt ? bd[s] : d;
                        const uint32_t tc = lt ? bc[s] : id;
                        bd[s] = lt ? d : bd[s];
                        bc[s] = lt ? id : bc[s];
                        d = td;
                        id = tc;
                    }
                }
            }
#pragma unroll
            for (int s = 0; s < DENSE_K; s++)
            {
                const uint32_t c = bc[s];
                if (c == 0xFFFFFFFFu || c == src)
                    continue;
                double px[2];
                const dense_cam cam = cams[c];
                dense_project(hit, cam, px);
                if (px[0] < 0 || px[0] >= cam.model[8] || px[1] < 0 || px[1] >= cam.model[9])
                    continue;
                cand_img[slot0 + used] = c;
                cand_px[slot0 + used] = make_double2(px[0], px[1]);
                used++;
            }
        }
        emitted = (unsigned int)used;
        for (int s = used; s < DENSE_K; s++)
            cand_img[slot0 + s] = 0xFFFFFFFFu;
    }
    // the batch's number of queries: one atomic per workgroup
    __shared__ unsigned int wsum[4];
    for (int off = 32; off >= 1; off >>= 1)
        emitted += (unsigned int)__shfl_xor((int)emitted, off);
    if ((threadIdx.x & 63) == 0)
        wsum[threadIdx.x >> 6] = emitted;
    __syncthreads();
    if (threadIdx.x == 0 && wsum[0] + wsum[1] + wsum[2] + wsum[3])
        atomicAdd(n_queries + ((blockIdx.x + 31 * blockIdx.y) & (DENSE_COUNTERS - 1)), (unsigned long long)(wsum[0] + wsum[1] + wsum[2] + wsum[3]));
}

// Lock-free union-find over all measurements of the survey.  Invariant: parent[x] <= x, and a parent pointer only ever moves to
// a smaller member of the same component - so ANY value parent[x] ever held is an ancestor of x.  The walk towards the root
// can therefore read through the caches (an XCD's L2 is not coherent with the others': a stale parent is an ancestor, the walk
// is at worst longer); only the link - root under a smaller node - is a device-scope compare-and-swap, and when it fails it
// says where the node points now.  Round 5: with every step of the walk a device-scope atomic load and every halving a
// device-scope CAS, the kernel ran at the rate the memory side executes those (~0.5 G per 130 ms), whatever the search cost.
__device__ __forceinline__ uint32_t uf_cached_parent(const uint32_t *parent, uint32_t x)
{
    return __hip_atomic_load(&parent[x], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT); // (a plain load the compiler may not hoist)
}
__device__ __forceinline__ uint32_t uf_find(uint32_t *parent, uint32_t x)
{
    while (true)
    {
        const uint32_t p = uf_cached_parent(parent, x);
        if (p == x)
            return x;
        const uint32_t gp = uf_cached_parent(parent, p);
        if (gp != p)
            __hip_atomic_store(&parent[x], gp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT); // path halving: an ancestor, whoever wins
        x = p;
    }
}
__device__ __forceinline__ void uf_unite(uint32_t *parent, uint32_t a, uint32_t b)
{
    while (true)
    {
        a = uf_find(parent, a);
        b = uf_find(parent, b);
        if (a == b)
            return;
        if (a < b)
        {
            const uint32_t t = a;
            a = b;
            b = t;
        }
        // the larger (apparent) root goes under the smaller node: a component's root is its smallest member
        const uint32_t was = atomicCAS(&parent[a], a, b);
        if (was == a)
            return;
        a = was; // not a root any more (this XCD's copy was stale): go on from where it points now
    }
}

// one wavefront per SLOTS_PER_WAVE query slots: search, accept, unite
// (measured and dropped, round 5: a slot's first trip of locations requested while the slot before it is searched - one of a
// search's two dependent round trips off its critical path, but 132 registers with four rounds in flight (three waves per SIMD):
// 4.25 ms per launch, 3.58 with two rounds (108 registers), against 2.81 for this form at 88)
constexpr int SLOTS_PER_WAVE = 8;
__global__ __launch_bounds__(256) void dense_link_kernel(const dense_image_meta *__restrict__ meta, const uint64_t *__restrict__ desc,
                                                         const double2 *__restrict__ loc, const uint32_t *__restrict__ cell_start,
                                                         const uint32_t *__restrict__ cand_img, const double2 *__restrict__ cand_px,
                                                         uint64_t n_slots, uint64_t batch_feat_base, const uint32_t *__restrict__ id_of_pos,
                                                         double radius_sq, double cell_size, double inv_bits, double ratio, double max_abs,
                                                         uint32_t *__restrict__ parent, uint8_t *__restrict__ matched,
                                                         uint32_t *__restrict__ slot_dst, unsigned long long *__restrict__ n_matches)
{
    __shared__ __attribute__((aligned(8))) uint32_t inside_lists[4][64];
    const uint64_t w = (uint64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    // an accepted match is a union of two measurements: a chain of dependent loads along the parents of both (and a CAS).  Done
    // by lane 0 slot after slot it was most of the kernel's time; the wave's SLOTS_PER_WAVE matches are kept, one per lane, and
    // united side by side after the searches (the partition does not depend on the order of the unions)
    uint32_t my_src_pos = 0xFFFFFFFFu, my_dst_pos = 0xFFFFFFFFu;
    // lane j plans slot j: candidate image, its record, the predicted pixel's cell, the runs' bounds
    const uint64_t plan_slot = w * SLOTS_PER_WAVE + (uint64_t)lane;
    const bool planner = lane < SLOTS_PER_WAVE && plan_slot < n_slots;
    const uint32_t plan_cand = planner ? cand_img[plan_slot] : 0xFFFFFFFFu;
    double2 plan_q = make_double2(0.0, 0.0);
    dense_plan_t plan{};
    if (plan_cand != 0xFFFFFFFFu)
    {
        plan_q = cand_px[plan_slot];
        plan = dense_plan(meta[plan_cand], cell_start, plan_q.x, plan_q.y, cell_size);
    }
    for (int j = 0; j < SLOTS_PER_WAVE; j++)
    {
        const uint64_t slot = w * SLOTS_PER_WAVE + j;
        if (slot >= n_slots)
            break;
        const uint32_t cand = (uint32_t)__builtin_amdgcn_readlane((int)plan_cand, j);
        uint32_t src_pos = 0xFFFFFFFFu, dst_pos = 0xFFFFFFFFu;
        if (cand != 0xFFFFFFFFu)
        {
            const uint32_t src_feature = (uint32_t)(batch_feat_base + slot / DENSE_K);
            const dense_plan_t P = dense_plan_from(plan, j);
            const double qx = bcast64(plan_q.x, j), qy = bcast64(plan_q.y, j);
            const lane_state s = dense_scan(P, desc, loc, src_feature, qx, qy, radius_sq, lane, inside_lists[threadIdx.x >> 6]);
            // (every lane holds the same state: the decision is wave-uniform)
            if (s.count != 0)
            {
                const double best_dist = s.best * inv_bits;
                const double second_best_dist = s.second == NONE_COUNT ? INFINITY : s.second * inv_bits;
                const bool good = s.count >= 2 ? best_dist < ratio * second_best_dist : best_dist < max_abs;
                if (good)
                {
                    src_pos = src_feature;
                    dst_pos = (uint32_t)(P.feat_base + s.idx);
                }
            }
        }
        if (lane == j)
        {
            my_src_pos = src_pos;
            my_dst_pos = dst_pos;
        }
    }
    uint32_t dst = 0xFFFFFFFFu;
    if (my_dst_pos != 0xFFFFFFFFu)
    {
        const uint32_t src_id = id_of_pos[my_src_pos];
        dst = id_of_pos[my_dst_pos];
        matched[src_id] = 1;
        matched[dst] = 1;
        uf_unite(parent, src_id, dst);
    }
    const uint64_t my_slot = w * SLOTS_PER_WAVE + (uint64_t)lane;
    if (slot_dst && lane < SLOTS_PER_WAVE && my_slot < n_slots)
        slot_dst[my_slot] = dst;
    const unsigned int accepted = (unsigned int)__popcll(__ballot(dst != 0xFFFFFFFFu));
    // (one counter for every wavefront of the launch was 700 k atomic adds to ONE address per launch, which the memory side
    // takes one after the other: 8.6 ms per launch whatever the searches cost - round 5's counters found the wait, five
    // rewrites of the search did not move it.  DENSE_COUNTERS words, a workgroup adds to its own.)
    if (lane == 0 && accepted)
        atomicAdd(n_matches + (blockIdx.x & (DENSE_COUNTERS - 1)), (unsigned long long)accepted);
}

__global__ void dense_uf_init_kernel(uint32_t *parent, uint8_t *matched, uint64_t n)
{
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n)
    {
        parent[i] = (uint32_t)i;
        matched[i] = 0;
    }
}
__global__ void dense_uf_roots_kernel(uint32_t *parent, const uint8_t *matched, uint64_t n, uint32_t *root)
{
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n)
        root[i] = matched[i] ? uf_find(parent, (uint32_t)i) : 0xFFFFFFFFu;
}

// ---- the tracks' points (dense_stereo.cpp:299-340): one thread per track ---------------------------------------------
// triangulateTrack (:112-172) as the host wrote it (host/dense_stereo.cpp): the first two members' rays meet in a point;
// members whose reprojection is within 8 px are inliers; fewer than two: no point; fewer than all: the point again from the
// first two inliers.  Same operations in the same order as the host's (undistort.hpp is shared, the rotation, the intersection
// and the projection are the ones the link uses), -ffp-contract=off on both sides.
__global__ void dense_pos_of_id_kernel(const uint32_t *__restrict__ id_of_pos, uint32_t *__restrict__ pos_of_id, uint64_t n)
{
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n && id_of_pos[i] < n) // (id_of_pos is a permutation of [0, n): ochip_dense_link rejects anything else)
        pos_of_id[id_of_pos[i]] = (uint32_t)i;
}
__device__ __forceinline__ void dense_ray_intersection(const dv3 &d1, const dv3 &o1, const dv3 &d2, const dv3 &o2, dv3 *mid, double *err)
{
    auto dot = [](const dv3 &a, const dv3 &b) { return a.x * b.x + a.y * b.y + a.z * b.z; };
    const double nan = __longlong_as_double(0x7ff8000000000000ll);
    *mid = {nan, nan, nan};
    *err = nan;
    const double n11 = dot(d1, d1), n12 = dot(d1, d2), n22 = dot(d2, d2);
    const double denom = n11 * n22 - n12 * n12;
    if (fabs(denom) > 1e-9)
    {
        const dv3 off{o1.x - o2.x, o1.y - o2.y, o1.z - o2.z};
        const double od1 = dot(off, d1), od2 = dot(off, d2);
        const double t = (n12 * od2 - n22 * od1) / denom;
        const double s = (n11 * od2 - n12 * od1) / denom;
        const dv3 p1{o1.x + d1.x * t, o1.y + d1.y * t, o1.z + d1.z * t}, p2{o2.x + d2.x * s, o2.y + d2.y * s, o2.z + d2.z * s};
        *mid = {(p1.x + p2.x) * 0.5, (p1.y + p2.y) * 0.5, (p1.z + p2.z) * 0.5};
        const dv3 g{p1.x - p2.x, p1.y - p2.y, p1.z - p2.z};
        *err = dot(g, g) * (t >= 0 && s >= 0 ? 1 : -1);
    }
}
struct dense_tri_args
{
    const dense_image_meta *meta;
    uint32_t n_images, n_tracks;
    const dense_cam *cams;
    const double *cam_q; // [n_images][4] orientation (the records hold its inverse)
    const double2 *loc;
    const uint32_t *pos_of_id, *track_start, *track_member;
    double max_err_sq;
    double *points;  // [n_tracks][3]
    uint8_t *valid;  // [n_tracks]
    uint32_t *large, *n_large; // tracks of more than TRI_LARGE members: a wavefront each, behind the one-thread-per-track pass
    uint32_t lds_images;       // the images' first positions go to LDS up to this many images (TRI_LDS_IMAGES; 0: the test hook)
};
constexpr uint32_t TRI_LARGE = 32, TRI_LDS_IMAGES = 2048;
// the images' first positions in LDS (a member's image is a binary search over them: ten dependent reads per member)
struct dense_tri_images
{
    const dense_image_meta *meta;
    const uint32_t *first; // LDS copy, or nullptr above TRI_LDS_IMAGES images
    uint32_t n_images;
    __device__ uint32_t of(uint32_t pos) const // the image whose positions hold pos (every image of the index has features)
    {
        uint32_t lo = 0, hi = n_images;
        while (hi - lo > 1)
        {
            const uint32_t mid = (lo + hi) / 2;
            const bool below = first ? first[mid] <= pos : meta[mid].feat_base <= pos;
            lo = below ? mid : lo;
            hi = below ? hi : mid;
        }
        return lo;
    }
};
__device__ __forceinline__ dense_tri_images dense_tri_stage(const dense_tri_args &A, uint32_t *lds)
{
    const bool staged = A.n_images <= A.lds_images;
    if (staged)
    {
        for (uint32_t i = threadIdx.x; i < A.n_images; i += blockDim.x)
            lds[i] = (uint32_t)A.meta[i].feat_base;
        __syncthreads();
    }
    return dense_tri_images{A.meta, staged ? lds : nullptr, A.n_images};
}
struct dense_tri_member
{
    uint32_t image;
    double2 px;
};
__device__ __forceinline__ dense_tri_member dense_tri_member_at(const dense_tri_args &A, const dense_tri_images &I, uint32_t m)
{
    const uint32_t pos = A.pos_of_id[A.track_member[m]];
    return dense_tri_member{I.of(pos), A.loc[pos]};
}
__device__ __forceinline__ void dense_tri_ray(const dense_tri_args &A, const dense_tri_member &M, dv3 *dir, dv3 *origin)
{
    const dense_cam &c = A.cams[M.image];
    const double key[2] = {M.px.x, M.px.y};
    double ray[3];
    ochip_ud::image_to_3d(key, c.model, ray);
    *dir = drotate(A.cam_q + 4 * (size_t)M.image, dv3{ray[0], ray[1], ray[2]});
    *origin = dv3{c.pos[0], c.pos[1], c.pos[2]};
}
__device__ __forceinline__ bool dense_tri_point(const dense_tri_args &A, const dense_tri_images &I, uint32_t ma, uint32_t mb, dv3 *point)
{
    dv3 d1, o1, d2, o2;
    double err;
    dense_tri_ray(A, dense_tri_member_at(A, I, ma), &d1, &o1);
    dense_tri_ray(A, dense_tri_member_at(A, I, mb), &d2, &o2);
    dense_ray_intersection(d1, o1, d2, o2, point, &err);
    return isfinite(point->x) && isfinite(point->y) && isfinite(point->z) && !(err < 0);
}
__device__ __forceinline__ bool dense_tri_inlier(const dense_tri_args &A, const dense_tri_images &I, const dv3 &point, uint32_t m)
{
    const dense_tri_member M = dense_tri_member_at(A, I, m);
    double reproj[2];
    dense_project(point, A.cams[M.image], reproj);
    const double ex = reproj[0] - M.px.x, ey = reproj[1] - M.px.y;
    return ex * ex + ey * ey <= A.max_err_sq;
}
__global__ __launch_bounds__(128) void dense_triangulate_kernel(dense_tri_args A)
{
    __shared__ uint32_t first_pos[TRI_LDS_IMAGES];
    const dense_tri_images I = dense_tri_stage(A, first_pos);
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= A.n_tracks)
        return;
    A.valid[t] = 0;
    const uint32_t m0 = A.track_start[t], n = A.track_start[t + 1] - m0;
    if (n < 2)
        return;
    if (n > TRI_LARGE) // (a merged track of thousands of members would keep this thread for the whole launch)
    {
        A.large[atomicAdd(A.n_large, 1u)] = t;
        return;
    }
    dv3 point;
    if (!dense_tri_point(A, I, m0, m0 + 1, &point)) // only the first two rays (:145-161)
        return;
    uint32_t n_inliers = 0, f0 = 0, f1 = 0;
    for (uint32_t i = 0; i < n; i++)
        if (dense_tri_inlier(A, I, point, m0 + i))
        {
            f0 = n_inliers == 0 ? i : f0;
            f1 = n_inliers == 1 ? i : f1;
            n_inliers++;
        }
    if (n_inliers < 2)
        return;
    if (n_inliers < n && !dense_tri_point(A, I, m0 + f0, m0 + f1, &point))
        return;
    A.points[3 * (size_t)t] = point.x;
    A.points[3 * (size_t)t + 1] = point.y;
    A.points[3 * (size_t)t + 2] = point.z;
    A.valid[t] = 1;
}
// the same for a track of many members, by a wavefront: the members' reprojections lane by lane, the number of inliers and the
// two smallest inlier indices by wavefront reductions
__global__ __launch_bounds__(256) void dense_triangulate_large_kernel(dense_tri_args A)
{
    __shared__ uint32_t first_pos[TRI_LDS_IMAGES];
    const dense_tri_images I = dense_tri_stage(A, first_pos);
    const uint32_t w = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (w >= *A.n_large)
        return;
    const uint32_t t = A.large[w];
    const uint32_t m0 = A.track_start[t], n = A.track_start[t + 1] - m0;
    dv3 point;
    if (!dense_tri_point(A, I, m0, m0 + 1, &point)) // (every lane: the same point)
        return;
    uint32_t mine = 0, first = 0xFFFFFFFFu, second = 0xFFFFFFFFu; // this lane's inliers: how many, the two smallest indices
    for (uint32_t i = lane; i < n; i += 64)
        if (dense_tri_inlier(A, I, point, m0 + i))
        {
            second = mine == 1 ? i : second;
            first = mine == 0 ? i : first;
            mine++;
        }
    uint32_t n_inliers = mine;
    for (int off = 32; off >= 1; off >>= 1)
        n_inliers += (uint32_t)__shfl_xor((int)n_inliers, off);
    if (n_inliers < 2)
        return;
    if (n_inliers < n)
    {
        const uint32_t f0 = wave_min_u32(first);
        const uint32_t f1 = wave_min_u32(first == f0 ? second : first);
        if (!dense_tri_point(A, I, m0 + f0, m0 + f1, &point))
            return;
    }
    if (lane == 0)
    {
        A.points[3 * (size_t)t] = point.x;
        A.points[3 * (size_t)t + 1] = point.y;
        A.points[3 * (size_t)t + 2] = point.z;
        A.valid[t] = 1;
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
    void *cams = nullptr;    // ochip_dense_link's camera records and measurement ids (by position), kept for
    uint32_t *ids = nullptr; // ochip_dense_triangulate
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
    std::vector<dense_image_meta> meta((size_t)n_images + 1); // (one more: where the last image's features end)
    meta[n_images] = dense_image_meta{feat_off[n_images], cell_off[n_images], 0, 0, 0.0, 0.0};
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

int ochip_dense_link(ochip_dense_index *ix, const double *cams17, const uint32_t *id_of_pos, const double *hits3, double radius,
                     uint32_t max_candidates, uint32_t descriptor_bits, double ratio, double max_abs, uint32_t *root_out,
                     uint64_t *counts2, uint32_t *slot_dst_out)
{
    if (!ix || !cams17 || !id_of_pos || !hits3 || !root_out || !counts2)
        return OCHIP_EINVAL;
    ochip_ctx *ctx = ix->ctx;
    counts2[0] = counts2[1] = 0;
    if (!(radius > 0) || !(radius < ix->cell_size))
        return ochip_fail(ctx, OCHIP_EINVAL, "ochip_dense_link: the search radius %g must be below the index's cell size %g", radius,
                          ix->cell_size);
    if (max_candidates + 1 != (uint32_t)DENSE_K || descriptor_bits == 0)
        return ochip_fail(ctx, OCHIP_EINVAL, "ochip_dense_link: built for %d candidate images per feature", DENSE_K - 1);
    const uint64_t total = ix->total_features;
    const uint32_t n_images = ix->n_images;
    if (total == 0 || n_images == 0)
        return OCHIP_OK;
    {
        // id_of_pos is a permutation of [0, total): the kernels scatter and gather through it and through its inverse
        std::vector<uint8_t> seen((size_t)total, 0);
        for (uint64_t i = 0; i < total; i++)
        {
            if (id_of_pos[i] >= total || seen[id_of_pos[i]])
                return ochip_fail(ctx, OCHIP_EINVAL, "ochip_dense_link: id_of_pos is not a permutation of the %llu measurements (position %llu)",
                                  (unsigned long long)total, (unsigned long long)i);
            seen[id_of_pos[i]] = 1;
        }
    }
    OCHIP_HIP(ctx, hipSetDevice(ctx->device));
    hipStream_t st = ctx->stream;
    std::vector<std::pair<void *, size_t>> blocks;
    int rc = OCHIP_OK;
    auto get = [&](size_t bytes) -> void * {
        size_t got = 0;
        void *d = ochip_pool_get(ctx, bytes ? bytes : 16, &got);
        if (!d)
            rc = ochip_fail(ctx, OCHIP_ENOMEM, "ochip_dense_link: device allocation of %zu bytes failed", bytes);
        else
            blocks.emplace_back(d, got);
        return d;
    };
    auto done = [&](int code) {
        (void)ochip_stream_wait(ctx, st);
        for (auto &b : blocks)
            ochip_pool_put(ctx, b.first, b.second);
        return code;
    };
    // batches of source images: bounded slot arrays (the reference walks its images in batches of OpenMP tasks too).
    // (measured and dropped, round 5: a batch's nearest-camera pass on the context's second stream under the search of the batch
    // before it, two sets of candidate arrays - the device phase stayed at 56.5 ms: the search fills the device by itself)
    std::vector<uint64_t> feat_base(n_images + 1);
    {
        std::vector<dense_image_meta> meta((size_t)n_images + 1);
        if (hipMemcpyAsync(meta.data(), ix->meta, meta.size() * sizeof(dense_image_meta), hipMemcpyDeviceToHost, st) != hipSuccess ||
            ochip_stream_wait(ctx, st) != hipSuccess)
            return ochip_fail(ctx, OCHIP_EHIP, "ochip_dense_link: reading the index failed");
        for (uint32_t i = 0; i <= n_images; i++)
            feat_base[i] = meta[i].feat_base;
    }
    const uint32_t BATCH = 64;
    uint64_t max_batch_feats = 0, max_image_feats = 0;
    for (uint32_t b0 = 0; b0 < n_images; b0 += BATCH)
        max_batch_feats = std::max(max_batch_feats, feat_base[std::min(n_images, b0 + BATCH)] - feat_base[b0]);
    for (uint32_t i = 0; i < n_images; i++)
        max_image_feats = std::max(max_image_feats, feat_base[i + 1] - feat_base[i]);
    // (the camera records and the ids stay with the index: ochip_dense_triangulate reads them)
    auto keep = [&](size_t bytes) -> void * {
        size_t got = 0;
        void *d = ochip_pool_get(ctx, bytes ? bytes : 16, &got);
        if (!d)
            rc = ochip_fail(ctx, OCHIP_ENOMEM, "ochip_dense_link: device allocation of %zu bytes failed", bytes);
        else
            ix->blocks.emplace_back(d, got);
        return d;
    };
    if (!ix->cams)
        ix->cams = keep((size_t)n_images * sizeof(dense_cam));
    if (!ix->ids)
        ix->ids = (uint32_t *)keep(total * 4);
    dense_cam *cams = (dense_cam *)ix->cams;
    uint32_t *ids = ix->ids;
    double *hits = (double *)get(total * 24);
    uint32_t *parent = (uint32_t *)get(total * 4), *root = (uint32_t *)get(total * 4);
    uint8_t *matched = (uint8_t *)get(total);
    uint32_t *cand_img = (uint32_t *)get(max_batch_feats * DENSE_K * 4);
    double2 *cand_px = (double2 *)get(max_batch_feats * DENSE_K * 16);
    uint32_t *slot_dst = slot_dst_out ? (uint32_t *)get(max_batch_feats * DENSE_K * 4) : nullptr;
    unsigned long long *counters = (unsigned long long *)get(2 * DENSE_COUNTERS * 8);
    if (rc != OCHIP_OK)
        return done(rc);
    static_assert(sizeof(dense_cam) == 17 * sizeof(double), "cams17 is the kernel's camera record");
    if (hipMemcpyAsync(cams, cams17, (size_t)n_images * sizeof(dense_cam), hipMemcpyHostToDevice, st) != hipSuccess ||
        hipMemcpyAsync(ids, id_of_pos, total * 4, hipMemcpyHostToDevice, st) != hipSuccess ||
        hipMemcpyAsync(hits, hits3, total * 24, hipMemcpyHostToDevice, st) != hipSuccess ||
        hipMemsetAsync(counters, 0, 2 * DENSE_COUNTERS * 8, st) != hipSuccess)
        return done(ochip_fail(ctx, OCHIP_EHIP, "ochip_dense_link: upload failed"));
    hipLaunchKernelGGL(dense_uf_init_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, parent, matched, total);
    const double inv_bits = 1.0 / (double)descriptor_bits;
    const bool unstaged_hook = ochip_test_hook("dense_predict_unstaged"); // (tests: the route of surveys above PRED_LDS_CAMS cameras)
    for (uint32_t b0 = 0; b0 < n_images; b0 += BATCH)
    {
        const uint32_t b1 = std::min(n_images, b0 + BATCH);
        const uint64_t base = feat_base[b0], feats = feat_base[b1] - base, slots = feats * DENSE_K;
        if (feats == 0)
            continue;
        uint64_t widest = 0;
        for (uint32_t i = b0; i < b1; i++)
            widest = std::max(widest, feat_base[i + 1] - feat_base[i]);
        if (n_images <= PRED_LDS_CAMS && !unstaged_hook)
            hipLaunchKernelGGL(dense_predict_kernel<true>, dim3((unsigned)((widest + 255) / 256), b1 - b0), dim3(256), (size_t)n_images * 24, st,
                               ix->meta, cams, n_images, b0, hits, base, cand_img, cand_px, counters);
        else
            hipLaunchKernelGGL(dense_predict_kernel<false>, dim3((unsigned)((widest + 255) / 256), b1 - b0), dim3(256), 0, st, ix->meta, cams,
                               n_images, b0, hits, base, cand_img, cand_px, counters);
        hipEvent_t e0, e1;
        ochip_prof_begin(ctx, OCHIP_K_DENSE, &e0, &e1);
        const uint64_t waves = (slots + SLOTS_PER_WAVE - 1) / SLOTS_PER_WAVE;
        hipLaunchKernelGGL(dense_link_kernel, dim3((unsigned)((waves + 3) / 4)), dim3(256), 0, st, ix->meta, ix->desc, ix->loc, ix->cell_start,
                           cand_img, cand_px, slots, base, ids, radius * radius, ix->cell_size, inv_bits, ratio, max_abs, parent, matched,
                           slot_dst, counters + DENSE_COUNTERS);
        ochip_prof_end(ctx, OCHIP_K_DENSE, e0, e1);
        if (hipGetLastError() != hipSuccess)
            return done(ochip_fail(ctx, OCHIP_EHIP, "ochip_dense_link: launch failed"));
        if (slot_dst_out)
        {
            if (hipMemcpyAsync(slot_dst_out + base * DENSE_K, slot_dst, slots * 4, hipMemcpyDeviceToHost, st) != hipSuccess ||
                ochip_stream_wait(ctx, st) != hipSuccess)
                return done(ochip_fail(ctx, OCHIP_EHIP, "ochip_dense_link: reading the slots failed"));
        }
    }
    hipLaunchKernelGGL(dense_uf_roots_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, parent, matched, total, root);
    std::vector<unsigned long long> host_counts(2 * DENSE_COUNTERS, 0);
    if (hipGetLastError() != hipSuccess || hipMemcpyAsync(root_out, root, total * 4, hipMemcpyDeviceToHost, st) != hipSuccess ||
        hipMemcpyAsync(host_counts.data(), counters, 2 * DENSE_COUNTERS * 8, hipMemcpyDeviceToHost, st) != hipSuccess ||
        ochip_stream_wait(ctx, st) != hipSuccess)
        return done(ochip_fail(ctx, OCHIP_EHIP, "ochip_dense_link: %s", hipGetErrorString(hipGetLastError())));
    counts2[0] = counts2[1] = 0;
    for (int i = 0; i < DENSE_COUNTERS; i++)
    {
        counts2[0] += host_counts[i];
        counts2[1] += host_counts[DENSE_COUNTERS + i];
    }
    return done(OCHIP_OK);
}

int ochip_dense_triangulate(ochip_dense_index *ix, const double *cam_q4, uint32_t n_tracks, const uint32_t *track_start,
                            const uint32_t *track_member, double max_reprojection_error, double *points3_out, uint8_t *valid_out)
{
    if (!ix || !cam_q4 || (n_tracks && (!track_start || !track_member || !points3_out || !valid_out)))
        return OCHIP_EINVAL;
    ochip_ctx *ctx = ix->ctx;
    if (n_tracks == 0)
        return OCHIP_OK;
    if (!ix->cams || !ix->ids)
        return ochip_fail(ctx, OCHIP_ESTATE, "ochip_dense_triangulate: ochip_dense_link has not run on this index");
    OCHIP_HIP(ctx, hipSetDevice(ctx->device));
    hipStream_t st = ctx->stream;
    const uint64_t total = ix->total_features;
    const uint64_t n_members = track_start[n_tracks];
    for (uint32_t t = 0; t < n_tracks; t++)
        if (track_start[t] > track_start[t + 1])
            return ochip_fail(ctx, OCHIP_EINVAL, "ochip_dense_triangulate: track_start does not ascend at track %u", t);
    for (uint64_t m = 0; m < n_members; m++) // (a member is a measurement id of the index: the kernels gather pos_of_id[member])
        if (track_member[m] >= total)
            return ochip_fail(ctx, OCHIP_EINVAL, "ochip_dense_triangulate: track member %llu is measurement %u of %llu", (unsigned long long)m,
                              track_member[m], (unsigned long long)total);
    std::vector<std::pair<void *, size_t>> blocks;
    int rc = OCHIP_OK;
    auto get = [&](size_t bytes) -> void * {
        size_t got = 0;
        void *d = ochip_pool_get(ctx, bytes ? bytes : 16, &got);
        if (!d)
            rc = ochip_fail(ctx, OCHIP_ENOMEM, "ochip_dense_triangulate: device allocation of %zu bytes failed", bytes);
        else
            blocks.emplace_back(d, got);
        return d;
    };
    auto done = [&](int code) {
        (void)ochip_stream_wait(ctx, st);
        for (auto &b : blocks)
            ochip_pool_put(ctx, b.first, b.second);
        return code;
    };
    double *q = (double *)get((size_t)ix->n_images * 32);
    uint32_t *pos_of_id = (uint32_t *)get(total * 4);
    uint32_t *start = (uint32_t *)get(((size_t)n_tracks + 1) * 4), *members = (uint32_t *)get(n_members * 4);
    double *points = (double *)get((size_t)n_tracks * 24);
    uint8_t *valid = (uint8_t *)get(n_tracks);
    const uint32_t max_large = (uint32_t)(n_members / (TRI_LARGE + 1)) + 1;
    uint32_t *large = (uint32_t *)get(((size_t)max_large + 1) * 4);
    if (rc != OCHIP_OK)
        return done(rc);
    if (hipMemsetAsync(large + max_large, 0, 4, st) != hipSuccess)
        return done(ochip_fail(ctx, OCHIP_EHIP, "ochip_dense_triangulate: memset failed"));
    if (hipMemcpyAsync(q, cam_q4, (size_t)ix->n_images * 32, hipMemcpyHostToDevice, st) != hipSuccess ||
        hipMemcpyAsync(start, track_start, ((size_t)n_tracks + 1) * 4, hipMemcpyHostToDevice, st) != hipSuccess ||
        (n_members && hipMemcpyAsync(members, track_member, n_members * 4, hipMemcpyHostToDevice, st) != hipSuccess))
        return done(ochip_fail(ctx, OCHIP_EHIP, "ochip_dense_triangulate: upload failed"));
    hipLaunchKernelGGL(dense_pos_of_id_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, ix->ids, pos_of_id, total);
    dense_tri_args A{};
    A.meta = ix->meta;
    A.n_images = ix->n_images;
    A.n_tracks = n_tracks;
    A.cams = (const dense_cam *)ix->cams;
    A.cam_q = q;
    A.loc = ix->loc;
    A.pos_of_id = pos_of_id;
    A.track_start = start;
    A.track_member = members;
    A.max_err_sq = max_reprojection_error * max_reprojection_error;
    A.points = points;
    A.valid = valid;
    A.large = large;
    A.lds_images = ochip_test_hook("dense_predict_unstaged") ? 0u : TRI_LDS_IMAGES; // (tests: the route of surveys above 2 048 images)
    A.n_large = large + max_large;
    hipLaunchKernelGGL(dense_triangulate_kernel, dim3((n_tracks + 127) / 128), dim3(128), 0, st, A);
    hipLaunchKernelGGL(dense_triangulate_large_kernel, dim3((max_large + 3) / 4), dim3(256), 0, st, A);
    if (hipGetLastError() != hipSuccess || hipMemcpyAsync(points3_out, points, (size_t)n_tracks * 24, hipMemcpyDeviceToHost, st) != hipSuccess ||
        hipMemcpyAsync(valid_out, valid, n_tracks, hipMemcpyDeviceToHost, st) != hipSuccess || ochip_stream_wait(ctx, st) != hipSuccess)
        return done(ochip_fail(ctx, OCHIP_EHIP, "ochip_dense_triangulate: %s", hipGetErrorString(hipGetLastError())));
    return done(OCHIP_OK);
}

} // extern "C"
