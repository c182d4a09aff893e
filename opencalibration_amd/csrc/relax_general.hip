// libochip.so — relax (bundle adjustment) in its general form on the device (gfx950): ground mesh with per-vertex
// heights, residual blocks of 2..5 rays, mesh priors, downward prior, shared intrinsics.  C ABI: ochip_relaxg_* (ochip.h).
//
// Replaces what ceres::Solver::Solve does for RelaxProblem::setupGroundMeshProblem / setupGroundPlaneProblem
// (src/relax/relax_problem.cpp:61-120,1390-1420).  Data-parallel structure:
//   * a RECORD per residual block: the packed upper triangle of its J'J (in the block's own columns), its J'r and its cost.
//     Ray blocks are evaluated by (block, pass) lanes: a pass seeds forward-mode duals (Dual<3>) for one group of three
//     columns (one camera's quaternion tangent, the three heights, f + principal point, the radial coefficients), the
//     passes of a block sit in neighbouring lanes of one wavefront and meet in LDS, where the block's J'J entries are
//     formed as dot products of Jacobian columns.  HBM-bound: one pass over the observation arrays, one record written.
//   * ASSEMBLY without atomics: every unknown group (camera, vertex, intrinsic group) owns the rows of the dense
//     system it stands for.  One wavefront per owner walks the owner's records in a fixed order and adds their
//     contributions into an LDS strip that covers the columns the owner is coupled to (its band of the block envelope
//     plus the dense tail), then stores the strip.  Sums are bitwise reproducible.
//   * unknowns that couple to (nearly) everything - the heights of a coarse mesh, the shared intrinsics - form the dense
//     TAIL of the system (last rows); their own rows are reduced by chunked owners + a merge.
//   * the linear solve and the Levenberg-Marquardt loop are the shared ones (relax_lm.hip).
#include "ctx.hpp"
#include "dual.hpp"
#include "relax_functors.hpp"
#include "relax_lm.hpp"

#include <algorithm>
#include <cmath>
#include <cstring>
#include <numeric>
#include <vector>

using namespace ochip;

namespace
{

constexpr int W = 64;
constexpr int NB = LM_NB;
constexpr int MAXV = 11; // unknown groups of one record: <= 5 cameras + 3 vertices + focal, principal point, radial

// record types: ray blocks = number of rays (2..5), + 8 with the shared lens model; then the priors
enum : uint8_t
{
    T_INTR = 8,
    T_DOWN = 16,
    T_DIFF = 17,
    T_ANCHOR = 18,
    T_SMOOTH = 19,
    T_MONO = 20,
    T_REL = 21, // MultiDecomposedRotationCost between two cameras (setupDecompositionProblem)
    T_NULL = 31 // padding between the ray blocks and the priors of a sharded problem: no unknowns, no data, cost 0
};

__host__ __device__ inline int rec_dim(int type)
{
    if (type < T_DOWN)
        return 3 * (type & 7) + 3 + ((type & T_INTR) ? 6 : 0);
    if (type == T_NULL)
        return 0;
    if (type == T_REL)
        return 6;
    return type == T_DOWN ? 3 : type == T_DIFF ? 2 : type == T_ANCHOR ? 1 : type == T_SMOOTH ? 4 : 3;
}
__host__ __device__ inline int rec_nvars(int type)
{
    if (type < T_DOWN)
        return (type & 7) + 3 + ((type & T_INTR) ? 3 : 0);
    if (type == T_NULL)
        return 0;
    if (type == T_REL)
        return 2;
    return type == T_DOWN ? 1 : type == T_DIFF ? 2 : type == T_ANCHOR ? 1 : type == T_SMOOTH ? 4 : 1;
}
// local column -> (slot of the record's unknown group, offset inside the group)
__host__ __device__ inline void rec_col(int type, int c, int *slot, int *off)
{
    if (type < T_DOWN)
    {
        const int N = type & 7;
        if (c < 3 * N)
        {
            *slot = c / 3;
            *off = c % 3;
        }
        else if (c < 3 * N + 3)
        {
            *slot = N + (c - 3 * N);
            *off = 0;
        }
        else if (c == 3 * N + 3)
        {
            *slot = N + 3;
            *off = 0;
        }
        else if (c < 3 * N + 6)
        {
            *slot = N + 4;
            *off = c - (3 * N + 4);
        }
        else
        {
            *slot = N + 5;
            *off = c - (3 * N + 6);
        }
        return;
    }
    if (type == T_DOWN || type == T_MONO)
    {
        *slot = 0;
        *off = c;
        return;
    }
    if (type == T_REL)
    {
        *slot = c / 3;
        *off = c % 3;
        return;
    }
    *slot = c; // DIFF, ANCHOR, SMOOTH: one column per vertex
    *off = 0;
}
// first local column of a slot
__host__ __device__ inline int rec_slot_col(int type, int slot)
{
    if (type < T_DOWN)
    {
        const int N = type & 7;
        if (slot < N)
            return 3 * slot;
        if (slot < N + 3)
            return 3 * N + (slot - N);
        return slot == N + 3 ? 3 * N + 3 : slot == N + 4 ? 3 * N + 4 : 3 * N + 6;
    }
    if (type == T_DOWN || type == T_MONO)
        return 0;
    if (type == T_REL)
        return 3 * slot;
    return slot;
}
__host__ __device__ inline int tri_idx(int i, int j, int d) // i <= j, packed upper triangle of a d x d matrix
{
    return i * d - i * (i - 1) / 2 + (j - i);
}

struct g_dev
{
    uint32_t n_cams, n_verts;
    double *cam_pos, *cam_q, *cam_q2;
    double *vert_xy, *vert_z, *vert_z2, *vert_z0;
    double *model, *model2; // 8 each
    int32_t *var_t;         // first unknown of a group or -1
    uint8_t *var_ts;        // its number of unknowns
    // ray blocks, sorted by type
    uint32_t *blk_ray_off, *ray_cam, *blk_tri;
    double *ray_dir, *ray_px;
    // records
    uint32_t n_rec;
    uint8_t *rec_type;
    uint64_t *rec_off;
    uint32_t *rec_var; // [n_rec][MAXV]
    double *rec_data, *rec_cost;
    // priors
    uint32_t n_down, n_diff, n_anchor, n_smooth, n_mono, n_rel, prior_base; // record ids: prior_base + [down | diff | anchor | smooth | mono | rel]
    uint32_t *down_cam, *diff_v, *smooth_v, *rel_cam;
    double *rel_pose; // [n_rel][4][8] decompositions: q xyzw, t xyz, score
    double rel_huber_a;
    double down_w, diff_w, anchor_w, smooth_w, mono_w, mono_rmax, huber_a, f_lo, f_hi;
    uint8_t n_k_free;
    int32_t *fail;
};

// ---- ray blocks ------------------------------------------------------------------------------------------------
template <typename T, int N, bool INTR>
__device__ __forceinline__ bool ray_block_residuals(const g_dev &P, uint32_t blk, int which, int pass, T *res)
{
    // pass (only for T = Dual<3>): which group of three columns carries the dual parts: 0..N-1 camera, N heights,
    // N + 1 focal + principal point, N + 2 radial; -1: none
    const double *Q = which ? P.cam_q2 : P.cam_q;
    const double *Z = which ? P.vert_z2 : P.vert_z;
    const double *M = which ? P.model2 : P.model;
    const uint32_t r0 = P.blk_ray_off[blk];
    T q[N][4];
    Vec3T<T> ray[N];
    double loc[N][3];
    T z[3];
    double txy[6];
    for (int j = 0; j < 3; j++)
    {
        const uint32_t v = P.blk_tri[3 * (size_t)blk + j];
        txy[2 * j] = P.vert_xy[2 * (size_t)v];
        txy[2 * j + 1] = P.vert_xy[2 * (size_t)v + 1];
        z[j] = T(Z[v]);
    }
    T m[8];
    if (INTR)
        for (int k = 0; k < 8; k++)
            m[k] = T(M[k]);
    if constexpr (!std::is_same<T, double>::value)
    {
        if (pass == N)
            for (int j = 0; j < 3; j++)
                z[j].v[j] = 1.0;
        if (INTR && pass == N + 1)
            for (int j = 0; j < 3; j++)
                m[j].v[j] = 1.0;
        if (INTR && pass == N + 2)
            for (int j = 0; j < 3; j++)
                m[3 + j].v[j] = 1.0;
    }
    for (int i = 0; i < N; i++)
    {
        const uint32_t c = P.ray_cam[r0 + i];
        const double *qc = Q + 4 * (size_t)c;
        if constexpr (!std::is_same<T, double>::value)
        {
            if (pass == i)
                gseed_quat(qc, q[i]);
            else
                for (int k = 0; k < 4; k++)
                    q[i][k] = T(qc[k]);
        }
        else
            for (int k = 0; k < 4; k++)
                q[i][k] = qc[k];
        for (int k = 0; k < 3; k++)
            loc[i][k] = P.cam_pos[3 * (size_t)c + k];
        if (INTR)
            ray[i] = gimage_to_3d_inverse<T>(P.ray_px + 2 * (size_t)(r0 + i), m);
        else
        {
            const double *d = P.ray_dir + 3 * (size_t)(r0 + i);
            ray[i] = {T(d[0]), T(d[1]), T(d[2])};
        }
    }
    return gmulti_ray_residuals<T, N>(q, ray, loc, txy, z, res);
}

// Huber loss + Triggs corrector of a block with squared norm s (rho'' <= 0 for Huber: plain sqrt(rho') scaling)
__device__ __forceinline__ void huber(double s, double a, bool with_loss, double *rho1, double *cost)
{
    *rho1 = 1.0;
    *cost = 0.5 * s;
    if (with_loss && s > a * a)
    {
        const double rn = sqrt(s);
        *rho1 = fmax(2.2250738585072014e-308, a / rn);
        *cost = 0.5 * (2.0 * a * rn - a * a);
    }
}

// cost of every block of one type at state `which`; one thread per block
template <int N, bool INTR> __global__ __launch_bounds__(W) void ray_cost_kernel(g_dev P, uint32_t first, uint32_t count, int which)
{
    const uint32_t i = blockIdx.x * W + threadIdx.x;
    if (i >= count)
        return;
    const uint32_t blk = first + i;
    double r[3 * N];
    bool failed = !ray_block_residuals<double, N, INTR>(P, blk, which, -1, r);
    double s = 0;
    for (int k = 0; k < 3 * N; k++)
    {
        s += r[k] * r[k];
        if (!(r[k] - r[k] == 0.0))
            failed = true;
    }
    double rho1, cost;
    huber(s, P.huber_a, N == 2, &rho1, &cost);
    P.rec_cost[blk] = cost;
    if (failed)
        atomicOr(P.fail, 1);
}

// records of every block of one type at the current state: (block, pass) lanes
// DT: the dual type of the pass (Dual<3>: everything in fp64; Dual<3, float>: the Jacobian propagated in fp32, for the C5
// precision sweep only)
template <int N, bool INTR, typename DT = Dual<3>>
__global__ __launch_bounds__(W) void ray_record_kernel(g_dev P, uint32_t first, uint32_t count, int which)
{
    constexpr int PASSES = N + 1 + (INTR ? 2 : 0);
    constexpr int G = W / PASSES; // blocks per wavefront
    constexpr int D = 3 * N + 3 + (INTR ? 6 : 0);
    constexpr int R = 3 * N;
    constexpr int TRI = D * (D + 1) / 2;
    __shared__ double Jt[G][D][R];
    __shared__ uint16_t lut[TRI];
    const int lane = threadIdx.x;
    for (int i = lane; i < D; i += W)
        for (int j = i; j < D; j++)
            lut[tri_idx(i, j, D)] = (uint16_t)((i << 8) | j);
    const int g = lane / PASSES, pass = lane % PASSES;
    const uint32_t bi = blockIdx.x * G + g;
    const bool active = g < G && bi < count;
    const uint32_t blk = first + (active ? bi : 0);
    double r[R];
    double rho1 = 1.0, cost = 0.0;
    bool failed = false;
    if (active)
    {
        DT rd[R];
        if (!ray_block_residuals<DT, N, INTR>(P, blk, which, pass, rd))
            failed = true;
        double s = 0;
        for (int k = 0; k < R; k++)
        {
            r[k] = rd[k].a;
            s += r[k] * r[k];
            if (!(r[k] - r[k] == 0.0))
                failed = true;
        }
        huber(s, P.huber_a, N == 2, &rho1, &cost);
        const int c0 = 3 * pass; // the pass order is the column order: cameras, heights, (f, pp), radial
        for (int cidx = 0; cidx < 3; cidx++)
            for (int k = 0; k < R; k++)
            {
                const double v = (double)rd[k].v[cidx];
                Jt[g][c0 + cidx][k] = v;
                if (!(v - v == 0.0))
                    failed = true;
            }
    }
    __syncthreads();
    if (active)
    {
        double *o = P.rec_data + P.rec_off[blk];
        for (int e = pass; e < TRI + D; e += PASSES)
        {
            double v = 0;
            if (e < TRI)
            {
                const int i = lut[e] >> 8, j = lut[e] & 255;
                for (int k = 0; k < R; k++)
                    v += Jt[g][i][k] * Jt[g][j][k];
            }
            else
            {
                const int i = e - TRI;
                for (int k = 0; k < R; k++)
                    v += Jt[g][i][k] * r[k];
            }
            o[e] = v * rho1;
        }
        if (pass == 0)
            P.rec_cost[blk] = cost;
    }
    if (failed)
        atomicOr(P.fail, 1);
}

// ---- priors: one thread per prior --------------------------------------------------------------------------------
__global__ void prior_kernel(g_dev P, int which, int with_jac)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t n_all = P.n_down + P.n_diff + P.n_anchor + P.n_smooth + P.n_mono + P.n_rel;
    if (i >= n_all)
        return;
    const uint32_t rec = P.prior_base + i;
    const double *Q = which ? P.cam_q2 : P.cam_q;
    const double *Z = which ? P.vert_z2 : P.vert_z;
    const double *M = which ? P.model2 : P.model;
    double *o = with_jac ? P.rec_data + P.rec_off[rec] : nullptr;
    double cost = 0;
    bool failed = false;
    uint32_t k = i;
    if (k < P.n_down)
    {
        double r, j3[3];
        gdownward_prior(Q + 4 * (size_t)P.down_cam[k], P.down_w, &r, j3);
        cost = 0.5 * r * r;
        if (with_jac)
        {
            int e = 0;
            for (int a = 0; a < 3; a++)
                for (int b = a; b < 3; b++)
                    o[e++] = j3[a] * j3[b];
            for (int a = 0; a < 3; a++)
                o[6 + a] = j3[a] * r;
        }
        failed = !(r - r == 0.0);
    }
    else if ((k -= P.n_down) < P.n_diff)
    {
        const double w = P.diff_w;
        const double r = w * (Z[P.diff_v[2 * k]] - Z[P.diff_v[2 * k + 1]]);
        cost = 0.5 * r * r;
        if (with_jac)
        {
            o[0] = w * w;
            o[1] = -w * w;
            o[2] = w * w;
            o[3] = w * r;
            o[4] = -w * r;
        }
    }
    else if ((k -= P.n_diff) < P.n_anchor)
    {
        const double w = P.anchor_w;
        const double r = w * (Z[k] - P.vert_z0[k]);
        cost = 0.5 * r * r;
        if (with_jac)
        {
            o[0] = w * w;
            o[1] = w * r;
        }
    }
    else if ((k -= P.n_anchor) < P.n_smooth)
    {
        double xy[8];
        Dual<4> z[4];
        for (int a = 0; a < 4; a++)
        {
            const uint32_t v = P.smooth_v[4 * k + a];
            xy[2 * a] = P.vert_xy[2 * (size_t)v];
            xy[2 * a + 1] = P.vert_xy[2 * (size_t)v + 1];
            z[a] = Dual<4>(Z[v]);
            z[a].v[a] = 1.0;
        }
        const Dual<4> r = gadjacent_triangle_normal<Dual<4>>(xy, z, P.smooth_w);
        cost = 0.5 * r.a * r.a;
        if (with_jac)
        {
            int e = 0;
            for (int a = 0; a < 4; a++)
                for (int b = a; b < 4; b++)
                    o[e++] = r.v[a] * r.v[b];
            for (int a = 0; a < 4; a++)
            {
                o[10 + a] = r.v[a] * r.a;
                if (!(r.v[a] - r.v[a] == 0.0))
                    failed = true;
            }
        }
        failed |= !(r.a - r.a == 0.0);
    }
    else if ((k -= P.n_smooth) >= P.n_mono)
    {
        // MultiDecomposedRotationCost (relax_cost_function.hpp:253-307) with HuberLoss(10 degrees): two passes of duals,
        // one per camera's tangent
        k -= P.n_mono;
        const uint32_t c1 = P.rel_cam[2 * k], c2 = P.rel_cam[2 * k + 1];
        const double *poses = P.rel_pose + 32 * (size_t)k;
        double r[3], J[3][6];
        for (int pass = 0; pass < 2; pass++)
        {
            Dual<3> q1[4], q2[4], rd[3];
            if (pass == 0)
                gseed_quat(Q + 4 * (size_t)c1, q1);
            else
                for (int a = 0; a < 4; a++)
                    q1[a] = Dual<3>(Q[4 * (size_t)c1 + a]);
            if (pass == 1)
                gseed_quat(Q + 4 * (size_t)c2, q2);
            else
                for (int a = 0; a < 4; a++)
                    q2[a] = Dual<3>(Q[4 * (size_t)c2 + a]);
            if (!gmulti_decomposed_rotation<Dual<3>>(q1, q2, poses, P.cam_pos + 3 * (size_t)c1, P.cam_pos + 3 * (size_t)c2, rd))
            {
                failed = true;
                for (int a = 0; a < 3; a++)
                    rd[a] = Dual<3>(0.0);
            }
            for (int a = 0; a < 3; a++)
            {
                r[a] = rd[a].a;
                for (int c = 0; c < 3; c++)
                {
                    J[a][3 * pass + c] = rd[a].v[c];
                    if (!(rd[a].v[c] - rd[a].v[c] == 0.0))
                        failed = true;
                }
            }
        }
        const double s = r[0] * r[0] + r[1] * r[1] + r[2] * r[2];
        double rho1;
        huber(s, P.rel_huber_a, true, &rho1, &cost);
        if (with_jac)
        {
            int e = 0;
            for (int a = 0; a < 6; a++)
                for (int b = a; b < 6; b++)
                    o[e++] = rho1 * (J[0][a] * J[0][b] + J[1][a] * J[1][b] + J[2][a] * J[2][b]);
            for (int a = 0; a < 6; a++)
                o[21 + a] = rho1 * (J[0][a] * r[0] + J[1][a] * r[1] + J[2][a] * r[2]);
        }
        failed |= !(s - s == 0.0);
    }
    else
    {
        // DistortionMonotonicityCost (relax_cost_function.hpp:157-185): 10 hinge residuals on d(r_d)/dr
        double acc[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
        for (int t = 0; t < 10; t++)
        {
            const double rr = P.mono_rmax * (t + 1.0) / 10;
            const double r2 = rr * rr, r4 = r2 * r2, r6 = r4 * r2;
            const double deriv = 1.0 + 3.0 * M[3] * r2 + 5.0 * M[4] * r4 + 7.0 * M[5] * r6;
            if (deriv < 0.0)
            {
                const double res = P.mono_w * (-deriv);
                const double j[3] = {-P.mono_w * 3.0 * r2, -P.mono_w * 5.0 * r4, -P.mono_w * 7.0 * r6};
                cost += 0.5 * res * res;
                int e = 0;
                for (int a = 0; a < 3; a++)
                    for (int b = a; b < 3; b++)
                        acc[e++] += j[a] * j[b];
                for (int a = 0; a < 3; a++)
                    acc[6 + a] += j[a] * res;
            }
        }
        if (with_jac)
            for (int e = 0; e < 9; e++)
                o[e] = acc[e];
    }
    P.rec_cost[rec] = cost;
    if (failed)
        atomicOr(P.fail, 1);
}

// total cost in two levels, both in a fixed order: a workgroup per slice of COST_SLICE ray-block costs (strided sums + tree),
// then one workgroup over the slices' sums and the priors.  scal[0] = cost.  The ray blocks [0, n_ray) and the priors
// [prior_base, prior_base + n_prior) are walked relative to their own starts, so the order of the additions does not
// depend on the padding a sharded problem puts between the two (sharded and unsharded solves add the same numbers in the
// same order).  (One workgroup over 650 k costs took 0.2 ms with 1 024 threads and 0.75 ms with 256: the largest kernel
// of a mesh relax's iteration.)
constexpr uint32_t COST_SLICE = 4096;
__global__ __launch_bounds__(LM_TG) void cost_slices_kernel(const double *rec_cost, uint32_t n_ray, double *slice_sum)
{
    __shared__ double sh[LM_TG];
    const int t = threadIdx.x;
    const uint32_t lo = blockIdx.x * COST_SLICE, hi = min(lo + COST_SLICE, n_ray);
    double v = 0;
    for (uint32_t i = lo + t; i < hi; i += LM_TG)
        v += rec_cost[i];
    sh[t] = v;
    __syncthreads();
    for (int s = LM_TG / 2; s > 0; s >>= 1)
    {
        if (t < s)
            sh[t] += sh[t + s];
        __syncthreads();
    }
    if (t == 0)
        slice_sum[blockIdx.x] = sh[0];
}
__global__ __launch_bounds__(LM_TG) void cost_reduce_kernel(const double *rec_cost, const double *slice_sum, uint32_t n_slices,
                                                           uint32_t prior_base, uint32_t n_prior, double *scal)
{
    __shared__ double sh[LM_TG];
    const int t = threadIdx.x;
    double v = 0;
    for (uint32_t i = t; i < n_slices; i += LM_TG)
        v += slice_sum[i];
    for (uint32_t i = t; i < n_prior; i += LM_TG)
        v += rec_cost[prior_base + i];
    sh[t] = v;
    __syncthreads();
    for (int s = LM_TG / 2; s > 0; s >>= 1)
    {
        if (t < s)
            sh[t] += sh[t + s];
        __syncthreads();
    }
    if (t == 0)
        scal[0] = sh[0];
}

// ---- assembly ------------------------------------------------------------------------------------------------------
struct work_item
{
    uint32_t var;      // owner
    uint32_t lo, hi;   // range of var_rec entries
    int32_t col_lo;    // band strip = columns [col_lo, col_hi)  (tail owners: empty)
    int32_t col_hi;
    int32_t partial;   // >= 0: store the strip as partial #partial of a tail owner (tail columns only)
    int64_t band_off;  // >= 0: a chunk of a band owner with many records: the whole strip goes to band_partials + band_off
};

// One wavefront per work item.  strip[a][x]: a < rows of the owner, x over [band strip | tail | gradient].
__global__ __launch_bounds__(W) void gather_kernel(g_dev P, const work_item *items, const uint32_t *var_rec, lm_matrix A, double *g,
                                                   int n, int tail_begin, double *partials, int strip_cap, double *band_partials)
{
    extern __shared__ double strip[];
    __shared__ int16_t colmap[W][24];
    __shared__ uint64_t roff[W];
    __shared__ uint8_t rdim[W], rlo[W];
    const int lane = threadIdx.x;
    const work_item it = items[blockIdx.x];
    const int tu = P.var_t[it.var], su = P.var_ts[it.var];
    const int T = n - tail_begin;
    const int wb = it.col_hi - it.col_lo; // band part
    const int ws = wb + T + 1;            // + tail + gradient
    for (int i = lane; i < su * ws; i += W)
        strip[i] = 0.0;
    __syncthreads();
    for (uint32_t e0 = it.lo; e0 < it.hi; e0 += W)
    {
        const uint32_t cnt = min((uint32_t)W, it.hi - e0);
        // phase A: one record per lane: where do its columns land in the strip?
        if ((uint32_t)lane < cnt)
        {
            const uint32_t ent = var_rec[e0 + lane];
            const uint32_t r = ent >> 4, slot = ent & 15;
            const int type = P.rec_type[r], d = rec_dim(type);
            roff[lane] = P.rec_off[r];
            rdim[lane] = (uint8_t)d;
            rlo[lane] = (uint8_t)rec_slot_col(type, slot);
            // the unknown groups of the record: all their ids first, then all their places in the system (three dependent
            // round trips for the whole record instead of two per group; unused slots hold group 0, a valid load)
            uint32_t vv[MAXV];
            int tvs[MAXV], tszs[MAXV];
#pragma unroll
            for (int s = 0; s < MAXV; s++)
                vv[s] = P.rec_var[(size_t)r * MAXV + s];
#pragma unroll
            for (int s = 0; s < MAXV; s++)
            {
                tvs[s] = P.var_t[vv[s]];
                tszs[s] = P.var_ts[vv[s]];
            }
            for (int c = 0; c < d; c++)
            {
                int s, off;
                rec_col(type, c, &s, &off);
                int tv = 0, tsz = 0;
#pragma unroll
                for (int k = 0; k < MAXV; k++) // (a select chain: the arrays stay in registers)
                    if (k == s)
                    {
                        tv = tvs[k];
                        tsz = tszs[k];
                    }
                int idx = -1;
                if (tv >= 0 && off < tsz)
                {
                    const int x = tv + off;
                    if (x >= tail_begin)
                        idx = wb + (x - tail_begin);
                    else if (x >= it.col_lo && x < it.col_hi)
                        idx = x - it.col_lo;
                }
                colmap[lane][c] = (int16_t)idx;
            }
        }
        __syncthreads();
        // phase B: the records one after the other (fixed order), their entries spread over the lanes.  What bounds this
        // loop is the latency of the record loads, so the values of 16 records are requested together before the first
        // of them is added (the adds keep the record order: the sums stay reproducible).
        constexpr int KB = 16;
        for (uint32_t i0 = 0; i0 < cnt; i0 += KB)
        {
            double v[KB];
            int sidx[KB];
#pragma unroll
            for (int k = 0; k < KB; k++)
            {
                v[k] = 0.0;
                sidx[k] = -1;
                const uint32_t i = i0 + k;
                if (i >= cnt)
                    continue;
                const int d = rdim[i], lo_u = rlo[i];
                if (lane >= su * (d + 1))
                    continue;
                const int a = (lane >= d + 1) + (lane >= 2 * (d + 1)), c = lane - a * (d + 1);
                const int row = lo_u + a;
                const double *rec = P.rec_data + roff[i];
                if (c == d)
                {
                    sidx[k] = a * ws + ws - 1;
                    v[k] = rec[d * (d + 1) / 2 + row];
                }
                else
                {
                    const int idx = colmap[i][c];
                    if (idx >= 0)
                    {
                        sidx[k] = a * ws + idx;
                        v[k] = rec[row <= c ? tri_idx(row, c, d) : tri_idx(c, row, d)];
                    }
                }
            }
#pragma unroll
            for (int k = 0; k < KB; k++)
            {
                if (sidx[k] >= 0)
                    strip[sidx[k]] += v[k];
                // entries beyond the first 64 of a wide record (3 rows x more than 21 columns: tracks of 5 rays with the
                // shared lens model): a second round, same order
                const uint32_t i = i0 + k;
                if (i < cnt)
                {
                    const int d = rdim[i], lo_u = rlo[i];
                    for (int l = lane + W; l < su * (d + 1); l += W)
                    {
                        const int a = l / (d + 1), c = l % (d + 1);
                        const int row = lo_u + a;
                        const double *rec = P.rec_data + roff[i];
                        if (c == d)
                            strip[a * ws + ws - 1] += rec[d * (d + 1) / 2 + row];
                        else
                        {
                            const int idx = colmap[i][c];
                            if (idx >= 0)
                                strip[a * ws + idx] += rec[row <= c ? tri_idx(row, c, d) : tri_idx(c, row, d)];
                        }
                    }
                }
            }
        }
        __syncthreads();
    }
    if (it.band_off >= 0)
    {
        // a chunk of a band owner: the strip as it is, merged with the owner's other chunks by band_merge_kernel
        double *o = band_partials + it.band_off;
        for (int i = lane; i < su * ws; i += W)
            o[i] = strip[i];
        return;
    }
    if (it.partial >= 0)
    {
        // tail owner: its tail columns and gradient go to the partial buffer [partial][su][T + 1]
        double *o = partials + (size_t)it.partial * 3 * (T + 1);
        for (int i = lane; i < su * (T + 1); i += W)
        {
            const int a = i / (T + 1), x = i % (T + 1);
            o[a * (T + 1) + x] = strip[a * ws + wb + x];
        }
        return;
    }
    // (the system keeps its lower triangle only, relax_lm.hpp: of the band part the columns up to the diagonal; the
    // columns of the tail are the tail rows' band part, mirrored)
    for (int a = 0; a < su; a++)
    {
        const int r = tu + a;
        for (int x = lane; x < wb && it.col_lo + x <= r; x += W)
            A.tiles[lm_at(A, r, it.col_lo + x)] = strip[a * ws + x];
        for (int x = lane; x < T; x += W)
            A.tiles[lm_at(A, tail_begin + x, r)] = strip[a * ws + wb + x];
        if (lane == 0)
            g[tu + a] = strip[a * ws + ws - 1];
    }
}

// tail rows: sum of the chunk partials in chunk order.  One workgroup per tail owner.
__global__ void tail_merge_kernel(g_dev P, const uint32_t *tail_var, const uint32_t *tail_first, const uint32_t *tail_count,
                                  const double *partials, lm_matrix A, double *g, int n, int tail_begin)
{
    const uint32_t u = tail_var[blockIdx.x];
    const int tu = P.var_t[u], su = P.var_ts[u];
    const int T = n - tail_begin;
    for (int i = threadIdx.x; i < su * (T + 1); i += blockDim.x)
    {
        const int a = i / (T + 1), x = i % (T + 1);
        double v = 0;
        for (uint32_t c = 0; c < tail_count[blockIdx.x]; c++)
            v += partials[(size_t)(tail_first[blockIdx.x] + c) * 3 * (T + 1) + a * (T + 1) + x];
        if (x == T)
            g[tu + a] = v;
        else if (tail_begin + x <= tu + a)
            A.tiles[lm_at(A, tu + a, tail_begin + x)] = v;
    }
}

// band owners with many records (cameras of a large group, well-observed vertices) are cut into chunks like the tail
// owners - a wavefront per chunk instead of one wavefront walking thousands of records - and their strips are added here
// in chunk order (fixed, so the sums stay reproducible).  One workgroup per such owner.
struct band_owner
{
    uint32_t var;
    int32_t col_lo, col_hi;
    uint32_t chunks;
    int64_t first_off; // of its first chunk's strip in band_partials; the chunks follow each other
};
__global__ void band_merge_kernel(g_dev P, const band_owner *owners, const double *band_partials, lm_matrix A, double *g, int n,
                                  int tail_begin)
{
    const band_owner bo = owners[blockIdx.x];
    const int tu = P.var_t[bo.var], su = P.var_ts[bo.var];
    const int T = n - tail_begin, wb = bo.col_hi - bo.col_lo, ws = wb + T + 1;
    const size_t stride = (size_t)su * ws;
    for (int i = threadIdx.x; i < su * ws; i += blockDim.x)
    {
        const int a = i / ws, x = i % ws, r = tu + a;
        double v = 0;
        for (uint32_t c = 0; c < bo.chunks; c++)
            v += band_partials[bo.first_off + (int64_t)(c * stride) + i];
        if (x == ws - 1)
            g[r] = v;
        else if (x >= wb)
            A.tiles[lm_at(A, tail_begin + (x - wb), r)] = v; // the tail rows' band part, mirrored
        else if (bo.col_lo + x <= r)
            A.tiles[lm_at(A, r, bo.col_lo + x)] = v;
    }
}

// ---- state -------------------------------------------------------------------------------------------------------
// candidate = x (+) delta, delta = -y .* scale.  One workgroup.  scal[2] = |x - candidate|^2, scal[3] = |candidate|^2
__global__ __launch_bounds__(LM_TG) void general_candidate_kernel(g_dev P, const double *scale, const double *y, double alpha, double *scal)
{
    __shared__ double sh[LM_TG];
    const int t = threadIdx.x;
    double sn = 0, xn = 0;
    for (uint32_t c = t; c < P.n_cams; c += LM_TG)
    {
        const int tc = P.var_t[c];
        const double *q = P.cam_q + (size_t)c * 4;
        double *o = P.cam_q2 + (size_t)c * 4;
        if (tc < 0)
        {
            for (int k = 0; k < 4; k++)
                o[k] = q[k];
            continue;
        }
        double d[3];
        for (int k = 0; k < 3; k++)
            d[k] = alpha * (-y[tc + k] * scale[tc + k]);
        const double nrm = sqrt(d[0] * d[0] + d[1] * d[1] + d[2] * d[2]);
        if (nrm == 0.0)
        {
            for (int k = 0; k < 4; k++)
                o[k] = q[k];
        }
        else
        {
            const double s = sin(nrm) / nrm;
            const double dx = s * d[0], dy = s * d[1], dz = s * d[2], dw = cos(nrm);
            const double qx = q[0], qy = q[1], qz = q[2], qw = q[3];
            o[3] = dw * qw - dx * qx - dy * qy - dz * qz;
            o[0] = dw * qx + dx * qw + dy * qz - dz * qy;
            o[1] = dw * qy + dy * qw + dz * qx - dx * qz;
            o[2] = dw * qz + dz * qw + dx * qy - dy * qx;
        }
        for (int k = 0; k < 4; k++)
        {
            sn += (q[k] - o[k]) * (q[k] - o[k]);
            xn += o[k] * o[k];
        }
    }
    for (uint32_t v = t; v < P.n_verts; v += LM_TG)
    {
        const int tz = P.var_t[P.n_cams + v];
        const double z0 = P.vert_z[v];
        const double z1 = tz >= 0 ? z0 + alpha * (-y[tz] * scale[tz]) : z0;
        P.vert_z2[v] = z1;
        if (tz >= 0)
        {
            sn += (z0 - z1) * (z0 - z1);
            xn += z1 * z1;
        }
    }
    if (t == 0)
    {
        // shared lens model: focal (projected onto its bounds, ceres ParameterBlock::Plus), principal point, radial
        const uint32_t vf = P.n_cams + P.n_verts;
        for (int k = 0; k < 8; k++)
            P.model2[k] = P.model[k];
        const int tf = P.var_t[vf], tp = P.var_t[vf + 1], tk = P.var_t[vf + 2];
        if (tf >= 0)
        {
            double f = P.model[0] + alpha * (-y[tf] * scale[tf]);
            f = fmin(fmax(f, P.f_lo), P.f_hi);
            P.model2[0] = f;
            sn += (P.model[0] - f) * (P.model[0] - f);
            xn += f * f;
        }
        if (tp >= 0)
            for (int k = 0; k < 2; k++)
            {
                const double v = P.model[1 + k] + alpha * (-y[tp + k] * scale[tp + k]);
                P.model2[1 + k] = v;
                sn += (P.model[1 + k] - v) * (P.model[1 + k] - v);
                xn += v * v;
            }
        if (tk >= 0)
            for (int k = 0; k < 3; k++) // SubsetManifold: the trailing coefficients stay, but count in |x|
            {
                const double v = k < P.n_k_free ? P.model[3 + k] + alpha * (-y[tk + k] * scale[tk + k]) : P.model[3 + k];
                P.model2[3 + k] = v;
                sn += (P.model[3 + k] - v) * (P.model[3 + k] - v);
                xn += v * v;
            }
    }
    for (int q = 0; q < 2; q++)
    {
        sh[t] = q == 0 ? sn : xn;
        __syncthreads();
        for (int s = LM_TG / 2; s > 0; s >>= 1)
        {
            if (t < s)
                sh[t] += sh[t + s];
            __syncthreads();
        }
        if (t == 0)
            scal[2 + q] = sh[0];
        __syncthreads();
    }
}

__global__ void general_accept_kernel(g_dev P)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < P.n_cams * 4)
        P.cam_q[i] = P.cam_q2[i];
    if (i < P.n_verts)
        P.vert_z[i] = P.vert_z2[i];
    if (i < 8)
        P.model[i] = P.model2[i];
}

// p.second->orientation.normalize() for every node of _nodes_to_optimize (relax_problem.cpp:1410-1413)
__global__ void general_normalize_kernel(g_dev P, const uint8_t *cam_optimize)
{
    const uint32_t c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= P.n_cams || !cam_optimize[c])
        return;
    double *q = P.cam_q + (size_t)c * 4;
    const double n = sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
    for (int k = 0; k < 4; k++)
        q[k] = q[k] / n;
}

} // namespace

// ---------------------------------------------------------------------------------------------------------------------
struct ochip_relaxg_problem
{
    ochip_ctx *ctx = nullptr;
    g_dev dev{};
    std::vector<std::pair<void *, size_t>> allocs;
    lm_system sys;
    uint32_t n_cams = 0, n_verts = 0, n_vars = 0, n_blocks = 0, n_rec = 0;
    bool structure_only = false;
    std::vector<uint8_t> cam_optimize, vert_optimize;
    uint8_t opt_f = 0, opt_pp = 0, n_k_free = 0;
    bool has_intr_blocks = false;
    // host copies used by the ordering
    std::vector<double> cam_xy, vert_xy;
    std::vector<uint8_t> rec_type;
    std::vector<uint32_t> rec_var; // [n_rec][MAXV]
    std::vector<uint32_t> var_rec_off, var_rec; // CSR unknown group -> (record << 4 | slot)
    std::vector<int32_t> var_t;
    std::vector<uint8_t> var_ts;
    int n_tangent = 0, tail_begin = 0, n_padding = 0; // (n_padding: unowned unknowns that align the band's regions to tiles)
    // per block type: first record and count
    struct type_range
    {
        int type;
        uint32_t first, count;
    };
    std::vector<type_range> ranges;
    // assembly plan (rebuilt by assign)
    uint32_t n_items = 0, n_tail_owners = 0, n_partials = 0, n_band_owners = 0;
    band_owner *band_owners_dev = nullptr;
    double *band_partials_dev = nullptr, *cost_slices_dev = nullptr;
    int max_strip = 0;
    work_item *items_dev = nullptr;
    uint32_t *var_rec_dev = nullptr, *tail_var_dev = nullptr, *tail_first_dev = nullptr, *tail_count_dev = nullptr;
    double *partials_dev = nullptr;
    uint8_t *cam_optimize_dev = nullptr;
    // sharded evaluation (ochip_relaxg_desc.shard_world > 1): this rank evaluates the ray blocks [shard_lo, shard_hi); the
    // record data of rank r lies at r * shard_stride doubles, its costs at r * shard_chunk
    uint32_t shard_rank = 0, shard_world = 1, shard_chunk = 0, shard_lo = 0, shard_hi = 0;
    uint64_t shard_stride = 0;
    int32_t *fail_ranks = nullptr;
    ochip_relax_exchange_fn exchange = nullptr;
    void *exchange_user = nullptr;
};

namespace
{

constexpr int STRIP_CAP = 2300;       // band + tail columns one owner's LDS strip may span (3 rows x 2301 doubles = 55 KB)
constexpr uint32_t TAIL_CHUNK = 1024;  // most records per chunk of a tail owner (a chunk is one wavefront walking its records in order; the length is chosen
                                      // per owner, assign())
constexpr uint32_t BAND_CHUNK = 512;  // a band owner with more than twice this many records is cut into chunks of this size
constexpr uint32_t DENSE_VERTS = 8;   // a mesh of at most this many vertices is dense (plane, minimal mesh): tail

template <typename T> int up(ochip_relaxg_problem *p, T **dst, const T *src, size_t n)
{
    return lm_dev_upload(p->ctx, &p->allocs, dst, src, n);
}
template <typename T> int up(ochip_relaxg_problem *p, T **dst, const std::vector<T> &v)
{
    return lm_dev_upload(p->ctx, &p->allocs, dst, v.data(), v.size());
}

// Which unknown groups are variable, in which order, with which envelope; and the assembly plan.
int assign(ochip_relaxg_problem *p)
{
    const uint32_t nc = p->n_cams, nv = p->n_verts, vf = nc + nv;
    std::vector<char> variable(p->n_vars, 0), is_tail(p->n_vars, 0);
    auto has_records = [&](uint32_t u) { return p->var_rec_off[u + 1] > p->var_rec_off[u]; };
    for (uint32_t c = 0; c < nc; c++)
        variable[c] = p->cam_optimize[c] && !p->structure_only && has_records(c);
    uint32_t n_var_verts = 0;
    for (uint32_t v = 0; v < nv; v++)
    {
        variable[nc + v] = p->vert_optimize[v] && has_records(nc + v);
        n_var_verts += variable[nc + v];
    }
    variable[vf] = p->opt_f && !p->structure_only && has_records(vf);
    variable[vf + 1] = p->opt_pp && !p->structure_only && has_records(vf + 1);
    variable[vf + 2] = p->n_k_free > 0 && !p->structure_only && has_records(vf + 2);
    for (uint32_t k = 0; k < 3; k++)
        is_tail[vf + k] = 1;
    if (n_var_verts <= DENSE_VERTS)
        for (uint32_t v = 0; v < nv; v++)
            is_tail[nc + v] = 1;
    auto size_of = [&](uint32_t u) -> int { return u < nc ? 3 : u < vf ? 1 : u == vf ? 1 : u == vf + 1 ? 2 : (int)p->n_k_free; };

    // band order: along the long axis of the bounding box of the band unknowns (cameras and vertices live in the same
    // ground coordinates; an unknown is coupled to what lies within a camera footprint of it).  Regions (round 4): the band
    // cut into stretches of the sweep that are coupled through the tail only (below); a region starts on a tile boundary,
    // the gap in front of it is padding - unknowns nobody owns: zero rows of J'J that the damping makes positive, zero step.
    std::vector<uint32_t> band, tail;
    std::vector<int> lo, hi, region_of(p->n_vars, 0), region_first_block;
    int n_regions = 1, n_padding = 0;
    // one layout: order, tangent offsets, coupling ranges; unknowns whose strip would not fit join the tail and it repeats
    auto layout = [&]() -> int {
        for (int round = 0;; round++)
        {
            band.clear();
            tail.clear();
            for (uint32_t u = 0; u < p->n_vars; u++)
                if (variable[u])
                    (is_tail[u] ? tail : band).push_back(u);
            auto xy = [&](uint32_t u, int a) { return u < nc ? p->cam_xy[2 * u + a] : p->vert_xy[2 * (u - nc) + a]; };
            double mn[2] = {1e300, 1e300}, mx[2] = {-1e300, -1e300};
            for (uint32_t u : band)
                for (int a = 0; a < 2; a++)
                {
                    mn[a] = std::min(mn[a], xy(u, a));
                    mx[a] = std::max(mx[a], xy(u, a));
                }
            const int ax = (mx[0] - mn[0]) >= (mx[1] - mn[1]) ? 0 : 1;
            std::stable_sort(band.begin(), band.end(), [&](uint32_t a, uint32_t b) {
                if (region_of[a] != region_of[b])
                    return region_of[a] < region_of[b];
                const double ka = xy(a, ax), kb = xy(b, ax);
                if (ka != kb)
                    return ka < kb;
                return xy(a, 1 - ax) < xy(b, 1 - ax);
            });
            p->var_t.assign(p->n_vars, -1);
            p->var_ts.assign(p->n_vars, 0);
            region_first_block.clear();
            n_padding = 0;
            int t = 0, current = -1;
            for (uint32_t u : band)
            {
                if (n_regions > 1 && region_of[u] != current)
                {
                    const int aligned = (t + NB - 1) / NB * NB;
                    n_padding += aligned - t;
                    t = aligned;
                    region_first_block.push_back(t / NB);
                    current = region_of[u];
                }
                p->var_t[u] = t;
                p->var_ts[u] = (uint8_t)size_of(u);
                t += size_of(u);
            }
            p->tail_begin = t;
            for (uint32_t u : tail)
            {
                p->var_t[u] = t;
                p->var_ts[u] = (uint8_t)size_of(u);
                t += size_of(u);
            }
            p->n_tangent = t;
            // coupling ranges of the band unknowns: over every record, the span of its band unknowns
            lo.assign(p->n_vars, INT32_MAX);
            hi.assign(p->n_vars, -1);
            for (uint32_t r = 0; r < p->n_rec; r++)
            {
                const int nvr = rec_nvars(p->rec_type[r]);
                int rlo = INT32_MAX, rhi = -1;
                for (int s = 0; s < nvr; s++)
                {
                    const uint32_t u = p->rec_var[(size_t)r * MAXV + s];
                    if (p->var_t[u] >= 0 && !is_tail[u])
                    {
                        rlo = std::min(rlo, p->var_t[u]);
                        rhi = std::max(rhi, p->var_t[u] + p->var_ts[u]);
                    }
                }
                if (rhi < 0)
                    continue;
                for (int s = 0; s < nvr; s++)
                {
                    const uint32_t u = p->rec_var[(size_t)r * MAXV + s];
                    if (p->var_t[u] >= 0 && !is_tail[u])
                    {
                        lo[u] = std::min(lo[u], rlo);
                        hi[u] = std::max(hi[u], rhi);
                    }
                }
            }
            // an unknown coupled to more columns than an LDS strip holds joins the tail; then order again
            const int T = p->n_tangent - p->tail_begin;
            bool moved = false;
            for (uint32_t u : band)
                if (hi[u] - lo[u] + T > STRIP_CAP)
                {
                    is_tail[u] = 1;
                    moved = true;
                }
            if (!moved)
                return OCHIP_OK;
            if (round == 3)
                return OCHIP_EINVAL;
        }
    };
    if (layout() != OCHIP_OK)
        return ochip_fail(p->ctx, OCHIP_EINVAL, "relax: the coupling structure does not fit the assembly strips");
    // Dissection of the band (as the plane engine's camera graph, relax.hip): cut the sweep at R - 1 places; what lies
    // behind a cut and is coupled to something in front of it becomes a separator - tail, dense rows -, and the stretches
    // between are regions whose chains of diagonal tiles factor side by side.  R minimises the longest region plus 1.5 x
    // the tail; no dissection unless that is below 0.8 of the single chain and the tail stays within its 1 024 unknowns.
    if (!ochip_test_hook("no_dissect") && p->tail_begin >= 8 * NB)
    {
        const int B = p->tail_begin, T0 = p->n_tangent - p->tail_begin;
        int best_r = 0, best_path = B + T0 + T0 / 2;
        std::vector<char> best_sep;
        for (int R = 2; R <= 16; R++)
        {
            std::vector<int> cuts;
            for (int i = 1; i < R; i++)
                cuts.push_back((int)((long)B * i / R));
            std::vector<char> sep(p->n_vars, 0);
            int seps = 0;
            for (uint32_t u : band)
                for (int c : cuts)
                    if (p->var_t[u] >= c && lo[u] < c)
                    {
                        sep[u] = 1;
                        seps += p->var_ts[u];
                        break;
                    }
            if (T0 + seps > 1024)
                continue;
            int longest = 0, prev = 0;
            for (size_t i = 0; i <= cuts.size(); i++)
            {
                const int end = i < cuts.size() ? cuts[i] : B;
                int size = 0;
                for (uint32_t u : band)
                    if (!sep[u] && p->var_t[u] >= prev && p->var_t[u] < end)
                        size += p->var_ts[u];
                longest = std::max(longest, size);
                prev = end;
            }
            const int path = longest + (T0 + seps) + (T0 + seps) / 2;
            if (path < best_path)
            {
                best_path = path;
                best_r = R;
                best_sep.swap(sep);
            }
        }
        if (best_r >= 2 && 10 * best_path <= 8 * (B + T0 + T0 / 2))
        {
            const std::vector<char> tail_before = is_tail;
            for (uint32_t u : band)
            {
                region_of[u] = std::min(best_r - 1, (int)((long)p->var_t[u] * best_r / B));
                // (the cut places are B i / R: the region of tangent offset t is the number of cuts at or below it)
                int r = 0;
                for (int i = 1; i < best_r; i++)
                    if (p->var_t[u] >= (int)((long)B * i / best_r))
                        r = i;
                region_of[u] = r;
                if (best_sep[u])
                    is_tail[u] = 1;
            }
            n_regions = best_r;
            bool ok = layout() == OCHIP_OK && p->n_tangent - p->tail_begin <= 1024 && (int)region_first_block.size() == n_regions;
            if (ok) // every band unknown is coupled inside its own region only
                for (uint32_t u : band)
                {
                    const int r = region_of[u];
                    const int r_lo = region_first_block[r] * NB, r_hi = r + 1 < n_regions ? region_first_block[r + 1] * NB : p->tail_begin;
                    if (hi[u] >= 0 && (lo[u] < r_lo || hi[u] > r_hi))
                        ok = false;
                }
            if (!ok)
            {
                is_tail = tail_before;
                std::fill(region_of.begin(), region_of.end(), 0);
                n_regions = 1;
                if (layout() != OCHIP_OK)
                    return ochip_fail(p->ctx, OCHIP_EINVAL, "relax: the coupling structure does not fit the assembly strips");
            }
        }
    }
    p->n_padding = n_padding;
    const int n = p->n_tangent, T = n - p->tail_begin;
    if (T > 1024)
        return ochip_fail(p->ctx, OCHIP_EINVAL, "relax: %d dense unknowns (limit 1024)", T);

    // block envelope for the factorisation
    lm_envelope env;
    {
        const int band_end = p->tail_begin;
        const int n_all = std::max(n, 1), nblk = (n_all + NB - 1) / NB;
        env.tail_begin = band_end;
        if (n_regions > 1)
            env.region_begin = region_first_block;
        env.env_end.assign(nblk, 0);
        for (int k = 0; k < nblk; k++)
            env.env_end[k] = std::min((k + 1) * NB, band_end);
        for (uint32_t u : band)
            for (int k = p->var_t[u] / NB; k <= (p->var_t[u] + p->var_ts[u] - 1) / NB; k++)
                env.env_end[k] = std::max(env.env_end[k], hi[u]);
        for (int k = 1; k < nblk; k++)
            env.env_end[k] = std::max(env.env_end[k], std::min(env.env_end[k - 1], band_end));
        env.first_col.assign(nblk, 0);
        for (int k = 0; k < nblk; k++)
        {
            const int k0 = k * NB;
            int first = k0;
            if (k0 + NB > band_end)
                first = 0;
            else
                for (int c = 0; c < k; c++)
                    if (env.env_end[c] > k0)
                    {
                        first = c * NB;
                        break;
                    }
            env.first_col[k] = first;
        }
        if (ochip_verbose("relax"))
        {
            long bandsum = 0;
            for (int k = 0; k < nblk; k++)
                bandsum += std::max(0, env.env_end[k] - (k + 1) * NB);
            fprintf(stderr, "[ochip relaxg] n=%d band=%d tail=%d blocks=%d mean envelope rows below a block %.1f; %d regions, %d padding\n",
                    n, band_end, T, nblk, (double)bandsum / nblk, n_regions, n_padding);
        }
    }
    p->sys.ctx = p->ctx;
    p->sys.allocs = &p->allocs;
    p->sys.speculative = true; // (the candidate is evaluated with its Jacobian into the second set: general_model::evaluate_candidate_jac)
    int rc = lm_system_resize(&p->sys, n, env);
    if (rc != OCHIP_OK)
        return rc;

    // assembly plan
    std::vector<work_item> items;
    std::vector<uint32_t> tail_var, tail_first, tail_count;
    int max_strip = 1;
    std::vector<band_owner> band_owners;
    int64_t band_doubles = 0;
    constexpr bool no_band_chunks = false; // (true: whole bands, the round-3 A/B)
    constexpr uint32_t band_chunk = BAND_CHUNK;
    for (uint32_t u : band)
    {
        const uint32_t r0 = p->var_rec_off[u], r1 = p->var_rec_off[u + 1];
        const int64_t strip = (int64_t)p->var_ts[u] * (hi[u] - lo[u] + T + 1);
        max_strip = std::max(max_strip, (int)strip);
        if (no_band_chunks || r1 - r0 <= 2 * band_chunk)
        {
            items.push_back(work_item{u, r0, r1, lo[u], hi[u], -1, -1});
            continue;
        }
        band_owner bo{u, lo[u], hi[u], 0, band_doubles};
        for (uint32_t e = r0; e < r1; e += band_chunk)
        {
            items.push_back(work_item{u, e, std::min(e + band_chunk, r1), lo[u], hi[u], -1, band_doubles});
            band_doubles += strip;
            bo.chunks++;
        }
        band_owners.push_back(bo);
    }
    uint32_t n_partials = 0;
    for (uint32_t u : tail)
    {
        tail_var.push_back(u);
        tail_first.push_back(n_partials);
        uint32_t cnt = 0;
        // chunk length: a chunk is walked record by record by one wavefront (~0.5 us each) and the owner's chunks are summed
        // one after the other (~0.3 us each): both chains are shortest near sqrt(0.6 records) - 96 for the 16 k records of a
        // 50-camera group (whose evaluation was one 540 us walk with chunks of 1 024), 530 for the 480 k of a whole survey
        const uint32_t n_rec_u = p->var_rec_off[u + 1] - p->var_rec_off[u];
        const uint32_t chunk = std::min(TAIL_CHUNK, std::max(64u, ((uint32_t)std::sqrt(0.6 * (double)n_rec_u) + 31u) / 32u * 32u));
        for (uint32_t e = p->var_rec_off[u]; e < p->var_rec_off[u + 1] || cnt == 0; e += chunk)
        {
            work_item it{u, e, std::min(e + chunk, p->var_rec_off[u + 1]), 0, 0, (int32_t)n_partials, -1};
            items.push_back(it);
            n_partials++;
            cnt++;
            if (p->var_rec_off[u + 1] == p->var_rec_off[u])
                break;
        }
        tail_count.push_back(cnt);
        max_strip = std::max(max_strip, (int)p->var_ts[u] * (T + 1));
    }
    p->n_items = (uint32_t)items.size();
    p->n_tail_owners = (uint32_t)tail_var.size();
    p->n_partials = n_partials;
    p->max_strip = max_strip;
    auto chk = [&](int r) {
        if (rc == OCHIP_OK)
            rc = r;
    };
    chk(up(p, &p->items_dev, items));
    chk(up(p, &p->tail_var_dev, tail_var));
    chk(up(p, &p->tail_first_dev, tail_first));
    chk(up(p, &p->tail_count_dev, tail_count));
    chk(up<double>(p, &p->partials_dev, nullptr, (size_t)std::max<uint32_t>(n_partials, 1) * 3 * (T + 1)));
    p->n_band_owners = (uint32_t)band_owners.size();
    chk(up(p, &p->band_owners_dev, band_owners));
    chk(up<double>(p, &p->band_partials_dev, nullptr, (size_t)std::max<int64_t>(band_doubles, 1)));
    chk(up<double>(p, &p->cost_slices_dev, nullptr, (size_t)p->n_blocks / COST_SLICE + 2));
    if (rc != OCHIP_OK)
        return rc;
    if (hipMemcpy(p->dev.var_t, p->var_t.data(), p->n_vars * 4, hipMemcpyHostToDevice) != hipSuccess ||
        hipMemcpy(p->dev.var_ts, p->var_ts.data(), p->n_vars, hipMemcpyHostToDevice) != hipSuccess)
        return ochip_fail(p->ctx, OCHIP_EHIP, "hipMemcpy failed (unknown map)");
    return OCHIP_OK;
}

template <int N, bool INTR> void launch_ray(const g_dev &D, hipStream_t st, uint32_t first, uint32_t count, bool with_jac, int which)
{
    if (with_jac)
    {
        constexpr int G = W / (N + 1 + (INTR ? 2 : 0));
        // SURVEY §8 / BASELINE C5: "fp32-vs-fp64 Jacobian sweep".  The path computes Jacobians in fp64 as Ceres does; this
        // switch (read once) propagates the dual parts of the ray blocks in fp32 instead - residual values, J'J accumulation
        // and the solve stay fp64 - so that scripts/sweep_jacobian_precision.py can measure what fp32 derivatives would cost
        static const bool jac32 = ochip_test_hook("jacobian_fp32");
        if (jac32)
            hipLaunchKernelGGL((ray_record_kernel<N, INTR, Dual<3, float>>), dim3((count + G - 1) / G), dim3(W), 0, st, D, first, count, which);
        else
            hipLaunchKernelGGL((ray_record_kernel<N, INTR>), dim3((count + G - 1) / G), dim3(W), 0, st, D, first, count, which);
    }
    else
        hipLaunchKernelGGL((ray_cost_kernel<N, INTR>), dim3((count + W - 1) / W), dim3(W), 0, st, D, first, count, which);
}

struct general_model final : lm_model
{
    ochip_relaxg_problem *p;
    bool mail_by_diag = false, fail_is_clear = false;
    explicit general_model(ochip_relaxg_problem *prob) : p(prob)
    {
    }
    int evaluate(bool with_jac, int which, double *cost) override
    {
        ochip_ctx *ctx = p->ctx;
        hipStream_t st = ctx->stream;
        g_dev &D = p->dev;
        const int n = p->n_tangent;
        // (this rank's failure flag: the kernel that mails an evaluation's results clears it again - evaluate_candidate_jac -,
        // any other evaluation leaves it to this memset)
        if (!fail_is_clear)
            OCHIP_HIP(ctx, hipMemsetAsync(D.fail, 0, 4, st));
        const bool mailed = mail_by_diag && p->shard_world <= (uint32_t)(2 * (lm_system::BOX_VECTORS - lm_system::BOX_FAILS));
        fail_is_clear = mailed;
        hipEvent_t e0, e1;
        ochip_prof_begin(ctx, OCHIP_K_RELAX_EVAL, &e0, &e1);
        for (const auto &full : p->ranges)
        {
            // this rank's part of the type's blocks (everything when the problem is not sharded)
            ochip_relaxg_problem::type_range r = full;
            const uint32_t a = std::max(full.first, p->shard_lo), b = std::min(full.first + full.count, p->shard_hi);
            r.first = a;
            r.count = b > a ? b - a : 0;
            if (r.count == 0)
                continue;
            const bool intr = (r.type & T_INTR) != 0;
            switch (r.type & 7)
            {
            case 2:
                intr ? launch_ray<2, true>(D, st, r.first, r.count, with_jac, which) : launch_ray<2, false>(D, st, r.first, r.count, with_jac, which);
                break;
            case 3:
                intr ? launch_ray<3, true>(D, st, r.first, r.count, with_jac, which) : launch_ray<3, false>(D, st, r.first, r.count, with_jac, which);
                break;
            case 4:
                intr ? launch_ray<4, true>(D, st, r.first, r.count, with_jac, which) : launch_ray<4, false>(D, st, r.first, r.count, with_jac, which);
                break;
            default:
                intr ? launch_ray<5, true>(D, st, r.first, r.count, with_jac, which) : launch_ray<5, false>(D, st, r.first, r.count, with_jac, which);
                break;
            }
        }
        const uint32_t n_prior = D.n_down + D.n_diff + D.n_anchor + D.n_smooth + D.n_mono + D.n_rel;
        if (n_prior)
            hipLaunchKernelGGL(prior_kernel, dim3((n_prior + 255) / 256), dim3(256), 0, st, D, which, with_jac ? 1 : 0);
        ochip_prof_end(ctx, OCHIP_K_RELAX_EVAL, e0, e1);
        if (p->exchange)
        {
            // the ranks' records (costs, failure flags) are all-gathered in place; from here on every rank holds the same
            // arrays and runs the same deterministic assembly.  A failed exchange is a hard error: the ranks would
            // otherwise leave the solve on different schedules and the next collective would hang.
            if (p->exchange != ochip_rccl_relax_exchange) // (the native transport is stream-ordered)
                OCHIP_HIP(ctx, ochip_stream_wait(ctx, st));
            const int xrc = p->exchange(p->exchange_user, D.rec_data, with_jac ? p->shard_stride * 8 : 0, D.rec_cost,
                                        (uint64_t)p->shard_chunk * 8, p->fail_ranks, 4);
            if (xrc != 0)
                return ochip_fail(ctx, OCHIP_EHIP, "relax exchange callback failed (%d)", xrc);
        }
        if (with_jac)
        {
            // the assembly ASSIGNS the same entries of A and g at every evaluation (fixed by the layout) and nothing else
            // writes them: what it leaves alone - fill-in positions, padding unknowns - is cleared once per layout and set
            if (!p->sys.A_clean)
            {
                OCHIP_HIP(ctx, hipMemsetAsync(p->sys.A, 0, p->sys.matrix_bytes(), st));
                OCHIP_HIP(ctx, hipMemsetAsync(p->sys.g, 0, (size_t)n * 8, st));
                p->sys.A_clean = true;
            }
            if (p->n_items)
                hipLaunchKernelGGL(gather_kernel, dim3(p->n_items), dim3(W), (size_t)p->max_strip * 8, st, D, p->items_dev,
                                   p->var_rec_dev, p->sys.matA(), p->sys.g, n, p->tail_begin, p->partials_dev, STRIP_CAP,
                                   p->band_partials_dev);
            if (p->n_band_owners)
                hipLaunchKernelGGL(band_merge_kernel, dim3(p->n_band_owners), dim3(256), 0, st, D, (const band_owner *)p->band_owners_dev,
                                   (const double *)p->band_partials_dev, p->sys.matA(), p->sys.g, n, p->tail_begin);
            if (p->n_tail_owners)
                hipLaunchKernelGGL(tail_merge_kernel, dim3(p->n_tail_owners), dim3(256), 0, st, D, p->tail_var_dev, p->tail_first_dev,
                                   p->tail_count_dev, p->partials_dev, p->sys.matA(), p->sys.g, n, p->tail_begin);
        }
        const uint32_t n_slices = (p->n_blocks + COST_SLICE - 1) / COST_SLICE;
        if (n_slices)
            hipLaunchKernelGGL(cost_slices_kernel, dim3(n_slices), dim3(LM_TG), 0, st, (const double *)D.rec_cost, p->n_blocks,
                               p->cost_slices_dev);
        hipLaunchKernelGGL(cost_reduce_kernel, dim3(1), dim3(LM_TG), 0, st, (const double *)D.rec_cost, (const double *)p->cost_slices_dev,
                           n_slices, D.prior_base, n_prior, p->sys.scal);
        OCHIP_HIP(ctx, hipGetLastError());
        // (read-backs into the system's page-locked block: a copy to pageable memory would make the host wait for it)
        std::vector<int32_t> hfails_pageable;
        double *const h0 = p->sys.box + lm_system::BOX_COST;
        int32_t *hfails = reinterpret_cast<int32_t *>(p->sys.box + lm_system::BOX_FAILS);
        if (p->shard_world > 2 * (lm_system::BOX_VECTORS - lm_system::BOX_FAILS))
        {
            hfails_pageable.assign(p->shard_world, 0);
            hfails = hfails_pageable.data();
        }
        if (!mailed) // (mailed: the solver's diagonal kernel, enqueued by before_wait, posts the cost and the flags too)
        {
            OCHIP_HIP(ctx, hipMemcpyAsync(h0, p->sys.scal, 8, hipMemcpyDeviceToHost, st));
            OCHIP_HIP(ctx, hipMemcpyAsync(hfails, p->fail_ranks, (size_t)p->shard_world * 4, hipMemcpyDeviceToHost, st));
        }
        if (before_wait)
            before_wait();
        OCHIP_HIP(ctx, ochip_stream_wait(ctx, st));
        *cost = *h0;
        int hfail = 0;
        for (int r = 0; r < (int)p->shard_world; r++)
            hfail |= hfails[r];
        return hfail ? 1 : 0;
    }
    void launch_candidate(const double *y, const double *scale, double alpha, double *scal) override
    {
        hipLaunchKernelGGL(general_candidate_kernel, dim3(1), dim3(LM_TG), 0, p->ctx->stream, p->dev, scale, y, alpha, scal);
    }
    // The candidate with its Jacobian, through the generic route: the system's sets are exchanged around a plain
    // evaluate(true, 1), so J'J, J'r and - by lm_launch_diag riding on the evaluation's wait - the clamped diagonal and
    // max |g| of the candidate land in the second set; an accepted step copies the state (launch_accept) and the solver
    // swaps the sets.  One evaluation and one wait per iteration instead of two of each.  (Any non-finite value counts
    // as a failed cost: the step is rejected.)
    bool speculates() override
    {
        return p->sys.A2 != nullptr;
    }
    int evaluate_candidate_jac(const double *scale, double *cost, int *fail_mask) override
    {
        lm_system &S = p->sys;
        S.swap_sets();
        ochip_relaxg_problem *const prob = p;
        before_wait = [&S, scale, prob]() { lm_launch_diag(S, scale, prob->fail_ranks, (int)prob->shard_world, 1, prob->dev.fail); };
        mail_by_diag = true;
        const int rc = evaluate(true, 1, cost);
        mail_by_diag = false;
        before_wait = nullptr;
        S.swap_sets();
        *fail_mask = rc > 0 ? 1 : 0;
        return rc < 0 ? rc : OCHIP_OK;
    }
    void accept_swap() override
    {
        launch_accept();
    }
    void launch_accept() override
    {
        const uint32_t m = std::max<uint32_t>(std::max(p->n_cams * 4, p->n_verts), 8);
        hipLaunchKernelGGL(general_accept_kernel, dim3((m + 255) / 256), dim3(256), 0, p->ctx->stream, p->dev);
    }
    void launch_normalize() override
    {
        if (p->n_cams)
            hipLaunchKernelGGL(general_normalize_kernel, dim3((p->n_cams + 255) / 256), dim3(256), 0, p->ctx->stream, p->dev,
                               p->cam_optimize_dev);
    }
    int x_norm(double *out) override
    {
        ochip_ctx *ctx = p->ctx;
        std::vector<double> q((size_t)p->n_cams * 4), z(p->n_verts);
        double m[8], s = 0;
        if (p->n_cams)
            OCHIP_HIP(ctx, hipMemcpy(q.data(), p->dev.cam_q, q.size() * 8, hipMemcpyDeviceToHost));
        if (p->n_verts)
            OCHIP_HIP(ctx, hipMemcpy(z.data(), p->dev.vert_z, z.size() * 8, hipMemcpyDeviceToHost));
        OCHIP_HIP(ctx, hipMemcpy(m, p->dev.model, 64, hipMemcpyDeviceToHost));
        for (uint32_t c = 0; c < p->n_cams; c++)
            if (p->var_t[c] >= 0)
                for (int k = 0; k < 4; k++)
                    s += q[c * 4 + k] * q[c * 4 + k];
        for (uint32_t v = 0; v < p->n_verts; v++)
            if (p->var_t[p->n_cams + v] >= 0)
                s += z[v] * z[v];
        const uint32_t vf = p->n_cams + p->n_verts;
        if (p->var_t[vf] >= 0)
            s += m[0] * m[0];
        if (p->var_t[vf + 1] >= 0)
            s += m[1] * m[1] + m[2] * m[2];
        if (p->var_t[vf + 2] >= 0)
            s += m[3] * m[3] + m[4] * m[4] + m[5] * m[5];
        *out = std::sqrt(s);
        return OCHIP_OK;
    }
    int num_residual_blocks() override
    {
        return (int)p->n_rec;
    }
    bool is_constrained() override
    {
        return p->var_t[p->n_cams + p->n_verts] >= 0;
    }
};

} // namespace

extern "C"
{

int ochip_relaxg_problem_create(ochip_ctx *ctx, const ochip_relaxg_desc *d, ochip_relaxg_problem **out)
{
    if (!ctx || !d || !out)
        return OCHIP_EINVAL;
    *out = nullptr;
    OCHIP_HIP(ctx, hipSetDevice(ctx->device));
    for (uint32_t b = 0; b < d->n_blocks; b++)
    {
        const int N = d->blk_n[b];
        if (N < 2 || N > 5 || d->blk_ray_off[b + 1] - d->blk_ray_off[b] != (uint32_t)N)
            return ochip_fail(ctx, OCHIP_EINVAL, "residual block %u: %d rays", b, N);
        for (uint32_t r = d->blk_ray_off[b]; r < d->blk_ray_off[b + 1]; r++)
        {
            if (d->ray_cam[r] >= d->n_cams)
                return ochip_fail(ctx, OCHIP_EINVAL, "residual block %u has a bad camera index", b);
            for (uint32_t r2 = d->blk_ray_off[b]; r2 < r; r2++)
                if (d->ray_cam[r2] == d->ray_cam[r])
                    return ochip_fail(ctx, OCHIP_EINVAL, "residual block %u names a camera twice", b);
        }
        for (int j = 0; j < 3; j++)
            if (d->blk_tri[3 * (size_t)b + j] >= d->n_verts)
                return ochip_fail(ctx, OCHIP_EINVAL, "residual block %u has a bad vertex index", b);
        if (d->blk_intr && d->blk_intr[b] && !d->ray_px)
            return ochip_fail(ctx, OCHIP_EINVAL, "intrinsics blocks need ray_px");
    }
    auto *p = new (std::nothrow) ochip_relaxg_problem();
    if (!p)
        return ochip_fail(ctx, OCHIP_ENOMEM, "host allocation failed");
    p->ctx = ctx;
    const uint32_t nc = d->n_cams, nv = d->n_verts, vf = nc + nv;
    p->n_cams = nc;
    p->n_verts = nv;
    p->n_vars = vf + 3;
    p->n_blocks = d->n_blocks;
    p->cam_optimize.assign(d->cam_optimize, d->cam_optimize + nc);
    p->vert_optimize.assign(d->vert_optimize, d->vert_optimize + nv);
    p->opt_f = d->opt_focal;
    p->opt_pp = d->opt_principal;
    p->n_k_free = std::min<uint8_t>(d->n_radial_free, 3);
    p->cam_xy.resize(2 * (size_t)nc);
    for (uint32_t c = 0; c < nc; c++)
    {
        p->cam_xy[2 * c] = d->cam_pos[3 * c];
        p->cam_xy[2 * c + 1] = d->cam_pos[3 * c + 1];
    }
    p->vert_xy.assign(d->vert_xy, d->vert_xy + 2 * (size_t)nv);

    // ray blocks sorted by type (stable: the order inside a type is the caller's)
    std::vector<uint32_t> order(d->n_blocks);
    std::iota(order.begin(), order.end(), 0u);
    auto type_of = [&](uint32_t b) { return (int)d->blk_n[b] | ((d->blk_intr && d->blk_intr[b]) ? T_INTR : 0); };
    std::stable_sort(order.begin(), order.end(), [&](uint32_t a, uint32_t b) { return type_of(a) < type_of(b); });
    std::vector<uint32_t> blk_ray_off(d->n_blocks + 1, 0), ray_cam, blk_tri(3 * (size_t)d->n_blocks);
    std::vector<double> ray_dir, ray_px;
    const uint32_t n_rays = d->n_blocks ? d->blk_ray_off[d->n_blocks] : 0;
    ray_cam.reserve(n_rays);
    ray_dir.reserve(3 * (size_t)n_rays);
    ray_px.reserve(2 * (size_t)n_rays);
    const uint32_t n_anchor = d->anchor_weight != 0.0 ? nv : 0;
    const uint32_t n_mono = d->mono_observations > 0 ? 1 : 0;
    // sharded evaluation: the ray blocks (in their type-sorted order) are cut into `world` equal runs of `chunk` records;
    // the record list is padded to world * chunk entries before the priors, which every rank evaluates itself
    const uint32_t world = std::max<uint32_t>(1, d->shard_world);
    if (d->shard_rank >= world || world > (uint32_t)lm_system::BOX_MAX_RANKS)
    {
        // (the ranks' failure flags travel in the solver's page-locked block: BOX_MAX_RANKS of them - four 8-GPU nodes)
        delete p;
        return ochip_fail(ctx, OCHIP_EINVAL, "bad shard (rank %u of %u; at most %d ranks)", d->shard_rank, world, lm_system::BOX_MAX_RANKS);
    }
    const uint32_t chunk = std::max<uint32_t>(1, (d->n_blocks + world - 1) / world);
    const uint32_t n_ray_pad = world > 1 ? world * chunk : d->n_blocks;
    p->shard_rank = d->shard_rank;
    p->shard_world = world;
    p->shard_chunk = chunk;
    p->shard_lo = world > 1 ? std::min(d->shard_rank * chunk, d->n_blocks) : 0;
    p->shard_hi = world > 1 ? std::min((d->shard_rank + 1) * chunk, d->n_blocks) : d->n_blocks;
    p->n_rec = n_ray_pad + d->n_down + d->n_diff + n_anchor + d->n_smooth + n_mono + d->n_rel;
    p->rec_type.resize(p->n_rec);
    p->rec_var.assign((size_t)p->n_rec * MAXV, 0);
    std::vector<uint64_t> rec_off(p->n_rec + 1, 0);
    for (uint32_t i = 0; i < d->n_blocks; i++)
    {
        const uint32_t b = order[i];
        const int type = type_of(b), N = d->blk_n[b];
        if (p->ranges.empty() || p->ranges.back().type != type)
            p->ranges.push_back({type, i, 0});
        p->ranges.back().count++;
        p->rec_type[i] = (uint8_t)type;
        blk_ray_off[i + 1] = blk_ray_off[i] + N;
        for (uint32_t r = d->blk_ray_off[b]; r < d->blk_ray_off[b + 1]; r++)
        {
            ray_cam.push_back(d->ray_cam[r]);
            for (int k = 0; k < 3; k++)
                ray_dir.push_back(d->ray_dir ? d->ray_dir[3 * (size_t)r + k] : NAN);
            for (int k = 0; k < 2; k++)
                ray_px.push_back(d->ray_px ? d->ray_px[2 * (size_t)r + k] : NAN);
            p->rec_var[(size_t)i * MAXV + (r - d->blk_ray_off[b])] = d->ray_cam[r];
        }
        for (int j = 0; j < 3; j++)
        {
            blk_tri[3 * (size_t)i + j] = d->blk_tri[3 * (size_t)b + j];
            p->rec_var[(size_t)i * MAXV + N + j] = nc + d->blk_tri[3 * (size_t)b + j];
        }
        if (type & T_INTR)
        {
            p->has_intr_blocks = true;
            for (int k = 0; k < 3; k++)
                p->rec_var[(size_t)i * MAXV + N + 3 + k] = vf + k;
        }
    }
    {
        uint32_t r = d->n_blocks;
        for (; r < n_ray_pad; r++)
            p->rec_type[r] = T_NULL;
        for (uint32_t i = 0; i < d->n_down; i++, r++)
        {
            if (d->down_cam[i] >= nc)
            {
                delete p;
                return ochip_fail(ctx, OCHIP_EINVAL, "downward prior %u has a bad camera index", i);
            }
            p->rec_type[r] = T_DOWN;
            p->rec_var[(size_t)r * MAXV] = d->down_cam[i];
        }
        for (uint32_t i = 0; i < d->n_diff; i++, r++)
        {
            p->rec_type[r] = T_DIFF;
            for (int k = 0; k < 2; k++)
            {
                if (d->diff_v[2 * i + k] >= nv)
                {
                    delete p;
                    return ochip_fail(ctx, OCHIP_EINVAL, "height-difference prior %u has a bad vertex index", i);
                }
                p->rec_var[(size_t)r * MAXV + k] = nc + d->diff_v[2 * i + k];
            }
        }
        for (uint32_t i = 0; i < n_anchor; i++, r++)
        {
            p->rec_type[r] = T_ANCHOR;
            p->rec_var[(size_t)r * MAXV] = nc + i;
        }
        for (uint32_t i = 0; i < d->n_smooth; i++, r++)
        {
            p->rec_type[r] = T_SMOOTH;
            for (int k = 0; k < 4; k++)
            {
                if (d->smooth_v[4 * i + k] >= nv)
                {
                    delete p;
                    return ochip_fail(ctx, OCHIP_EINVAL, "smoothness prior %u has a bad vertex index", i);
                }
                p->rec_var[(size_t)r * MAXV + k] = nc + d->smooth_v[4 * i + k];
            }
        }
        for (uint32_t i = 0; i < n_mono; i++, r++)
        {
            p->rec_type[r] = T_MONO;
            p->rec_var[(size_t)r * MAXV] = vf + 2;
        }
        for (uint32_t i = 0; i < d->n_rel; i++, r++)
        {
            p->rec_type[r] = T_REL;
            for (int k = 0; k < 2; k++)
            {
                if (d->rel_cam[2 * i + k] >= nc || d->rel_cam[2 * i] == d->rel_cam[2 * i + 1])
                {
                    delete p;
                    return ochip_fail(ctx, OCHIP_EINVAL, "relation block %u has bad cameras", i);
                }
                p->rec_var[(size_t)r * MAXV + k] = d->rel_cam[2 * i + k];
            }
        }
    }
    auto rec_size = [&](uint32_t r) {
        const int dd = rec_dim(p->rec_type[r]);
        return (uint64_t)(dd * (dd + 1) / 2 + dd);
    };
    if (world == 1)
        for (uint32_t r = 0; r < p->n_rec; r++)
            rec_off[r + 1] = rec_off[r] + rec_size(r);
    else
    {
        // rank k's records start at k * stride (stride = the largest run), so that the exchange is a plain in-place
        // all-gather of `stride` doubles per rank; the priors' records follow the last run
        uint64_t stride = 0;
        for (uint32_t k = 0; k < world; k++)
        {
            uint64_t run = 0;
            for (uint32_t r = k * chunk; r < std::min((k + 1) * chunk, d->n_blocks); r++)
                run += rec_size(r);
            stride = std::max(stride, run);
        }
        p->shard_stride = stride;
        for (uint32_t k = 0; k < world; k++)
        {
            uint64_t at = (uint64_t)k * stride;
            for (uint32_t r = k * chunk; r < (k + 1) * chunk; r++)
            {
                rec_off[r] = at;
                at += rec_size(r); // (padding records: size 0)
            }
        }
        rec_off[n_ray_pad] = (uint64_t)world * stride;
        for (uint32_t r = n_ray_pad; r < p->n_rec; r++)
            rec_off[r + 1] = rec_off[r] + rec_size(r);
    }
    // CSR unknown group -> records, in record order (the fixed summation order of the assembly)
    p->var_rec_off.assign(p->n_vars + 1, 0);
    for (uint32_t r = 0; r < p->n_rec; r++)
        for (int s = 0; s < rec_nvars(p->rec_type[r]); s++)
            p->var_rec_off[p->rec_var[(size_t)r * MAXV + s] + 1]++;
    for (uint32_t u = 0; u < p->n_vars; u++)
        p->var_rec_off[u + 1] += p->var_rec_off[u];
    p->var_rec.resize(p->var_rec_off[p->n_vars]);
    {
        std::vector<uint32_t> fill(p->var_rec_off.begin(), p->var_rec_off.end() - 1);
        for (uint32_t r = 0; r < p->n_rec; r++)
            for (int s = 0; s < rec_nvars(p->rec_type[r]); s++)
                p->var_rec[fill[p->rec_var[(size_t)r * MAXV + s]]++] = (r << 4) | (uint32_t)s;
    }
    if (p->n_rec >= (1u << 28))
    {
        delete p;
        return ochip_fail(ctx, OCHIP_EINVAL, "too many residual blocks");
    }

    g_dev &D = p->dev;
    D.n_cams = nc;
    D.n_verts = nv;
    D.n_rec = p->n_rec;
    D.n_down = d->n_down;
    D.n_diff = d->n_diff;
    D.n_anchor = n_anchor;
    D.n_smooth = d->n_smooth;
    D.n_mono = n_mono;
    D.n_rel = d->n_rel;
    D.rel_huber_a = d->rel_huber_a;
    D.prior_base = n_ray_pad;
    D.down_w = d->down_weight;
    D.diff_w = d->diff_weight;
    D.anchor_w = d->anchor_weight;
    D.smooth_w = d->smooth_weight;
    D.mono_w = std::sqrt(d->mono_observations / 10.0);
    D.mono_rmax = d->mono_r_max;
    D.huber_a = d->huber_a;
    D.f_lo = d->focal_lo;
    D.f_hi = d->focal_hi;
    D.n_k_free = p->n_k_free;
    int rc = OCHIP_OK;
    auto chk = [&](int r) {
        if (rc == OCHIP_OK)
            rc = r;
    };
    chk(up(p, &D.cam_pos, d->cam_pos, (size_t)nc * 3));
    chk(up(p, &D.cam_q, d->cam_q, (size_t)nc * 4));
    chk(up(p, &D.cam_q2, d->cam_q, (size_t)nc * 4));
    chk(up(p, &D.vert_xy, d->vert_xy, (size_t)nv * 2));
    chk(up(p, &D.vert_z, d->vert_z, (size_t)nv));
    chk(up(p, &D.vert_z2, d->vert_z, (size_t)nv));
    chk(up(p, &D.vert_z0, d->vert_z, (size_t)nv));
    chk(up(p, &D.model, d->model, 8));
    chk(up(p, &D.model2, d->model, 8));
    chk(up<int32_t>(p, &D.var_t, nullptr, p->n_vars));
    chk(up<uint8_t>(p, &D.var_ts, nullptr, p->n_vars));
    chk(up(p, &D.blk_ray_off, blk_ray_off));
    chk(up(p, &D.ray_cam, ray_cam));
    chk(up(p, &D.blk_tri, blk_tri));
    chk(up(p, &D.ray_dir, ray_dir));
    chk(up(p, &D.ray_px, ray_px));
    chk(up(p, &D.rec_type, p->rec_type));
    chk(up(p, &D.rec_off, rec_off));
    chk(up(p, &D.rec_var, p->rec_var));
    chk(up<double>(p, &D.rec_data, nullptr, (size_t)rec_off[p->n_rec]));
    chk(up<double>(p, &D.rec_cost, nullptr, p->n_rec));
    chk(up<int32_t>(p, &p->fail_ranks, nullptr, world));
    chk(up(p, &D.down_cam, d->down_cam, d->n_down));
    chk(up(p, &D.diff_v, d->diff_v, (size_t)d->n_diff * 2));
    chk(up(p, &D.smooth_v, d->smooth_v, (size_t)d->n_smooth * 4));
    chk(up(p, &D.rel_cam, d->rel_cam, (size_t)d->n_rel * 2));
    chk(up(p, &D.rel_pose, d->rel_pose, (size_t)d->n_rel * 32));
    chk(up(p, &p->var_rec_dev, p->var_rec));
    chk(up(p, &p->cam_optimize_dev, p->cam_optimize));
    if (rc == OCHIP_OK)
    {
        D.fail = p->fail_ranks + p->shard_rank;
        if (hipMemset(p->fail_ranks, 0, (size_t)world * 4) != hipSuccess || hipMemset(D.rec_cost, 0, (size_t)p->n_rec * 8) != hipSuccess)
            rc = ochip_fail(ctx, OCHIP_EHIP, "hipMemset failed (relax records)");
    }
    if (rc == OCHIP_OK)
        rc = assign(p);
    if (rc != OCHIP_OK)
    {
        ochip_relaxg_problem_destroy(p);
        return rc;
    }
    *out = p;
    return OCHIP_OK;
}

int ochip_relaxg_set_exchange(ochip_relaxg_problem *p, ochip_relax_exchange_fn fn, void *user)
{
    if (!p)
        return OCHIP_EINVAL;
    if (p->shard_world > 1 && !fn)
        return ochip_fail(p->ctx, OCHIP_EINVAL, "a problem sharded over %u ranks needs an exchange function", p->shard_world);
    p->exchange = fn;
    p->exchange_user = user;
    return OCHIP_OK;
}

void ochip_relaxg_problem_destroy(ochip_relaxg_problem *p)
{
    if (!p)
        return;
    (void)hipSetDevice(p->ctx->device);
    (void)ochip_stream_wait(p->ctx, p->ctx->stream);
    for (auto &a : p->allocs)
        ochip_pool_put(p->ctx, a.first, a.second);
    delete p;
}

int ochip_relaxg_set_structure_only(ochip_relaxg_problem *p, int on)
{
    if (!p)
        return OCHIP_EINVAL;
    OCHIP_HIP(p->ctx, hipSetDevice(p->ctx->device));
    p->structure_only = on != 0;
    return assign(p);
}

int ochip_relaxg_get_state(ochip_relaxg_problem *p, double *cam_q, double *vert_z, double *model)
{
    if (!p)
        return OCHIP_EINVAL;
    ochip_ctx *ctx = p->ctx;
    OCHIP_HIP(ctx, ochip_stream_wait(ctx, ctx->stream));
    if (cam_q && p->n_cams)
        OCHIP_HIP(ctx, hipMemcpy(cam_q, p->dev.cam_q, (size_t)p->n_cams * 32, hipMemcpyDeviceToHost));
    if (vert_z && p->n_verts)
        OCHIP_HIP(ctx, hipMemcpy(vert_z, p->dev.vert_z, (size_t)p->n_verts * 8, hipMemcpyDeviceToHost));
    if (model)
        OCHIP_HIP(ctx, hipMemcpy(model, p->dev.model, 64, hipMemcpyDeviceToHost));
    return OCHIP_OK;
}

int ochip_relaxg_evaluate(ochip_relaxg_problem *p, double *cost, int *n_out, double *JtJ, double *Jtr, int32_t *order_out)
{
    if (!p || !cost)
        return OCHIP_EINVAL;
    ochip_ctx *ctx = p->ctx;
    OCHIP_HIP(ctx, hipSetDevice(ctx->device));
    general_model model(p);
    const int n = p->n_tangent;
    if (n_out)
        *n_out = n;
    if (order_out)
        for (uint32_t u = 0; u < p->n_vars; u++)
            order_out[u] = p->var_t[u];
    const int rc = model.evaluate(JtJ != nullptr || Jtr != nullptr, 0, cost);
    if (rc < 0)
        return rc;
    if (JtJ && n)
    {
        const int drc = lm_download_dense(p->sys, JtJ);
        if (drc)
            return drc;
    }
    if (Jtr && n)
        OCHIP_HIP(ctx, hipMemcpy(Jtr, p->sys.g, (size_t)n * 8, hipMemcpyDeviceToHost));
    return rc;
}

int ochip_relaxg_solve(ochip_relaxg_problem *p, const ochip_relax_options *opt, ochip_relax_summary *sum)
{
    if (!p || !opt || !sum)
        return OCHIP_EINVAL;
    ochip_ctx *ctx = p->ctx;
    OCHIP_HIP(ctx, hipSetDevice(ctx->device));
    *sum = ochip_relax_summary{};
    general_model model(p);
    sum->num_parameters = p->n_tangent - p->n_padding;
    sum->num_residual_blocks = model.num_residual_blocks();
    if (p->n_rec == 0)
    {
        sum->termination = OCHIP_RELAX_NO_PARAMETERS; // RelaxProblem::solve returns before Solve (:1398-1402)
        return OCHIP_OK;
    }
    if (p->n_tangent == 0)
    {
        sum->termination = OCHIP_RELAX_NO_PARAMETERS;
        model.launch_normalize();
        OCHIP_HIP(ctx, ochip_stream_wait(ctx, ctx->stream));
        return OCHIP_OK;
    }
    return lm_solve(p->sys, model, opt, sum);
}

} // extern "C"
