// libochip.so — relax (bundle adjustment) for the ground-plane problem on the device (gfx950).
//
// Replaces what ceres::Solver::Solve does for RelaxProblem::setupGroundPlaneProblem
// (src/relax/relax_problem.cpp:61-81,1390-1420): evaluation of the 2-ray plane-intersection residual
// blocks (include/opencalibration/relax/relax_cost_function.hpp:601-684) and downward priors (:21-49)
// with forward-mode derivatives in the quaternion tangent space, Huber corrector, assembly of the
// normal equations, and the Levenberg-Marquardt trust-region loop with a dense
// Cholesky solve of (J'J + D'D) y = J'r.
//
// Data-parallel structure (DESIGN.md "relax kernels"):
//   * residual blocks are sorted by unordered camera pair; one wavefront owns one pair segment,
//     evaluates its blocks lane-strided, keeps the 9x9 (J'J), 9 (J'r) and cost partial sums in
//     registers and reduces them with a fixed shuffle tree -> one 55-double record per pair.  No
//     atomics: the sums are bitwise reproducible run to run.
//   * a per-camera gather (CSR camera -> pairs) writes the 3x3 diagonal / camera-plane blocks and the
//     prior; a per-pair scatter writes the off-diagonal 3x3 blocks; one workgroup reduces the plane
//     block, gradient and cost.  Streams observation arrays once per evaluation: HBM bound.
//   * the reduced system is dense (3 dof per camera + 3 plane heights); blocked right-looking Cholesky
//     (64-wide panels: diagonal factor in LDS, row-parallel panel solve, 64x64 tiled trailing update).
#include "ctx.hpp"
#include "dual.hpp"
#include "relax_lm.hpp"
#include "relax_lm_back.hpp"

#include <algorithm>
#include <cmath>
#include <map>
#include <thread>
#include <vector>

using namespace ochip;

#include "relax_plane_functor.hpp"

namespace
{

constexpr int W = 64;
constexpr int NB = LM_NB;
constexpr int ACC = 55; // 45 upper-triangular entries of the 9x9 [p|q|z] block + 9 gradient + cost

__host__ __device__ inline int tri(int i, int j) // i <= j, 9x9 upper triangle
{
    return i * 9 - i * (i - 1) / 2 + (j - i);
}

struct relax_dev
{
    // cameras
    uint32_t n_cams;
    double *cam_pos;  // [n][3]
    double *cam_q;    // [n][4] current state (x y z w)
    double *cam_q2;   // candidate state
    int32_t *cam_t;   // tangent offset or -1
    double z[2][3];   // unused on device (kept on host)
    // plane
    double *plane;    // [0..5] xy of 3 corners, [6..8] and [9..11]: the heights of the current and the candidate state
    int zcur;         // where the current state's heights start (6 or 9); the candidate's start at 15 - zcur
    int32_t *z_t;     // [3]
    // blocks sorted by pair
    uint32_t n_blocks, n_pairs, n_prior;
    uint32_t *blk_a, *blk_b; // camera indices
    double *blk_rays;        // [n][6] camera-frame unit rays a, b
    uint32_t *pair_off;      // [n_pairs+1]
    uint32_t *pair_p, *pair_q;
    uint32_t *prior_cam;
    // CSR camera -> (pair, role)
    uint32_t *cam_pair_off, *cam_pair_idx; // idx = pair*2 + role (0: camera is p, 1: camera is q)
    // outputs
    double *pair_acc;  // [n_pairs (padded to world * chunk when sharded)][ACC]
    double *pair_cost; // [same]
    int32_t *fail;     // this rank's flag inside fail_ranks[world]
    uint32_t pair_lo;  // first pair this rank evaluates (0 unless sharded)
    double huber_a, prior_weight;
};

template <bool WITH_JAC>
__global__ __launch_bounds__(W) void relax_pair_eval_kernel(relax_dev P, int which_state)
{
    // One wavefront per camera pair, one lane per residual block (a pair has ~50).  With the Jacobian, a lane used to keep
    // its block's 6 x 9 Jacobian AND 54 running sums of J'J / J'r in registers (218 VGPRs before the functor's own
    // temporaries: 896 bytes of scratch per lane, one wavefront per SIMD), followed by a 54-value shuffle tree.  Now a
    // lane leaves its scaled Jacobian and residuals in LDS (role-normalised columns [p | q | z]) and 54 lanes each form
    // ONE entry of J'J or J'r, adding the blocks in order; the entries go out as one coalesced store.
    constexpr int JP = 61; // doubles per lane in LDS: 54 of J, 6 of r, 1 of padding (bank spread)
    __shared__ double Jl[WITH_JAC ? W : 1][JP];
    const int lane = threadIdx.x;
    const uint32_t pair = P.pair_lo + blockIdx.x;
    const uint32_t b0 = P.pair_off[pair], b1 = P.pair_off[pair + 1];
    const uint32_t p = P.pair_p[pair];
    const double *Q = which_state ? P.cam_q2 : P.cam_q;
    const double *Z = P.plane + (which_state ? 15 - P.zcur : P.zcur);
    const double a2 = P.huber_a * P.huber_a;

    // this lane's entry of the pair's 9 x 9 block (upper triangle, tri(i, j)) or of its gradient (45 + i)
    int ei = 0, ej = 0;
    if (WITH_JAC)
    {
        if (lane < 45)
        {
            int rem = lane;
            while (rem >= 9 - ei)
            {
                rem -= 9 - ei;
                ei++;
            }
            ej = ei + rem;
        }
        else
            ei = lane - 45;
    }
    double entry = 0;
    double cost = 0;
    bool failed = false, failed_jac = false; // a residual / a derivative that is not finite

    for (uint32_t first = b0; first < b1; first += W)
    {
        const uint32_t blk = first + lane;
        if (blk < b1)
        {
            const uint32_t ca = P.blk_a[blk], cb = P.blk_b[blk];
            const double *rays = P.blk_rays + (size_t)blk * 6;
            const double *la = P.cam_pos + (size_t)ca * 3, *lb = P.cam_pos + (size_t)cb * 3;
            const double *qa = Q + (size_t)ca * 4, *qb = Q + (size_t)cb * 4;
            double r[6];
            {
                functor_io<double> in;
                for (int k = 0; k < 4; k++)
                {
                    in.qa[k] = qa[k];
                    in.qb[k] = qb[k];
                }
                for (int k = 0; k < 3; k++)
                    in.z[k] = Z[k];
                if (!plane_intersection_residuals<double>(in, la, lb, rays, P.plane, r))
                    failed = true;
            }
            double s = 0;
            for (int k = 0; k < 6; k++)
            {
                s += r[k] * r[k];
                if (!(r[k] - r[k] == 0.0))
                    failed = true;
            }
            // Huber + Triggs corrector (rho'' <= 0 for Huber: plain sqrt(rho') scaling)
            double sqrt_rho1 = 1.0, c = 0.5 * s;
            if (s > a2)
            {
                const double rn = sqrt(s);
                const double rho1 = fmax(2.2250738585072014e-308, P.huber_a / rn);
                sqrt_rho1 = sqrt(rho1);
                c = 0.5 * (2.0 * P.huber_a * rn - a2);
            }
            cost += c;
            if (WITH_JAC)
            {
                // one pass per parameter block, the other two as plain doubles (plane_intersection_residuals_mixed);
                // role normalisation: the pair's columns are [p | q | z], this block's cameras may be (q, p)
                const int first_col = ca != p ? 3 : 0, second_col = ca != p ? 0 : 3;
                double *mine = Jl[lane];
                D3 rd[6], seeded[4];
                auto take = [&](int col0) {
                    for (int k = 0; k < 6; k++)
                        for (int cidx = 0; cidx < 3; cidx++)
                        {
                            const double v = rd[k].v[cidx] * sqrt_rho1;
                            mine[k * 9 + col0 + cidx] = v;
                            if (!(v - v == 0.0))
                                failed_jac = true;
                        }
                };
                seed_quat(qa, seeded);
                plane_intersection_residuals_mixed(seeded, qb, Z, la, lb, rays, P.plane, rd);
                take(first_col);
                seed_quat(qb, seeded);
                plane_intersection_residuals_mixed(qa, seeded, Z, la, lb, rays, P.plane, rd);
                take(second_col);
                D3 zs[3];
                for (int k = 0; k < 3; k++)
                {
                    zs[k] = D3(Z[k]);
                    zs[k].v[k] = 1.0;
                }
                plane_intersection_residuals_mixed(qa, qb, zs, la, lb, rays, P.plane, rd);
                take(6);
                for (int k = 0; k < 6; k++)
                    mine[54 + k] = r[k] * sqrt_rho1;
            }
        }
        if (WITH_JAC)
        {
            __syncthreads();
            const int blocks = (int)min((uint32_t)W, b1 - first);
            if (lane < 54)
            {
                const int cb2 = lane < 45 ? ej : 54; // second factor: column ej of J, or the residual
                for (int b = 0; b < blocks; b++)
                {
                    const double *jb = Jl[b];
                    double m = 0;
                    for (int k = 0; k < 6; k++)
                        m += jb[k * 9 + ei] * (lane < 45 ? jb[k * 9 + cb2] : jb[54 + k]);
                    entry += m;
                }
            }
            __syncthreads();
        }
    }
    // fixed shuffle tree: bitwise reproducible
    for (int off = 32; off >= 1; off >>= 1)
        cost += __shfl_xor(cost, off);
    const int fail_bits = (__ballot(failed) ? 1 : 0) | (__ballot(failed_jac) ? 2 : 0);
    if (fail_bits && lane == 0)
        atomicOr(P.fail, fail_bits);
    if (lane == 0)
        P.pair_cost[pair] = cost;
    if (WITH_JAC)
    {
        double *o = P.pair_acc + (size_t)pair * ACC;
        if (lane < 54)
            o[lane] = entry;
        if (lane == 0)
            o[54] = cost;
    }
}

// Per-camera gather: diagonal 3x3, camera-plane 3x3, gradient; plus the prior.  One wavefront per camera, one lane per
// pair the camera is in (~18): one trip to memory for all of them instead of one per pair, then a fixed shuffle tree
// (a thread per camera walking its pairs took 26 us of an LM iteration).
__device__ __forceinline__ void scatter_cam(const relax_dev &P, const lm_matrix &A, double *g, int n, const uint8_t *cam_has_prior,
                                            uint32_t c, int lane)
{
    if (c >= P.n_cams)
        return;
    const int tc = P.cam_t[c];
    if (tc < 0)
        return;
    double v[18]; // D (6), CZ (9), G (3)
    for (int k = 0; k < 18; k++)
        v[k] = 0;
    for (uint32_t e = P.cam_pair_off[c] + lane; e < P.cam_pair_off[c + 1]; e += W)
    {
        const uint32_t idx = P.cam_pair_idx[e];
        const double *a = P.pair_acc + (size_t)(idx >> 1) * ACC;
        const int o = (idx & 1) ? 3 : 0;
        int k = 0;
        for (int i = 0; i < 3; i++)
            for (int j = i; j < 3; j++)
                v[k++] += a[tri(o + i, o + j)];
        for (int i = 0; i < 3; i++)
            for (int j = 0; j < 3; j++)
                v[6 + i * 3 + j] += a[tri(o + i, 6 + j)];
        for (int i = 0; i < 3; i++)
            v[15 + i] += a[45 + o + i];
    }
    for (int off = 32; off >= 1; off >>= 1)
        for (int k = 0; k < 18; k++)
            v[k] += __shfl_xor(v[k], off);
    if (lane != 0)
        return;
    double *D = v, *CZ = v + 6, *G = v + 15;
    if (cam_has_prior[c])
    {
        double r, j3[3];
        downward_prior(P.cam_q + (size_t)c * 4, P.prior_weight, &r, j3);
        int k = 0;
        for (int i = 0; i < 3; i++)
        {
            for (int j = i; j < 3; j++)
                D[k++] += j3[i] * j3[j];
            G[i] += j3[i] * r;
        }
    }
    int k = 0;
    for (int i = 0; i < 3; i++)
        for (int j = i; j < 3; j++)
        {
            A.tiles[lm_at(A, tc + j, tc + i)] = D[k]; // (lower triangle only: relax_lm.hpp)
            k++;
        }
    for (int i = 0; i < 3; i++)
    {
        g[tc + i] = G[i];
        for (int j = 0; j < 3; j++)
        {
            const int tz = P.z_t[j];
            if (tz >= 0)
            {
                A.tiles[lm_at(A, tz, tc + i)] = CZ[i * 3 + j]; // (the plane unknowns are the tail: tz > tc)
            }
        }
    }
}

// Per-pair scatter of the off-diagonal camera-camera block.  One thread per pair.
__device__ __forceinline__ void scatter_pair(const relax_dev &P, const lm_matrix &A, int n, uint32_t pr)
{
    if (pr >= P.n_pairs)
        return;
    const int tp = P.cam_t[P.pair_p[pr]], tq = P.cam_t[P.pair_q[pr]];
    if (tp < 0 || tq < 0)
        return;
    const double *a = P.pair_acc + (size_t)pr * ACC;
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++)
        {
            const double v = a[tri(i, 3 + j)];
            if (tp > tq)
                A.tiles[lm_at(A, tp + i, tq + j)] = v;
            else
                A.tiles[lm_at(A, tq + j, tp + i)] = v;
        }
}

// Plane-plane block, plane gradient, total cost (pairs + priors); scal[0] = cost.  REDUCE_GROUPS workgroups each sum a
// fixed share of the pairs and cameras (one workgroup took 33 us of an iteration: 35 dependent trips to memory per thread),
// the one that finishes last adds the groups' sums in group order - the same result whoever that is.
constexpr int REDUCE_GROUPS = 32;
__global__ __launch_bounds__(256) void relax_reduce_plane_kernel(relax_dev P, lm_matrix A, double *g, int n,
                                                                 const uint8_t *cam_has_prior, double *scal,
                                                                 int with_jac, int which_state, double *partials /*[groups][10]*/,
                                                                 unsigned int *arrived, lm_mail mail, const double *diag_scale,
                                                                 double *diagonal, uint32_t cam_blocks, uint32_t scatter_blocks)
{
    // The last `scatter_blocks` workgroups of the launch do what relax_scatter_kernel does (cameras, then pairs): one launch
    // per Jacobian evaluation less.  Whichever workgroup of the whole launch arrives last finishes the evaluation.
    __shared__ int s_last;
    __shared__ double wsum[4][10];
    const int t = threadIdx.x, groups = (int)(gridDim.x - scatter_blocks), b = blockIdx.x;
    if (b >= groups)
    {
        const uint32_t sb = (uint32_t)(b - groups);
        if (sb < cam_blocks)
            scatter_cam(P, A, g, n, cam_has_prior, sb * (256 / W) + t / W, t % W);
        else
            scatter_pair(P, A, n, (sb - cam_blocks) * 256 + t);
        // every wavefront's stores have left it before the barrier in front of the arrival (a workgroup barrier alone does not
        // wait for the other wavefronts' vmcnt: MI355X_MICROARCH.md, producer side of a cross-CU hand-off)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (t == 0)
        {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); // (the compiler may drop the fence's own wait: MI355X_MICROARCH.md, compiler hazard)
            s_last = __hip_atomic_fetch_add(arrived, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT) == gridDim.x - 1;
        }
        __syncthreads();
        if (!s_last)
            return;
    }
    else
    {
        double v[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0}; // zz (6), gz (3), cost
        for (uint32_t pr = b * 256 + t; pr < P.n_pairs; pr += 256 * groups)
        {
            if (with_jac)
            {
                const double *a = P.pair_acc + (size_t)pr * ACC;
                int k = 0;
                for (int i = 0; i < 3; i++)
                    for (int j = i; j < 3; j++)
                        v[k++] += a[tri(6 + i, 6 + j)];
                for (int i = 0; i < 3; i++)
                    v[6 + i] += a[45 + 6 + i];
            }
            v[9] += P.pair_cost[pr];
        }
        const double *Q = which_state ? P.cam_q2 : P.cam_q;
        for (uint32_t c = b * 256 + t; c < P.n_cams; c += 256 * groups)
            if (cam_has_prior[c] && P.cam_t[c] >= 0) // priors of constant cameras are fixed cost (not in the reduced program)
            {
                double r, j3[3];
                downward_prior(Q + (size_t)c * 4, P.prior_weight, &r, j3);
                v[9] += 0.5 * r * r;
            }
        // within the workgroup: a fixed shuffle tree per wavefront, then the four wavefronts' sums in order
        for (int q = with_jac ? 0 : 9; q < 10; q++)
        {
            double x = v[q];
            for (int off = 32; off >= 1; off >>= 1)
                x += __shfl_xor(x, off);
            if ((t & 63) == 0)
                wsum[t >> 6][q] = x;
        }
        __syncthreads();
        if (t < 10)
        {
            const double sum = (with_jac || t == 9) ? ((wsum[0][t] + wsum[1][t]) + wsum[2][t]) + wsum[3][t] : 0.0;
            __hip_atomic_store(&partials[b * 10 + t], sum, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        if (t == 0)
        {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent"); // (lanes 0 .. 9 of this wavefront stored the sums)
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            s_last = __hip_atomic_fetch_add(arrived, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT) == gridDim.x - 1;
        }
        __syncthreads();
        if (!s_last)
            return;
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent"); // (the other workgroups' entries of A and g, their partial sums)
    // the last workgroup: every group's sums in one trip to memory, then ten lanes add them in group order
    __shared__ double part[REDUCE_GROUPS * 10];
    __shared__ double total[10];
    for (int i = t; i < groups * 10; i += 256)
        part[i] = __hip_atomic_load(&partials[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();
    if (t < 10)
    {
        double sum = 0;
        for (int w = 0; w < groups; w++)
            sum += part[w * 10 + t];
        total[t] = sum;
    }
    __syncthreads();
    if (t == 0)
    {
        __hip_atomic_store(arrived, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        scal[0] = total[9];
        if (with_jac)
        {
            int k = 0;
            for (int i = 0; i < 3; i++)
                for (int j = i; j < 3; j++)
                {
                    const int ti = P.z_t[i], tj = P.z_t[j];
                    if (ti >= 0 && tj >= 0)
                    {
                        A.tiles[lm_at(A, ti > tj ? ti : tj, ti > tj ? tj : ti)] = total[k];
                    }
                    k++;
                }
            for (int i = 0; i < 3; i++)
                if (P.z_t[i] >= 0)
                    g[P.z_t[i]] = total[6 + i];
        }
    }
    if (diag_scale) // (uniform) what lm_diag_kernel would do behind this kernel: max |g| and the damping's clamped diagonal
    {
        __shared__ double shd[256];
        __threadfence(); // thread 0's entries of A and g, for the other wavefronts of this workgroup
        __syncthreads();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        const double gmax = lm_diag_pass<256>(A, g, nullptr, n, diag_scale, diagonal, shd);
        if (t == 0)
            scal[4] = gmax;
    }
    if (t == 0)
    {
        // the evaluation's results go to the host block from here (no copies behind this kernel), and this rank's
        // failure flag is clear again for the next evaluation (no memset in front of it)
        lm_mail_post(mail);
        if (mail.box) // (not mailed - no page-locked block -: the host copies the flags behind this kernel and clears them itself)
            *P.fail = 0;
    }
}

// step = -y with (As + D) y = gs; delta = S step; candidate state = x (+) delta; step_norm^2 in ambient space.
// One workgroup.  scal: [2] step_norm^2, [3] x_norm^2 (candidate)
__device__ __forceinline__ void plane_candidate_body(const relax_dev &P, const double *scale, const double *y, double alpha, double *scal)
{
    __shared__ double sh[LM_TG];
    const int t = threadIdx.x;
    // candidate state
    double sn = 0, xn = 0;
    for (uint32_t c = t; c < P.n_cams; c += LM_TG)
    {
        const int tc = P.cam_t[c];
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
    if (t < 3)
    {
        const int tz = P.z_t[t];
        const double z0 = P.plane[P.zcur + t];
        const double z1 = tz >= 0 ? z0 + alpha * (-y[tz] * scale[tz]) : z0;
        P.plane[15 - P.zcur + t] = z1;
        if (tz >= 0)
        {
            sn += (z0 - z1) * (z0 - z1);
            xn += z1 * z1;
        }
    }
    __syncthreads();
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

__global__ __launch_bounds__(LM_TG) void plane_candidate_kernel(relax_dev P, const double *scale, const double *y, double alpha, double *scal)
{
    plane_candidate_body(P, scale, y, alpha, scal);
}

// the backward substitution (relax_lm_back.hpp) with the candidate state as its tail: the workgroup that finishes last has
// the whole step and computes the candidate (alpha = 1) - one launch instead of two
__global__ __launch_bounds__(LM_TG) void plane_back_solve_candidate_kernel(lm_matrix Lm, int n, const double *Linv, double *x, double *work,
                                                                          const int *first_blk, int n_blocks, const int *region, int tb,
                                                                          const double *lm_diag, const double *gs, double *scal,
                                                                          unsigned int *arrived, int x_in_lds, relax_dev P, const double *scale)
{
    back_solve_regions_body(Lm, n, Linv, x, work, first_blk, n_blocks, region, tb, lm_diag, gs, scal, arrived, x_in_lds,
                            [&]() { plane_candidate_body(P, scale, x, 1.0, scal); });
}

__global__ void lm_accept_kernel(relax_dev P)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < P.n_cams * 4)
        P.cam_q[i] = P.cam_q2[i];
    if (i < 3)
        P.plane[P.zcur + i] = P.plane[15 - P.zcur + i];
}

// p.second->orientation.normalize() for every node of _nodes_to_optimize (relax_problem.cpp:1410-1413)
__global__ void normalize_kernel(relax_dev P, const uint8_t *cam_optimize)
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

// ---------------------------------------------------------------------------------------------------
struct ochip_relax_problem
{
    ochip_ctx *ctx = nullptr;
    relax_dev dev{};
    std::vector<std::pair<void *, size_t>> allocs; // device blocks from the context's pool (returned on destroy)
    int n_tangent = 0;
    std::vector<int32_t> cam_t;
    int32_t z_t[3] = {-1, -1, -1};
    std::vector<uint8_t> cam_optimize, cam_has_prior_host;
    uint8_t z_optimize[3] = {1, 1, 1};
    bool cams_frozen = false;
    uint8_t *cam_has_prior = nullptr, *cam_optimize_dev = nullptr;
    double *reduce_partials = nullptr; // relax_reduce_plane_kernel: the groups' sums, then its arrival counter
    unsigned int *reduce_arrived = nullptr;
    lm_system sys; // the reduced normal equations and their block envelope (assign_tangent), relax_lm.hpp
    std::vector<uint32_t> pair_p_h, pair_q_h; // host copies of the camera pairs
    uint32_t n_cams = 0;
    std::vector<uint32_t> cam_pair_count;
    // sharded evaluation (ochip_relax_set_shard): this rank evaluates pairs [pair_lo, pair_hi)
    uint32_t shard_rank = 0, shard_world = 1, shard_chunk = 0, pair_hi = 0;
    ochip_relax_exchange_fn exchange = nullptr;
    void *exchange_user = nullptr;
    int32_t *fail_ranks = nullptr;
};

namespace
{
template <typename T> int dev_upload(ochip_relax_problem *p, T **dst, const T *src, size_t n)
{
    size_t got = 0;
    void *d = ochip_pool_get(p->ctx, (n ? n : 1) * sizeof(T), &got);
    if (!d)
        return ochip_fail(p->ctx, OCHIP_ENOMEM, "device allocation of %zu bytes failed in relax problem", n * sizeof(T));
    p->allocs.emplace_back(d, got);
    if (n && src) // on the context's stream: the device's default stream is a queue shared with every other context
        if (hipMemcpyAsync(d, src, n * sizeof(T), hipMemcpyHostToDevice, p->ctx->stream) != hipSuccess ||
            ochip_stream_wait(p->ctx, p->ctx->stream) != hipSuccess)
            return ochip_fail(p->ctx, OCHIP_EHIP, "hipMemcpy failed in relax problem");
    *dst = (T *)d;
    return OCHIP_OK;
}

int assign_tangent(ochip_relax_problem *p)
{
    int t = 0;
    p->cam_t.assign(p->n_cams, -1);
    std::vector<uint32_t> active; // cameras that get unknowns, in the order of their unknowns
    int n_separators = 0;         // the last cameras of `active` when the camera graph was dissected (below)
    std::vector<int> region_first_block;
    for (uint32_t c = 0; c < p->n_cams; c++)
        if (p->cam_optimize[c] && !p->cams_frozen && p->cam_pair_count[c] + p->cam_has_prior_host[c] > 0)
            active.push_back(c);
    if (active.size() > 2 * (size_t)NB / 3)
    {
        // Reverse Cuthill-McKee over the camera graph (a link = a pair with residual blocks): the block envelope the
        // factorisation works in then does not depend on the order the images happen to arrive in.  Per connected
        // component: start from a pseudo-peripheral camera (two breadth-first sweeps from the lowest-degree one),
        // visit neighbours by increasing degree (ties: camera index), reverse the whole order at the end.
        std::vector<int> slot(p->n_cams, -1);
        for (size_t i = 0; i < active.size(); i++)
            slot[active[i]] = (int)i;
        std::vector<std::vector<uint32_t>> adj(active.size());
        for (size_t i = 0; i < p->pair_p_h.size(); i++)
        {
            const int a = slot[p->pair_p_h[i]], b = slot[p->pair_q_h[i]];
            if (a >= 0 && b >= 0)
            {
                adj[a].push_back((uint32_t)b);
                adj[b].push_back((uint32_t)a);
            }
        }
        for (auto &l : adj)
            std::sort(l.begin(), l.end(), [&](uint32_t x, uint32_t y) {
                return adj[x].size() != adj[y].size() ? adj[x].size() < adj[y].size() : x < y;
            });
        std::vector<char> seen(active.size(), 0);
        std::vector<uint32_t> order;
        order.reserve(active.size());
        auto bfs = [&](uint32_t start, std::vector<uint32_t> &out, std::vector<char> &mark) {
            const size_t first = out.size();
            out.push_back(start);
            mark[start] = 1;
            for (size_t h = first; h < out.size(); h++)
                for (uint32_t v : adj[out[h]])
                    if (!mark[v])
                    {
                        mark[v] = 1;
                        out.push_back(v);
                    }
        };
        for (uint32_t s0 = 0; s0 < active.size(); s0++)
        {
            if (seen[s0])
                continue;
            // the component of s0, then its lowest-degree member, then two sweeps towards the periphery
            std::vector<uint32_t> comp;
            std::vector<char> tmp(seen);
            bfs(s0, comp, tmp);
            uint32_t start = comp[0];
            for (uint32_t v : comp)
                if (adj[v].size() < adj[start].size() || (adj[v].size() == adj[start].size() && v < start))
                    start = v;
            for (int sweep = 0; sweep < 2; sweep++)
            {
                std::vector<uint32_t> lv;
                std::vector<char> tmp2(seen);
                bfs(start, lv, tmp2);
                start = lv.back();
            }
            bfs(start, order, seen);
        }
        std::reverse(order.begin(), order.end());
        // Dissection (round 3): with the cameras in one band, the factorisation is one chain of diagonal tiles - 47 at
        // n = 3003, ~22 us each, the whole device waiting on one workgroup.  Cut the order into regions of g cameras
        // (g a multiple of 64, so that a region starts on a tile boundary: 3 g unknowns = 3 g / 64 tiles): the next g
        // cameras of the order form a region, every camera further down that is linked to one of them becomes a
        // separator, and so on; the separators go to the end of the order, in front of the plane unknowns, and are the
        // tail of the envelope (dense rows).  Regions are then linked to separators only, their chains of tiles are
        // independent (lm_envelope::region_begin), and the critical path is the longest region plus the tail.  g is
        // chosen to make that shortest (a separator camera counted 1.5 times); no dissection when it does not shorten the
        // path to 0.8 of the single chain.
        const bool use_dissect = !ochip_test_hook("no_dissect");
        if (use_dissect && order.size() >= 4 * (size_t)NB)
        {
            const int N = (int)order.size();
            auto cut = [&](int g, std::vector<int> *state_out, int *n_regions) -> int { // -> path length in cameras (or -1)
                std::vector<int> state((size_t)N, -1);                                  // by slot: region, or -2 = separator
                int left = N, regions = 0, longest = 0, seps = 0;
                size_t pos = 0;
                while (left > 0)
                {
                    const bool last = left <= g + g / 2;
                    const int want = last ? left : g;
                    int got = 0;
                    std::vector<uint32_t> mine;
                    for (; pos < order.size() && got < want; pos++)
                        if (state[order[pos]] == -1)
                        {
                            state[order[pos]] = regions;
                            mine.push_back(order[pos]);
                            got++;
                        }
                    left -= got;
                    longest = std::max(longest, got);
                    regions++;
                    if (last)
                        break;
                    for (uint32_t v : mine)
                        for (uint32_t u : adj[v])
                            if (state[u] == -1)
                            {
                                state[u] = -2;
                                seps++;
                                left--;
                            }
                }
                if (regions < 2)
                    return -1;
                if (state_out)
                    state_out->swap(state);
                *n_regions = regions;
                // (a column of the tail costs more than a column of a region: its tiles first sum over the whole band.
                // Measured at n = 3003: g = 128 / 192 / 256 / 320 / 448 cameras -> 684 / 605 / 598 / 592 / 764 us.)
                return longest + seps + seps / 2;
            };
            int best_g = 0, best_path = N, regions = 0;
            const int forced_g = getenv("OCHIP_RELAX_DISSECT_G") ? atoi(getenv("OCHIP_RELAX_DISSECT_G")) / NB * NB : 0; // A/B knob
            for (int g = forced_g > 0 ? forced_g : NB; g <= (forced_g > 0 ? forced_g : N / 2); g += NB)
            {
                int r = 0;
                const int path = cut(g, nullptr, &r);
                if (path >= 0 && path < best_path)
                    best_path = path, best_g = g;
            }
            if (best_g > 0 && 10 * best_path <= 8 * N)
            {
                std::vector<int> state;
                cut(best_g, &state, &regions);
                std::vector<uint32_t> cut_order;
                cut_order.reserve(order.size());
                for (uint32_t v : order) // regions were handed out in the order's direction: region numbers ascend
                    if (state[v] >= 0)
                        cut_order.push_back(v);
                n_separators = 0;
                for (uint32_t v : order)
                    if (state[v] == -2)
                    {
                        cut_order.push_back(v);
                        n_separators++;
                    }
                // (a stable partition by region: a region's cameras may be interleaved with separators of earlier cuts)
                std::stable_sort(cut_order.begin(), cut_order.end() - n_separators,
                                 [&](uint32_t a, uint32_t b) { return state[a] < state[b]; });
                order.swap(cut_order);
                for (int r = 0; r < regions; r++)
                    region_first_block.push_back(r * (3 * best_g / NB));
            }
        }
        std::vector<uint32_t> reordered(active.size());
        for (size_t i = 0; i < order.size(); i++)
            reordered[i] = active[order[i]];
        active.swap(reordered);
    }
    for (uint32_t c : active)
    {
        p->cam_t[c] = t;
        t += 3;
    }
    for (int i = 0; i < 3; i++)
    {
        p->z_t[i] = -1;
        if (p->z_optimize[i] && p->dev.n_blocks > 0)
            p->z_t[i] = t++;
    }
    p->n_tangent = t;
    lm_envelope env;
    {
        // block envelope of the reduced system J'J: camera unknowns in camera order, then the plane unknowns (coupled to
        // every camera: the tail).  A pair (p, q) puts a 3 x 3 block at rows t_q.., columns t_p..; Cholesky fill stays
        // inside the column envelope once that is made monotone.
        // (dissected graph: the separator cameras belong to the tail as well, and a link into the tail leaves the band's
        // envelope alone - the tail's rows are dense under every column)
        const int cam_end = 3 * ((int)active.size() - n_separators);
        const int n_all = std::max(t, 1), nblk = (n_all + NB - 1) / NB;
        env.tail_begin = cam_end;
        env.region_begin = region_first_block;
        env.env_end.assign(nblk, 0);
        for (int k = 0; k < nblk; k++)
            env.env_end[k] = std::min((k + 1) * NB, cam_end);
        for (uint32_t c = 0; c < p->n_cams; c++) // a camera's own 3 x 3 block may straddle two column blocks
            if (p->cam_t[c] >= 0 && p->cam_t[c] < cam_end)
                for (int k = p->cam_t[c] / NB; k <= (p->cam_t[c] + 2) / NB; k++)
                    env.env_end[k] = std::max(env.env_end[k], p->cam_t[c] + 3);
        for (size_t i = 0; i < p->pair_p_h.size(); i++)
        {
            const int ta = p->cam_t[p->pair_p_h[i]], tb = p->cam_t[p->pair_q_h[i]];
            if (ta < 0 || tb < 0 || std::max(ta, tb) >= cam_end)
                continue;
            const int lo = std::min(ta, tb), hi = std::max(ta, tb) + 3;
            for (int k = lo / NB; k <= (lo + 2) / NB; k++)
                env.env_end[k] = std::max(env.env_end[k], hi);
        }
        for (int k = 1; k < nblk; k++)
            env.env_end[k] = std::max(env.env_end[k], std::min(env.env_end[k - 1], cam_end));
        // row envelope for the backward solve: the first column block whose envelope reaches into block row k
        env.first_col.assign(nblk, 0);
        for (int k = 0; k < nblk; k++)
        {
            const int k0 = k * NB;
            int first = k0;
            if (k0 + NB > cam_end) // the block holds tail rows: dense
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
            long band = 0;
            for (int k = 0; k < nblk; k++)
                band += std::max(0, env.env_end[k] - (k + 1) * NB);
            fprintf(stderr, "[ochip relax] n=%d blocks=%d tail_begin=%d mean envelope rows below a block %.1f (dense: %.1f); %zu regions, %d separator cameras\n", n_all, nblk,
                    cam_end, (double)band / nblk, (double)n_all / 2, std::max<size_t>(region_first_block.size(), 1), n_separators);
        }
    }
    if (hipMemcpy(p->dev.cam_t, p->cam_t.data(), p->n_cams * 4, hipMemcpyHostToDevice) != hipSuccess ||
        hipMemcpy(p->dev.z_t, p->z_t, 12, hipMemcpyHostToDevice) != hipSuccess)
        return ochip_fail(p->ctx, OCHIP_EHIP, "hipMemcpy failed (tangent map)");
    p->sys.ctx = p->ctx;
    p->sys.allocs = &p->allocs;
    p->sys.speculative = true; // (the candidate is evaluated with its Jacobian: plane_model::evaluate_candidate_jac)
    const int rrc = lm_system_resize(&p->sys, t, env);
    if (rrc != OCHIP_OK)
        return rrc;
    return OCHIP_OK;
}
} // namespace

extern "C"
{

int ochip_relax_problem_create(ochip_ctx *ctx, const ochip_relax_desc *d, ochip_relax_problem **out)
{
    if (!ctx || !d || !out)
        return OCHIP_EINVAL;
    *out = nullptr;
    OCHIP_HIP(ctx, hipSetDevice(ctx->device));
    auto *p = new (std::nothrow) ochip_relax_problem();
    if (!p)
        return ochip_fail(ctx, OCHIP_ENOMEM, "host allocation failed");
    p->ctx = ctx;
    p->n_cams = d->n_cams;
    for (uint32_t b = 0; b < d->n_blocks; b++)
        if (d->blk_cam_a[b] >= d->n_cams || d->blk_cam_b[b] >= d->n_cams || d->blk_cam_a[b] == d->blk_cam_b[b])
        {
            delete p;
            return ochip_fail(ctx, OCHIP_EINVAL, "residual block %u has bad camera indices", b);
        }
    // sort blocks by unordered camera pair (stable), build segments and the camera -> pair CSR
    auto key = [&](uint32_t b) {
        const uint32_t a = d->blk_cam_a[b], c = d->blk_cam_b[b];
        return ((uint64_t)std::min(a, c) << 32) | std::max(a, c);
    };
    // blocks arrive edge by edge, i.e. in runs of equal key: sort the runs (stable), then expand
    std::vector<uint32_t> order(d->n_blocks);
    {
        struct run
        {
            uint64_t key;
            uint32_t start, len;
        };
        std::vector<run> runs;
        for (uint32_t i = 0; i < d->n_blocks; i++)
        {
            const uint64_t k = key(i);
            if (runs.empty() || runs.back().key != k)
                runs.push_back(run{k, i, 1});
            else
                runs.back().len++;
        }
        std::stable_sort(runs.begin(), runs.end(), [](const run &x, const run &y) { return x.key < y.key; });
        uint32_t o = 0;
        for (const run &r : runs)
            for (uint32_t i = 0; i < r.len; i++)
                order[o++] = r.start + i;
    }
    std::vector<uint32_t> blk_a(d->n_blocks), blk_b(d->n_blocks), pair_off, pair_p, pair_q;
    // the permuted rays (tens of MB) are written straight into page-locked memory: the upload then runs at link speed
    // instead of through the runtime's bounce buffers
    double *rays = nullptr;
    {
        void *pinned = nullptr;
        if (ochip_host_alloc(ctx, std::max<size_t>((size_t)d->n_blocks * 48, 8), &pinned) != OCHIP_OK)
        {
            delete p;
            return OCHIP_ENOMEM;
        }
        rays = (double *)pinned;
    }
    {
        // the permuted copy of the blocks (tens of MB) on a few threads; the segment boundaries afterwards, in order
        const uint32_t nthr = d->n_blocks > (1u << 16) ? 8u : 1u;
        auto copy_range = [&](uint32_t lo, uint32_t hi) {
            for (uint32_t i = lo; i < hi; i++)
            {
                const uint32_t b = order[i];
                blk_a[i] = d->blk_cam_a[b];
                blk_b[i] = d->blk_cam_b[b];
                for (int k = 0; k < 6; k++)
                    rays[(size_t)i * 6 + k] = d->blk_rays[(size_t)b * 6 + k];
            }
        };
        std::vector<std::thread> workers;
        const uint32_t per = (d->n_blocks + nthr - 1) / nthr;
        for (uint32_t w = 1; w < nthr; w++)
            workers.emplace_back(copy_range, std::min(d->n_blocks, w * per), std::min(d->n_blocks, (w + 1) * per));
        copy_range(0, std::min(d->n_blocks, per));
        for (auto &t : workers)
            t.join();
    }
    for (uint32_t i = 0; i < d->n_blocks; i++)
    {
        const uint32_t b = order[i];
        if (i == 0 || key(order[i - 1]) != key(b))
        {
            pair_off.push_back(i);
            pair_p.push_back(std::min(blk_a[i], blk_b[i]));
            pair_q.push_back(std::max(blk_a[i], blk_b[i]));
        }
    }
    pair_off.push_back(d->n_blocks);
    const uint32_t n_pairs = (uint32_t)pair_p.size();
    p->cam_pair_count.assign(d->n_cams, 0);
    for (uint32_t pr = 0; pr < n_pairs; pr++)
    {
        p->cam_pair_count[pair_p[pr]]++;
        p->cam_pair_count[pair_q[pr]]++;
    }
    std::vector<uint32_t> cpo(d->n_cams + 1, 0), cpi(2 * (size_t)n_pairs);
    for (uint32_t c = 0; c < d->n_cams; c++)
        cpo[c + 1] = cpo[c] + p->cam_pair_count[c];
    {
        std::vector<uint32_t> fill(cpo.begin(), cpo.end() - 1);
        for (uint32_t pr = 0; pr < n_pairs; pr++)
        {
            cpi[fill[pair_p[pr]]++] = pr * 2;
            cpi[fill[pair_q[pr]]++] = pr * 2 + 1;
        }
    }
    p->cam_optimize.assign(d->cam_optimize, d->cam_optimize + d->n_cams);
    p->cam_has_prior_host.assign(d->n_cams, 0);
    for (uint32_t i = 0; i < d->n_prior; i++)
        if (d->prior_cam[i] < d->n_cams)
            p->cam_has_prior_host[d->prior_cam[i]] = 1;
    for (int i = 0; i < 3; i++)
        p->z_optimize[i] = d->z_optimize[i];

    relax_dev &D = p->dev;
    D.n_cams = d->n_cams;
    D.n_blocks = d->n_blocks;
    D.n_pairs = n_pairs;
    D.n_prior = d->n_prior;
    D.huber_a = d->huber_a;
    D.prior_weight = d->prior_weight;
    double plane[12];
    for (int i = 0; i < 6; i++)
        plane[i] = d->plane_xy[i];
    for (int i = 0; i < 3; i++)
        plane[6 + i] = plane[9 + i] = d->plane_z[i];
    int rc = OCHIP_OK;
    auto chk = [&](int r) {
        if (rc == OCHIP_OK)
            rc = r;
    };
    chk(dev_upload(p, &D.cam_pos, d->cam_pos, (size_t)d->n_cams * 3));
    chk(dev_upload(p, &D.cam_q, d->cam_q, (size_t)d->n_cams * 4));
    chk(dev_upload(p, &D.cam_q2, d->cam_q, (size_t)d->n_cams * 4));
    chk(dev_upload<int32_t>(p, &D.cam_t, nullptr, d->n_cams));
    chk(dev_upload(p, &D.plane, plane, 12));
    D.zcur = 6;
    chk(dev_upload<int32_t>(p, &D.z_t, nullptr, 3));
    chk(dev_upload(p, &D.blk_a, blk_a.data(), blk_a.size()));
    chk(dev_upload(p, &D.blk_b, blk_b.data(), blk_b.size()));
    chk(dev_upload(p, &D.blk_rays, (const double *)rays, (size_t)d->n_blocks * 6));
    ochip_host_free(ctx, rays);
    chk(dev_upload(p, &D.pair_off, pair_off.data(), pair_off.size()));
    p->pair_p_h = pair_p;
    p->pair_q_h = pair_q;
    chk(dev_upload(p, &D.pair_p, pair_p.data(), pair_p.size()));
    chk(dev_upload(p, &D.pair_q, pair_q.data(), pair_q.size()));
    chk(dev_upload(p, &D.cam_pair_off, cpo.data(), cpo.size()));
    chk(dev_upload(p, &D.cam_pair_idx, cpi.data(), cpi.size()));
    chk(dev_upload<double>(p, &D.pair_acc, nullptr, (size_t)n_pairs * ACC));
    chk(dev_upload<double>(p, &D.pair_cost, nullptr, n_pairs));
    chk(dev_upload<int32_t>(p, &p->fail_ranks, nullptr, 1));
    if (rc == OCHIP_OK && hipMemsetAsync(p->fail_ranks, 0, 4, ctx->stream) != hipSuccess) // (evaluations expect it clear and leave it clear)
        rc = ochip_fail(ctx, OCHIP_EHIP, "hipMemsetAsync failed in relax problem");
    D.fail = p->fail_ranks;
    D.pair_lo = 0;
    p->pair_hi = n_pairs;
    p->shard_chunk = n_pairs;
    chk(dev_upload(p, &p->cam_has_prior, p->cam_has_prior_host.data(), p->cam_has_prior_host.size()));
    chk(dev_upload(p, &p->cam_optimize_dev, p->cam_optimize.data(), p->cam_optimize.size()));
    if (rc == OCHIP_OK)
        rc = assign_tangent(p);
    if (rc != OCHIP_OK)
    {
        ochip_relax_problem_destroy(p);
        return rc;
    }
    *out = p;
    return OCHIP_OK;
}

void ochip_relax_problem_destroy(ochip_relax_problem *p)
{
    if (!p)
        return;
    (void)hipSetDevice(p->ctx->device);
    (void)ochip_stream_wait(p->ctx, p->ctx->stream);
    for (auto &a : p->allocs)
        ochip_pool_put(p->ctx, a.first, a.second);
    delete p;
}

int ochip_relax_set_cameras_constant(ochip_relax_problem *p, int constant)
{
    if (!p)
        return OCHIP_EINVAL;
    p->cams_frozen = constant != 0;
    return assign_tangent(p);
}

int ochip_relax_set_shard(ochip_relax_problem *p, uint32_t rank, uint32_t world, ochip_relax_exchange_fn fn, void *user)
{
    if (!p)
        return OCHIP_EINVAL;
    ochip_ctx *ctx = p->ctx;
    if (world == 0 || rank >= world || (world > 1 && !fn) || world > (uint32_t)lm_system::BOX_MAX_RANKS)
        return ochip_fail(ctx, OCHIP_EINVAL, "bad shard (rank %u of %u%s; at most %d ranks: their failure flags travel in the solver's page-locked block)",
                          rank, world, fn ? "" : ", no exchange function", lm_system::BOX_MAX_RANKS);
    OCHIP_HIP(ctx, hipSetDevice(ctx->device));
    relax_dev &D = p->dev;
    const uint32_t n_pairs = D.n_pairs;
    const uint32_t chunk = std::max<uint32_t>(1, (n_pairs + world - 1) / world);
    int rc = OCHIP_OK;
    auto chk = [&](int r) {
        if (rc == OCHIP_OK)
            rc = r;
    };
    // record arrays padded to world equal slices so that the exchange is a plain in-place all-gather
    chk(dev_upload<double>(p, &D.pair_acc, nullptr, (size_t)world * chunk * ACC));
    chk(dev_upload<double>(p, &D.pair_cost, nullptr, (size_t)world * chunk));
    chk(dev_upload<int32_t>(p, &p->fail_ranks, nullptr, world));
    if (rc != OCHIP_OK)
        return rc;
    OCHIP_HIP(ctx, hipMemset(D.pair_acc, 0, (size_t)world * chunk * ACC * 8));
    OCHIP_HIP(ctx, hipMemset(D.pair_cost, 0, (size_t)world * chunk * 8));
    OCHIP_HIP(ctx, hipMemset(p->fail_ranks, 0, (size_t)world * 4));
    D.pair_lo = std::min(rank * chunk, n_pairs);
    p->pair_hi = std::min((rank + 1) * chunk, n_pairs);
    D.fail = p->fail_ranks + rank;
    p->shard_rank = rank;
    p->shard_world = world;
    p->shard_chunk = chunk;
    p->exchange = fn;
    p->exchange_user = user;
    return OCHIP_OK;
}

int ochip_relax_get_state(ochip_relax_problem *p, double *cam_q, double *plane_z)
{
    if (!p)
        return OCHIP_EINVAL;
    ochip_ctx *ctx = p->ctx;
    OCHIP_HIP(ctx, ochip_stream_wait(ctx, ctx->stream));
    if (cam_q)
        OCHIP_HIP(ctx, hipMemcpy(cam_q, p->dev.cam_q, (size_t)p->n_cams * 32, hipMemcpyDeviceToHost));
    if (plane_z)
        OCHIP_HIP(ctx, hipMemcpy(plane_z, p->dev.plane + p->dev.zcur, 24, hipMemcpyDeviceToHost));
    return OCHIP_OK;
}

} // extern "C"

// The ground-plane flavour as an lm_model (relax_lm.hpp): evaluation = pair records (sharded + exchanged when
// ochip_relax_set_shard is in effect), deterministic assembly into the dense system.
namespace
{
struct plane_model final : lm_model
{
    ochip_relax_problem *p;
    explicit plane_model(ochip_relax_problem *prob) : p(prob)
    {
    }
    int evaluate(bool with_jac, int which, double *cost) override
    {
        int mask = 0;
        const int rc = run(p->dev, with_jac, which, false, nullptr, cost, &mask);
        return rc < 0 ? rc : (mask ? 1 : 0);
    }
    // The candidate as the current state of a view of the problem (state buffers exchanged), evaluated with its Jacobian
    // into the system's second set; accept_swap() makes the view the problem.
    relax_dev candidate_view() const
    {
        relax_dev V = p->dev;
        std::swap(V.cam_q, V.cam_q2);
        V.zcur = 15 - V.zcur;
        return V;
    }
    bool speculates() override
    {
        return mails_results() && p->sys.A2 != nullptr;
    }
    int evaluate_candidate_jac(const double *scale, double *cost, int *fail_mask) override
    {
        return run(candidate_view(), true, 0, true, scale, cost, fail_mask);
    }
    void accept_swap() override
    {
        p->dev = candidate_view();
    }
    bool launch_back_solve_candidate(const back_args &a, const double *scale) override
    {
        hipLaunchKernelGGL(plane_back_solve_candidate_kernel, dim3((unsigned)a.n_regions), dim3(LM_TG), 0, p->ctx->stream, a.W, a.n, a.linv,
                           a.y, a.work, a.first_blk, a.n_blocks, a.region, a.tb, a.lm_diag, a.gs, a.scal, a.arrived, a.x_in_lds, p->dev,
                           scale);
        return true;
    }
    // one evaluation of state `which` of D: pair records (sharded + exchanged when ochip_relax_set_shard is in effect),
    // deterministic assembly into the system's first or second set, results mailed to the host block
    int run(const relax_dev &D, bool with_jac, int which, bool second_set, const double *diag_scale, double *cost, int *fail_mask)
    {
        ochip_ctx *ctx = p->ctx;
        hipStream_t st = ctx->stream;
        const int n = p->n_tangent;
        const lm_matrix Am = second_set ? p->sys.matA2() : p->sys.matA();
        double *const gv = second_set ? p->sys.g2 : p->sys.g;
        bool &clean = second_set ? p->sys.A2_clean : p->sys.A_clean;
        // (D.fail is zero here: cleared at creation and by the reduction that ended the evaluation before this one)
        hipEvent_t e0, e1;
        ochip_prof_begin(ctx, OCHIP_K_RELAX_EVAL, &e0, &e1);
        const uint32_t my_pairs = p->pair_hi > D.pair_lo ? p->pair_hi - D.pair_lo : 0;
        if (my_pairs)
        {
            if (with_jac)
                hipLaunchKernelGGL(relax_pair_eval_kernel<true>, dim3(my_pairs), dim3(W), 0, st, D, which);
            else
                hipLaunchKernelGGL(relax_pair_eval_kernel<false>, dim3(my_pairs), dim3(W), 0, st, D, which);
        }
        ochip_prof_end(ctx, OCHIP_K_RELAX_EVAL, e0, e1);
        (with_jac ? ctx->relax_blocks_jac : ctx->relax_blocks_cost) += D.n_blocks ? (uint64_t)D.n_blocks * my_pairs / std::max(D.n_pairs, 1u) : 0;
        if (p->exchange)
        {
            // the ranks' pair records (and failure flags) are all-gathered in place; from here on every rank holds
            // the same arrays and runs the same deterministic assembly.  A failed exchange is a hard error: the ranks
            // would otherwise leave the solve on different schedules and the next collective would hang.
            // (the native transport enqueues its all-gathers on this stream: stream order is all it needs; a callback
            // transport reads the arrays from outside the stream and gets them complete)
            if (p->exchange != ochip_rccl_relax_exchange)
                OCHIP_HIP(ctx, ochip_stream_wait(ctx, st));
            const int xrc = p->exchange(p->exchange_user, D.pair_acc, with_jac ? (uint64_t)p->shard_chunk * ACC * 8 : 0,
                                        D.pair_cost, (uint64_t)p->shard_chunk * 8, p->fail_ranks, 4);
            if (xrc != 0)
                return ochip_fail(ctx, OCHIP_EHIP, "relax exchange callback failed (%d)", xrc);
        }
        if (with_jac)
        {
            // the entries of A that the scatter and the reduction write are the same from one evaluation to the next (fixed
            // by the layout), and nothing else writes A: the rest is cleared once per layout, not once per Jacobian (23 MB)
            if (!clean)
            {
                OCHIP_HIP(ctx, hipMemsetAsync(Am.tiles, 0, p->sys.matrix_bytes(), st));
                clean = true;
            }
            // (J'r needs no clearing: every unknown belongs to an active camera or a free plane height, whose owners write it)
        }
        if (!p->reduce_partials)
        {
            const int arc = dev_upload<double>(p, &p->reduce_partials, nullptr, (size_t)REDUCE_GROUPS * 10 + 2);
            if (arc != OCHIP_OK)
                return arc;
            p->reduce_arrived = reinterpret_cast<unsigned int *>(p->reduce_partials + (size_t)REDUCE_GROUPS * 10);
            OCHIP_HIP(ctx, hipMemsetAsync(p->reduce_arrived, 0, 8, st));
        }
        const bool mailed = mails_results();
        lm_mail mail{};
        if (mailed)
            mail = lm_mail{p->sys.box, p->sys.scal, p->sys.fail_chol, p->fail_ranks, (int)p->shard_world, 1, nullptr};
        // (with the Jacobian: the scatter of the pair records into A and g rides in the same launch)
        const uint32_t cam_blocks = with_jac ? (D.n_cams + 3) / 4 : 0, pair_blocks = with_jac ? (D.n_pairs + 255) / 256 : 0;
        hipLaunchKernelGGL(relax_reduce_plane_kernel, dim3(REDUCE_GROUPS + cam_blocks + pair_blocks), dim3(256), 0, st, D, Am, gv, n,
                           p->cam_has_prior, p->sys.scal, with_jac ? 1 : 0, which, p->reduce_partials, p->reduce_arrived, mail, diag_scale,
                           second_set ? p->sys.diagonal2 : p->sys.diagonal, cam_blocks, cam_blocks + pair_blocks);
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
        if (!mailed)
        {
            OCHIP_HIP(ctx, hipMemcpyAsync(h0, p->sys.scal, 8, hipMemcpyDeviceToHost, st));
            OCHIP_HIP(ctx, hipMemcpyAsync(hfails, p->fail_ranks, (size_t)p->shard_world * 4, hipMemcpyDeviceToHost, st));
            OCHIP_HIP(ctx, hipMemsetAsync(D.fail, 0, 4, st)); // this rank's flag, clear for the next evaluation
        }
        if (before_wait)
            before_wait();
        OCHIP_HIP(ctx, ochip_stream_wait(ctx, st));
        *cost = *h0;
        int hfail = 0;
        for (int r = 0; r < (int)p->shard_world; r++)
            hfail |= hfails[r];
        *fail_mask = hfail;
        return OCHIP_OK;
    }
    bool mails_results() override
    {
        return p->shard_world <= (uint32_t)(2 * (lm_system::BOX_VECTORS - lm_system::BOX_FAILS)) && p->sys.box != nullptr;
    }
    void launch_candidate(const double *y, const double *scale, double alpha, double *scal) override
    {
        hipLaunchKernelGGL(plane_candidate_kernel, dim3(1), dim3(LM_TG), 0, p->ctx->stream, p->dev, scale, y, alpha, scal);
    }
    void launch_accept() override
    {
        hipLaunchKernelGGL(lm_accept_kernel, dim3((p->n_cams * 4 + 255) / 256 + 1), dim3(256), 0, p->ctx->stream, p->dev);
    }
    void launch_normalize() override
    {
        if (p->dev.n_cams)
            hipLaunchKernelGGL(normalize_kernel, dim3((p->dev.n_cams + 255) / 256), dim3(256), 0, p->ctx->stream, p->dev,
                               p->cam_optimize_dev);
    }
    int x_norm(double *out) override
    {
        ochip_ctx *ctx = p->ctx;
        std::vector<double> q((size_t)p->n_cams * 4);
        double z[3], x_norm = 0;
        OCHIP_HIP(ctx, hipMemcpy(q.data(), p->dev.cam_q, q.size() * 8, hipMemcpyDeviceToHost));
        OCHIP_HIP(ctx, hipMemcpy(z, p->dev.plane + p->dev.zcur, 24, hipMemcpyDeviceToHost));
        for (uint32_t c = 0; c < p->n_cams; c++)
            if (p->cam_t[c] >= 0)
                for (int k = 0; k < 4; k++)
                    x_norm += q[c * 4 + k] * q[c * 4 + k];
        for (int i = 0; i < 3; i++)
            if (p->z_t[i] >= 0)
                x_norm += z[i] * z[i];
        *out = std::sqrt(x_norm);
        return OCHIP_OK;
    }
    int num_residual_blocks() override
    {
        return (int)(p->dev.n_blocks + p->dev.n_prior);
    }
};
} // namespace

extern "C"
{

int ochip_relax_solve(ochip_relax_problem *p, const ochip_relax_options *opt, ochip_relax_summary *sum)
{
    if (!p || !opt || !sum)
        return OCHIP_EINVAL;
    ochip_ctx *ctx = p->ctx;
    OCHIP_HIP(ctx, hipSetDevice(ctx->device));
    *sum = ochip_relax_summary{};
    plane_model model(p);
    sum->num_parameters = p->n_tangent;
    sum->num_residual_blocks = model.num_residual_blocks();
    if (p->dev.n_blocks == 0 && p->dev.n_prior == 0)
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
