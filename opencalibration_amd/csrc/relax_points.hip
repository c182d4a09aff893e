// libochip.so — relax with 3-D points on the device (gfx950): reprojection bundle adjustment with the points eliminated
// by a per-point 3 x 3 Schur complement.  C ABI: ochip_relaxp_* (ochip.h).
//
// Replaces what ceres::Solver::Solve does with SPARSE_SCHUR for RelaxProblem::setup3dPointProblem
// (src/relax/relax_problem.cpp:122-145,986-1187; functors PixelErrorCost_Orientation[Focal[Radial[Tangential]]],
// include/opencalibration/relax/relax_cost_function.hpp:309-500; DistortionMonotonicityCost :157-185).  In the reference
// every whitelisted inlier of an edge is a 3-D point of its own seen by exactly the edge's two cameras, so the points of
// one edge form a GROUP with one pair of cameras; the structure below relies on it:
//   * an OBSERVATION (camera, point, pixel) is one thread: forward-mode duals (Dual<3>) through image_from_3d in passes
//     (camera tangent, point, (f, pp), radial, tangential), Huber loss with Ceres' corrector; its corrected Jacobian rows
//     and residual are kept (2 x 14 + 2 doubles) - everything else is formed from them;
//   * the normal equations are [U W; W' V]: V is block diagonal with one 3 x 3 block per point.  The LM step solves the
//     REDUCED system (U_s + D_c^2 - sum_p W_p (V_p + D_p^2)^-1 W_p') dc = -(g_c - sum_p W_p (V_p + D_p^2)^-1 g_p) in the
//     camera (+ shared lens model) unknowns - dense, factored by the shared block Cholesky (relax_lm.hip) - and
//     back-substitutes dp = -(V_p + D_p^2)^-1 (g_p + W_p' dc) per point.  Damping, Jacobi scaling and the step-quality test
//     are those of the full system, as Ceres applies them (every column of J, point columns included).
//   * a group's contribution to U, and per LM iteration to the Schur complement, is a RECORD over the columns
//     [camera a (3) | camera b (3) | lens model (<= 8)]: entry (i, j) of the record is summed over the group's points in
//     their order by one thread; records are added into the system by the row owners (one thread per camera row / lens
//     row walking its groups in order): no atomics, bitwise reproducible.
#include "ctx.hpp"
#include "dual.hpp"
#include "relax_functors.hpp"
#include "relax_lm.hpp"

#include <algorithm>
#include <cmath>
#include <cstring>
#include <vector>

using namespace ochip;

namespace
{

constexpr int KI = 8;        // lens model columns: f | ppx ppy | k1 k2 k3 | p1 p2
constexpr int JW = 6 + KI;   // columns of an observation's Jacobian rows: camera tangent 3 | point 3 | lens 8
constexpr int RD = 6 + KI;   // columns of a group record: camera a 3 | camera b 3 | lens 8
constexpr int RTRI = RD * (RD + 1) / 2;
constexpr int RLEN = RTRI + RD; // packed upper triangle + right-hand side

struct p_dev
{
    uint32_t n_cams, n_points, n_groups, n_obs;
    double *cam_pos, *cam_q, *cam_q2;
    double *model, *model2; // 8 each
    double *X, *X2;         // points
    int32_t *cam_t;         // first reduced unknown of a camera or -1
    int32_t lens_t[KI];     // reduced unknown of every lens column or -1
    // groups: points [grp_first[g], grp_first[g + 1]) seen by cameras grp_cam[2g], grp_cam[2g + 1]; point p's two
    // observations are 2p (camera a) and 2p + 1 (camera b)
    uint32_t *grp_first, *grp_cam, *pt_group;
    double *obs_px;         // [n_obs][2]
    double *obs_J;          // [n_obs][2][JW] corrected Jacobian rows
    double *obs_r;          // [n_obs][2] corrected residuals
    double *obs_cost;       // [n_obs]
    double *pt_V;           // [n_points][6] J_p' J_p (xx xy xz yy yz zz)
    double *pt_g;           // [n_points][3] J_p' r
    double *pt_scale;       // [n_points][3] Jacobi scaling of the point columns (fixed per solve)
    double *pt_Vinv;        // [n_points][6] (S V S + D^2)^-1
    double *pt_d;           // [n_points][3] the last full step of the point (unscaled), for the line search's slopes
    double *rec;            // [n_groups][RLEN] group records (U after an evaluation, the Schur term inside an iteration)
    int functor;            // 0 orientation, 1 + focal / principal, 2 + radial, 3 + tangential
    double huber_a, f_lo, f_hi;
    int f_bounded;
    double mono_w, mono_rmax; // DistortionMonotonicityCost; mono_w = 0: none
    double *mono;           // [1 + 6 + 3] cost, k x k triangle, rhs
    int32_t *fail;
};

// image_from_3d (distort_keypoints.hpp:26-69) of the camera-frame ray (x, y, z) through model m = f ppx ppy k1 k2 k3 p1 p2
template <typename T> __device__ __forceinline__ void project(const Vec3T<T> &ray, const T *m, T *px)
{
    const double min_z = 1e-3;
    const T cz = (value_of(ray.z) < min_z) ? T(min_z) : ray.z;
    const T u[2] = {ray.x / cz, ray.y / cz};
    T r2[3];
    r2[0] = u[0] * u[0] + u[1] * u[1];
    r2[1] = r2[0] * r2[0];
    r2[2] = r2[1] * r2[0];
    const T radial_dot = m[3] * r2[0] + m[4] * r2[1] + m[5] * r2[2];
    const T prod = u[0] * u[1];
    for (int i = 0; i < 2; i++)
    {
        const T d = (T(1.0) + radial_dot) * u[i] + T(2.0) * prod * m[6 + i] + m[6 + (1 - i)] * (r2[0] + T(2.0) * u[i] * u[i]);
        px[i] = d * m[0] + m[1 + i];
    }
}

// the inverse rotation of a quaternion applied to v: Eigen's q.inverse() * v (conjugate / squared norm)
template <typename T> __device__ __forceinline__ Vec3T<T> inverse_rotate(const T *q, const Vec3T<T> &v)
{
    const T n2 = q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3];
    T qi[4];
    if (value_of(n2) > 0.0)
    {
        qi[0] = (T(0.0) - q[0]) / n2;
        qi[1] = (T(0.0) - q[1]) / n2;
        qi[2] = (T(0.0) - q[2]) / n2;
        qi[3] = q[3] / n2;
    }
    else
        qi[0] = qi[1] = qi[2] = qi[3] = T(0.0);
    return gquat_rotate(qi, v);
}

// one observation: residual, cost and (with_jac) the corrected Jacobian rows.  pass: which group of three columns carries
// the dual parts (0 camera tangent, 1 point, 2 f + pp, 3 radial, 4 tangential); -1: values only
__device__ void observation(const p_dev &P, uint32_t o, int which, int pass, Dual<3> *res)
{
    const uint32_t p = o >> 1, g = P.pt_group[p], c = P.grp_cam[2 * g + (o & 1)];
    const double *Q = (which ? P.cam_q2 : P.cam_q) + 4 * (size_t)c;
    const double *M = which ? P.model2 : P.model;
    const double *X = (which ? P.X2 : P.X) + 3 * (size_t)p;
    Dual<3> q[4], m[8], x[3];
    if (pass == 0)
        gseed_quat(Q, q);
    else
        for (int k = 0; k < 4; k++)
            q[k] = Dual<3>(Q[k]);
    for (int k = 0; k < 3; k++)
    {
        x[k] = Dual<3>(X[k]);
        if (pass == 1)
            x[k].v[k] = 1.0;
    }
    for (int k = 0; k < 8; k++)
        m[k] = Dual<3>(M[k]);
    if (pass == 2)
        m[0].v[0] = 1.0, m[1].v[1] = 1.0, m[2].v[2] = 1.0;
    if (pass == 3)
        m[3].v[0] = 1.0, m[4].v[1] = 1.0, m[5].v[2] = 1.0;
    if (pass == 4)
        m[6].v[0] = 1.0, m[7].v[1] = 1.0;
    const double *L = P.cam_pos + 3 * (size_t)c;
    const Vec3T<Dual<3>> d{x[0] - Dual<3>(L[0]), x[1] - Dual<3>(L[1]), x[2] - Dual<3>(L[2])};
    const Vec3T<Dual<3>> ray = inverse_rotate(q, d);
    Dual<3> px[2];
    project(ray, m, px);
    res[0] = px[0] - Dual<3>(P.obs_px[2 * (size_t)o]);
    res[1] = px[1] - Dual<3>(P.obs_px[2 * (size_t)o + 1]);
}

__global__ __launch_bounds__(64) void obs_kernel(p_dev P, int which, int with_jac)
{
    const uint32_t o = blockIdx.x * 64 + threadIdx.x;
    if (o >= P.n_obs)
        return;
    Dual<3> r[2];
    observation(P, o, which, -1, r);
    const double s = r[0].a * r[0].a + r[1].a * r[1].a;
    bool failed = !(s - s == 0.0);
    // HuberLoss(a) on s = |r|^2 and Ceres' corrector: rho'' <= 0, so residual and Jacobian are scaled by sqrt(rho')
    double rho1 = 1.0, cost = 0.5 * s;
    if (s > P.huber_a * P.huber_a)
    {
        const double rn = sqrt(s);
        rho1 = fmax(2.2250738585072014e-308, P.huber_a / rn);
        cost = 0.5 * (2.0 * P.huber_a * rn - P.huber_a * P.huber_a);
    }
    P.obs_cost[o] = cost;
    if (with_jac)
    {
        const double sr = sqrt(rho1);
        double *J = P.obs_J + (size_t)o * 2 * JW;
        const int n_pass = P.functor == 0 ? 2 : P.functor == 1 ? 3 : P.functor == 2 ? 4 : 5;
        for (int k = 0; k < 2 * JW; k++)
            J[k] = 0.0;
        for (int pass = 0; pass < n_pass; pass++)
        {
            Dual<3> rd[2];
            observation(P, o, which, pass, rd);
            const int c0 = pass == 0 ? 0 : pass == 1 ? 3 : pass == 2 ? 6 : pass == 3 ? 9 : 12, w = pass == 4 ? 2 : 3;
            for (int row = 0; row < 2; row++)
                for (int k = 0; k < w; k++)
                {
                    const double v = rd[row].v[k];
                    if (!(v - v == 0.0))
                        failed = true;
                    J[row * JW + c0 + k] = sr * v;
                }
        }
        P.obs_r[2 * (size_t)o] = sr * r[0].a;
        P.obs_r[2 * (size_t)o + 1] = sr * r[1].a;
    }
    if (failed)
        atomicOr(P.fail, 1);
}

// per point: V = sum over its two observations of J_p' J_p, g = J_p' r
__global__ void point_kernel(p_dev P)
{
    const uint32_t p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= P.n_points)
        return;
    double V[6] = {0, 0, 0, 0, 0, 0}, g[3] = {0, 0, 0};
    for (int s = 0; s < 2; s++)
    {
        const double *J = P.obs_J + ((size_t)2 * p + s) * 2 * JW, *r = P.obs_r + 2 * ((size_t)2 * p + s);
        for (int row = 0; row < 2; row++)
        {
            const double *jp = J + row * JW + 3;
            int e = 0;
            for (int a = 0; a < 3; a++)
            {
                for (int b = a; b < 3; b++)
                    V[e++] += jp[a] * jp[b];
                g[a] += jp[a] * r[row];
            }
        }
    }
    for (int e = 0; e < 6; e++)
        P.pt_V[6 * (size_t)p + e] = V[e];
    for (int a = 0; a < 3; a++)
        P.pt_g[3 * (size_t)p + a] = g[a];
}

// column i of a group record -> (side: 0 camera a's observation, 1 camera b's, 2 both; column of the observation's rows)
__device__ __forceinline__ void rec_col(int i, int *side, int *col)
{
    if (i < 3)
        *side = 0, *col = i;
    else if (i < 6)
        *side = 1, *col = i - 3;
    else
        *side = 2, *col = i; // lens columns sit at 6.. in the observation's rows as well
}
__device__ __forceinline__ int tri_at(int i, int j) // i <= j
{
    return i * RD - i * (i - 1) / 2 + (j - i);
}

// U part of the normal equations, group by group: entry (i, j) = sum over the group's observations of J_i' J_j over the
// camera / lens columns, rhs_i = J_i' r.  One thread per (group, entry), points in order.
__global__ __launch_bounds__(128) void group_u_kernel(p_dev P)
{
    const uint32_t g = blockIdx.x;
    const int e = threadIdx.x;
    if (e >= RLEN)
        return;
    int i, j = -1;
    if (e < RTRI)
    {
        int rem = e;
        i = 0;
        while (rem >= RD - i)
        {
            rem -= RD - i;
            i++;
        }
        j = i + rem;
    }
    else
        i = e - RTRI;
    int si, ci, sj = 2, cj = 0;
    rec_col(i, &si, &ci);
    if (j >= 0)
        rec_col(j, &sj, &cj);
    double acc = 0;
    for (uint32_t p = P.grp_first[g]; p < P.grp_first[g + 1]; p++)
        for (int s = 0; s < 2; s++)
        {
            if ((si != 2 && si != s) || (j >= 0 && sj != 2 && sj != s))
                continue;
            const double *J = P.obs_J + ((size_t)2 * p + s) * 2 * JW, *r = P.obs_r + 2 * ((size_t)2 * p + s);
            for (int row = 0; row < 2; row++)
                acc += J[row * JW + ci] * (j >= 0 ? J[row * JW + cj] : r[row]);
        }
    P.rec[(size_t)g * RLEN + e] = acc;
}

// per LM iteration and point: S V S + D^2 and its inverse.  fix_scale: this is the first Jacobian of the solve - the
// point's Jacobi scaling 1 / (1 + sqrt(diag)) is taken from it.  D^2 = clamp(diag * scale^2, 1e-6, 1e32) / radius.
__global__ void point_prepare_kernel(p_dev P, double radius, int fix_scale, int *fail)
{
    const uint32_t p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= P.n_points)
        return;
    const double *V = P.pt_V + 6 * (size_t)p;
    double *S = P.pt_scale + 3 * (size_t)p;
    const double dg[3] = {V[0], V[3], V[5]};
    if (fix_scale)
        for (int a = 0; a < 3; a++)
            S[a] = 1.0 / (1.0 + sqrt(dg[a]));
    double A[3][3];
    const int at[3][3] = {{0, 1, 2}, {1, 3, 4}, {2, 4, 5}};
    for (int a = 0; a < 3; a++)
        for (int b = 0; b < 3; b++)
            A[a][b] = V[at[a][b]] * S[a] * S[b];
    for (int a = 0; a < 3; a++)
    {
        const double dd = sqrt(fmin(fmax(dg[a] * S[a] * S[a], 1e-6), 1e32) / radius);
        A[a][a] += dd * dd;
    }
    // inverse of the symmetric positive definite 3 x 3 by its Cholesky factor
    const double l00 = sqrt(A[0][0]), l10 = A[1][0] / l00, l20 = A[2][0] / l00;
    const double l11 = sqrt(A[1][1] - l10 * l10), l21 = (A[2][1] - l20 * l10) / l11;
    const double l22 = sqrt(A[2][2] - l20 * l20 - l21 * l21);
    const double i00 = 1.0 / l00, i11 = 1.0 / l11, i22 = 1.0 / l22;
    const double i10 = -l10 * i00 * i11, i21 = -l21 * i11 * i22, i20 = -(l20 * i00 + l21 * i10) * i22;
    double *W = P.pt_Vinv + 6 * (size_t)p; // L^-T L^-1
    W[0] = i00 * i00 + i10 * i10 + i20 * i20;
    W[1] = i10 * i11 + i20 * i21;
    W[2] = i20 * i22;
    W[3] = i11 * i11 + i21 * i21;
    W[4] = i21 * i22;
    W[5] = i22 * i22;
    if (!(W[0] - W[0] == 0.0) || !(W[3] - W[3] == 0.0) || !(W[5] - W[5] == 0.0))
        atomicOr(fail, 1);
}

// w_i of a point: (scaled) coupling of reduced column i with the point, S_i (J_i' J_p) S_p summed over the observations
// that carry column i
__device__ __forceinline__ void coupling(const p_dev &P, uint32_t p, int side, int col, double si, const double *Sp, double *w)
{
    w[0] = w[1] = w[2] = 0;
    for (int s = 0; s < 2; s++)
    {
        if (side != 2 && side != s)
            continue;
        const double *J = P.obs_J + ((size_t)2 * p + s) * 2 * JW;
        for (int row = 0; row < 2; row++)
            for (int a = 0; a < 3; a++)
                w[a] += J[row * JW + col] * J[row * JW + 3 + a];
    }
    for (int a = 0; a < 3; a++)
        w[a] *= si * Sp[a];
}
__device__ __forceinline__ double quad(const double *Vi, const double *a, const double *b)
{
    const double t0 = Vi[0] * b[0] + Vi[1] * b[1] + Vi[2] * b[2];
    const double t1 = Vi[1] * b[0] + Vi[3] * b[1] + Vi[4] * b[2];
    const double t2 = Vi[2] * b[0] + Vi[4] * b[1] + Vi[5] * b[2];
    return a[0] * t0 + a[1] * t1 + a[2] * t2;
}
// reduced unknown (or -1) and scale of record column i of group g
__device__ __forceinline__ int rec_unknown(const p_dev &P, uint32_t g, int i)
{
    if (i < 3)
        return P.cam_t[P.grp_cam[2 * g]] < 0 ? -1 : P.cam_t[P.grp_cam[2 * g]] + i;
    if (i < 6)
        return P.cam_t[P.grp_cam[2 * g + 1]] < 0 ? -1 : P.cam_t[P.grp_cam[2 * g + 1]] + (i - 3);
    return P.lens_t[i - 6];
}

// the Schur term of a group in scaled unknowns: entry (i, j) = sum_p w_i' Vinv w_j, rhs_i = sum_p w_i' Vinv (S_p g_p)
__global__ __launch_bounds__(128) void group_schur_kernel(p_dev P, const double *scale)
{
    const uint32_t g = blockIdx.x;
    const int e = threadIdx.x;
    if (e >= RLEN)
        return;
    int i, j = -1;
    if (e < RTRI)
    {
        int rem = e;
        i = 0;
        while (rem >= RD - i)
        {
            rem -= RD - i;
            i++;
        }
        j = i + rem;
    }
    else
        i = e - RTRI;
    const int ui = rec_unknown(P, g, i), uj = j >= 0 ? rec_unknown(P, g, j) : 0;
    double acc = 0;
    if (ui >= 0 && uj >= 0)
    {
        int si, ci, sj = 2, cj = 0;
        rec_col(i, &si, &ci);
        if (j >= 0)
            rec_col(j, &sj, &cj);
        const double sci = scale[ui], scj = j >= 0 ? scale[uj] : 0.0;
        for (uint32_t p = P.grp_first[g]; p < P.grp_first[g + 1]; p++)
        {
            const double *Sp = P.pt_scale + 3 * (size_t)p, *Vi = P.pt_Vinv + 6 * (size_t)p;
            double wi[3], wj[3];
            coupling(P, p, si, ci, sci, Sp, wi);
            if (j >= 0)
                coupling(P, p, sj, cj, scj, Sp, wj);
            else
                for (int a = 0; a < 3; a++)
                    wj[a] = P.pt_g[3 * (size_t)p + a] * Sp[a];
            acc += quad(Vi, wi, wj);
        }
    }
    P.rec[(size_t)g * RLEN + e] = acc;
}

// records -> system: one thread per (camera, row of its block) and one per lens row, each walking its groups in order.
// sign +1: U into A and g; sign -1: the Schur term out of Wm and its augmented row n (rhs == nullptr).  The matrices keep
// their lower triangle (relax_lm.hpp): a row owner adds the columns up to its diagonal.
__device__ __forceinline__ void add_rhs(const lm_matrix &A, double *rhs, int n, int row, double v)
{
    if (rhs)
        rhs[row] += v;
    else
        A.tiles[lm_at(A, n, row)] += v;
}
__global__ void apply_cam_kernel(p_dev P, const uint32_t *cam_grp_off, const uint32_t *cam_grp, lm_matrix A, double *rhs, int n,
                                 double sign)
{
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t c = t / 3;
    const int r = (int)(t % 3);
    if (c >= P.n_cams || P.cam_t[c] < 0)
        return;
    const int row = P.cam_t[c] + r;
    for (uint32_t k = cam_grp_off[c]; k < cam_grp_off[c + 1]; k++)
    {
        const uint32_t g = cam_grp[k] >> 1, side = cam_grp[k] & 1;
        const double *R = P.rec + (size_t)g * RLEN;
        const int i = (int)side * 3 + r;
        for (int j = 0; j < RD; j++)
        {
            const int u = rec_unknown(P, g, j);
            if (u < 0)
                continue;
            if (u > row)
                continue;
            const double v = R[i <= j ? tri_at(i, j) : tri_at(j, i)];
            A.tiles[lm_at(A, row, u)] += sign * v;
        }
        add_rhs(A, rhs, n, row, sign * R[RTRI + i]);
    }
}
__global__ void apply_lens_kernel(p_dev P, lm_matrix A, double *rhs, int n, double sign)
{
    const int k = threadIdx.x; // lens column
    if (k >= KI || P.lens_t[k] < 0)
        return;
    const int row = P.lens_t[k], i = 6 + k;
    for (uint32_t g = 0; g < P.n_groups; g++)
    {
        const double *R = P.rec + (size_t)g * RLEN;
        for (int j = 0; j < RD; j++)
        {
            const int u = rec_unknown(P, g, j);
            if (u < 0)
                continue;
            if (u > row)
                continue;
            A.tiles[lm_at(A, row, u)] += sign * R[i <= j ? tri_at(i, j) : tri_at(j, i)];
        }
        add_rhs(A, rhs, n, row, sign * R[RTRI + i]);
    }
}

// DistortionMonotonicityCost (relax_cost_function.hpp:157-185) on the radial coefficients of state `which`
__global__ void mono_kernel(p_dev P, int which)
{
    const double *M = which ? P.model2 : P.model;
    double cost = 0, acc[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
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
    P.mono[0] = cost;
    for (int e = 0; e < 9; e++)
        P.mono[1 + e] = acc[e];
}
__global__ void mono_apply_kernel(p_dev P, lm_matrix A, double *g, int n)
{
    const int at[3][3] = {{0, 1, 2}, {1, 3, 4}, {2, 4, 5}};
    for (int a = 0; a < 3; a++)
    {
        const int ua = P.lens_t[3 + a];
        if (ua < 0)
            continue;
        for (int b = 0; b < 3; b++)
            if (P.lens_t[3 + b] >= 0 && P.lens_t[3 + b] <= ua)
                A.tiles[lm_at(A, ua, P.lens_t[3 + b])] += P.mono[1 + at[a][b]];
        g[ua] += P.mono[7 + a];
    }
}

// total cost (fixed order), and the largest |g| over the point columns -> scal[0], scal[5]
__global__ __launch_bounds__(LM_TG) void p_reduce_kernel(p_dev P, int with_gmax, double *scal)
{
    __shared__ double sh[LM_TG];
    const int t = threadIdx.x;
    double v = 0, m = 0;
    for (uint32_t i = t; i < P.n_obs; i += LM_TG)
        v += P.obs_cost[i];
    if (with_gmax)
        for (uint32_t i = t; i < 3 * P.n_points; i += LM_TG)
            m = fmax(m, fabs(P.pt_g[i]));
    sh[t] = v;
    __syncthreads();
    for (int s = LM_TG / 2; s > 0; s >>= 1)
    {
        if (t < s)
            sh[t] += sh[t + s];
        __syncthreads();
    }
    const double total = sh[0] + (P.mono_w > 0 ? P.mono[0] : 0.0);
    __syncthreads();
    sh[t] = m;
    __syncthreads();
    for (int s = LM_TG / 2; s > 0; s >>= 1)
    {
        if (t < s)
            sh[t] = fmax(sh[t], sh[t + s]);
        __syncthreads();
    }
    if (t == 0)
    {
        scal[0] = total;
        if (with_gmax)
            scal[5] = sh[0];
    }
}

// candidate = x (+) alpha * delta: cameras and lens model from the reduced solution (delta = -y .* scale), the points by
// back-substitution dp = -Vinv (S g_p - W' y) in scaled unknowns.  One workgroup.  scal[1] += the points' share of the
// model cost change (alpha = 1), scal[2] = |x - candidate|^2, scal[3] = |candidate|^2, scal[6] = g_p . d_p (slope)
__global__ __launch_bounds__(LM_TG) void p_candidate_kernel(p_dev P, const double *scale, const double *y, double alpha, double radius,
                                                           int fresh, double *scal)
{
    __shared__ double sh[LM_TG];
    const int t = threadIdx.x;
    double sn = 0, xn = 0, mc = 0, slope = 0;
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
    if (t == 0)
    {
        // parameter blocks: focal (1, bounded), principal point (2), radial (3, SubsetManifold: trailing constants count
        // in |x|), tangential (2)
        const int first[4] = {0, 1, 3, 6}, size[4] = {1, 2, 3, 2};
        for (int k = 0; k < 8; k++)
            P.model2[k] = P.model[k];
        for (int b = 0; b < 4; b++)
        {
            bool variable = false;
            for (int k = 0; k < size[b]; k++)
                variable = variable || P.lens_t[first[b] + k] >= 0;
            if (!variable)
                continue;
            for (int k = 0; k < size[b]; k++)
            {
                const int u = P.lens_t[first[b] + k];
                double v = P.model[first[b] + k];
                if (u >= 0)
                    v += alpha * (-y[u] * scale[u]);
                if (b == 0 && P.f_bounded)
                    v = fmin(fmax(v, P.f_lo), P.f_hi);
                P.model2[first[b] + k] = v;
                sn += (P.model[first[b] + k] - v) * (P.model[first[b] + k] - v);
                xn += v * v;
            }
        }
    }
    for (uint32_t p = t; p < P.n_points && !fresh; p += LM_TG)
        for (int a = 0; a < 3; a++)
        {
            const double x0 = P.X[3 * (size_t)p + a], x1 = x0 + alpha * P.pt_d[3 * (size_t)p + a];
            P.X2[3 * (size_t)p + a] = x1;
            sn += (x0 - x1) * (x0 - x1);
            xn += x1 * x1;
        }
    for (uint32_t p = t; p < P.n_points && fresh; p += LM_TG)
    {
        const uint32_t g = P.pt_group[p];
        const double *Sp = P.pt_scale + 3 * (size_t)p, *Vi = P.pt_Vinv + 6 * (size_t)p, *V = P.pt_V + 6 * (size_t)p;
        double rhs[3];
        for (int a = 0; a < 3; a++)
            rhs[a] = P.pt_g[3 * (size_t)p + a] * Sp[a];
        for (int i = 0; i < RD; i++)
        {
            const int u = rec_unknown(P, g, i);
            if (u < 0)
                continue;
            int side, col;
            rec_col(i, &side, &col);
            double w[3];
            coupling(P, p, side, col, scale[u], Sp, w);
            for (int a = 0; a < 3; a++)
                rhs[a] -= w[a] * y[u]; // W' dc with dc = -y
        }
        const double dp[3] = {-(Vi[0] * rhs[0] + Vi[1] * rhs[1] + Vi[2] * rhs[2]), -(Vi[1] * rhs[0] + Vi[3] * rhs[1] + Vi[4] * rhs[2]),
                              -(Vi[2] * rhs[0] + Vi[4] * rhs[1] + Vi[5] * rhs[2])};
        const double dg[3] = {V[0], V[3], V[5]};
        for (int a = 0; a < 3; a++)
        {
            const double d2 = fmin(fmax(dg[a] * Sp[a] * Sp[a], 1e-6), 1e32) / radius;
            mc += 0.5 * (d2 * dp[a] * dp[a] - P.pt_g[3 * (size_t)p + a] * Sp[a] * dp[a]);
            const double full = Sp[a] * dp[a], x0 = P.X[3 * (size_t)p + a], x1 = x0 + alpha * full;
            P.pt_d[3 * (size_t)p + a] = full;
            P.X2[3 * (size_t)p + a] = x1;
            sn += (x0 - x1) * (x0 - x1);
            xn += x1 * x1;
            slope += P.pt_g[3 * (size_t)p + a] * full;
        }
    }
    double out[4];
    const double part[4] = {sn, xn, mc, slope};
    for (int q = 0; q < 4; q++)
    {
        sh[t] = part[q];
        __syncthreads();
        for (int s = LM_TG / 2; s > 0; s >>= 1)
        {
            if (t < s)
                sh[t] += sh[t + s];
            __syncthreads();
        }
        out[q] = sh[0];
        __syncthreads();
    }
    if (t == 0)
    {
        scal[2] = out[0];
        scal[3] = out[1];
        if (fresh)
        {
            scal[1] += out[2];
            scal[6] = out[3];
        }
    }
}

// g_p . d_p with the gradient of the last evaluated Jacobian and the stored full step (line search slopes)
__global__ __launch_bounds__(LM_TG) void p_slope_kernel(p_dev P, double *scal)
{
    __shared__ double sh[LM_TG];
    const int t = threadIdx.x;
    double v = 0;
    for (uint32_t i = t; i < 3 * P.n_points; i += LM_TG)
        v += P.pt_g[i] * P.pt_d[i];
    sh[t] = v;
    __syncthreads();
    for (int s = LM_TG / 2; s > 0; s >>= 1)
    {
        if (t < s)
            sh[t] += sh[t + s];
        __syncthreads();
    }
    if (t == 0)
        scal[6] = sh[0];
}

__global__ void p_accept_kernel(p_dev P)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < P.n_cams * 4)
        P.cam_q[i] = P.cam_q2[i];
    if (i < P.n_points * 3)
        P.X[i] = P.X2[i];
    if (i < 8)
        P.model[i] = P.model2[i];
}

__global__ void p_normalize_kernel(p_dev P, const uint8_t *cam_optimize)
{
    const uint32_t c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= P.n_cams || !cam_optimize[c])
        return;
    double *q = P.cam_q + 4 * (size_t)c;
    const double n2 = q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3];
    if (n2 > 0.0) // Eigen's normalize()
    {
        const double nn = sqrt(n2);
        for (int k = 0; k < 4; k++)
            q[k] /= nn;
    }
}

} // namespace

struct ochip_relaxp_problem
{
    ochip_ctx *ctx = nullptr;
    p_dev dev{};
    std::vector<std::pair<void *, size_t>> allocs;
    lm_system sys;
    uint32_t n_cams = 0, n_points = 0, n_groups = 0;
    std::vector<uint8_t> cam_optimize;
    std::vector<uint8_t> cam_used; // appears in a group
    uint8_t opt_f = 0, opt_pp = 0, n_k_free = 0, opt_tan = 0;
    int functor = 0;
    bool structure_only = false;
    int n = 0;
    std::vector<int32_t> cam_t;
    uint32_t *cam_grp_off = nullptr, *cam_grp = nullptr;
    uint8_t *cam_optimize_dev = nullptr;
};

namespace
{

template <typename T> int up(ochip_relaxp_problem *p, T **dst, const T *src, size_t n)
{
    return lm_dev_upload(p->ctx, &p->allocs, dst, src, n);
}

// which unknowns are variable and where they sit in the reduced system (cameras in index order, then the lens model)
int assign(ochip_relaxp_problem *p)
{
    p->cam_t.assign(p->n_cams, -1);
    int t = 0;
    if (!p->structure_only)
        for (uint32_t c = 0; c < p->n_cams; c++)
            if (p->cam_optimize[c] && p->cam_used[c])
            {
                p->cam_t[c] = t;
                t += 3;
            }
    int32_t lens_t[KI];
    for (int k = 0; k < KI; k++)
        lens_t[k] = -1;
    if (!p->structure_only && p->n_groups > 0)
    {
        if (p->functor >= 1 && p->opt_f)
            lens_t[0] = t++;
        if (p->functor >= 1 && p->opt_pp)
            lens_t[1] = t++, lens_t[2] = t++;
        if (p->functor >= 2)
            for (int k = 0; k < (int)p->n_k_free; k++)
                lens_t[3 + k] = t++;
        if (p->functor >= 3)
            lens_t[6] = t++, lens_t[7] = t++;
    }
    p->n = t;
    std::memcpy(p->dev.lens_t, lens_t, sizeof lens_t);
    if (p->n_cams && hipMemcpy(p->dev.cam_t, p->cam_t.data(), p->n_cams * 4, hipMemcpyHostToDevice) != hipSuccess)
        return ochip_fail(p->ctx, OCHIP_EHIP, "hipMemcpy failed (unknown map)");
    // the Schur complement couples every pair of cameras that share an edge, and the lens model couples everything: the
    // reduced system is taken dense (it is small next to the points it stands for)
    lm_envelope env;
    const int nblk = (std::max(t, 1) + LM_NB - 1) / LM_NB;
    env.tail_begin = 0;
    env.env_end.assign(nblk, 0);
    env.first_col.assign(nblk, 0);
    p->sys.ctx = p->ctx;
    p->sys.allocs = &p->allocs;
    return lm_system_resize(&p->sys, t, env);
}

struct points_model final : lm_model
{
    ochip_relaxp_problem *p;
    bool scale_fixed = false, step_fresh = false;
    double radius_now = 1.0;
    explicit points_model(ochip_relaxp_problem *prob) : p(prob)
    {
    }
    void begin_solve() override
    {
        scale_fixed = false;
    }
    bool has_eliminated() override
    {
        return true;
    }
    int evaluate(bool with_jac, int which, double *cost) override
    {
        ochip_ctx *ctx = p->ctx;
        hipStream_t st = ctx->stream;
        p_dev &D = p->dev;
        const int n = p->n;
        OCHIP_HIP(ctx, hipMemsetAsync(D.fail, 0, 4, st));
        hipEvent_t e0, e1;
        ochip_prof_begin(ctx, OCHIP_K_RELAX_EVAL, &e0, &e1);
        if (D.n_obs)
            hipLaunchKernelGGL(obs_kernel, dim3((D.n_obs + 63) / 64), dim3(64), 0, st, D, which, with_jac ? 1 : 0);
        if (D.mono_w > 0)
            hipLaunchKernelGGL(mono_kernel, dim3(1), dim3(1), 0, st, D, which);
        if (with_jac)
        {
            if (D.n_points)
                hipLaunchKernelGGL(point_kernel, dim3((D.n_points + 255) / 256), dim3(256), 0, st, D);
            if (n > 0)
            {
                OCHIP_HIP(ctx, hipMemsetAsync(p->sys.A, 0, p->sys.matrix_bytes(), st));
                OCHIP_HIP(ctx, hipMemsetAsync(p->sys.g, 0, (size_t)n * 8, st));
                if (D.n_groups)
                {
                    hipLaunchKernelGGL(group_u_kernel, dim3(D.n_groups), dim3(128), 0, st, D);
                    hipLaunchKernelGGL(apply_cam_kernel, dim3((3 * D.n_cams + 255) / 256), dim3(256), 0, st, D, p->cam_grp_off, p->cam_grp,
                                       p->sys.matA(), p->sys.g, n, 1.0);
                    hipLaunchKernelGGL(apply_lens_kernel, dim3(1), dim3(KI), 0, st, D, p->sys.matA(), p->sys.g, n, 1.0);
                }
                if (D.mono_w > 0)
                    hipLaunchKernelGGL(mono_apply_kernel, dim3(1), dim3(1), 0, st, D, p->sys.matA(), p->sys.g, n);
            }
        }
        ochip_prof_end(ctx, OCHIP_K_RELAX_EVAL, e0, e1);
        hipLaunchKernelGGL(p_reduce_kernel, dim3(1), dim3(LM_TG), 0, st, D, with_jac ? 1 : 0, p->sys.scal);
        OCHIP_HIP(ctx, hipGetLastError());
        // (read-backs into the system's page-locked block: a copy to pageable memory would make the host wait for it)
        double *const h0 = p->sys.box + lm_system::BOX_COST;
        int32_t *const hfail = reinterpret_cast<int32_t *>(p->sys.box + lm_system::BOX_FAILS);
        OCHIP_HIP(ctx, hipMemcpyAsync(h0, p->sys.scal, 8, hipMemcpyDeviceToHost, st));
        OCHIP_HIP(ctx, hipMemcpyAsync(hfail, D.fail, 4, hipMemcpyDeviceToHost, st));
        if (before_wait)
            before_wait();
        OCHIP_HIP(ctx, ochip_stream_wait(ctx, st));
        *cost = *h0;
        return *hfail ? 1 : 0;
    }
    int gradient_max_extra(double *out) override
    {
        double v = 0;
        OCHIP_HIP(p->ctx, hipMemcpy(&v, p->sys.scal + 5, 8, hipMemcpyDeviceToHost));
        *out = v;
        return OCHIP_OK;
    }
    void launch_schur(double radius, const double *scale, lm_matrix Wm, int n, int *fail) override
    {
        hipStream_t st = p->ctx->stream;
        p_dev &D = p->dev;
        radius_now = radius;
        step_fresh = true;
        if (D.n_points)
            hipLaunchKernelGGL(point_prepare_kernel, dim3((D.n_points + 255) / 256), dim3(256), 0, st, D, radius, scale_fixed ? 0 : 1, fail);
        scale_fixed = true;
        if (n > 0 && D.n_groups)
        {
            hipLaunchKernelGGL(group_schur_kernel, dim3(D.n_groups), dim3(128), 0, st, D, scale);
            hipLaunchKernelGGL(apply_cam_kernel, dim3((3 * D.n_cams + 255) / 256), dim3(256), 0, st, D, p->cam_grp_off, p->cam_grp, Wm,
                               (double *)nullptr, n, -1.0);
            hipLaunchKernelGGL(apply_lens_kernel, dim3(1), dim3(KI), 0, st, D, Wm, (double *)nullptr, n, -1.0);
        }
    }
    void launch_candidate(const double *y, const double *scale, double alpha, double *scal) override
    {
        // the first call after a solve back-substitutes the points (and keeps their full step); the line search's later calls
        // rescale that step - by then the kept Jacobians may be those of a trial point
        hipLaunchKernelGGL(p_candidate_kernel, dim3(1), dim3(LM_TG), 0, p->ctx->stream, p->dev, scale, y, alpha, radius_now,
                           step_fresh ? 1 : 0, scal);
        step_fresh = false;
    }
    int slope_extra(bool from_candidate, double *out) override
    {
        // from_candidate: the value launch_candidate left (gradient of the current point); otherwise recompute with the
        // gradient of the Jacobian evaluated last
        ochip_ctx *ctx = p->ctx;
        if (!from_candidate)
            hipLaunchKernelGGL(p_slope_kernel, dim3(1), dim3(LM_TG), 0, ctx->stream, p->dev, p->sys.scal);
        double v = 0;
        OCHIP_HIP(ctx, hipMemcpyAsync(&v, p->sys.scal + 6, 8, hipMemcpyDeviceToHost, ctx->stream));
        OCHIP_HIP(ctx, ochip_stream_wait(ctx, ctx->stream));
        *out = v;
        return OCHIP_OK;
    }
    void launch_accept() override
    {
        const uint32_t m = std::max<uint32_t>(std::max(p->n_cams * 4, p->n_points * 3), 8);
        hipLaunchKernelGGL(p_accept_kernel, dim3((m + 255) / 256), dim3(256), 0, p->ctx->stream, p->dev);
    }
    void launch_normalize() override
    {
        if (p->n_cams)
            hipLaunchKernelGGL(p_normalize_kernel, dim3((p->n_cams + 255) / 256), dim3(256), 0, p->ctx->stream, p->dev, p->cam_optimize_dev);
    }
    int x_norm(double *out) override
    {
        ochip_ctx *ctx = p->ctx;
        std::vector<double> q((size_t)p->n_cams * 4), X((size_t)p->n_points * 3);
        double m[8], s = 0;
        if (p->n_cams)
            OCHIP_HIP(ctx, hipMemcpy(q.data(), p->dev.cam_q, q.size() * 8, hipMemcpyDeviceToHost));
        if (p->n_points)
            OCHIP_HIP(ctx, hipMemcpy(X.data(), p->dev.X, X.size() * 8, hipMemcpyDeviceToHost));
        OCHIP_HIP(ctx, hipMemcpy(m, p->dev.model, 64, hipMemcpyDeviceToHost));
        for (uint32_t c = 0; c < p->n_cams; c++)
            if (p->cam_t[c] >= 0)
                for (int k = 0; k < 4; k++)
                    s += q[c * 4 + k] * q[c * 4 + k];
        for (double v : X)
            s += v * v;
        const int first[4] = {0, 1, 3, 6}, size[4] = {1, 2, 3, 2};
        for (int b = 0; b < 4; b++)
        {
            bool variable = false;
            for (int k = 0; k < size[b]; k++)
                variable = variable || p->dev.lens_t[first[b] + k] >= 0;
            if (variable)
                for (int k = 0; k < size[b]; k++)
                    s += m[first[b] + k] * m[first[b] + k];
        }
        *out = std::sqrt(s);
        return OCHIP_OK;
    }
    int num_residual_blocks() override
    {
        return (int)p->dev.n_obs + (p->dev.mono_w > 0 ? 1 : 0);
    }
    bool is_constrained() override
    {
        return p->dev.f_bounded && p->dev.lens_t[0] >= 0;
    }
};

} // namespace

extern "C"
{

int ochip_relaxp_problem_create(ochip_ctx *ctx, const ochip_relaxp_desc *d, ochip_relaxp_problem **out)
{
    if (!ctx || !d || !out)
        return OCHIP_EINVAL;
    *out = nullptr;
    OCHIP_HIP(ctx, hipSetDevice(ctx->device));
    if (d->functor < 0 || d->functor > 3)
        return ochip_fail(ctx, OCHIP_EINVAL, "relax (points): unknown functor %d", d->functor);
    for (uint32_t g = 0; g < d->n_groups; g++)
    {
        if (d->grp_cam[2 * g] >= d->n_cams || d->grp_cam[2 * g + 1] >= d->n_cams || d->grp_cam[2 * g] == d->grp_cam[2 * g + 1])
            return ochip_fail(ctx, OCHIP_EINVAL, "relax (points): group %u has bad cameras", g);
        if (d->grp_first[g + 1] < d->grp_first[g] || d->grp_first[g + 1] > d->n_points)
            return ochip_fail(ctx, OCHIP_EINVAL, "relax (points): group %u has a bad point range", g);
    }
    if (d->n_groups && (d->grp_first[0] != 0 || d->grp_first[d->n_groups] != d->n_points))
        return ochip_fail(ctx, OCHIP_EINVAL, "relax (points): the groups do not cover the points");
    auto *p = new (std::nothrow) ochip_relaxp_problem();
    if (!p)
        return ochip_fail(ctx, OCHIP_ENOMEM, "host allocation failed");
    p->ctx = ctx;
    p->n_cams = d->n_cams;
    p->n_points = d->n_points;
    p->n_groups = d->n_groups;
    p->functor = d->functor;
    p->cam_optimize.assign(d->cam_optimize, d->cam_optimize + d->n_cams);
    p->cam_used.assign(d->n_cams, 0);
    p->opt_f = d->opt_focal;
    p->opt_pp = d->opt_principal;
    p->n_k_free = std::min<uint8_t>(d->n_radial_free, 3);
    std::vector<uint32_t> pt_group(d->n_points, 0);
    std::vector<std::vector<uint32_t>> per_cam(d->n_cams);
    for (uint32_t g = 0; g < d->n_groups; g++)
    {
        for (uint32_t q = d->grp_first[g]; q < d->grp_first[g + 1]; q++)
            pt_group[q] = g;
        if (d->grp_first[g + 1] > d->grp_first[g])
            for (uint32_t s = 0; s < 2; s++)
            {
                p->cam_used[d->grp_cam[2 * g + s]] = 1;
                per_cam[d->grp_cam[2 * g + s]].push_back(g << 1 | s);
            }
    }
    std::vector<uint32_t> cam_grp_off(d->n_cams + 1, 0), cam_grp;
    for (uint32_t c = 0; c < d->n_cams; c++)
    {
        cam_grp.insert(cam_grp.end(), per_cam[c].begin(), per_cam[c].end());
        cam_grp_off[c + 1] = (uint32_t)cam_grp.size();
    }
    p_dev &D = p->dev;
    D.n_cams = d->n_cams;
    D.n_points = d->n_points;
    D.n_groups = d->n_groups;
    D.n_obs = 2 * d->n_points;
    D.functor = d->functor;
    D.huber_a = d->huber_a;
    D.f_lo = d->focal_lo;
    D.f_hi = d->focal_hi;
    D.f_bounded = d->opt_focal ? 1 : 0;
    D.mono_w = d->mono_observations > 0 && d->functor >= 2 ? std::sqrt(d->mono_observations / 10.0) : 0.0;
    D.mono_rmax = d->mono_r_max;
    int rc = OCHIP_OK;
    auto chk = [&](int r) {
        if (rc == OCHIP_OK)
            rc = r;
    };
    const size_t np = d->n_points, nobs = 2 * np;
    chk(up(p, &D.cam_pos, d->cam_pos, (size_t)d->n_cams * 3));
    chk(up(p, &D.cam_q, d->cam_q, (size_t)d->n_cams * 4));
    chk(up(p, &D.cam_q2, d->cam_q, (size_t)d->n_cams * 4));
    chk(up(p, &D.model, d->model, 8));
    chk(up(p, &D.model2, d->model, 8));
    chk(up(p, &D.X, d->point_xyz, np * 3));
    chk(up(p, &D.X2, d->point_xyz, np * 3));
    chk(up<int32_t>(p, &D.cam_t, nullptr, d->n_cams));
    chk(up(p, &D.grp_first, d->grp_first, (size_t)d->n_groups + 1));
    chk(up(p, &D.grp_cam, d->grp_cam, (size_t)d->n_groups * 2));
    chk(up(p, &D.pt_group, pt_group.data(), np));
    chk(up(p, &D.obs_px, d->obs_px, nobs * 2));
    chk(up<double>(p, &D.obs_J, nullptr, nobs * 2 * JW));
    chk(up<double>(p, &D.obs_r, nullptr, nobs * 2));
    chk(up<double>(p, &D.obs_cost, nullptr, nobs));
    chk(up<double>(p, &D.pt_V, nullptr, np * 6));
    chk(up<double>(p, &D.pt_g, nullptr, np * 3));
    chk(up<double>(p, &D.pt_scale, nullptr, np * 3));
    chk(up<double>(p, &D.pt_Vinv, nullptr, np * 6));
    chk(up<double>(p, &D.pt_d, nullptr, np * 3));
    chk(up<double>(p, &D.rec, nullptr, (size_t)std::max<uint32_t>(d->n_groups, 1) * RLEN));
    chk(up<double>(p, &D.mono, nullptr, 16));
    chk(up<int32_t>(p, &D.fail, nullptr, 1));
    chk(up(p, &p->cam_grp_off, cam_grp_off.data(), cam_grp_off.size()));
    chk(up(p, &p->cam_grp, cam_grp.data(), cam_grp.size()));
    chk(up(p, &p->cam_optimize_dev, p->cam_optimize.data(), p->cam_optimize.size()));
    if (rc == OCHIP_OK && np && hipMemset(D.pt_d, 0, np * 24) != hipSuccess)
        rc = ochip_fail(ctx, OCHIP_EHIP, "hipMemset failed");
    if (rc == OCHIP_OK)
        rc = assign(p);
    if (rc != OCHIP_OK)
    {
        ochip_relaxp_problem_destroy(p);
        return rc;
    }
    *out = p;
    return OCHIP_OK;
}

void ochip_relaxp_problem_destroy(ochip_relaxp_problem *p)
{
    if (!p)
        return;
    (void)hipSetDevice(p->ctx->device);
    (void)ochip_stream_wait(p->ctx, p->ctx->stream);
    for (auto &a : p->allocs)
        ochip_pool_put(p->ctx, a.first, a.second);
    delete p;
}

int ochip_relaxp_set_structure_only(ochip_relaxp_problem *p, int on)
{
    if (!p)
        return OCHIP_EINVAL;
    OCHIP_HIP(p->ctx, hipSetDevice(p->ctx->device));
    p->structure_only = on != 0;
    return assign(p);
}

int ochip_relaxp_solve(ochip_relaxp_problem *p, const ochip_relax_options *opt, ochip_relax_summary *sum)
{
    if (!p || !opt || !sum)
        return OCHIP_EINVAL;
    ochip_ctx *ctx = p->ctx;
    OCHIP_HIP(ctx, hipSetDevice(ctx->device));
    std::memset(sum, 0, sizeof *sum);
    sum->num_parameters = p->n + 3 * (int)p->n_points;
    sum->num_residual_blocks = (int)p->dev.n_obs + (p->dev.mono_w > 0 ? 1 : 0);
    if (p->n == 0 && p->n_points == 0)
    {
        sum->termination = OCHIP_RELAX_NO_PARAMETERS;
        return OCHIP_OK;
    }
    points_model M(p);
    return lm_solve(p->sys, M, opt, sum);
}

int ochip_relaxp_get_state(ochip_relaxp_problem *p, double *cam_q, double *point_xyz, double *model)
{
    if (!p)
        return OCHIP_EINVAL;
    ochip_ctx *ctx = p->ctx;
    OCHIP_HIP(ctx, ochip_stream_wait(ctx, ctx->stream));
    if (cam_q && p->n_cams)
        OCHIP_HIP(ctx, hipMemcpy(cam_q, p->dev.cam_q, (size_t)p->n_cams * 32, hipMemcpyDeviceToHost));
    if (point_xyz && p->n_points)
        OCHIP_HIP(ctx, hipMemcpy(point_xyz, p->dev.X, (size_t)p->n_points * 24, hipMemcpyDeviceToHost));
    if (model)
        OCHIP_HIP(ctx, hipMemcpy(model, p->dev.model, 64, hipMemcpyDeviceToHost));
    return OCHIP_OK;
}

} // extern "C"
