// libochip.so — homography RANSAC for every image pair of a link batch, one wavefront per pair (gfx950).
//
// Replaces, per directed pair, the whole body of ransac<homography_model>
// (src/model_inliers/ransac.cpp:53-257) together with homography_model::fit / fitInliers / error /
// evaluate / checkSampleDegeneracy (src/model_inliers/homography_model.cpp:19-136), and the ray
// normalisation half of distort_keypoints (src/distort/distort_keypoints.cpp:48-103, hoisted to once
// per image because image_to_3d only depends on the keypoint and its camera model).
//
// Why one wave per pair: the reference loop is sequential in three places that decide its results
// bit for bit — the minstd_rand0 sample stream, the SPRT early exit that compares a running fp64 MSAC
// sum (in shuffled evaluation order) with the best score so far, and the local-optimisation /
// adaptive-termination chain that follows every improvement.  A wavefront keeps that control flow
// wave-uniform and parallelises inside each step: 64 symmetric-transfer errors per chunk, ballot for
// the SPRT exit and the inlier masks, lane-strided rows for the (2n+1)x9 full-pivot LU of
// fitInliers.  The fp64 score is accumulated strictly in evaluation order (one add per inlier), so
// scores, inlier sets and H are bit-identical to the CPU restatement; pairs are independent, so the
// chip is filled with thousands of resident waves.  All arithmetic is fp64 with -ffp-contract=off
// (the reference is built without FMA); device division and sqrt are correctly rounded
// (tests/test_gpu_fp64.py).
//
// libstdc++ pieces on the result path: std::default_random_engine (= minstd_rand0) and
// std::uniform_int_distribution<size_t> are re-implemented here exactly (bits/random.tcc,
// bits/uniform_int_dist.h "downscaling" branch); std::sort (PROSAC order) and std::shuffle
// (evaluation order) are executed on the host with the real libstdc++ and handed in.
#include "ctx.hpp"
#include "undistort.hpp"

#include <cmath>

namespace
{

constexpr int W = 64;
constexpr uint32_t MIN_ITERATIONS = 20, MAX_ITERATIONS = 10000, MAX_INNER_ITERATIONS = 5;

struct rays_view
{
    const double *rays;      // [total][3]
    const uint64_t *img_off; // per image
};

__device__ __forceinline__ double bcast(double v, int src_lane)
{
    union {
        double d;
        uint32_t u[2];
    } x;
    x.d = v;
    x.u[0] = __builtin_amdgcn_readlane(x.u[0], src_lane);
    x.u[1] = __builtin_amdgcn_readlane(x.u[1], src_lane);
    return x.d;
}

__device__ __forceinline__ uint32_t minstd_next(uint32_t &x) // std::minstd_rand0: x = 16807 x mod (2^31 - 1)
{
    x = (uint32_t)(((uint64_t)x * 16807ull) % 2147483647ull);
    return x;
}

// std::uniform_int_distribution<size_t>(0, hi)(minstd_rand0) — bits/uniform_int_dist.h, urngrange > urange
__device__ __forceinline__ uint32_t uniform_int(uint32_t &x, uint32_t hi)
{
    const uint64_t urngrange = 2147483645ull; // max() - min() = 2147483646 - 1
    const uint64_t uerange = (uint64_t)hi + 1;
    const uint64_t scaling = urngrange / uerange;
    const uint64_t past = uerange * scaling;
    uint64_t ret;
    do
        ret = (uint64_t)minstd_next(x) - 1ull;
    while (ret >= past);
    return (uint32_t)(ret / scaling);
}

struct model_t // homography_model state, wave-uniform
{
    double H[9], Hi[9];
};

__device__ __forceinline__ void set_nan(model_t &m)
{
    for (int i = 0; i < 9; i++)
        m.H[i] = m.Hi[i] = __builtin_nan("");
}

// Eigen compute_inverse_size3 (cofactors, 1/det) — same order as the restatement
__device__ __forceinline__ double cof(const double *m, int i, int j)
{
    const int i1 = (i + 1) % 3, i2 = (i + 2) % 3, j1 = (j + 1) % 3, j2 = (j + 2) % 3;
    return m[i1 * 3 + j1] * m[i2 * 3 + j2] - m[i1 * 3 + j2] * m[i2 * 3 + j1];
}

__device__ __forceinline__ void model_from_solution(model_t &m, const double *h) // homography_model.cpp:45-49
{
    const double s = h[8];
    for (int i = 0; i < 9; i++)
        m.H[i] = h[i] / s;
    const double c00 = cof(m.H, 0, 0), c10 = cof(m.H, 1, 0), c20 = cof(m.H, 2, 0);
    const double d = c00 * m.H[0] + c10 * m.H[3] + c20 * m.H[6];
    const double invdet = 1.0 / d;
    for (int r = 0; r < 3; r++)
        for (int c = 0; c < 3; c++)
            m.Hi[r * 3 + c] = cof(m.H, c, r) * invdet;
}

// homography_model::error (homography_model.cpp:89-97) on pre-divided coordinates (x/z, y/z, z/z == 1)
__device__ __forceinline__ double transfer_error(const model_t &m, double x1, double y1, double x2, double y2)
{
    const double fx_ = m.H[0] * x1 + m.H[1] * y1 + m.H[2] * 1.0;
    const double fy_ = m.H[3] * x1 + m.H[4] * y1 + m.H[5] * 1.0;
    const double fz_ = m.H[6] * x1 + m.H[7] * y1 + m.H[8] * 1.0;
    const double bx_ = m.Hi[0] * x2 + m.Hi[1] * y2 + m.Hi[2] * 1.0;
    const double by_ = m.Hi[3] * x2 + m.Hi[4] * y2 + m.Hi[5] * 1.0;
    const double bz_ = m.Hi[6] * x2 + m.Hi[7] * y2 + m.Hi[8] * 1.0;
    const double fx = fx_ / fz_ - x2, fy = fy_ / fz_ - y2;
    const double bx = bx_ / bz_ - x1, by = by_ / bz_ - y1;
    const double fwd = fx * fx + fy * fy;
    const double bwd = bx * bx + by * by;
    return sqrt((fwd + bwd) / 2.0);
}

struct pair_data // per-pair scratch in HBM (L2 resident while the pair is being processed)
{
    const double *x1, *y1, *x2, *y2; // [M] normalised coordinates
    uint8_t *cand, *inl;             // [M] candidate / current inlier flags, indexed by correspondence
    double *P;                       // column-major (2M+1) x 9 system of fitInliers
    uint32_t M;
};

// MSAC scoring of one model.  ORDERED: walk `order` (the shuffled eval_order) and apply the SPRT early
// exit of ransac.cpp:187-203; otherwise natural order, no exit (homography_model::evaluate :99-118).
// The running sum is accumulated one inlier at a time in walk order.  Returns the score; *rejected.
template <bool ORDERED>
__device__ double score_model(const model_t &m, const pair_data &pd, const uint32_t *__restrict__ order, uint8_t *flags,
                              double thr, double best_score, bool *rejected, uint32_t *n_inliers)
{
    const int lane = threadIdx.x;
    const uint32_t M = pd.M;
    double s = 0;
    uint32_t count = 0;
    *rejected = false;
    for (uint32_t base = 0; base < M; base += W)
    {
        const uint32_t pos = base + lane;
        const bool valid = pos < M;
        const uint32_t idx = valid ? (ORDERED ? order[pos] : pos) : 0;
        const double e = transfer_error(m, pd.x1[idx], pd.y1[idx], pd.x2[idx], pd.y2[idx]);
        const bool inl = valid && (e < thr);
        double term = 0;
        if (inl)
        {
            const double ratio = e / thr;
            term = 1.0 - ratio * ratio;
        }
        if (valid)
            flags[idx] = inl ? 1 : 0;
        unsigned long long mask = __ballot(inl);
        count += __popcll(mask);
        double pref = s; // running sum as seen right after this lane's element
        while (mask)
        {
            const int l = __builtin_ctzll(mask);
            mask &= mask - 1;
            s = s + bcast(term, l);
            if (lane >= l)
                pref = s;
        }
        if (ORDERED)
        {
            const uint32_t checked = pos + 1;
            const bool rej = valid && checked > 20 && best_score > 0 &&
                             pref < best_score * (double)checked / (double)M * 0.6;
            if (__ballot(rej))
            {
                *rejected = true;
                return s;
            }
        }
    }
    *n_inliers = count;
    return s;
}

// Eigen FullPivLU<Matrix<double, rows, 9>>::solve(e_last) with the wave cooperating: lanes stride over
// rows.  A is column-major with leading dimension ld (generic pointer: LDS for the 9x9 fit, HBM for
// fitInliers).  Pivot search order / ties, rank threshold, substitution order: exactly the restated
// Eigen algorithm of the oracle (column-by-column scan, strict '>').
__device__ void full_piv_lu_solve9(double *A, uint32_t rows, uint32_t ld, double *sol /*[9], uniform*/)
{
    const int lane = threadIdx.x;
    const uint32_t cols = 9;
    const uint32_t size = rows < cols ? rows : cols;
    uint32_t rowT[9], colT[9];
    uint32_t nonzero_pivots = size;
    double maxpivot = 0;

    for (uint32_t k = 0; k < size; k++)
    {
        __syncthreads();
        // ---- pivot search over the bottom-right corner
        double bv = -1.0;
        uint32_t bi = k, bj = k;
        for (uint32_t j = k; j < cols; j++)
            for (uint32_t i = k + lane; i < rows; i += W)
            {
                const double v = fabs(A[(size_t)j * ld + i]);
                if (v > bv) // within a lane (j, i) ascend, so strict '>' keeps the first maximum
                {
                    bv = v;
                    bi = i;
                    bj = j;
                }
            }
        for (int off = 32; off >= 1; off >>= 1)
        {
            const double ov = __shfl_xor(bv, off);
            const uint32_t oi = __shfl_xor(bi, off), oj = __shfl_xor(bj, off);
            const bool better = ov > bv || (ov == bv && (oj < bj || (oj == bj && oi < bi)));
            if (better)
            {
                bv = ov;
                bi = oi;
                bj = oj;
            }
        }
        const double akk = A[(size_t)k * ld + k];
        if (akk != akk) // a NaN in the first scanned cell sticks (nothing compares greater than NaN)
        {
            bv = akk;
            bi = k;
            bj = k;
        }
        else if (bv < 0) // every candidate was NaN except none: keep (k,k)
        {
            bv = fabs(akk);
            bi = k;
            bj = k;
        }
        if (bv == 0.0)
        {
            nonzero_pivots = k;
            for (uint32_t i = k; i < size; i++)
            {
                rowT[i] = i;
                colT[i] = i;
            }
            break;
        }
        if (bv > maxpivot)
            maxpivot = bv;
        rowT[k] = bi;
        colT[k] = bj;
        // ---- row swap k <-> bi (lanes over the 9 columns), then column swap k <-> bj (lanes over rows)
        if (k != bi && lane < (int)cols)
        {
            const double t = A[(size_t)lane * ld + k];
            A[(size_t)lane * ld + k] = A[(size_t)lane * ld + bi];
            A[(size_t)lane * ld + bi] = t;
        }
        __syncthreads();
        if (k != bj)
            for (uint32_t i = lane; i < rows; i += W)
            {
                const double t = A[(size_t)k * ld + i];
                A[(size_t)k * ld + i] = A[(size_t)bj * ld + i];
                A[(size_t)bj * ld + i] = t;
            }
        __syncthreads();
        // ---- eliminate: col(k).tail /= pivot; block(k+1,k+1) -= col(k).tail * row(k).tail
        const double p = A[(size_t)k * ld + k];
        double rowk[9];
        for (uint32_t j = k + 1; j < cols; j++)
            rowk[j] = A[(size_t)j * ld + k];
        __syncthreads();
        if (k < rows - 1)
            for (uint32_t i = k + 1 + lane; i < rows; i += W)
            {
                const double l = A[(size_t)k * ld + i] / p;
                A[(size_t)k * ld + i] = l;
                if (k < size - 1)
                    for (uint32_t j = k + 1; j < cols; j++)
                        A[(size_t)j * ld + i] -= l * rowk[j];
            }
    }
    __syncthreads();

    // ---- rank, then the solve steps on the leading 9x9 block (uniform work)
    const double premult = fabs(maxpivot) * (2.220446049250313e-16 * (double)size);
    uint32_t rank = 0;
    for (uint32_t i = 0; i < nonzero_pivots; i++)
        rank += (fabs(A[(size_t)i * ld + i]) > premult) ? 1 : 0;
    for (int i = 0; i < 9; i++)
        sol[i] = 0;
    if (rank == 0)
        return;
    // c = P * e_last: follow the single 1 through the row transpositions
    uint32_t pos = rows - 1;
    for (uint32_t k = 0; k < size; k++)
    {
        if (pos == k)
            pos = rowT[k];
        else if (pos == rowT[k])
            pos = k;
    }
    double c[9];
    for (uint32_t i = 0; i < 9; i++)
        c[i] = (i == pos) ? 1.0 : 0.0;
    for (uint32_t j = 0; j < size; j++) // unit-lower forward substitution, column oriented
    {
        const double cj = c[j];
        for (uint32_t i = j + 1; i < size; i++)
            c[i] -= cj * A[(size_t)j * ld + i];
    }
    for (uint32_t jj = rank; jj-- > 0;) // upper back substitution, column oriented
    {
        c[jj] /= A[(size_t)jj * ld + jj];
        const double cj = c[jj];
        for (uint32_t i = 0; i < jj; i++)
            c[i] -= cj * A[(size_t)jj * ld + i];
    }
    uint32_t perm[9] = {0, 1, 2, 3, 4, 5, 6, 7, 8};
    for (uint32_t k = 0; k < size; k++)
    {
        const uint32_t t = perm[k];
        perm[k] = perm[colT[k]];
        perm[colT[k]] = t;
    }
    for (uint32_t i = 0; i < rank; i++)
        sol[perm[i]] = c[i];
}

// the two DLT rows of one correspondence (homography_model.cpp:26-35), written column-major
__device__ __forceinline__ void write_dlt_rows(double *A, uint32_t ld, uint32_t r, double x, double y, double x_,
                                               double y_)
{
    const double a[9] = {-x, -y, -1, 0, 0, 0, x * x_, y * x_, x_};
    const double b[9] = {0, 0, 0, -x, -y, -1, x * y_, y * y_, y_};
    for (int j = 0; j < 9; j++)
    {
        A[(size_t)j * ld + r] = a[j];
        A[(size_t)j * ld + r + 1] = b[j];
    }
}

template <int OCC> __global__ __launch_bounds__(W, OCC) void ransac_homography_kernel(
    const ochip_ransac_job *__restrict__ jobs, const ochip_ransac_match *__restrict__ matches,
    const uint32_t *__restrict__ sorted_idx_all, const uint32_t *__restrict__ eval_order_all, rays_view rv,
    double *__restrict__ coord_scratch /*4 x total*/, uint8_t *__restrict__ flag_scratch /*total*/,
    double *__restrict__ P_scratch /*9 x (2 total + n_jobs)*/, uint64_t total, double thr,
    ochip_ransac_result *__restrict__ results, uint8_t *__restrict__ inliers_out)
{
    __shared__ double P9[81];
    const int lane = threadIdx.x;
    const uint32_t job_id = blockIdx.x;
    const ochip_ransac_job job = jobs[job_id];
    const uint32_t M = job.n;
    const uint64_t mo = job.match_offset;

    ochip_ransac_result res;
    for (int i = 0; i < 9; i++)
        res.H[i] = __builtin_nan("");
    res.score = 0;
    res.iterations = 0;
    res.n_inliers = 0;
    res.improvements = 0;
    res.reserved = 0;

    uint8_t *inl = inliers_out + mo;
    for (uint32_t i = lane; i < M; i += W)
        inl[i] = 0;
    if (M < 4) // ransac.cpp:69-72
    {
        if (lane == 0)
            results[job_id] = res;
        return;
    }

    // ---- prologue: gather the unit rays of the matched keypoints, divide by z once (error() and fit()
    //      both start from measurement / measurement.z), detect has_quality (ransac.cpp:74-82)
    pair_data pd;
    double *cx1 = coord_scratch + mo, *cy1 = coord_scratch + total + mo, *cx2 = coord_scratch + 2 * total + mo,
           *cy2 = coord_scratch + 3 * total + mo;
    const double *r1 = rv.rays + rv.img_off[job.image_1] * 3, *r2 = rv.rays + rv.img_off[job.image_2] * 3;
    bool hq_lane = false;
    for (uint32_t i = lane; i < M; i += W)
    {
        const ochip_ransac_match mt = matches[mo + i];
        const double ax = r1[(size_t)mt.k1 * 3], ay = r1[(size_t)mt.k1 * 3 + 1], az = r1[(size_t)mt.k1 * 3 + 2];
        const double bx = r2[(size_t)mt.k2 * 3], by = r2[(size_t)mt.k2 * 3 + 1], bz = r2[(size_t)mt.k2 * 3 + 2];
        cx1[i] = ax / az;
        cy1[i] = ay / az;
        cx2[i] = bx / bz;
        cy2[i] = by / bz;
        hq_lane |= (mt.count != 0); // quality = count * (1/486) != 0  <=>  count != 0
    }
    const bool has_quality = __ballot(hq_lane) != 0;
    __syncthreads();
    pd.x1 = cx1;
    pd.y1 = cy1;
    pd.x2 = cx2;
    pd.y2 = cy2;
    pd.cand = flag_scratch + mo;
    pd.inl = inl;
    pd.P = P_scratch + 9 * (2 * mo + job_id);
    pd.M = M;

    const uint32_t *sorted_idx = sorted_idx_all + mo;
    const uint32_t *eval_order = eval_order_all + job.eval_offset;

    model_t model, best_model;
    set_nan(model);
    set_nan(best_model);
    double best_score = 0;
    uint32_t rng = job.rng_state;
    uint32_t prosac_n = has_quality ? 4u : M;
    uint32_t probability_iterations = MAX_ITERATIONS;
    const double log_1m_p = log(1 - 0.999);
    uint32_t it = 0;

    for (; it < probability_iterations; it++)
    {
        if (has_quality && prosac_n < M && it > 0 && it % 10 == 0)
            prosac_n++;

        // ---- minimal sample (ransac.cpp:104-154)
        uint32_t s4[4];
        if (has_quality && prosac_n < M && prosac_n > 4)
        {
            s4[0] = sorted_idx[prosac_n - 1];
            for (int j = 1; j < 4; j++)
            {
                uint32_t cand;
                bool unique;
                do
                {
                    cand = sorted_idx[uniform_int(rng, prosac_n - 2)];
                    unique = true;
                    for (int k = 0; k < j; k++)
                        if (s4[k] == cand)
                        {
                            unique = false;
                            break;
                        }
                } while (!unique);
                s4[j] = cand;
            }
        }
        else
        {
            const uint32_t pool = has_quality ? prosac_n : M;
            for (int j = 0; j < 4; j++)
            {
                uint32_t cand;
                bool unique;
                do
                {
                    const uint32_t c = uniform_int(rng, pool - 1);
                    cand = has_quality ? sorted_idx[c] : c;
                    unique = true;
                    for (int k = 0; k < j; k++)
                        if (s4[k] == cand)
                        {
                            unique = false;
                            break;
                        }
                } while (!unique);
                s4[j] = cand;
            }
        }

        // ---- checkSampleDegeneracy (homography_model.cpp:120-136) on measurement1.hnormalized()
        double px[4], py[4], qx[4], qy[4];
        for (int j = 0; j < 4; j++)
        {
            px[j] = pd.x1[s4[j]];
            py[j] = pd.y1[s4[j]];
            qx[j] = pd.x2[s4[j]];
            qy[j] = pd.y2[s4[j]];
        }
        bool degenerate = false;
        for (int a = 0; a < 4; a++)
            for (int b = a + 1; b < 4; b++)
                for (int c = b + 1; c < 4; c++)
                {
                    const double v1x = px[b] - px[a], v1y = py[b] - py[a];
                    const double v2x = px[c] - px[a], v2y = py[c] - py[a];
                    if (fabs(v1x * v2y - v1y * v2x) < 1e-10)
                        degenerate = true;
                }
        if (degenerate)
            continue;

        // ---- fit (homography_model.cpp:19-50): 9x9 DLT system in LDS
        __syncthreads();
        if (lane < 4)
            write_dlt_rows(P9, 9, 2 * lane, px[lane], py[lane], qx[lane], qy[lane]);
        if (lane < 9)
            P9[lane * 9 + 8] = lane == 8 ? 1.0 : 0.0;
        double sol[9];
        full_piv_lu_solve9(P9, 9, 9, sol);
        model_from_solution(model, sol);

        // ---- SPRT-pruned MSAC scoring in shuffled order (ransac.cpp:177-205)
        bool rejected;
        uint32_t n_inl = 0;
        const double score = score_model<true>(model, pd, eval_order, pd.cand, thr, best_score, &rejected, &n_inl);
        if (rejected)
            continue;

        if (score > best_score)
        {
            res.improvements++;
            best_model = model;
            best_score = score;
            __syncthreads();
            for (uint32_t i = lane; i < M; i += W)
                pd.inl[i] = pd.cand[i];
            __syncthreads();

            // local optimisation: fitInliers + evaluate, up to MAX_INNER_ITERATIONS (ransac.cpp:224-245)
            for (uint32_t j = 0; j < MAX_INNER_ITERATIONS; j++)
            {
                // build the (2 n_inl + 1) x 9 system in index order (homography_model.cpp:52-79)
                uint32_t n_in = 0;
                for (uint32_t base = 0; base < M; base += W)
                {
                    const uint32_t i = base + lane;
                    n_in += __popcll(__ballot(i < M && pd.inl[i]));
                }
                const uint32_t rows = 2 * n_in + 1, ld = rows;
                uint32_t before = 0;
                for (uint32_t base = 0; base < M; base += W)
                {
                    const uint32_t i = base + lane;
                    const bool f = i < M && pd.inl[i];
                    const unsigned long long mask = __ballot(f);
                    if (f)
                    {
                        const uint32_t r = before + __popcll(mask & ((1ull << lane) - 1ull));
                        write_dlt_rows(pd.P, ld, 2 * r, pd.x1[i], pd.y1[i], pd.x2[i], pd.y2[i]);
                    }
                    before += __popcll(mask);
                }
                if (lane < 9)
                    pd.P[(size_t)lane * ld + rows - 1] = lane == 8 ? 1.0 : 0.0;
                full_piv_lu_solve9(pd.P, rows, ld, sol);
                model_from_solution(model, sol);
                bool dummy;
                uint32_t cnt = 0;
                __syncthreads();
                const double inlier_score = score_model<false>(model, pd, nullptr, pd.inl, thr, 0.0, &dummy, &cnt);
                __syncthreads();
                if (inlier_score > best_score)
                {
                    best_model = model;
                    best_score = inlier_score;
                }
                else
                    break;
            }

            const double omega = best_score / (double)M;
            double t = omega * omega;
            const double omega_n = t * t; // fast_pow<4>
            const double log_1m_omega_n = log(1 - omega_n);
            // static_cast<size_t>(log_1m_p / log_1m_omega_n), then clamp to [MIN, MAX]
            const double q = log_1m_p / log_1m_omega_n;
            uint64_t qi;
            if (!(q == q)) // NaN -> x86 cvttsd2si gives 0x8000000000000000: huge as size_t
                qi = 0x8000000000000000ull;
            else if (q >= 9223372036854775808.0)
                qi = (q >= 18446744073709551616.0) ? 0ull : (uint64_t)q; // never reached with p = 0.999
            else if (q <= -1.0)
                qi = (uint64_t)(int64_t)q; // negative wraps like the x86 conversion
            else
                qi = (uint64_t)q;
            const uint64_t clamped = qi < MAX_ITERATIONS ? qi : MAX_ITERATIONS;
            probability_iterations = (uint32_t)(clamped > MIN_ITERATIONS ? clamped : MIN_ITERATIONS);
        }
    }

    // ---- model = best_model; return model.evaluate(matches, inliers) / matches.size()
    bool dummy;
    uint32_t cnt = 0;
    __syncthreads();
    const double final_score = score_model<false>(best_model, pd, nullptr, pd.inl, thr, 0.0, &dummy, &cnt);
    for (int i = 0; i < 9; i++)
        res.H[i] = best_model.H[i];
    res.score = final_score / (double)M;
    res.iterations = it;
    res.n_inliers = cnt;
    if (lane == 0)
        results[job_id] = res;
}

// pixel -> unit ray, distort_keypoints.cpp:68-103 (csrc/undistort.hpp: lens distortion is inverted per keypoint with
// the restated TinySolver; one thread per keypoint, the solver's <= 10 iterations are a few hundred flops)
__global__ void keypoints_to_rays_kernel(const double *__restrict__ xy, const double *__restrict__ models /*[img][8]*/,
                                         const uint32_t *__restrict__ kp_image, double *__restrict__ rays, uint64_t n)
{
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n)
        return;
    const double *m = models + (size_t)kp_image[i] * 8;
    double model8[8], kp[2] = {xy[2 * i], xy[2 * i + 1]}, ray[3];
    for (int k = 0; k < 8; k++)
        model8[k] = m[k];
    ochip_ud::image_to_3d(kp, model8, ray);
    rays[3 * i] = ray[0];
    rays[3 * i + 1] = ray[1];
    rays[3 * i + 2] = ray[2];
}

} // namespace

extern "C"
{

int ochip_upload_keypoints(ochip_ctx *ctx, uint32_t image_id, const double *xy, uint32_t n, const double *model8)
{
    if (!ctx)
        return OCHIP_EINVAL;
    if (image_id >= ctx->n_images || !ctx->img_set[image_id])
        return ochip_fail(ctx, OCHIP_ESTATE, "upload the descriptors of image %u first", image_id);
    if (n != ctx->img_n[image_id])
        return ochip_fail(ctx, OCHIP_EINVAL, "image %u: %u keypoints but %u descriptors", image_id, n,
                          ctx->img_n[image_id]);
    if (!model8 || (n && !xy))
        return ochip_fail(ctx, OCHIP_EINVAL, "NULL argument");
    OCHIP_HIP(ctx, hipSetDevice(ctx->device));
    if (!ctx->kp_store_ready)
    {
        const int rc = ochip_ensure_keypoint_store(ctx, ctx->desc_capacity, ctx->n_images);
        if (rc)
            return rc;
        ctx->kp_set.assign(ctx->n_images, 0);
    }
    const uint64_t off = ctx->img_off[image_id];
    std::vector<uint32_t> ids(n, image_id);
    if (n)
    {
        OCHIP_HIP(ctx, hipMemcpyAsync(ctx->kp_xy_dev + 2 * off, xy, (size_t)n * 16, hipMemcpyHostToDevice, ctx->stream));
        OCHIP_HIP(ctx, hipMemcpyAsync(ctx->kp_image_dev + off, ids.data(), (size_t)n * 4, hipMemcpyHostToDevice,
                                      ctx->stream));
    }
    OCHIP_HIP(ctx, hipMemcpyAsync(ctx->models_dev + (size_t)image_id * 8, model8, 64, hipMemcpyHostToDevice,
                                  ctx->stream));
    OCHIP_HIP(ctx, hipStreamSynchronize(ctx->stream));
    ctx->kp_set[image_id] = 1;
    ctx->rays_dirty = true;
    return OCHIP_OK;
}

int ochip_ransac_homography_batch(ochip_ctx *ctx, const ochip_ransac_job *jobs, uint32_t n_jobs,
                                  const ochip_ransac_match *matches, const uint32_t *sorted_idx, uint64_t total_matches,
                                  const uint32_t *eval_order, uint64_t eval_total, double inlier_threshold,
                                  ochip_ransac_result *results, uint8_t *inliers)
{
    if (!ctx)
        return OCHIP_EINVAL;
    if (n_jobs == 0)
        return OCHIP_OK;
    if (!jobs || !results || (total_matches && (!matches || !sorted_idx || !inliers)) || (eval_total && !eval_order))
        return ochip_fail(ctx, OCHIP_EINVAL, "NULL argument");
    if (!ctx->kp_store_ready)
        return ochip_fail(ctx, OCHIP_ESTATE, "ochip_upload_keypoints has not been called");
    OCHIP_HIP(ctx, hipSetDevice(ctx->device));
    for (uint32_t j = 0; j < n_jobs; j++)
    {
        const ochip_ransac_job &jb = jobs[j];
        if (jb.image_1 >= ctx->n_images || jb.image_2 >= ctx->n_images || !ctx->kp_set[jb.image_1] ||
            !ctx->kp_set[jb.image_2])
            return ochip_fail(ctx, OCHIP_ESTATE, "job %u references an image without keypoints", j);
        if (jb.match_offset + jb.n > total_matches || jb.eval_offset + jb.n > eval_total)
            return ochip_fail(ctx, OCHIP_EINVAL, "job %u: offsets exceed the arrays", j);
    }
    if (ctx->rays_dirty)
    {
        const uint64_t n = ctx->desc_used;
        if (n)
            hipLaunchKernelGGL(keypoints_to_rays_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream,
                               ctx->kp_xy_dev, ctx->models_dev, ctx->kp_image_dev, ctx->rays_dev, n);
        OCHIP_HIP(ctx, hipGetLastError());
        ctx->rays_dirty = false;
    }
    if (ctx->img_tables_dirty)
    {
        OCHIP_HIP(ctx, hipMemcpyAsync(ctx->img_off_dev, ctx->img_off.data(), (size_t)ctx->n_images * 8,
                                      hipMemcpyHostToDevice, ctx->stream));
        OCHIP_HIP(ctx, hipMemcpyAsync(ctx->img_n_dev, ctx->img_n.data(), (size_t)ctx->n_images * 4,
                                      hipMemcpyHostToDevice, ctx->stream));
        ctx->img_tables_dirty = false;
    }
    const uint64_t T = total_matches ? total_matches : 1;
    enum
    {
        S_JOBS,
        S_MATCH,
        S_SORTED,
        S_EVAL,
        S_COORD,
        S_FLAGS,
        S_P,
        S_OUT
    };
    const size_t sizes[8] = {(size_t)n_jobs * sizeof(ochip_ransac_job),
                             (size_t)T * sizeof(ochip_ransac_match),
                             (size_t)T * 4,
                             (size_t)(eval_total ? eval_total : 1) * 4,
                             (size_t)T * 32,
                             (size_t)T,
                             (size_t)(2 * T + n_jobs) * 72,
                             (size_t)n_jobs * sizeof(ochip_ransac_result) + T};
    for (int i = 0; i < 8; i++)
    {
        int rc = ochip_ensure(ctx, &ctx->scratch_dev[i], &ctx->scratch_cap[i], sizes[i]);
        if (rc)
            return rc;
    }
    OCHIP_HIP(ctx, hipMemcpyAsync(ctx->scratch_dev[S_JOBS], jobs, sizes[S_JOBS], hipMemcpyHostToDevice, ctx->stream));
    if (total_matches)
    {
        OCHIP_HIP(ctx, hipMemcpyAsync(ctx->scratch_dev[S_MATCH], matches, (size_t)total_matches * sizeof(ochip_ransac_match),
                                      hipMemcpyHostToDevice, ctx->stream));
        OCHIP_HIP(ctx, hipMemcpyAsync(ctx->scratch_dev[S_SORTED], sorted_idx, (size_t)total_matches * 4,
                                      hipMemcpyHostToDevice, ctx->stream));
    }
    if (eval_total)
        OCHIP_HIP(ctx, hipMemcpyAsync(ctx->scratch_dev[S_EVAL], eval_order, (size_t)eval_total * 4, hipMemcpyHostToDevice,
                                      ctx->stream));
    rays_view rv{ctx->rays_dev, ctx->img_off_dev};
    ochip_ransac_result *res_dev = (ochip_ransac_result *)ctx->scratch_dev[S_OUT];
    uint8_t *inl_dev = (uint8_t *)ctx->scratch_dev[S_OUT] + (size_t)n_jobs * sizeof(ochip_ransac_result);
    hipEvent_t e0, e1;
    ochip_prof_begin(ctx, OCHIP_K_RANSAC, &e0, &e1);
    static const int occ = []() {
        const char *e = getenv("OCHIP_RANSAC_OCC"); // waves per SIMD the register allocator targets (tuning knob)
        // 2: with ~1 200 matches per pair a pair's LU workspace is ~170 KB, and more than ~2 000 resident pairs push
        // the combined working set out of the 256 MB Infinity Cache (C3: 160 ms per 9 000 pairs at 2, 209 ms at 4)
        const int v = e ? atoi(e) : 2;
        return (v == 1 || v == 2 || v == 4) ? v : 2;
    }();
    auto launch = [&](auto kernel) {
        hipLaunchKernelGGL(kernel, dim3(n_jobs), dim3(W), 0, ctx->stream,
                           (const ochip_ransac_job *)ctx->scratch_dev[S_JOBS],
                           (const ochip_ransac_match *)ctx->scratch_dev[S_MATCH],
                           (const uint32_t *)ctx->scratch_dev[S_SORTED], (const uint32_t *)ctx->scratch_dev[S_EVAL], rv,
                           (double *)ctx->scratch_dev[S_COORD], (uint8_t *)ctx->scratch_dev[S_FLAGS],
                           (double *)ctx->scratch_dev[S_P], (uint64_t)T, inlier_threshold, res_dev, inl_dev);
    };
    if (occ == 1)
        launch(ransac_homography_kernel<1>);
    else if (occ == 2)
        launch(ransac_homography_kernel<2>);
    else
        launch(ransac_homography_kernel<4>);
    ochip_prof_end(ctx, OCHIP_K_RANSAC, e0, e1);
    OCHIP_HIP(ctx, hipGetLastError());
    OCHIP_HIP(ctx, hipMemcpyAsync(results, res_dev, (size_t)n_jobs * sizeof(ochip_ransac_result), hipMemcpyDeviceToHost,
                                  ctx->stream));
    if (total_matches)
        OCHIP_HIP(ctx, hipMemcpyAsync(inliers, inl_dev, (size_t)total_matches, hipMemcpyDeviceToHost, ctx->stream));
    OCHIP_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return OCHIP_OK;
}

} // extern "C"
