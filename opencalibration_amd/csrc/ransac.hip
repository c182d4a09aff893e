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
#include "decompose.hpp"

#include <cmath>

namespace
{

// The pivot search's reduction over the wavefront: the largest value, ties to the smaller (column, row) - the first maximum
// of the column-by-column scan, whatever order the candidates meet in.  Every lane ends with the result.  The six
// butterfly stages used to be four LDS-crossbar shuffles each (__shfl_xor of a double and two indices: 24 dependent
// ds_bpermute per pivot, 9 pivots per hypothesis - the critical path of a kernel that is one wavefront per image pair);
// now the indices travel as one word and the four stages inside a row of 16 lanes are DPP moves (quad permutes, then
// rotations by 4 and 8: any pairing that ends with every lane having met every other serves a commutative, associative
// choice), two stages across the rows remain shuffles.
template <int CTRL> __device__ __forceinline__ uint32_t dpp_word(uint32_t x)
{
    return (uint32_t)__builtin_amdgcn_update_dpp((int)x, (int)x, CTRL, 0xf, 0xf, false);
}
__device__ __forceinline__ void first_maximum_take(double &bv, uint32_t &key, double ov, uint32_t ok)
{
    if (ov > bv || (ov == bv && ok < key))
    {
        bv = ov;
        key = ok;
    }
}
template <int CTRL> __device__ __forceinline__ void first_maximum_stage(double &bv, uint32_t &key)
{
    const unsigned long long bits = (unsigned long long)__double_as_longlong(bv);
    const uint32_t lo = dpp_word<CTRL>((uint32_t)bits), hi = dpp_word<CTRL>((uint32_t)(bits >> 32));
    const uint32_t ok = dpp_word<CTRL>(key);
    first_maximum_take(bv, key, __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo)), ok);
}
__device__ __forceinline__ void wave_first_maximum(double &bv, uint32_t &bi, uint32_t &bj)
{
    uint32_t key = (bj << 28) | bi; // (column, row): a column is below 9 (4 bits), which leaves 28 bits to the row
    first_maximum_stage<0xB1>(bv, key);  // quad_perm [1, 0, 3, 2]
    first_maximum_stage<0x4E>(bv, key);  // quad_perm [2, 3, 0, 1]
    first_maximum_stage<0x124>(bv, key); // row_ror:4
    first_maximum_stage<0x128>(bv, key); // row_ror:8
#pragma unroll
    for (int off = 16; off <= 32; off <<= 1)
        first_maximum_take(bv, key, __shfl_xor(bv, off), __shfl_xor(key, off));
    bi = key & 0x0FFFFFFFu;
    bj = key >> 28;
}

constexpr int W = 64;
// -DOCHIP_RANSAC_PHASES: shader-clock cycles per phase of ransac_homography_kernel, summed over the wavefronts of a launch and
// printed by the host after it (how the kernel's time divides; not part of the product build)
#ifdef OCHIP_RANSAC_PHASES
__device__ unsigned long long g_phase[16];
#define OCHIP_PHASE_T0(t) const unsigned long long t = clock64()
#define OCHIP_PHASE_ADD(t, slot)                                                                                       \
    do                                                                                                                 \
    {                                                                                                                  \
        if (threadIdx.x == 0)                                                                                          \
            atomicAdd(&g_phase[slot], clock64() - t);                                                                  \
    } while (0)
#define OCHIP_PHASE_COUNT(slot, n)                                                                                     \
    do                                                                                                                 \
    {                                                                                                                  \
        if (threadIdx.x == 0)                                                                                          \
            atomicAdd(&g_phase[slot], (unsigned long long)(n));                                                        \
    } while (0)
#else
#define OCHIP_PHASE_T0(t)
#define OCHIP_PHASE_ADD(t, slot)
#define OCHIP_PHASE_COUNT(slot, n)
#endif
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

// std::uniform_int_distribution<size_t>(0, hi)(minstd_rand0) — bits/uniform_int_dist.h, urngrange > urange.
// urngrange = max() - min() = 2147483646 - 1 and every operand stays below 2^31, so 32-bit arithmetic gives the
// 64-bit results; the scaling only depends on hi and is shared by the draws of one sample.
struct uniform_range
{
    uint32_t scaling, past;
};
__device__ __forceinline__ uniform_range make_range(uint32_t hi)
{
    const uint32_t uerange = hi + 1;
    uniform_range r;
    r.scaling = 2147483645u / uerange;
    r.past = uerange * r.scaling;
    return r;
}
__device__ __forceinline__ uint32_t uniform_int(uint32_t &x, const uniform_range &r)
{
    uint32_t ret;
    do
        ret = minstd_next(x) - 1u;
    while (ret >= r.past);
    return ret / r.scaling;
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
    const double *x1, *y1, *x2, *y2;     // [M] normalised coordinates, correspondence order
    const double *ex1, *ey1, *ex2, *ey2; // [M] the same in shuffled evaluation order (position p holds eval_order[p])
    uint8_t *cand, *inl;                 // [M] candidate / current inlier flags, indexed by correspondence
    double *P;                           // column-major (2M+1) x 9 system of fitInliers
    uint32_t M;
};

// MSAC scoring of one model.  ORDERED: walk the shuffled eval_order and apply the SPRT early
// exit of ransac.cpp:187-203; otherwise natural order, no exit (homography_model::evaluate :99-118).
// The running sum is accumulated one element at a time in walk order (non-inliers add +0.0, which changes nothing,
// so the chain is a fixed 64 adds per chunk with the addend taken from lane l by v_readlane).  The walk is
// software-pipelined: the next chunk's coordinates are requested before this chunk's arithmetic, and the ORDERED walk
// reads coordinates that the prologue stored in evaluation order, so every load of the 20+ scorings of a pair is a
// coalesced stream (the gather through eval_order cost two dependent memory round trips per 64 matches).
template <bool ORDERED>
__device__ double score_model(const model_t &m, const pair_data &pd, const uint32_t *__restrict__ order, uint8_t *flags,
                              double thr, double best_score, bool *rejected, uint32_t *n_inliers)
{
    const int lane = threadIdx.x;
    const uint32_t M = pd.M;
    const double *X1 = ORDERED ? pd.ex1 : pd.x1, *Y1 = ORDERED ? pd.ey1 : pd.y1, *X2 = ORDERED ? pd.ex2 : pd.x2,
                 *Y2 = ORDERED ? pd.ey2 : pd.y2;
    double s = 0;
    uint32_t count = 0;
    *rejected = false;
    double nx1, ny1, nx2, ny2;
    uint32_t nidx;
    {
        const bool nv = (uint32_t)lane < M;
        nx1 = nv ? X1[lane] : 0.0, ny1 = nv ? Y1[lane] : 0.0, nx2 = nv ? X2[lane] : 0.0, ny2 = nv ? Y2[lane] : 0.0;
        nidx = nv ? (ORDERED ? order[lane] : (uint32_t)lane) : 0;
    }
    for (uint32_t base = 0; base < M; base += W)
    {
        const uint32_t pos = base + lane;
        const bool valid = pos < M;
        const double x1 = nx1, y1 = ny1, x2 = nx2, y2 = ny2;
        const uint32_t idx = nidx;
        {
            const uint32_t np = pos + W;
            const bool nv = np < M;
            nx1 = nv ? X1[np] : 0.0, ny1 = nv ? Y1[np] : 0.0, nx2 = nv ? X2[np] : 0.0, ny2 = nv ? Y2[np] : 0.0;
            nidx = nv ? (ORDERED ? order[np] : np) : 0;
        }
        const double e = transfer_error(m, x1, y1, x2, y2);
        const bool inl = valid && (e < thr);
        double term = 0;
        if (inl)
        {
            const double ratio = e / thr;
            term = 1.0 - ratio * ratio;
        }
        if (valid)
            flags[idx] = inl ? 1 : 0;
        const unsigned long long mask = __ballot(inl);
        count += __popcll(mask);
        double pref = s; // running sum as seen right after this lane's element
        if (mask)
        {
#pragma unroll
            for (int l = 0; l < W; l++)
            {
                s = s + bcast(term, l);
                if (ORDERED)
                    pref = lane == l ? s : pref;
            }
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

// ---- Eigen FullPivLU<Matrix<double, rows, 9>>::solve(e_last).  Pivot search order / ties (column-by-column scan,
//      strict '>'), rank threshold and substitution order are the restated Eigen algorithm of the oracle.
struct lu_state
{
    uint32_t rowT[9], colT[9];
    uint32_t nonzero_pivots;
    double maxpivot;
};

// rank, then the solve steps on the factored leading 9 x 9 block B (column-major, ld 9, in LDS); uniform work
__device__ void lu_finish(const double *B, uint32_t rows, uint32_t size, const lu_state &st, double *sol /*[9], uniform*/)
{
    const uint32_t ld = 9;
    const double premult = fabs(st.maxpivot) * (2.220446049250313e-16 * (double)size);
    uint32_t rank = 0;
    for (uint32_t i = 0; i < st.nonzero_pivots; i++)
        rank += (fabs(B[(size_t)i * ld + i]) > premult) ? 1 : 0;
    for (int i = 0; i < 9; i++)
        sol[i] = 0;
    if (rank == 0)
        return;
    // c = P * e_last: follow the single 1 through the row transpositions
    uint32_t pos = rows - 1;
    for (uint32_t k = 0; k < size; k++)
    {
        if (pos == k)
            pos = st.rowT[k];
        else if (pos == st.rowT[k])
            pos = k;
    }
    double c[9];
    for (uint32_t i = 0; i < 9; i++)
        c[i] = (i == pos) ? 1.0 : 0.0;
    for (uint32_t j = 0; j < size; j++) // unit-lower forward substitution, column oriented
    {
        const double cj = c[j];
        for (uint32_t i = j + 1; i < size; i++)
            c[i] -= cj * B[(size_t)j * ld + i];
    }
    for (uint32_t jj = rank; jj-- > 0;) // upper back substitution, column oriented
    {
        c[jj] /= B[(size_t)jj * ld + jj];
        const double cj = c[jj];
        for (uint32_t i = 0; i < jj; i++)
            c[i] -= cj * B[(size_t)jj * ld + i];
    }
    uint32_t perm[9] = {0, 1, 2, 3, 4, 5, 6, 7, 8};
    for (uint32_t k = 0; k < size; k++)
    {
        const uint32_t t = perm[k];
        perm[k] = perm[st.colT[k]];
        perm[st.colT[k]] = t;
    }
    for (uint32_t i = 0; i < rank; i++)
        sol[perm[i]] = c[i];
}

// The minimal-sample fit: A is the 9 x 9 system in LDS (ld 9), rows <= 9; the wave cooperates, lanes over rows.
__device__ void full_piv_lu_solve9(double *A, uint32_t rows, double *sol /*[9], uniform*/)
{
    const int lane = threadIdx.x;
    const uint32_t cols = 9, ld = 9;
    const uint32_t size = rows < cols ? rows : cols;
    lu_state st;
    st.nonzero_pivots = size;
    st.maxpivot = 0;

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
        wave_first_maximum(bv, bi, bj);
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
            st.nonzero_pivots = k;
            for (uint32_t i = k; i < size; i++)
            {
                st.rowT[i] = i;
                st.colT[i] = i;
            }
            break;
        }
        if (bv > st.maxpivot)
            st.maxpivot = bv;
        st.rowT[k] = bi;
        st.colT[k] = bj;
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
    lu_finish(A, rows, size, st, sol);
}

// ---- the tall (2n+1) x 9 system of fitInliers (rows > 9): the same factorisation
struct piv_t
{
    double v;
    uint32_t i, j;
};
__device__ __forceinline__ void piv_reduce(piv_t &b)
{
    wave_first_maximum(b.v, b.i, b.j);
}

// ---- Round 5: the factorisation of the tall system WITHOUT the system.  Rounds 2 - 4 kept it column-major in HBM and went
//      over it once per elimination step: the (2 n + 1) x 9 matrix of a pair (170 KB at 1 180 matches) read and written
//      once per elimination step, 2.4 factorisations per pair, 2 048 pairs resident - 40 GB per 9 000-pair launch, 59 % of
//      the kernel's cycles.  A row of the system is a function of its correspondence's four coordinates, and what the
//      eliminations do to a row depends on nothing but the row and the pivot rows so far (which ARE the U part of the
//      factored leading block, 81 doubles in LDS).  So step k regenerates every row from its coordinates, applies the k
//      eliminations so far - the float operations of the stored form in its order, column for column - and looks for the
//      next pivot among the results; nothing is written.  36 eliminations per row instead of 9, 33 bytes read per
//      correspondence and step instead of ~1 600 per factorisation step.
//      Bookkeeping.  Positions: a row that never took part in a row transposition stands where it started (2 rank + a/b);
//      the rows that did - the nine that start in the leading block, every pivot row, the last row (0 .. 0 1), and the sibling
//      of a pivot row - live in a table of at most 29 entries in LDS with their position, and the sweep over the
//      correspondences skips them (rank < 5, or one of the <= 9 correspondences a pivot came from).  Columns: sigma maps
//      position -> original column (a nibble each); the U rows in LDS are kept in the current arrangement, so a regenerated
//      row is permuted once and every elimination is the stored form's.
__device__ __forceinline__ uint32_t nib_get(uint64_t p, uint32_t k)
{
    return (uint32_t)(p >> (4 * k)) & 15u;
}
__device__ __forceinline__ uint64_t nib_set(uint64_t p, uint32_t k, uint32_t v)
{
    return (p & ~(15ull << (4 * k))) | ((uint64_t)v << (4 * k));
}
typedef double dvec16 __attribute__((ext_vector_type(16))); // (indexed by a wave-uniform value: s_set_gpr_idx, no select chain)

struct tall_tab
{
    double xy[32][4];      // x, y, x', y' of the entry's correspondence
    uint32_t pos[32];      // where the row stands now
    uint32_t kind[32];     // 0 / 1: first / second row of its correspondence, 2: the last row; | 4: a pivot row (done)
    uint32_t top[9];       // entry at position 0 .. 8
    uint32_t piv_match[9]; // correspondence whose two rows entered the table at step t (~0u: none did)
    uint32_t n;
};
typedef __attribute__((address_space(3))) tall_tab lds_tab;
typedef __attribute__((address_space(3))) double lds_double;
typedef const __attribute__((address_space(3))) double lds_cdouble;
typedef const __attribute__((address_space(1))) double glb_cdouble;
typedef const __attribute__((address_space(1))) uint8_t glb_cbyte;
struct regen_in // what the factorisation reads of a pair, in global memory
{
    glb_cdouble *x1, *y1, *x2, *y2;
    glb_cbyte *inl;
    uint32_t M;
};

// A row of the system has eight values to draw from - write_dlt_rows' {-x, -y, -1, 0, x x', y x', x'} (second row: y' for x') and
// the last row's 1 - and a map from column to value: 0x654333210 / 0x654210333 / 0x733333333, a nibble per column.
typedef double dvec8 __attribute__((ext_vector_type(8)));
__device__ __forceinline__ dvec8 dlt_values(double x, double y, double xp)
{
    dvec8 o;
    o[0] = -x, o[1] = -y, o[2] = -1.0, o[3] = 0.0, o[4] = x * xp, o[5] = y * xp, o[6] = xp, o[7] = 1.0;
    return o;
}
__device__ __forceinline__ unsigned long long dlt_map(uint32_t kind)
{
    return kind == 0 ? 0x654333210ull : (kind == 1 ? 0x654210333ull : 0x733333333ull);
}

// a row in position order from its values.  `pick` = the value index of every position, a nibble each (wave-uniform: sigma
// through the row kind's map)
__device__ __forceinline__ void row_pick(const dvec8 &o, uint32_t pick_lo, uint32_t pick_hi, double (&v)[9])
{
#pragma unroll
    for (int pos = 0; pos < 9; pos++)
        v[pos] = o[pos < 8 ? (pick_lo >> (4 * pos)) & 7u : pick_hi & 7u];
}
// NR rows after K eliminations: v[t] for t < K becomes the row's multiplier of step t (what the stored form leaves in
// column t), v[K ..] the values the search for pivot K looks at.  The U row of the next step is requested while this
// step's arithmetic runs and no further ahead (left to itself the compiler asks for all 44 entries first: 88 registers)
// The multiplier v / p is a true division: what the compiler emits for one is v_div_scale x 2, v_rcp, four fma on the
// reciprocal, a product, a residual, v_div_fmas, v_div_fixup.  The reciprocal chain depends on p alone - wave-uniform here, one
// chain per step and trip instead of one per row - and when neither operand is near the ends of the exponent range
// v_div_scale passes both through, v_div_fmas is an fma and v_div_fixup sets the sign of the operands on the magnitude
// (also for a zero numerator): the FAST form below.  |v| <= |p| (full pivoting), so "near the ends" is: p outside
// [2^-100, 2^100] or not finite, or 0 < |v| < 2^-900; any lane that sees one reports it and the trip's rows are done
// again with the plain division.
__device__ __forceinline__ double refined_reciprocal(double p)
{
    const double r0 = __builtin_amdgcn_rcp(p);
    const double e0 = __builtin_fma(-p, r0, 1.0);
    const double r1 = __builtin_fma(r0, e0, r0);
    const double e1 = __builtin_fma(-p, r1, 1.0);
    return __builtin_fma(r1, e1, r1);
}
__device__ __forceinline__ double divide_in_range(double v, double p, double r2)
{
    const double q0 = v * r2;
    const double e2 = __builtin_fma(-p, q0, v);
    const double q1 = __builtin_fma(e2, r2, q0);
    const unsigned long long qb = (unsigned long long)__double_as_longlong(q1);
    const uint32_t sign = ((uint32_t)((unsigned long long)__double_as_longlong(v) >> 32) ^ (uint32_t)((unsigned long long)__double_as_longlong(p) >> 32)) & 0x80000000u;
    const uint32_t hi = ((uint32_t)(qb >> 32) & 0x7FFFFFFFu) | sign;
    return __longlong_as_double((long long)(((unsigned long long)hi << 32) | (qb & 0xFFFFFFFFull)));
}
template <int K, int NR, bool FAST>
__device__ __forceinline__ bool rows_eliminate(lds_cdouble *T9, double (&v)[NR][9])
{
    bool out_of_range = false;
    if (K == 0)
        return out_of_range;
    double u[9];
#pragma unroll
    for (int j = 0; j < 9; j++)
        u[j] = T9[j * 9];
#pragma unroll
    for (int t = 0; t < K; t++)
    {
        double nu[9];
#pragma unroll
        for (int j = 0; j < 9; j++)
            nu[j] = (t + 1 < K && j > t) ? T9[j * 9 + t + 1] : 0.0;
        asm volatile("" ::: "memory");
        const double p = u[t];
        double r2 = 0.0;
        if (FAST)
        {
            r2 = refined_reciprocal(p);
            out_of_range |= !((int)(fabs(p) >= 0x1p-100) & (int)(fabs(p) <= 0x1p100)); // (no short circuit: straight-line code)
        }
#pragma unroll
        for (int r = 0; r < NR; r++)
        {
            if (FAST)
                out_of_range |= (bool)((int)(fabs(v[r][t]) < 0x1p-900) & (int)(v[r][t] != 0.0));
            const double l = FAST ? divide_in_range(v[r][t], p, r2) : v[r][t] / p;
            v[r][t] = l;
#pragma unroll
            for (int j = t + 1; j < 9; j++)
                v[r][j] = v[r][j] - l * u[j];
        }
#pragma unroll
        for (int j = 0; j < 9; j++)
            u[j] = nu[j];
    }
    return out_of_range;
}
// sigma (position -> column) through a kind's map (column -> value): position -> value
__device__ __forceinline__ void compose_pick(uint32_t kind, uint32_t sig_lo, uint32_t sig_hi, uint32_t *pick_lo, uint32_t *pick_hi)
{
    const unsigned long long m = dlt_map(kind);
    uint32_t lo = 0;
#pragma unroll
    for (int pos = 0; pos < 8; pos++)
        lo |= (uint32_t)((m >> (4 * ((sig_lo >> (4 * pos)) & 15u))) & 15ull) << (4 * pos);
    *pick_lo = lo;
    *pick_hi = (uint32_t)((m >> (4 * (sig_hi & 15u))) & 15ull);
}

// the row's first maximum among positions K .. 8 (columns ascend, strict '>': the earliest column of the largest value;
// NaN never wins), offered to the lane's running first maximum
template <int K>
__device__ __forceinline__ bool row_offer(const double (&v)[9], uint32_t pos, piv_t &pv)
{
    double bv = -1.0;
    uint32_t bj = K;
#pragma unroll
    for (int j = K; j < 9; j++)
    {
        const double a = fabs(v[j]);
        if (a > bv)
        {
            bv = a;
            bj = j;
        }
    }
    const bool better = bv >= 0 && (bv > pv.v || (bv == pv.v && (bj < pv.j || (bj == pv.j && pos < pv.i))));
    if (better)
    {
        pv.v = bv;
        pv.i = pos;
        pv.j = bj;
    }
    return better;
}

// writes row `q` of the factored leading block: multipliers and remaining values of the table entry / correspondence row
template <int K>
__device__ __forceinline__ void leading_row(lds_double *T9, int q, uint32_t kind, double x, double y, double x_, double y_, uint32_t sig_lo,
                                            uint32_t sig_hi)
{
    double v[1][9];
    uint32_t pick_lo, pick_hi; // (kind and sigma are wave-uniform here)
    compose_pick((uint32_t)__builtin_amdgcn_readfirstlane((int)kind), sig_lo, sig_hi, &pick_lo, &pick_hi);
    row_pick(dlt_values(x, y, kind == 0 ? x_ : y_), pick_lo, pick_hi, v[0]);
    asm volatile("" ::: "memory");
    rows_eliminate<K, 1, false>(T9, v);
    if (threadIdx.x == 0)
    {
#pragma unroll
        for (int j = 0; j < 9; j++)
            T9[j * 9 + q] = v[0][j];
    }
}

template <int K>
__device__ __forceinline__ bool regen_lu_step(const regen_in &pd, uint32_t rows, lds_double *T9, lds_tab &tab, lu_state &st, uint32_t &sig_lo,
                                              uint32_t &sig_hi)
{
    const int lane = threadIdx.x;
    const uint32_t M = pd.M;
    __syncthreads(); // the table and the leading block of the previous step have landed
    sig_lo = (uint32_t)__builtin_amdgcn_readfirstlane((int)sig_lo); // (wave-uniform by construction; the register indices want it in SGPRs)
    sig_hi = (uint32_t)__builtin_amdgcn_readfirstlane((int)sig_hi);
    piv_t pv{-1.0, (uint32_t)K, (uint32_t)K};
    uint32_t cand = 0; // where the lane's best row came from: correspondence slot << 1 | a/b, or 0x80000000 | table entry
    uint32_t touched[K > 0 ? K : 1];
#pragma unroll
    for (int t = 0; t < K; t++)
        touched[t] = (uint32_t)__builtin_amdgcn_readfirstlane((int)tab.piv_match[t]);
    uint32_t pick_lo[3], pick_hi[3]; // position -> value index, per kind of row
#pragma unroll
    for (uint32_t kind = 0; kind < 3; kind++)
        compose_pick(kind, sig_lo, sig_hi, &pick_lo[kind], &pick_hi[kind]);
    // ---- the rows that never moved: a sweep over the correspondences, both rows of an inlier
    {
        // Two correspondences per lane and trip: four rows whose eliminations the scheduler interleaves (a row's steps are one
        // dependent chain of fp64 operations, and with two wavefronts on a SIMD there is little else to issue meanwhile).
        // The next trip's flags stay raw bytes until that trip (turned into predicates here they are waited for here, a
        // memory round trip per trip), and the loads take a clamped index instead of a branch around them (after a join
        // the compiler waits for every load).
        constexpr int SL = 2;
        uint32_t before = 0;
        uint32_t nflag[SL];
        double nx1[SL], ny1[SL], nx2[SL], ny2[SL];
#pragma unroll
        for (int q = 0; q < SL; q++)
        {
            const uint32_t c = min((uint32_t)(q * W + lane), M - 1);
            nflag[q] = pd.inl[c];
            nx1[q] = pd.x1[c], ny1[q] = pd.y1[c], nx2[q] = pd.x2[c], ny2[q] = pd.y2[c];
        }
        for (uint32_t base = 0; base < M; base += SL * W)
        {
            // (the U rows come from LDS in every trip: kept in registers across the loop they are up to 88 of them)
            asm volatile("" ::: "memory");
            bool take[SL];
            uint32_t rank[SL];
            double x1[SL], y1[SL], x2[SL], y2[SL];
#pragma unroll
            for (int q = 0; q < SL; q++)
            {
                const uint32_t i = base + q * W + lane;
                const bool f = i < M && nflag[q] != 0;
                x1[q] = nx1[q], y1[q] = ny1[q], x2[q] = nx2[q], y2[q] = ny2[q];
                const uint32_t nc = min(i + SL * W, M - 1);
                nflag[q] = pd.inl[nc];
                nx1[q] = pd.x1[nc], ny1[q] = pd.y1[nc], nx2[q] = pd.x2[nc], ny2[q] = pd.y2[nc];
                const unsigned long long mask = __ballot(f);
                rank[q] = before + __popcll(mask & ((1ull << lane) - 1ull));
                before += __popcll(mask);
                take[q] = f && rank[q] >= 5;
#pragma unroll
                for (int t = 0; t < K; t++)
                    take[q] = take[q] && i != touched[t];
            }
            if (take[0] || take[1])
            {
                double v[2 * SL][9];
                auto rows = [&]() {
#pragma unroll
                    for (int q = 0; q < SL; q++)
                    {
                        row_pick(dlt_values(x1[q], y1[q], x2[q]), pick_lo[0], pick_hi[0], v[2 * q]);
                        row_pick(dlt_values(x1[q], y1[q], y2[q]), pick_lo[1], pick_hi[1], v[2 * q + 1]);
                    }
                };
                rows();
                if (__ballot(rows_eliminate<K, 2 * SL, true>(T9, v)) != 0)
                {
#ifdef OCHIP_RANSAC_PHASES
                    if ((int)threadIdx.x == __builtin_ctzll(__ballot(true)))
                        atomicAdd(&g_phase[12], 1ull);
#endif
                    rows();
                    rows_eliminate<K, 2 * SL, false>(T9, v);
                }
#pragma unroll
                for (int q = 0; q < SL; q++)
                    if (take[q])
                    {
                        if (row_offer<K>(v[2 * q], 2 * rank[q], pv))
                            cand = (base / W + q) << 1;
                        if (row_offer<K>(v[2 * q + 1], 2 * rank[q] + 1, pv))
                            cand = (base / W + q) << 1 | 1u;
                    }
            }
        }
    }
    // ---- the rows of the table that are not pivots yet, one per lane
    double v_kk = 0.0; // (of the lane whose entry stands at position K: the value at (K, K))
    {
        // (the kind differs from lane to lane and the value picks are wave-uniform: a turn per kind, its lanes only)
        const bool mine = (uint32_t)lane < tab.n && !(tab.kind[lane] & 4u);
        const uint32_t kind = mine ? tab.kind[lane] : 3u;
        const double ex = mine ? tab.xy[lane][0] : 0.0, ey = mine ? tab.xy[lane][1] : 0.0;
        const double exp_ = mine ? (kind == 0 ? tab.xy[lane][2] : tab.xy[lane][3]) : 0.0;
        const uint32_t epos = mine ? tab.pos[lane] : 0u;
#pragma unroll
        for (uint32_t kd = 0; kd < 3; kd++)
            if (kind == kd)
            {
                double v[1][9];
                row_pick(dlt_values(ex, ey, exp_), pick_lo[kd], pick_hi[kd], v[0]);
                asm volatile("" ::: "memory");
                rows_eliminate<K, 1, false>(T9, v);
                v_kk = v[0][K];
                if (row_offer<K>(v[0], epos, pv))
                    cand = 0x80000000u | (uint32_t)lane;
            }
    }
    const double lv = pv.v;
    const uint32_t li = pv.i, lj = pv.j;
    piv_reduce(pv);
    double bv = pv.v;
    uint32_t bi = (uint32_t)__builtin_amdgcn_readfirstlane((int)pv.i), bj = (uint32_t)__builtin_amdgcn_readfirstlane((int)pv.j);
    const uint32_t at_k = (uint32_t)__builtin_amdgcn_readfirstlane((int)tab.top[K]);
    const double akk = bcast(v_kk, (int)at_k);
    uint32_t who; // the pivot row: cand of the lane that found it, or the entry at position K
    if (akk != akk) // a NaN in the first scanned cell sticks (nothing compares greater than NaN)
    {
        bv = akk;
        bi = K;
        bj = K;
        who = 0x80000000u | at_k;
    }
    else if (bv < 0) // every candidate was NaN: keep (k,k)
    {
        bv = fabs(akk);
        bi = K;
        bj = K;
        who = 0x80000000u | at_k;
    }
    else
    {
        const unsigned long long won = __ballot(lv == bv && li == bi && lj == bj);
        who = (uint32_t)__builtin_amdgcn_readlane((int)cand, __builtin_ctzll(won));
        if (!(who & 0x80000000u))
            who |= (uint32_t)__builtin_ctzll(won) << 24; // (a correspondence row: slot < 2^22, its lane beside it)
    }
    if (bv == 0.0)
        return false;
    if (bv > st.maxpivot)
        st.maxpivot = bv;
    st.rowT[K] = bi;
    st.colT[K] = bj;
    // ---- the pivot row: identity, then its multipliers and values into row K of the leading block (after the column
    //      transposition K <-> bj, which the rows above take too)
    uint32_t kind, entry = ~0u, match = ~0u;
    double x, y, x_, y_;
    if (who & 0x80000000u)
    {
        entry = who & 31u;
        kind = tab.kind[entry] & 3u;
        x = tab.xy[entry][0], y = tab.xy[entry][1], x_ = tab.xy[entry][2], y_ = tab.xy[entry][3];
    }
    else
    {
        match = ((who & 0x00FFFFFFu) >> 1) * W + (who >> 24);
        kind = who & 1u;
        x = pd.x1[match], y = pd.y1[match], x_ = pd.x2[match], y_ = pd.y2[match];
    }
    __syncthreads(); // every lane has read the U rows in the old arrangement
    if (bj != (uint32_t)K && lane < K)
    {
        const double t = T9[K * 9 + lane];
        T9[K * 9 + lane] = T9[bj * 9 + lane];
        T9[bj * 9 + lane] = t;
    }
    __syncthreads();
    if (bj != (uint32_t)K) // sigma: positions K and bj trade columns
    {
        unsigned long long sg = ((unsigned long long)sig_hi << 32) | sig_lo;
        const uint32_t a = nib_get(sg, K), b = nib_get(sg, bj);
        sg = nib_set(nib_set(sg, K, b), bj, a);
        sig_lo = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)sg);
        sig_hi = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(sg >> 32));
    }
    leading_row<K>(T9, K, kind, x, y, x_, y_, sig_lo, sig_hi); // (regenerated in the new arrangement: its value at position K is the pivot)
    // ---- the table: the pivot row goes to position K, the row that stood there to where the pivot row came from
    if (lane == 0)
    {
        uint32_t e = entry;
        if (e == ~0u)
        {
            e = tab.n;
            tab.xy[e][0] = x, tab.xy[e][1] = y, tab.xy[e][2] = x_, tab.xy[e][3] = y_;
            tab.xy[e + 1][0] = x, tab.xy[e + 1][1] = y, tab.xy[e + 1][2] = x_, tab.xy[e + 1][3] = y_;
            tab.kind[e + 1] = kind ^ 1u; // the sibling row stays where it is, but the sweep now skips its correspondence
            tab.pos[e + 1] = bi ^ 1u;
            tab.n = e + 2;
        }
        tab.piv_match[K] = match;
        tab.kind[e] = kind | 4u;
        tab.pos[e] = K;
        if (at_k != e)
        {
            tab.pos[at_k] = bi;
            if (bi < 9)
                tab.top[bi] = at_k;
        }
        tab.top[K] = e;
    }
    return true;
}

// (not inlined: the RANSAC kernels hold two models and a sample stream in registers around it, and with nine unrolled steps
// inside them the allocator gave up on two wavefronts per SIMD.  The pointers carry their address spaces across the call -
// generic ones would make every LDS access a flat one.)
__device__ __attribute__((noinline)) void regen_lu_solve9(regen_in pd, uint32_t n_in, lds_double *T9 /*LDS 81*/, lds_tab *tab_p, double *sol /*[9], uniform*/)
{
    lds_tab &tab = *tab_p;
    const int lane = threadIdx.x;
    const uint32_t M = pd.M, rows = 2 * n_in + 1;
    lu_state st;
    st.nonzero_pivots = 9;
    st.maxpivot = 0;
    __syncthreads();
    // the table starts with the rows of the first five inliers (rows 0 .. 9: the leading block and the row below it) and the last row
    {
        uint32_t before = 0;
        for (uint32_t base = 0; base < M && before < 5; base += W)
        {
            const uint32_t i = base + lane;
            const bool f = i < M && pd.inl[i];
            const unsigned long long mask = __ballot(f);
            const uint32_t rank = before + __popcll(mask & ((1ull << lane) - 1ull));
            before += __popcll(mask);
            if (f && rank < 5)
            {
                const double x = pd.x1[i], y = pd.y1[i], x_ = pd.x2[i], y_ = pd.y2[i];
                for (uint32_t ab = 0; ab < 2; ab++)
                {
                    const uint32_t e = 2 * rank + ab;
                    tab.xy[e][0] = x, tab.xy[e][1] = y, tab.xy[e][2] = x_, tab.xy[e][3] = y_;
                    tab.kind[e] = ab;
                    tab.pos[e] = e;
                }
            }
        }
        if (lane < 9)
        {
            tab.top[lane] = lane;
            tab.piv_match[lane] = ~0u;
        }
        if (lane == 0)
        {
            tab.xy[10][0] = tab.xy[10][1] = tab.xy[10][2] = tab.xy[10][3] = 0.0;
            tab.kind[10] = 2;
            tab.pos[10] = rows - 1;
            tab.n = 11;
        }
    }
    uint32_t sig_lo = 0x76543210u, sig_hi = 8u;
    int stop = 9;
    do
    {
#define OCHIP_LU_STEP(K)                                                                                               \
    if (!regen_lu_step<K>(pd, rows, T9, tab, st, sig_lo, sig_hi))                                                      \
    {                                                                                                                  \
        stop = K;                                                                                                      \
        break;                                                                                                         \
    }
        OCHIP_LU_STEP(0)
        OCHIP_LU_STEP(1)
        OCHIP_LU_STEP(2)
        OCHIP_LU_STEP(3)
        OCHIP_LU_STEP(4)
        OCHIP_LU_STEP(5)
        OCHIP_LU_STEP(6)
        OCHIP_LU_STEP(7)
        OCHIP_LU_STEP(8)
#undef OCHIP_LU_STEP
    } while (false);
    if (stop < 9)
    {
        // the remaining block is all zero: the rows standing at positions stop .. 8 close the leading block as they are
        st.nonzero_pivots = (uint32_t)stop;
        for (int i = stop; i < 9; i++)
        {
            st.rowT[i] = (uint32_t)i;
            st.colT[i] = (uint32_t)i;
        }
        __syncthreads();
        for (int q = stop; q < 9; q++)
        {
            const uint32_t e = tab.top[q];
            const uint32_t kind = tab.kind[e] & 3u;
            const double x = tab.xy[e][0], y = tab.xy[e][1], x_ = tab.xy[e][2], y_ = tab.xy[e][3];
            switch (stop)
            {
#define OCHIP_LEAD(K)                                                                                                  \
    case K: leading_row<K>(T9, q, kind, x, y, x_, y_, sig_lo, sig_hi); break;
                OCHIP_LEAD(0)
                OCHIP_LEAD(1)
                OCHIP_LEAD(2)
                OCHIP_LEAD(3)
                OCHIP_LEAD(4)
                OCHIP_LEAD(5)
                OCHIP_LEAD(6)
                OCHIP_LEAD(7)
                OCHIP_LEAD(8)
#undef OCHIP_LEAD
            }
        }
    }
    __syncthreads();
    lu_finish((const double *)T9, rows, 9, st, sol);
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

// homography_model::fitInliers (homography_model.cpp:52-87) on the flags pd.inl, of which n_in are set: the
// (2 n_in + 1) x 9 system in index order, its full-pivot LU, H from the solution.
__device__ __forceinline__ void fit_inliers(const pair_data &pd, uint32_t n_in, double *T9 /*LDS 81*/, tall_tab *tab /*LDS*/, model_t &model)
{
    const int lane = threadIdx.x;
    const uint32_t M = pd.M;
    double sol[9];
    const uint32_t rows = 2 * n_in + 1, ld = rows;
    if (rows > 9)
    {
        regen_in in;
        in.x1 = (glb_cdouble *)pd.x1, in.y1 = (glb_cdouble *)pd.y1, in.x2 = (glb_cdouble *)pd.x2, in.y2 = (glb_cdouble *)pd.y2;
        in.inl = (glb_cbyte *)pd.inl;
        in.M = pd.M;
        regen_lu_solve9(in, n_in, (lds_double *)T9, (lds_tab *)tab, sol);
        model_from_solution(model, sol);
        return;
    }
    // fewer than five inliers: the (2 n_in + 1) x 9 system in index order (homography_model.cpp:52-79), at most 9 x 9; the
    // small factorisation runs it from LDS
    uint32_t before = 0;
    bool nf = (uint32_t)lane < M && pd.inl[lane];
    double nx1 = nf ? pd.x1[lane] : 0.0, ny1 = nf ? pd.y1[lane] : 0.0, nx2 = nf ? pd.x2[lane] : 0.0,
           ny2 = nf ? pd.y2[lane] : 0.0;
    for (uint32_t base = 0; base < M; base += W)
    {
        const bool f = nf;
        const double x1 = nx1, y1 = ny1, x2 = nx2, y2 = ny2;
        {
            const uint32_t ni = base + W + lane;
            nf = ni < M && pd.inl[ni];
            nx1 = ni < M ? pd.x1[ni] : 0.0, ny1 = ni < M ? pd.y1[ni] : 0.0, nx2 = ni < M ? pd.x2[ni] : 0.0,
            ny2 = ni < M ? pd.y2[ni] : 0.0;
        }
        const unsigned long long mask = __ballot(f);
        if (f)
        {
            const uint32_t r = before + __popcll(mask & ((1ull << lane) - 1ull));
            write_dlt_rows(pd.P, ld, 2 * r, x1, y1, x2, y2);
        }
        before += __popcll(mask);
    }
    if (lane < 9)
        pd.P[(size_t)lane * ld + rows - 1] = lane == 8 ? 1.0 : 0.0;
    __syncthreads();
    for (uint32_t t = lane; t < rows * 9; t += W)
        T9[(t / rows) * 9 + (t % rows)] = pd.P[t];
    full_piv_lu_solve9(T9, rows, sol);
    model_from_solution(model, sol);
}

// four distinct positions of the sampling pool; distinct positions are distinct correspondences because the PROSAC
// order is a permutation, so the uniqueness test of ransac.cpp:118-150 needs no loaded value and the four
// sorted_idx reads go out together
__device__ __forceinline__ void draw_distinct(uint32_t &rng, uint32_t hi, int first, uint32_t *c)
{
    const uniform_range range = make_range(hi);
    for (int j = first; j < 4; j++)
    {
        uint32_t cand;
        bool unique;
        do
        {
            cand = uniform_int(rng, range);
            unique = true;
            for (int k = first; k < j; k++)
                if (c[k] == cand)
                    unique = false;
        } while (!unique);
        c[j] = cand;
    }
}

// ---- fast-forward over iterations that change nothing -------------------------------------------------------------
// An iteration of ransac.cpp:98-247 only touches the state when its model is neither degenerate nor SPRT-rejected and
// scores above the best so far; every other iteration just consumes random numbers.  Link pairs with few matches and
// a low inlier ratio run hundreds to thousands of such iterations (one C3 pair: 6 111, 33 us each when the whole wave
// works on one 9 x 9 fit at a time — that single pair WAS the kernel's duration).  So the next up to FB iterations
// are looked at side by side, one per lane: the sample stream is replayed (uniform, sequential, as it must be), every
// lane fits its own sample with a private 9 x 9 full-pivot LU in LDS and walks the matches in evaluation order with the
// SPRT test, exactly the arithmetic of the one-at-a-time path.  The iterations before the first lane that would
// improve are skipped (rng and PROSAC state advanced past them); that iteration itself then runs on the normal path.
constexpr int FB = 32;


// homography_model::fit of one lane's sample; element (i, j) of the lane's 9 x 9 system lives at Aq[(j * 9 + i) * FB]
__device__ void lane_fit(double *Aq /*already offset by the lane*/, const double *px, const double *py, const double *qx,
                         const double *qy, model_t &m)
{
#define AQ(i, j) Aq[(((j) * 9 + (i)) * FB)]
#pragma unroll
    for (int t = 0; t < 81; t++)
        Aq[t * FB] = 0.0;
#pragma unroll
    for (int p = 0; p < 4; p++)
    {
        const double x = px[p], y = py[p], x_ = qx[p], y_ = qy[p];
        AQ(2 * p, 0) = -x;
        AQ(2 * p, 1) = -y;
        AQ(2 * p, 2) = -1;
        AQ(2 * p, 6) = x * x_;
        AQ(2 * p, 7) = y * x_;
        AQ(2 * p, 8) = x_;
        AQ(2 * p + 1, 3) = -x;
        AQ(2 * p + 1, 4) = -y;
        AQ(2 * p + 1, 5) = -1;
        AQ(2 * p + 1, 6) = x * y_;
        AQ(2 * p + 1, 7) = y * y_;
        AQ(2 * p + 1, 8) = y_;
    }
    AQ(8, 8) = 1.0;

    uint64_t rowT = 0x876543210ull, colT = 0x876543210ull;
    uint32_t nonzero_pivots = 9;
    double maxpivot = 0;
    bool alive = true;
#pragma unroll
    for (int k = 0; k < 9; k++)
    {
        if (alive)
        {
            double bv = -1.0;
            uint32_t bi = k, bj = k;
#pragma unroll
            for (int j = k; j < 9; j++)
#pragma unroll
                for (int i = k; i < 9; i++)
                {
                    const double v = fabs(AQ(i, j));
                    if (v > bv) // column-by-column scan, strict '>' keeps the first maximum
                    {
                        bv = v;
                        bi = i;
                        bj = j;
                    }
                }
            const double akk = AQ(k, k);
            if (akk != akk)
            {
                bv = akk;
                bi = k;
                bj = k;
            }
            else if (bv < 0)
            {
                bv = fabs(akk);
                bi = k;
                bj = k;
            }
            if (bv == 0.0)
            {
                nonzero_pivots = k;
                alive = false; // the remaining transpositions stay the identity
            }
            else
            {
                if (bv > maxpivot)
                    maxpivot = bv;
                rowT = nib_set(rowT, k, bi);
                colT = nib_set(colT, k, bj);
                if ((uint32_t)k != bi)
                {
#pragma unroll
                    for (int j = 0; j < 9; j++)
                    {
                        const double t = AQ(k, j);
                        AQ(k, j) = AQ(bi, j);
                        AQ(bi, j) = t;
                    }
                }
                if ((uint32_t)k != bj)
                {
#pragma unroll
                    for (int i = 0; i < 9; i++)
                    {
                        const double t = AQ(i, k);
                        AQ(i, k) = AQ(i, bj);
                        AQ(i, bj) = t;
                    }
                }
                const double p = AQ(k, k);
                double rowk[9];
#pragma unroll
                for (int j = k + 1; j < 9; j++)
                    rowk[j] = AQ(k, j);
#pragma unroll
                for (int i = k + 1; i < 9; i++)
                {
                    const double l = AQ(i, k) / p;
                    AQ(i, k) = l;
#pragma unroll
                    for (int j = k + 1; j < 9; j++)
                        AQ(i, j) = AQ(i, j) - l * rowk[j];
                }
            }
        }
    }
    // rank, P e_last, the two substitutions, the column permutation (lu_finish, per lane)
    const double premult = fabs(maxpivot) * (2.220446049250313e-16 * (double)9);
    uint32_t rank = 0;
#pragma unroll
    for (int i = 0; i < 9; i++)
        if ((uint32_t)i < nonzero_pivots)
            rank += (fabs(AQ(i, i)) > premult) ? 1 : 0;
    double sol[9];
#pragma unroll
    for (int i = 0; i < 9; i++)
        sol[i] = 0;
    if (rank != 0)
    {
        uint32_t pos = 8;
#pragma unroll
        for (int k = 0; k < 9; k++)
        {
            const uint32_t r = nib_get(rowT, k);
            if (pos == (uint32_t)k)
                pos = r;
            else if (pos == r)
                pos = k;
        }
        double c[9];
#pragma unroll
        for (int i = 0; i < 9; i++)
            c[i] = ((uint32_t)i == pos) ? 1.0 : 0.0;
#pragma unroll
        for (int j = 0; j < 9; j++)
        {
            const double cj = c[j];
#pragma unroll
            for (int i = j + 1; i < 9; i++)
                c[i] -= cj * AQ(i, j);
        }
#pragma unroll
        for (int jj = 8; jj >= 0; jj--)
            if ((uint32_t)jj < rank)
            {
                c[jj] /= AQ(jj, jj);
                const double cj = c[jj];
#pragma unroll
                for (int i = 0; i < jj; i++)
                    c[i] -= cj * AQ(i, jj);
            }
        uint64_t perm = 0x876543210ull;
#pragma unroll
        for (int k = 0; k < 9; k++)
        {
            const uint32_t ck = nib_get(colT, k);
            const uint32_t t = nib_get(perm, k);
            perm = nib_set(perm, k, nib_get(perm, ck));
            perm = nib_set(perm, ck, t);
        }
#pragma unroll
        for (int i = 0; i < 9; i++)
            if ((uint32_t)i < rank)
            {
                const uint32_t pi = nib_get(perm, i);
#pragma unroll
                for (int mm = 0; mm < 9; mm++)
                    if (pi == (uint32_t)mm)
                        sol[mm] = c[i];
            }
    }
#undef AQ
    model_from_solution(m, sol);
}

// Returns the number of leading no-op iterations among the next n (rng / prosac_n advanced past them); *found says
// whether the iteration after them would improve the best model (it then runs on the normal path).
__device__ uint32_t fast_forward(const pair_data &pd, const uint32_t *__restrict__ sorted_idx, bool has_quality, uint32_t it,
                                 uint32_t n, uint32_t &rng, uint32_t &prosac_n, double best_score, double thr, double *AQs,
                                 bool *found)
{
    const int lane = threadIdx.x;
    const uint32_t M = pd.M;
    OCHIP_PHASE_COUNT(8, 1);
    OCHIP_PHASE_T0(t_fit);
    // ---- replay the sample stream of the next n iterations (ransac.cpp:100-154); lane L keeps iteration it + L
    uint32_t rng_s = rng, prosac_s = prosac_n;
    uint32_t my_rng = 0, my_prosac = 0, my_c[4] = {0, 0, 0, 0};
    for (uint32_t L = 0; L < n; L++)
    {
        const uint32_t itL = it + L, rb = rng_s, pb = prosac_s;
        if (has_quality && prosac_s < M && itL > 0 && itL % 10 == 0)
            prosac_s++;
        uint32_t c[4];
        if (has_quality && prosac_s < M && prosac_s > 4)
        {
            c[0] = prosac_s - 1;
            draw_distinct(rng_s, prosac_s - 2, 1, c);
        }
        else
            draw_distinct(rng_s, (has_quality ? prosac_s : M) - 1, 0, c);
        if ((uint32_t)lane == L)
        {
            my_rng = rb;
            my_prosac = pb;
            for (int j = 0; j < 4; j++)
                my_c[j] = c[j];
        }
    }
    // ---- every lane: its sample, the degeneracy test, the fit
    const bool active = (uint32_t)lane < n;
    bool live = active;
    model_t m;
    set_nan(m);
    if (active)
    {
        uint32_t s4[4];
        for (int j = 0; j < 4; j++)
            s4[j] = has_quality ? sorted_idx[my_c[j]] : my_c[j];
        double px[4], py[4], qx[4], qy[4];
        for (int j = 0; j < 4; j++)
        {
            px[j] = pd.x1[s4[j]];
            py[j] = pd.y1[s4[j]];
            qx[j] = pd.x2[s4[j]];
            qy[j] = pd.y2[s4[j]];
        }
        for (int a = 0; a < 4; a++)
            for (int b = a + 1; b < 4; b++)
                for (int c = b + 1; c < 4; c++)
                {
                    const double v1x = px[b] - px[a], v1y = py[b] - py[a];
                    const double v2x = px[c] - px[a], v2y = py[c] - py[a];
                    if (fabs(v1x * v2y - v1y * v2x) < 1e-10)
                        live = false; // degenerate: the iteration ends here
                }
        if (live)
            lane_fit(AQs + lane, px, py, qx, qy, m);
    }
    OCHIP_PHASE_ADD(t_fit, 1);
    OCHIP_PHASE_T0(t_walk);
    // ---- SPRT-pruned MSAC walk in evaluation order.  The error of a (model, position) is the expensive part (four divisions
    //      and a square root in fp64), the running sum of a model takes its terms one position after the other.  The n <= 32
    //      models sit in lanes 0 .. n - 1; the wave's other lanes evaluate the same models at the following positions - P =
    //      2 .. 4 positions per trip, as many as 64 lanes hold groups of n (19 models, the usual number behind the 20-iteration
    //      floor: 3 positions, 57 lanes busy; rounds 3 - 4 used two half-waves whatever n was) - and hand term and SPRT
    //      limit of their position to the model's lane, which adds and tests them in order.
    static_assert(FB == 32, "two groups of FB models fill the wave");
    const uint32_t P = n > 21 ? 2u : (n > 16 ? 3u : 4u);
    const uint32_t CH = ((uint32_t)W / P) * P; // positions per chunk: whole trips
    const uint32_t slot = ((uint32_t)lane >= n ? 1u : 0u) + ((uint32_t)lane >= 2 * n ? 1u : 0u) + ((uint32_t)lane >= 3 * n ? 1u : 0u);
    const bool evaluates = (uint32_t)lane < P * n;
    const int src = evaluates ? lane - (int)(slot * n) : 0;
    model_t mm;
    for (int i = 0; i < 9; i++)
    {
        mm.H[i] = __shfl(m.H[i], src);
        mm.Hi[i] = __shfl(m.Hi[i], src);
    }
    double s = 0;
    bool rej = false;
    // (round 5: the walk took its four coordinates of every position straight from memory - the same address in all
    // lanes, one L2 round trip per two positions with the exit test between them, 14 k loads per pair and two thirds of the
    // kernel's cycles in s_waitcnt.  Now a chunk of 64 positions is loaded once, a position per lane, and the walk takes a
    // position's coordinates from the lane that holds it: one round trip per chunk, and most walks of a pair with few
    // inliers end inside the first.)
    for (uint32_t base = 0; base < M; base += CH)
    {
        if (__ballot(lane < 32 && live && !rej) == 0)
            break;
        const uint32_t mine = base + (uint32_t)lane;
        const double c_x1 = mine < M ? pd.ex1[mine] : 0.0, c_y1 = mine < M ? pd.ey1[mine] : 0.0;
        const double c_x2 = mine < M ? pd.ex2[mine] : 0.0, c_y2 = mine < M ? pd.ey2[mine] : 0.0;
        const uint32_t chunk_end = min(base + CH, M);
        for (uint32_t pos0 = base; pos0 < chunk_end; pos0 += P)
        {
            if (__ballot(lane < 32 && live && !rej) == 0)
                break;
            const uint32_t pos = pos0 + (evaluates ? slot : 0u);
            const int holder = (int)(pos - base); // (< CH + P - 1 <= 64; a position past the chunk's last is past M or unused)
            const double px1 = __shfl(c_x1, holder), py1 = __shfl(c_y1, holder), px2 = __shfl(c_x2, holder), py2 = __shfl(c_y2, holder);
            double term = 0, limit = 0;
            if (evaluates && pos < chunk_end)
            {
                const double e = transfer_error(mm, px1, py1, px2, py2);
                if (e < thr)
                {
                    const double ratio = e / thr;
                    term = 1.0 - ratio * ratio;
                }
                limit = best_score * (double)(pos + 1) / (double)M * 0.6;
            }
#pragma unroll
            for (uint32_t sl = 0; sl < 4; sl++)
                if (sl < P) // (wave-uniform)
                {
                    const int from = (lane + (int)(sl * n)) & (W - 1);
                    const double t = sl == 0 ? term : __shfl(term, from), lim = sl == 0 ? limit : __shfl(limit, from);
                    if (pos0 + sl < chunk_end)
                    {
                        s = s + t; // position pos0 + sl (lanes 0 .. n - 1; the other lanes' sums are not used)
                        if (pos0 + sl + 1 > 20 && best_score > 0 && s < lim)
                            rej = true;
                    }
                }
        }
    }
    OCHIP_PHASE_ADD(t_walk, 2);
    const unsigned long long improving = __ballot(lane < 32 && live && !rej && s > best_score);
    if (improving == 0)
    {
        rng = rng_s;
        prosac_n = prosac_s;
        *found = false;
        return n;
    }
    const int first = __builtin_ctzll(improving);
    rng = (uint32_t)__builtin_amdgcn_readlane((int)my_rng, first);
    prosac_n = (uint32_t)__builtin_amdgcn_readlane((int)my_prosac, first);
    *found = true;
    return (uint32_t)first;
}

template <int OCC> __global__ __launch_bounds__(W, OCC) __attribute__((amdgpu_num_vgpr(248))) void ransac_homography_kernel(
    const ochip_ransac_job *__restrict__ jobs, const ochip_ransac_match *__restrict__ matches,
    const uint32_t *__restrict__ sorted_idx_all, const uint32_t *__restrict__ eval_order_all, rays_view rv,
    double *__restrict__ coord_scratch /*8 x total*/, uint8_t *__restrict__ flag_scratch /*2 x total*/,
    double *__restrict__ P_scratch /*81 x n_jobs*/, uint64_t total, double thr,
    ochip_ransac_result *__restrict__ results, uint8_t *__restrict__ inliers_out, const uint32_t *__restrict__ launch_order)
{
    __shared__ double P9[81], T9[81], AQs[81 * FB];
    static_assert(sizeof(tall_tab) <= sizeof(double) * 81 * FB, "the factorisation's table borrows the lane fits' LDS");
    const int lane = threadIdx.x;
    const uint32_t job_id = launch_order[blockIdx.x]; // (the pairs with the most matches first: see the launch)
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

    if (M < 4) // ransac.cpp:69-72
    {
        for (uint32_t i = lane; i < M; i += W)
            inliers_out[mo + i] = 0;
        if (lane == 0)
            results[job_id] = res;
        return;
    }

    // ---- prologue: gather the unit rays of the matched keypoints, divide by z once (error() and fit()
    //      both start from measurement / measurement.z), detect has_quality (ransac.cpp:74-82); the same
    //      coordinates once more in evaluation order for the SPRT walks
    OCHIP_PHASE_T0(t_total);
    const uint32_t *sorted_idx = sorted_idx_all + mo;
    const uint32_t *eval_order = eval_order_all + job.eval_offset;
    pair_data pd;
    double *cx1 = coord_scratch + mo, *cy1 = coord_scratch + total + mo, *cx2 = coord_scratch + 2 * total + mo,
           *cy2 = coord_scratch + 3 * total + mo;
    double *ex1 = coord_scratch + 4 * total + mo, *ey1 = coord_scratch + 5 * total + mo,
           *ex2 = coord_scratch + 6 * total + mo, *ey2 = coord_scratch + 7 * total + mo;
    const double *r1 = rv.rays + rv.img_off[job.image_1] * 3, *r2 = rv.rays + rv.img_off[job.image_2] * 3;
    bool hq_lane = false;
    for (uint32_t i = lane; i < M; i += W)
    {
        const ochip_ransac_match mt = matches[mo + i];
        const ochip_ransac_match me = matches[mo + eval_order[i]];
        const double ax = r1[(size_t)mt.k1 * 3], ay = r1[(size_t)mt.k1 * 3 + 1], az = r1[(size_t)mt.k1 * 3 + 2];
        const double bx = r2[(size_t)mt.k2 * 3], by = r2[(size_t)mt.k2 * 3 + 1], bz = r2[(size_t)mt.k2 * 3 + 2];
        const double eax = r1[(size_t)me.k1 * 3], eay = r1[(size_t)me.k1 * 3 + 1], eaz = r1[(size_t)me.k1 * 3 + 2];
        const double ebx = r2[(size_t)me.k2 * 3], eby = r2[(size_t)me.k2 * 3 + 1], ebz = r2[(size_t)me.k2 * 3 + 2];
        cx1[i] = ax / az;
        cy1[i] = ay / az;
        cx2[i] = bx / bz;
        cy2[i] = by / bz;
        ex1[i] = eax / eaz;
        ey1[i] = eay / eaz;
        ex2[i] = ebx / ebz;
        ey2[i] = eby / ebz;
        hq_lane |= (mt.count != 0); // quality = count * (1/486) != 0  <=>  count != 0
    }
    const bool has_quality = __ballot(hq_lane) != 0;
    __syncthreads();
    OCHIP_PHASE_ADD(t_total, 0);
    pd.x1 = cx1;
    pd.y1 = cy1;
    pd.x2 = cx2;
    pd.y2 = cy2;
    pd.ex1 = ex1;
    pd.ey1 = ey1;
    pd.ex2 = ex2;
    pd.ey2 = ey2;
    // two flag arrays that trade places on every improvement (the candidates of the improving model become the
    // inliers); the final evaluation writes the caller's array
    pd.cand = flag_scratch + mo;
    pd.inl = flag_scratch + total + mo;
    pd.P = P_scratch + 81 * (size_t)job_id;
    pd.M = M;

    // (the best model lives in LDS: it is written on an improvement and read once at the end, and 36 registers held across
    // the factorisations of the local optimisation were the difference between two wavefronts per SIMD and one)
    __shared__ double best_model_lds[18];
    auto keep_best = [&](const model_t &m) {
        __syncthreads();
        if (lane < 9)
        {
            best_model_lds[lane] = m.H[lane];
            best_model_lds[9 + lane] = m.Hi[lane];
        }
    };
    model_t model;
    set_nan(model);
    keep_best(model);
    double best_score = 0;
    uint32_t rng = job.rng_state;
    uint32_t prosac_n = has_quality ? 4u : M;
    uint32_t probability_iterations = MAX_ITERATIONS;
    const double log_1m_p = log(1 - 0.999);
    uint32_t it = 0;

    for (; it < probability_iterations; it++)
    {
        if (best_score > 0 && probability_iterations - it >= 4)
        {
            // look at the next iterations side by side and skip those that change nothing
            const uint32_t n = min((uint32_t)FB, probability_iterations - it);
            bool found;
            it += fast_forward(pd, sorted_idx, has_quality, it, n, rng, prosac_n, best_score, thr, AQs, &found);
            if (!found)
            {
                it--; // the loop's increment: the next iteration to look at is `it`
                continue;
            }
        }
        if (has_quality && prosac_n < M && it > 0 && it % 10 == 0)
            prosac_n++;

        OCHIP_PHASE_T0(t_sample);
        // ---- minimal sample (ransac.cpp:104-154)
        uint32_t s4[4];
        if (has_quality && prosac_n < M && prosac_n > 4)
        {
            uint32_t c[4];
            c[0] = prosac_n - 1; // never drawn again: the other three come from [0, prosac_n - 2]
            draw_distinct(rng, prosac_n - 2, 1, c);
            for (int j = 0; j < 4; j++)
                s4[j] = sorted_idx[c[j]];
        }
        else
        {
            uint32_t c[4];
            draw_distinct(rng, (has_quality ? prosac_n : M) - 1, 0, c);
            for (int j = 0; j < 4; j++)
                s4[j] = has_quality ? sorted_idx[c[j]] : c[j];
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
        full_piv_lu_solve9(P9, 9, sol);
        model_from_solution(model, sol);
        OCHIP_PHASE_ADD(t_sample, 3);

        // ---- SPRT-pruned MSAC scoring in shuffled order (ransac.cpp:177-205)
        bool rejected;
        uint32_t n_inl = 0;
        OCHIP_PHASE_T0(t_score);
        const double score = score_model<true>(model, pd, eval_order, pd.cand, thr, best_score, &rejected, &n_inl);
        OCHIP_PHASE_ADD(t_score, 4);
        OCHIP_PHASE_COUNT(9, 1);
        if (rejected)
            continue;

        if (score > best_score)
        {
            res.improvements++;
            keep_best(model);
            best_score = score;
            __syncthreads();
            {
                uint8_t *t = pd.cand; // inliers = candidate flags of this model
                pd.cand = pd.inl;
                pd.inl = t;
            }
            uint32_t n_in = n_inl; // the scoring was not cut short, so it counted every inlier it flagged

            // local optimisation: fitInliers + evaluate, up to MAX_INNER_ITERATIONS (ransac.cpp:224-245)
            for (uint32_t j = 0; j < MAX_INNER_ITERATIONS; j++)
            {
                OCHIP_PHASE_T0(t_fi);
                fit_inliers(pd, n_in, T9, reinterpret_cast<tall_tab *>(AQs), model); // (the lane fits' LDS is idle here)
                OCHIP_PHASE_ADD(t_fi, 5);
                OCHIP_PHASE_COUNT(10, 1);
                bool dummy;
                uint32_t cnt = 0;
                __syncthreads();
                OCHIP_PHASE_T0(t_sf);
                const double inlier_score = score_model<false>(model, pd, nullptr, pd.inl, thr, 0.0, &dummy, &cnt);
                __syncthreads();
                OCHIP_PHASE_ADD(t_sf, 6);
                n_in = cnt;
                if (inlier_score > best_score)
                {
                    keep_best(model);
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
    for (int i = 0; i < 9; i++)
    {
        model.H[i] = best_model_lds[i];
        model.Hi[i] = best_model_lds[9 + i];
    }
    const double final_score = score_model<false>(model, pd, nullptr, inliers_out + mo, thr, 0.0, &dummy, &cnt);
    for (int i = 0; i < 9; i++)
        res.H[i] = model.H[i];
    res.score = final_score / (double)M;
    res.iterations = it;
    res.n_inliers = cnt;
    if (lane == 0)
        results[job_id] = res;
    OCHIP_PHASE_ADD(t_total, 7);
    OCHIP_PHASE_COUNT(11, res.improvements);
#ifdef OCHIP_RANSAC_PHASES
    if (threadIdx.x == 0)
    {
        const unsigned long long mine = clock64() - t_total;
        atomicMax(&g_phase[13], mine);
        if (mine > 20000000ull)
            atomicAdd(&g_phase[14], 1ull);
        if (mine > 10000000ull)
            atomicAdd(&g_phase[15], 1ull);
    }
#endif
}

// ---- re-fit of an edge's homography on its previous inliers after the camera models changed
//      (RelaxGroup::finalize, src/relax/relax_group.cpp:137-177): rays from the current models, then `rounds` times
//      fitInliers + evaluate starting from the previous inlier set.  One wavefront per edge, the RANSAC kernel's
//      device functions.
__global__ __launch_bounds__(W, 2) __attribute__((amdgpu_num_vgpr(248))) void refit_homography_kernel(
    const ochip_ransac_job *__restrict__ jobs, const ochip_ransac_match *__restrict__ matches, rays_view rv,
    double *__restrict__ coord_scratch /*4 x total*/, double *__restrict__ P_scratch /*81 x n_jobs*/, uint64_t total,
    double thr, uint32_t rounds, ochip_ransac_result *__restrict__ results, uint8_t *__restrict__ inliers /*in: previous, out: new*/)
{
    __shared__ double T9[81];
    __shared__ tall_tab tall;
    const int lane = threadIdx.x;
    const uint32_t job_id = blockIdx.x;
    const ochip_ransac_job job = jobs[job_id];
    const uint32_t M = job.n;
    const uint64_t mo = job.match_offset;
    pair_data pd;
    double *cx1 = coord_scratch + mo, *cy1 = coord_scratch + total + mo, *cx2 = coord_scratch + 2 * total + mo,
           *cy2 = coord_scratch + 3 * total + mo;
    const double *r1 = rv.rays + rv.img_off[job.image_1] * 3, *r2 = rv.rays + rv.img_off[job.image_2] * 3;
    uint32_t n_in = 0;
    for (uint32_t base = 0; base < M; base += W)
    {
        const uint32_t i = base + lane;
        bool f = false;
        if (i < M)
        {
            const ochip_ransac_match mt = matches[mo + i];
            const double ax = r1[(size_t)mt.k1 * 3], ay = r1[(size_t)mt.k1 * 3 + 1], az = r1[(size_t)mt.k1 * 3 + 2];
            const double bx = r2[(size_t)mt.k2 * 3], by = r2[(size_t)mt.k2 * 3 + 1], bz = r2[(size_t)mt.k2 * 3 + 2];
            cx1[i] = ax / az;
            cy1[i] = ay / az;
            cx2[i] = bx / bz;
            cy2[i] = by / bz;
            f = inliers[mo + i] != 0;
        }
        n_in += __popcll(__ballot(f));
    }
    __syncthreads();
    pd.x1 = cx1;
    pd.y1 = cy1;
    pd.x2 = cx2;
    pd.y2 = cy2;
    pd.ex1 = pd.ey1 = pd.ex2 = pd.ey2 = nullptr;
    pd.cand = nullptr;
    pd.inl = inliers + mo;
    pd.P = P_scratch + 81 * (size_t)job_id;
    pd.M = M;
    model_t model;
    set_nan(model);
    double score = 0;
    uint32_t cnt = n_in;
    for (uint32_t r = 0; r < rounds; r++)
    {
        fit_inliers(pd, cnt, T9, &tall, model);
        bool dummy;
        __syncthreads();
        score = score_model<false>(model, pd, nullptr, pd.inl, thr, 0.0, &dummy, &cnt);
        __syncthreads();
    }
    ochip_ransac_result res;
    for (int i = 0; i < 9; i++)
        res.H[i] = model.H[i];
    res.score = M ? score / (double)M : 0.0;
    res.iterations = rounds;
    res.n_inliers = cnt;
    res.improvements = 0;
    res.reserved = 0;
    if (lane == 0)
        results[job_id] = res;
}

// ---- a9: the fundamental- and essential-matrix models (src/model_inliers/fundamental_matrix_model.cpp:12-217,
//      essential_matrix_model.cpp:12-123) under the same ransac<Model> loop (ransac.cpp:53-257).  No pipeline stage calls
//      them (link_stage.cpp:91-98 instantiates the homography model only); the reference's unit tests and benchmarks do.
//      One wavefront per job as above: the loop is wave-uniform; the 64 lanes share the scoring walks (Sampson error per
//      correspondence, MSAC sum in walk order, SPRT exit) and the 81 entries of A'A; the Jacobi SVDs (9 x 9 of A'A for
//      the null vector, 3 x 3 for the rank / singular-value constraint and DEGENSAC's epipole) are a few hundred
//      dependent rotations and run on one lane out of LDS.  Eigen::JacobiSVD is restated from its algorithm (two-sided
//      Jacobi on the scaled matrix, sweeps p > q until every off-diagonal pair is below precision * max |diagonal|,
//      singular values positive and sorted decreasing); DEGENSAC's homography fits are the kernel's own
//      (full_piv_lu_solve9, fit_inliers).
struct emodel_t
{
    double F[9]; // row-major
};

__device__ __forceinline__ double sampson_error(const emodel_t &m, double x1, double y1, double x2, double y2)
{
    const double *F = m.F;
    const double fx = F[0] * x1 + F[1] * y1 + F[2] * 1.0, fy = F[3] * x1 + F[4] * y1 + F[5] * 1.0,
                 fz = F[6] * x1 + F[7] * y1 + F[8] * 1.0;                                    // F x1
    const double tx = F[0] * x2 + F[3] * y2 + F[6] * 1.0, ty = F[1] * x2 + F[4] * y2 + F[7] * 1.0; // F' x2 (first two)
    const double x2tFx1 = x2 * fx + y2 * fy + 1.0 * fz;
    const double denom = fx * fx + fy * fy + tx * tx + ty * ty;
    if (denom < 1e-20)
        return 1.7976931348623157e308;
    return sqrt((x2tFx1 * x2tFx1) / denom);
}

// the scoring walk of score_model with the Sampson error (same accumulation: one element at a time in walk order)
template <bool ORDERED>
__device__ double score_epipolar(const emodel_t &m, const pair_data &pd, const uint32_t *__restrict__ order, uint8_t *flags, double thr,
                                 double best_score, bool *rejected, uint32_t *n_inliers)
{
    const int lane = threadIdx.x;
    const uint32_t M = pd.M;
    const double *X1 = ORDERED ? pd.ex1 : pd.x1, *Y1 = ORDERED ? pd.ey1 : pd.y1, *X2 = ORDERED ? pd.ex2 : pd.x2,
                 *Y2 = ORDERED ? pd.ey2 : pd.y2;
    double s = 0;
    uint32_t count = 0;
    *rejected = false;
    for (uint32_t base = 0; base < M; base += W)
    {
        const uint32_t pos = base + lane;
        const bool valid = pos < M;
        const uint32_t idx = valid ? (ORDERED ? order[pos] : pos) : 0;
        const double e = valid ? sampson_error(m, X1[pos], Y1[pos], X2[pos], Y2[pos]) : 1.7976931348623157e308;
        const bool inl = valid && (e < thr);
        double term = 0;
        if (inl)
        {
            const double ratio = e / thr;
            term = 1.0 - ratio * ratio;
        }
        if (valid)
            flags[idx] = inl ? 1 : 0;
        const unsigned long long mask = __ballot(inl);
        count += __popcll(mask);
        double pref = s;
        if (mask)
        {
#pragma unroll
            for (int l = 0; l < W; l++)
            {
                s = s + bcast(term, l);
                if (ORDERED)
                    pref = lane == l ? s : pref;
            }
        }
        if (ORDERED)
        {
            const uint32_t checked = pos + 1;
            const bool rej = valid && checked > 20 && best_score > 0 && pref < best_score * (double)checked / (double)M * 0.6;
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

// JacobiRotation::makeJacobi(x, y, z) for the real symmetric 2 x 2 block [x y; y z]
__device__ __forceinline__ void make_jacobi(double x, double y, double z, double *c, double *sn)
{
    const double deno = 2.0 * fabs(y);
    if (deno < 2.2250738585072014e-308)
    {
        *c = 1.0;
        *sn = 0.0;
        return;
    }
    const double tau = (x - z) / deno;
    const double w = sqrt(tau * tau + 1.0);
    const double t = tau > 0 ? 1.0 / (tau + w) : 1.0 / (tau - w);
    const double sign_t = t > 0 ? 1.0 : -1.0;
    const double n = 1.0 / sqrt(t * t + 1.0);
    *c = n;
    *sn = -sign_t * (y / fabs(y)) * fabs(t) * n;
}

// Eigen::JacobiSVD<Matrix<double, n, n>>(ComputeFullU | ComputeFullV) on row-major n x n arrays in LDS; ONE lane runs it.
// Wm: the matrix (destroyed), U, V: n x n, S: n.
__device__ void jacobi_svd_lane(double *Wm, int n, double *U, double *S, double *V)
{
    const double precision = 2.0 * 2.220446049250313e-16, consider_as_zero = 2.2250738585072014e-308;
    double scale = 0;
    for (int i = 0; i < n * n; i++)
        scale = fmax(scale, fabs(Wm[i]));
    if (scale == 0)
        scale = 1;
    for (int i = 0; i < n * n; i++)
    {
        Wm[i] /= scale;
        U[i] = V[i] = 0.0;
    }
    for (int i = 0; i < n; i++)
        U[i * n + i] = V[i * n + i] = 1.0;
    double max_diag = 0;
    for (int i = 0; i < n; i++)
        max_diag = fmax(max_diag, fabs(Wm[i * n + i]));
    bool finished = false;
    while (!finished)
    {
        finished = true;
        for (int p = 1; p < n; p++)
            for (int q = 0; q < p; q++)
            {
                const double threshold = fmax(consider_as_zero, precision * max_diag);
                if (fabs(Wm[p * n + q]) > threshold || fabs(Wm[q * n + p]) > threshold)
                {
                    finished = false;
                    // real_2x2_jacobi_svd: a rotation that makes the block symmetric, then makeJacobi
                    const double m00 = Wm[p * n + p], m01 = Wm[p * n + q], m10 = Wm[q * n + p], m11 = Wm[q * n + q];
                    const double t = m00 + m11, d = m10 - m01;
                    double r1c, r1s;
                    if (fabs(d) < 2.2250738585072014e-308)
                        r1c = 1.0, r1s = 0.0;
                    else
                    {
                        const double u = t / d, tmp = sqrt(1.0 + u * u);
                        r1c = u / tmp;
                        r1s = 1.0 / tmp;
                    }
                    const double n00 = r1c * m00 + r1s * m10, n01 = r1c * m01 + r1s * m11, n11 = -r1s * m01 + r1c * m11;
                    double jrc, jrs;
                    make_jacobi(n00, n01, n11, &jrc, &jrs);
                    // j_left = rot1 * j_right.transpose()
                    const double jtc = jrc, jts = -jrs;
                    const double jlc = r1c * jtc - r1s * jts, jls = r1c * jts + r1s * jtc;
                    for (int k = 0; k < n; k++) // Wm.applyOnTheLeft(p, q, j_left)
                    {
                        const double x = Wm[p * n + k], y = Wm[q * n + k];
                        Wm[p * n + k] = jlc * x + jls * y;
                        Wm[q * n + k] = -jls * x + jlc * y;
                    }
                    for (int k = 0; k < n; k++) // U.applyOnTheRight(p, q, j_left.transpose())
                    {
                        const double x = U[k * n + p], y = U[k * n + q];
                        U[k * n + p] = jlc * x - (-jls) * y;
                        U[k * n + q] = (-jls) * x + jlc * y;
                    }
                    for (int k = 0; k < n; k++) // Wm.applyOnTheRight(p, q, j_right)
                    {
                        const double x = Wm[k * n + p], y = Wm[k * n + q];
                        Wm[k * n + p] = jrc * x - jrs * y;
                        Wm[k * n + q] = jrs * x + jrc * y;
                    }
                    for (int k = 0; k < n; k++) // V.applyOnTheRight(p, q, j_right)
                    {
                        const double x = V[k * n + p], y = V[k * n + q];
                        V[k * n + p] = jrc * x - jrs * y;
                        V[k * n + q] = jrs * x + jrc * y;
                    }
                    max_diag = fmax(max_diag, fmax(fabs(Wm[p * n + p]), fabs(Wm[q * n + q])));
                }
            }
    }
    for (int i = 0; i < n; i++)
    {
        const double a = Wm[i * n + i];
        S[i] = fabs(a);
        if (a < 0)
            for (int k = 0; k < n; k++)
                U[k * n + i] = -U[k * n + i];
        S[i] *= scale;
    }
    for (int i = 0; i < n; i++) // decreasing singular values, the columns of U and V follow
    {
        int pos = i;
        for (int k = i + 1; k < n; k++)
            if (S[k] > S[pos])
                pos = k;
        if (S[pos] == 0)
            break;
        if (pos != i)
        {
            const double ts = S[i];
            S[i] = S[pos];
            S[pos] = ts;
            for (int k = 0; k < n; k++)
            {
                const double tu = U[k * n + i], tv = V[k * n + i];
                U[k * n + i] = U[k * n + pos];
                U[k * n + pos] = tu;
                V[k * n + i] = V[k * n + pos];
                V[k * n + pos] = tv;
            }
        }
    }
}

// LDS work area of the SVDs
struct svd_lds
{
    double A[81], U[81], V[81], S[9], B[9], U3[9], V3[9], S3[3], out[9];
};

// the rank-2 / equal-singular-value constraint (calculateFundamentalMatrix / calculateEssentialMatrix second halves) on the
// 3 x 3 matrix in L.B -> L.out.  One lane.
__device__ void enforce_rank2_lane(svd_lds &L, bool equal_singular_values)
{
    jacobi_svd_lane(L.B, 3, L.U3, L.S3, L.V3);
    double sv[3] = {L.S3[0], L.S3[1], 0.0};
    if (equal_singular_values)
        sv[0] = sv[1] = (L.S3[0] + L.S3[1]) / 2.0;
    for (int r = 0; r < 3; r++)
        for (int c = 0; c < 3; c++)
        {
            double v = 0;
            for (int k = 0; k < 3; k++)
                v += L.U3[r * 3 + k] * sv[k] * L.V3[c * 3 + k];
            L.out[r * 3 + c] = v;
        }
}

// entry i of the row  x x', x y', x, y x', y y', y, x', y', 1  of a correspondence
__device__ __forceinline__ double epi_row(double x, double y, double x_, double y_, int i)
{
    switch (i)
    {
    case 0:
        return x * x_;
    case 1:
        return x * y_;
    case 2:
        return x;
    case 3:
        return y * x_;
    case 4:
        return y * y_;
    case 5:
        return y;
    case 6:
        return x_;
    case 7:
        return y_;
    default:
        return 1.0;
    }
}

// model from A'A in L.A (all lanes arrive; lane 0 works): null vector, constraint -> m
__device__ void epipolar_from_gram(svd_lds &L, bool essential, emodel_t &m)
{
    __syncthreads();
    if (threadIdx.x == 0)
    {
        jacobi_svd_lane(L.A, 9, L.U, L.S, L.V);
        for (int e = 0; e < 9; e++)
            L.B[e] = L.V[e * 9 + 8];
        enforce_rank2_lane(L, essential);
    }
    __syncthreads();
    for (int e = 0; e < 9; e++)
        m.F[e] = L.out[e];
    __syncthreads();
}

// fit on a minimal sample (fit(), :41-61 / :39-58): rows in sample order
template <int K> __device__ void epipolar_fit_sample(const pair_data &pd, const uint32_t *s, svd_lds &L, bool essential, emodel_t &m)
{
    __syncthreads();
    for (int e = threadIdx.x; e < 81; e += W)
    {
        const int i = e / 9, j = e % 9;
        double v = 0;
        for (int k = 0; k < K; k++)
        {
            const double x = pd.x1[s[k]], y = pd.y1[s[k]], x_ = pd.x2[s[k]], y_ = pd.y2[s[k]];
            v += epi_row(x, y, x_, y_, i) * epi_row(x, y, x_, y_, j);
        }
        L.A[e] = v;
    }
    epipolar_from_gram(L, essential, m);
}

// fitInliers (:63-90 / :60-87) on the flags `inl`, of which n_in are set: returns false (model untouched) below K
template <int K>
__device__ bool epipolar_fit_inliers(const pair_data &pd, const uint8_t *inl, uint32_t n_in, svd_lds &L, bool essential, emodel_t &m)
{
    if (n_in < (uint32_t)K)
        return false;
    __syncthreads();
    for (int e = threadIdx.x; e < 81; e += W)
    {
        const int i = e / 9, j = e % 9;
        double v = 0;
        for (uint32_t k = 0; k < pd.M; k++)
            if (inl[k])
            {
                const double x = pd.x1[k], y = pd.y1[k], x_ = pd.x2[k], y_ = pd.y2[k];
                v += epi_row(x, y, x_, y_, i) * epi_row(x, y, x_, y_, j);
            }
        L.A[e] = v;
    }
    epipolar_from_gram(L, essential, m);
    return true;
}

template <int K> __device__ __forceinline__ void draw_distinct_k(uint32_t &rng, uint32_t hi, int first, uint32_t *c)
{
    const uniform_range range = make_range(hi);
    for (int j = first; j < K; j++)
    {
        uint32_t cand;
        bool unique;
        do
        {
            cand = uniform_int(rng, range);
            unique = true;
            for (int k = first; k < j; k++)
                if (c[k] == cand)
                    unique = false;
        } while (!unique);
        c[j] = cand;
    }
}

// fundamental_matrix_model::checkDegeneracy (DEGENSAC, :123-215).  inl: the model's inlier flags (rewritten when the
// candidate wins); f1 / f2 / f3: three more flag arrays of M bytes.  Returns the number of inliers of `inl` afterwards.
__device__ uint32_t degensac(const pair_data &pd, emodel_t &model, uint8_t *&inl, uint8_t *f1, uint8_t *&f2, uint8_t *f3, uint32_t n_f,
                             double thr, double *P9, double *T9, svd_lds &L)
{
    const int lane = threadIdx.x;
    const uint32_t M = pd.M;
    if (n_f < 4)
        return n_f;
    // the first four F inliers in index order
    uint32_t h4[4] = {0, 0, 0, 0};
    {
        uint32_t found = 0;
        for (uint32_t base = 0; base < M && found < 4; base += W)
        {
            const uint32_t i = base + lane;
            const unsigned long long mask = __ballot(i < M && inl[i]);
            unsigned long long mm = mask;
            while (mm && found < 4)
            {
                h4[found++] = base + (uint32_t)__builtin_ctzll(mm);
                mm &= mm - 1;
            }
        }
    }
    const double thr2 = thr * 2;
    model_t h;
    __syncthreads();
    if (lane < 4)
        write_dlt_rows(P9, 9, 2 * lane, pd.x1[h4[lane]], pd.y1[h4[lane]], pd.x2[h4[lane]], pd.y2[h4[lane]]);
    if (lane < 9)
        P9[lane * 9 + 8] = lane == 8 ? 1.0 : 0.0;
    double sol[9];
    full_piv_lu_solve9(P9, 9, sol);
    model_from_solution(h, sol);
    // F inliers that are also inliers of that homography
    uint32_t h_count = 0;
    for (uint32_t base = 0; base < M; base += W)
    {
        const uint32_t i = base + lane;
        bool f = false;
        if (i < M)
        {
            f = inl[i] && transfer_error(h, pd.x1[i], pd.y1[i], pd.x2[i], pd.y2[i]) < thr2;
            f1[i] = f ? 1 : 0;
        }
        h_count += __popcll(__ballot(f));
    }
    const double h_ratio = (double)h_count / (double)n_f;
    if (h_ratio < 0.7)
        return n_f;
    __syncthreads();
    {
        __shared__ tall_tab tab;
        pair_data ph = pd;
        ph.inl = f1;
        fit_inliers(ph, h_count, T9, &tab, h);
    }
    __syncthreads();
    // off-plane F inliers: their Gram matrix of (x2 x H x1)
    uint32_t non_h = 0;
    for (uint32_t base = 0; base < M; base += W)
    {
        const uint32_t i = base + lane;
        bool off = false;
        if (i < M)
        {
            off = inl[i] && !(transfer_error(h, pd.x1[i], pd.y1[i], pd.x2[i], pd.y2[i]) < thr2);
            f3[i] = off ? 1 : 0;
        }
        non_h += __popcll(__ballot(off));
    }
    if (non_h < 2)
        return n_f;
    __syncthreads();
    if (lane < 9)
    {
        const int a = lane / 3, b = lane % 3;
        double v = 0;
        for (uint32_t i = 0; i < M; i++)
            if (f3[i])
            {
                const double x1 = pd.x1[i], y1 = pd.y1[i], x2 = pd.x2[i], y2 = pd.y2[i];
                const double hx = h.H[0] * x1 + h.H[1] * y1 + h.H[2] * 1.0, hy = h.H[3] * x1 + h.H[4] * y1 + h.H[5] * 1.0,
                             hz = h.H[6] * x1 + h.H[7] * y1 + h.H[8] * 1.0;
                const double r[3] = {y2 * hz - 1.0 * hy, 1.0 * hx - x2 * hz, x2 * hy - y2 * hx}; // x2 x (H x1)
                v += r[a] * r[b];
            }
        L.B[lane] = v;
    }
    __syncthreads();
    if (lane == 0)
    {
        // the epipole: the right singular vector of the smallest singular value (the eigenvector of the Gram matrix's)
        jacobi_svd_lane(L.B, 3, L.U3, L.S3, L.V3);
        const double ex = L.V3[0 * 3 + 2], ey = L.V3[1 * 3 + 2], ez = L.V3[2 * 3 + 2];
        const double ec[9] = {0, -ez, ey, ez, 0, -ex, -ey, ex, 0};
        for (int r = 0; r < 3; r++)
            for (int c = 0; c < 3; c++)
            {
                double v = 0;
                for (int k = 0; k < 3; k++)
                    v += ec[r * 3 + k] * h.H[k * 3 + c];
                L.B[r * 3 + c] = v;
            }
        enforce_rank2_lane(L, false);
    }
    __syncthreads();
    emodel_t cand;
    for (int e = 0; e < 9; e++)
        cand.F[e] = L.out[e];
    __syncthreads();
    // keep the candidate only if it scores better (both evaluated in natural order)
    bool dummy;
    uint32_t n_c = 0, n_o = 0;
    const double candidate_score = score_epipolar<false>(cand, pd, nullptr, f2, thr, 0.0, &dummy, &n_c);
    __syncthreads();
    const double original_score = score_epipolar<false>(model, pd, nullptr, inl, thr, 0.0, &dummy, &n_o);
    __syncthreads();
    if (candidate_score > original_score)
    {
        model = cand;
        uint8_t *t = inl;
        inl = f2;
        f2 = t;
        return n_c;
    }
    return n_o;
}

struct ochip_epipolar_job_dev
{
    uint32_t n, rng_state, has_quality, reserved;
    uint64_t corr_offset, eval_offset;
};

template <int K, bool ESSENTIAL>
__global__ __launch_bounds__(W, 2) __attribute__((amdgpu_num_vgpr(248))) void ransac_epipolar_kernel(
    const ochip_epipolar_job_dev *__restrict__ jobs, const double *__restrict__ corr6, const uint32_t *__restrict__ sorted_idx_all,
    const uint32_t *__restrict__ eval_order_all, double *__restrict__ coord_scratch /*8 x total*/,
    uint8_t *__restrict__ flag_scratch /*5 x total*/, double *__restrict__ P_scratch /*81 x n_jobs*/, uint64_t total,
    double thr, ochip_ransac_result *__restrict__ results, uint8_t *__restrict__ inliers_out)
{
    __shared__ double P9[81], T9[81];
    __shared__ svd_lds L;
    const int lane = threadIdx.x;
    const uint32_t job_id = blockIdx.x;
    const ochip_epipolar_job_dev job = jobs[job_id];
    const uint32_t M = job.n;
    const uint64_t mo = job.corr_offset;
    ochip_ransac_result res;
    for (int i = 0; i < 9; i++)
        res.H[i] = __builtin_nan("");
    res.score = 0;
    res.iterations = 0;
    res.n_inliers = 0;
    res.improvements = 0;
    res.reserved = 0;
    if (M < (uint32_t)K) // ransac.cpp:69-72
    {
        for (uint32_t i = lane; i < M; i += W)
            inliers_out[mo + i] = 0;
        if (lane == 0)
            results[job_id] = res;
        return;
    }
    const uint32_t *sorted_idx = sorted_idx_all + mo;
    const uint32_t *eval_order = eval_order_all + job.eval_offset;
    pair_data pd;
    double *cx1 = coord_scratch + mo, *cy1 = coord_scratch + total + mo, *cx2 = coord_scratch + 2 * total + mo,
           *cy2 = coord_scratch + 3 * total + mo;
    double *ex1 = coord_scratch + 4 * total + mo, *ey1 = coord_scratch + 5 * total + mo, *ex2 = coord_scratch + 6 * total + mo,
           *ey2 = coord_scratch + 7 * total + mo;
    for (uint32_t i = lane; i < M; i += W)
    {
        const double *c = corr6 + 6 * (mo + i), *ce = corr6 + 6 * (mo + eval_order[i]);
        cx1[i] = c[0] / c[2];
        cy1[i] = c[1] / c[2];
        cx2[i] = c[3] / c[5];
        cy2[i] = c[4] / c[5];
        ex1[i] = ce[0] / ce[2];
        ey1[i] = ce[1] / ce[2];
        ex2[i] = ce[3] / ce[5];
        ey2[i] = ce[4] / ce[5];
    }
    const bool has_quality = job.has_quality != 0;
    __syncthreads();
    pd.x1 = cx1, pd.y1 = cy1, pd.x2 = cx2, pd.y2 = cy2;
    pd.ex1 = ex1, pd.ey1 = ey1, pd.ex2 = ex2, pd.ey2 = ey2;
    pd.cand = flag_scratch + mo;
    pd.inl = flag_scratch + total + mo;
    uint8_t *f1 = flag_scratch + 2 * total + mo, *f2 = flag_scratch + 3 * total + mo, *f3 = flag_scratch + 4 * total + mo;
    pd.P = P_scratch + 81 * (size_t)job_id;
    pd.M = M;

    emodel_t model, best_model;
    for (int e = 0; e < 9; e++)
        model.F[e] = best_model.F[e] = __builtin_nan("");
    double best_score = 0;
    uint32_t rng = job.rng_state;
    uint32_t prosac_n = has_quality ? (uint32_t)K : M;
    uint32_t probability_iterations = MAX_ITERATIONS;
    const double log_1m_p = log(1 - 0.999);
    uint32_t it = 0;
    for (; it < probability_iterations; it++)
    {
        if (has_quality && prosac_n < M && it > 0 && it % 10 == 0)
            prosac_n++;
        uint32_t sk[K], c[K];
        if (has_quality && prosac_n < M && prosac_n > (uint32_t)K)
        {
            c[0] = prosac_n - 1;
            draw_distinct_k<K>(rng, prosac_n - 2, 1, c);
            for (int j = 0; j < K; j++)
                sk[j] = sorted_idx[c[j]];
        }
        else
        {
            draw_distinct_k<K>(rng, (has_quality ? prosac_n : M) - 1, 0, c);
            for (int j = 0; j < K; j++)
                sk[j] = has_quality ? sorted_idx[c[j]] : c[j];
        }
        epipolar_fit_sample<K>(pd, sk, L, ESSENTIAL, model);
        bool rejected;
        uint32_t n_inl = 0;
        const double score = score_epipolar<true>(model, pd, eval_order, pd.cand, thr, best_score, &rejected, &n_inl);
        if (rejected)
            continue;
        if (score > best_score)
        {
            res.improvements++;
            best_model = model;
            best_score = score;
            __syncthreads();
            {
                uint8_t *t = pd.cand; // inliers = candidate flags of this model
                pd.cand = pd.inl;
                pd.inl = t;
            }
            uint32_t n_in = n_inl;
            if (!ESSENTIAL)
            {
                // checkDegeneracy, then model.evaluate(matches, inliers) (ransac.cpp:213-222)
                n_in = degensac(pd, model, pd.inl, f1, f2, f3, n_in, thr, P9, T9, L);
                bool dummy;
                uint32_t cnt = 0;
                __syncthreads();
                const double degen_score = score_epipolar<false>(model, pd, nullptr, pd.inl, thr, 0.0, &dummy, &cnt);
                __syncthreads();
                n_in = cnt;
                if (degen_score > best_score)
                {
                    best_model = model;
                    best_score = degen_score;
                }
            }
            // local optimisation (ransac.cpp:224-245)
            for (uint32_t j = 0; j < MAX_INNER_ITERATIONS; j++)
            {
                epipolar_fit_inliers<K>(pd, pd.inl, n_in, L, ESSENTIAL, model);
                bool dummy;
                uint32_t cnt = 0;
                __syncthreads();
                const double inlier_score = score_epipolar<false>(model, pd, nullptr, pd.inl, thr, 0.0, &dummy, &cnt);
                __syncthreads();
                n_in = cnt;
                if (inlier_score > best_score)
                {
                    best_model = model;
                    best_score = inlier_score;
                }
                else
                    break;
            }
            const double omega = best_score / (double)M;
            double omega_n;
            if (K == 8)
            {
                double t = omega * omega;
                t = t * t;
                omega_n = t * t; // fast_pow<8>
            }
            else
            {
                const double t = omega * omega;
                omega_n = t * t * omega; // fast_pow<5>
            }
            const double log_1m_omega_n = log(1 - omega_n);
            const double q = log_1m_p / log_1m_omega_n;
            uint64_t qi;
            if (!(q == q))
                qi = 0x8000000000000000ull;
            else if (q >= 9223372036854775808.0)
                qi = (q >= 18446744073709551616.0) ? 0ull : (uint64_t)q;
            else if (q <= -1.0)
                qi = (uint64_t)(int64_t)q;
            else
                qi = (uint64_t)q;
            const uint64_t clamped = qi < MAX_ITERATIONS ? qi : MAX_ITERATIONS;
            probability_iterations = (uint32_t)(clamped > MIN_ITERATIONS ? clamped : MIN_ITERATIONS);
        }
    }
    bool dummy;
    uint32_t cnt = 0;
    __syncthreads();
    const double final_score = score_epipolar<false>(best_model, pd, nullptr, inliers_out + mo, thr, 0.0, &dummy, &cnt);
    for (int i = 0; i < 9; i++)
        res.H[i] = best_model.F[i];
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
    OCHIP_HIP(ctx, ochip_stream_wait(ctx, ctx->stream));
    ctx->kp_set[image_id] = 1;
    ctx->rays_dirty = true;
    return OCHIP_OK;
}

} // extern "C"

namespace
{
// matches of job j out of ochip_match_sort's records: correspondence i = the i-th sorted record of the pair; and the keys
// of the PROSAC order (ransac.cpp:83-90: iota sorted by quality ASCENDING with std::sort - complemented counts)
__global__ void sorted_matches_kernel(const ochip_ransac_job *__restrict__ jobs, const unsigned int *__restrict__ seg_begin,
                                      const unsigned long long *__restrict__ recs, const ochip_match *__restrict__ raw,
                                      ochip_ransac_match *__restrict__ matches, unsigned long long *__restrict__ prosac,
                                      unsigned int *__restrict__ seg2)
{
    const unsigned int j = blockIdx.y, i = blockIdx.x * blockDim.x + threadIdx.x;
    const ochip_ransac_job jb = jobs[j];
    if (i == 0)
    {
        seg2[j] = (unsigned int)jb.match_offset;
        seg2[gridDim.y + j] = (unsigned int)jb.match_offset + jb.n;
    }
    if (i >= jb.n)
        return;
    const unsigned int off = seg_begin[j];
    const unsigned long long r = recs[off + i];
    const unsigned int a = (unsigned int)r, count = (unsigned int)(r >> 32);
    ochip_ransac_match m;
    m.k1 = a;
    m.k2 = raw[off + a].best_k;
    m.count = (uint16_t)count;
    m.reserved = 0;
    matches[jb.match_offset + i] = m;
    prosac[jb.match_offset + i] = ((unsigned long long)(0xFFFFFFFFu - count) << 32) | i;
}
// homography_model::decompose (homography_model.cpp:138-185) for every job: cv::decomposeHomographyMat by one lane
// (csrc/decompose.hpp, the host's code), the cheirality vote over the inliers' rays by the wavefront (integers: the order is
// free), the poses in std::stable_sort's order.  One wavefront per job.
__global__ __launch_bounds__(256) void decompose_vote_kernel(const ochip_ransac_job *__restrict__ jobs, unsigned int n_jobs,
                                                             const ochip_ransac_match *__restrict__ matches,
                                                             const ochip_ransac_result *__restrict__ results,
                                                             const uint8_t *__restrict__ inliers, rays_view rv,
                                                             ochip_decomposition *__restrict__ out)
{
    __shared__ ochip_dc::vote_plan plans[4];
    const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const unsigned int j = blockIdx.x * 4 + wv;
    if (j >= n_jobs)
        return;
    const ochip_ransac_job jb = jobs[j];
    ochip_dc::vote_plan &plan = plans[wv];
    if (lane == 0)
        ochip_dc::plan_votes(results[j].H, &plan);
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
    __builtin_amdgcn_wave_barrier();
    const int solutions = plan.solutions;
    const double *r1 = rv.rays + rv.img_off[jb.image_1] * 3, *r2 = rv.rays + rv.img_off[jb.image_2] * 3;
    const ochip_ransac_match *mm = matches + jb.match_offset;
    const uint8_t *inl = inliers + jb.match_offset;
    int votes[4] = {0, 0, 0, 0};
    unsigned int n_inl = 0;
    for (unsigned int i0 = 0; i0 < jb.n; i0 += 64)
    {
        const unsigned int i = i0 + lane;
        const bool on = i < jb.n && inl[i] != 0;
        bool ok[4] = {false, false, false, false};
        if (on)
        {
            const double *m1 = r1 + 3 * (size_t)mm[i].k1, *m2 = r2 + 3 * (size_t)mm[i].k2;
            const double a0 = m1[0], a1 = m1[1], a2 = m1[2], b0 = m2[0], b1 = m2[1], b2 = m2[2];
            for (int s = 0; s < 4; s++)
                if (s < solutions)
                {
                    const double dot1 = plan.N[s][0] * a0 + plan.N[s][1] * a1 + plan.N[s][2] * a2;
                    const double dot2 = plan.RN[s][0] * b0 + plan.RN[s][1] * b1 + plan.RN[s][2] * b2;
                    ok[s] = dot1 >= 0 && dot2 >= 0;
                }
        }
        n_inl += (unsigned int)__popcll(__ballot(on));
        for (int s = 0; s < 4; s++)
            votes[s] += (int)__popcll(__ballot(ok[s]));
    }
    if (lane == 0)
    {
        int score[4], order[4];
        for (int s = 0; s < 4; s++)
            score[s] = s < solutions ? votes[s] : -1;
        ochip_dc::order_by_votes(score, order);
        ochip_decomposition d;
        const double nan = __longlong_as_double(0x7FF8000000000000ll);
        for (int k = 0; k < 4; k++)
        {
            const int s = order[k];
            for (int c = 0; c < 4; c++)
                d.pose[k][c] = s < solutions ? plan.q[s][c] : nan;
            for (int c = 0; c < 3; c++)
                d.pose[k][4 + c] = s < solutions ? plan.t[s][c] : nan;
            d.pose[k][7] = (double)score[s];
        }
        d.can_decompose = score[order[0]] > 0 ? 1u : 0u;
        d.n_inliers = n_inl;
        out[j] = d;
    }
}

// The two lists an accepted edge keeps (camera_relations.matches and .inlier_matches, link_stage.cpp:99-107): pure gathers
// of what the device holds - the sorted correspondences, the inlier flags, the subsets' pixel locations and the
// keypoints' indices in their images' feature lists.  One wavefront per job; inliers in match order (ballot prefix).
struct edge_feature_match // types/feature_match.hpp:11-23
{
    uint64_t feature_index_1, feature_index_2;
    double distance;
};
struct edge_inlier_match // types/feature_match.hpp:26-39 (feature_match_denormalized)
{
    double pixel_1[2], pixel_2[2];
    uint64_t feature_index_1, feature_index_2, match_index;
};
__global__ __launch_bounds__(256) void edge_lists_kernel(const ochip_ransac_job *__restrict__ jobs, unsigned int n_jobs,
                                                         const ochip_ransac_match *__restrict__ matches,
                                                         const uint8_t *__restrict__ inliers, const uint64_t *__restrict__ img_off,
                                                         const double *__restrict__ kp_xy, const uint32_t *__restrict__ feature_index,
                                                         const uint64_t *__restrict__ inlier_offset, edge_feature_match *__restrict__ fm,
                                                         edge_inlier_match *__restrict__ fmd)
{
    const int lane = threadIdx.x & 63;
    const unsigned int j = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (j >= n_jobs)
        return;
    const ochip_ransac_job jb = jobs[j];
    const uint64_t o1 = img_off[jb.image_1], o2 = img_off[jb.image_2];
    const ochip_ransac_match *mm = matches + jb.match_offset;
    const uint8_t *inl = inliers + jb.match_offset;
    uint64_t at = inlier_offset[j];
    for (unsigned int i0 = 0; i0 < jb.n; i0 += 64)
    {
        const unsigned int i = i0 + lane;
        bool on = false;
        ochip_ransac_match m{};
        uint64_t f1 = 0, f2 = 0;
        if (i < jb.n)
        {
            m = mm[i];
            f1 = feature_index[o1 + m.k1];
            f2 = feature_index[o2 + m.k2];
            edge_feature_match r;
            r.feature_index_1 = f1;
            r.feature_index_2 = f2;
            r.distance = (double)m.count * (1.0 / 486); // count * (1.0 / feature_2d::DESCRIPTOR_BITS), match_features.cpp:90
            fm[jb.match_offset + i] = r;
            on = inl[i] != 0;
        }
        const unsigned long long mask = __ballot(on);
        if (on)
        {
            edge_inlier_match r;
            const double *p1 = kp_xy + 2 * (o1 + m.k1), *p2 = kp_xy + 2 * (o2 + m.k2);
            r.pixel_1[0] = p1[0], r.pixel_1[1] = p1[1];
            r.pixel_2[0] = p2[0], r.pixel_2[1] = p2[1];
            r.feature_index_1 = f1;
            r.feature_index_2 = f2;
            r.match_index = i;
            fmd[at + (uint64_t)__popcll(mask & ((1ull << lane) - 1ull))] = r;
        }
        at += (uint64_t)__popcll(mask);
    }
}

__global__ void prosac_order_kernel(const unsigned long long *__restrict__ prosac, uint32_t *__restrict__ sorted_idx, uint64_t total)
{
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < total)
        sorted_idx[i] = (uint32_t)prosac[i];
}

int ransac_homography_impl(ochip_ctx *ctx, const ochip_ransac_job *jobs, uint32_t n_jobs, const ochip_ransac_match *matches,
                           const uint32_t *sorted_idx, uint64_t total_matches, const uint32_t *eval_order, uint64_t eval_total,
                           double inlier_threshold, ochip_ransac_result *results, uint8_t *inliers, bool sorted_on_device,
                           ochip_ransac_match *matches_out, uint8_t *fallback_out, ochip_decomposition *decomp_out);
} // namespace

extern "C"
{

int ochip_ransac_homography_batch(ochip_ctx *ctx, const ochip_ransac_job *jobs, uint32_t n_jobs,
                                  const ochip_ransac_match *matches, const uint32_t *sorted_idx, uint64_t total_matches,
                                  const uint32_t *eval_order, uint64_t eval_total, double inlier_threshold,
                                  ochip_ransac_result *results, uint8_t *inliers)
{
    return ransac_homography_impl(ctx, jobs, n_jobs, matches, sorted_idx, total_matches, eval_order, eval_total, inlier_threshold,
                                  results, inliers, false, nullptr, nullptr, nullptr);
}

int ochip_ransac_homography_batch_sorted(ochip_ctx *ctx, const ochip_ransac_job *jobs, uint32_t n_jobs, uint64_t total_matches,
                                         const uint32_t *eval_order, uint64_t eval_total, double inlier_threshold,
                                         ochip_ransac_result *results, uint8_t *inliers, ochip_ransac_match *matches_out,
                                         uint8_t *fallback_out, ochip_decomposition *decomp_out)
{
    if (!ctx)
        return OCHIP_EINVAL;
    if (n_jobs && (!matches_out || !fallback_out))
        return ochip_fail(ctx, OCHIP_EINVAL, "NULL argument");
    if (n_jobs != ctx->ms_pairs)
        return ochip_fail(ctx, OCHIP_ESTATE, "ochip_ransac_homography_batch_sorted must follow ochip_match_sort of the same %u pairs", n_jobs);
    return ransac_homography_impl(ctx, jobs, n_jobs, nullptr, nullptr, total_matches, eval_order, eval_total, inlier_threshold, results,
                                  inliers, true, matches_out, fallback_out, decomp_out);
}

} // extern "C"

namespace
{
int ransac_homography_impl(ochip_ctx *ctx, const ochip_ransac_job *jobs, uint32_t n_jobs, const ochip_ransac_match *matches,
                           const uint32_t *sorted_idx, uint64_t total_matches, const uint32_t *eval_order, uint64_t eval_total,
                           double inlier_threshold, ochip_ransac_result *results, uint8_t *inliers, bool sorted_on_device,
                           ochip_ransac_match *matches_out, uint8_t *fallback_out, ochip_decomposition *decomp_out)
{
    if (!ctx)
        return OCHIP_EINVAL;
    if (n_jobs == 0)
        return OCHIP_OK;
    if (!jobs || !results || (total_matches && ((!sorted_on_device && (!matches || !sorted_idx)) || !inliers)) || (eval_total && !eval_order))
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
                             (size_t)T * 64,
                             (size_t)T * 2,
                             (size_t)n_jobs * 81 * 8,
                             (size_t)n_jobs * sizeof(ochip_ransac_result) + T};
    for (int i = 0; i < 8; i++)
    {
        int rc = ochip_ensure(ctx, &ctx->scratch_dev[i], &ctx->scratch_cap[i], sizes[i]);
        if (rc)
            return rc;
    }
    OCHIP_HIP(ctx, hipMemcpyAsync(ctx->scratch_dev[S_JOBS], jobs, sizes[S_JOBS], hipMemcpyHostToDevice, ctx->stream));
    std::vector<std::pair<void *, size_t>> allocs; // (std_sort's work space, returned to the pool after the wait below)
    struct put_back
    {
        ochip_ctx *ctx;
        std::vector<std::pair<void *, size_t>> *allocs;
        ~put_back()
        {
            // an early error return can leave kernels in flight that still touch these blocks: the pool must not hand
            // them to the next caller before the stream has drained (on the normal path it already has)
            if (!allocs->empty())
                (void)ochip_stream_wait(ctx, ctx->stream);
            for (auto &a : *allocs)
                ochip_pool_put(ctx, a.first, a.second);
        }
    } put_back_guard{ctx, &allocs};
    if (total_matches && !sorted_on_device)
    {
        OCHIP_HIP(ctx, hipMemcpyAsync(ctx->scratch_dev[S_MATCH], matches, (size_t)total_matches * sizeof(ochip_ransac_match),
                                      hipMemcpyHostToDevice, ctx->stream));
        OCHIP_HIP(ctx, hipMemcpyAsync(ctx->scratch_dev[S_SORTED], sorted_idx, (size_t)total_matches * 4,
                                      hipMemcpyHostToDevice, ctx->stream));
    }
    else if (total_matches)
    {
        // the correspondences out of ochip_match_sort's records, and their PROSAC order by the same device std::sort
        if (total_matches >= 0xFFFFFFFFull)
            return ochip_fail(ctx, OCHIP_EINVAL, "more than 2^32 matches in one batch");
        uint32_t max_n = 0;
        for (uint32_t j = 0; j < n_jobs; j++)
            max_n = std::max(max_n, jobs[j].n);
        size_t g0 = 0, g1 = 0, g2 = 0;
        unsigned long long *prosac = (unsigned long long *)ochip_pool_get(ctx, (size_t)total_matches * 8, &g0);
        unsigned int *seg2 = (unsigned int *)ochip_pool_get(ctx, (size_t)n_jobs * 8, &g1);
        unsigned char *fb2 = (unsigned char *)ochip_pool_get(ctx, std::max<size_t>(n_jobs, 16), &g2);
        if (prosac)
            allocs.emplace_back(prosac, g0);
        if (seg2)
            allocs.emplace_back(seg2, g1);
        if (fb2)
            allocs.emplace_back(fb2, g2);
        if (!prosac || !seg2 || !fb2)
            return ochip_fail(ctx, OCHIP_ENOMEM, "device allocation failed (PROSAC order)");
        hipLaunchKernelGGL(sorted_matches_kernel, dim3((max_n + 255) / 256, n_jobs), dim3(256), 0, ctx->stream,
                           (const ochip_ransac_job *)ctx->scratch_dev[S_JOBS], (const unsigned int *)ctx->ms_seg_dev,
                           (const unsigned long long *)ctx->ms_recs_dev, (const ochip_match *)ctx->match_out_dev,
                           (ochip_ransac_match *)ctx->scratch_dev[S_MATCH], prosac, seg2);
        const int src = ochip::std_sort_enqueue(ctx, &allocs, prosac, total_matches, seg2, seg2 + n_jobs, n_jobs, max_n, fb2);
        if (src != OCHIP_OK)
            return src;
        hipLaunchKernelGGL(prosac_order_kernel, dim3((unsigned)((total_matches + 255) / 256)), dim3(256), 0, ctx->stream,
                           (const unsigned long long *)prosac, (uint32_t *)ctx->scratch_dev[S_SORTED], total_matches);
        OCHIP_HIP(ctx, hipMemcpyAsync(matches_out, ctx->scratch_dev[S_MATCH], (size_t)total_matches * sizeof(ochip_ransac_match),
                                      hipMemcpyDeviceToHost, ctx->stream));
        OCHIP_HIP(ctx, hipMemcpyAsync(fallback_out, fb2, n_jobs, hipMemcpyDeviceToHost, ctx->stream));
    }
    else if (sorted_on_device)
        for (uint32_t j = 0; j < n_jobs; j++)
            fallback_out[j] = 0;
    if (eval_total)
        OCHIP_HIP(ctx, hipMemcpyAsync(ctx->scratch_dev[S_EVAL], eval_order, (size_t)eval_total * 4, hipMemcpyHostToDevice,
                                      ctx->stream));
    rays_view rv{ctx->rays_dev, ctx->img_off_dev};
    ochip_ransac_result *res_dev = (ochip_ransac_result *)ctx->scratch_dev[S_OUT];
    uint8_t *inl_dev = (uint8_t *)ctx->scratch_dev[S_OUT] + (size_t)n_jobs * sizeof(ochip_ransac_result);
    // A pair is one wavefront from its first sample to its last evaluation, 2 048 of them at a time, and what a pair costs
    // goes with its matches (every scoring, walk and factorisation is a sweep over them): taken in the caller's order the
    // launch ended with a few long pairs that had started late - 17.3 ms for 9 000 pairs whose work fills the device for
    // 9.8.  Longest first.
    uint32_t *order_dev = nullptr;
    std::vector<uint32_t> order(n_jobs); // (read by the copy below until the stream wait at the end)
    {
        for (uint32_t j = 0; j < n_jobs; j++)
            order[j] = j;
        std::stable_sort(order.begin(), order.end(), [&](uint32_t a, uint32_t b) { return jobs[a].n > jobs[b].n; });
        size_t go = 0;
        order_dev = (uint32_t *)ochip_pool_get(ctx, (size_t)n_jobs * 4, &go);
        if (!order_dev)
            return ochip_fail(ctx, OCHIP_ENOMEM, "device allocation failed (launch order)");
        allocs.emplace_back(order_dev, go);
        OCHIP_HIP(ctx, hipMemcpyAsync(order_dev, order.data(), (size_t)n_jobs * 4, hipMemcpyHostToDevice, ctx->stream));
    }
    hipEvent_t e0, e1;
    ochip_prof_begin(ctx, OCHIP_K_RANSAC, &e0, &e1);
    // the wave keeps two models, a sample and the LU rows of a round in VGPRs (uniform fp64 values have no scalar
    // home on gfx950): 2 waves per SIMD leaves it 256 registers, 1 leaves it 512 (measured slower)
    constexpr int occ = 2;
    auto launch = [&](auto kernel) {
        hipLaunchKernelGGL(kernel, dim3(n_jobs), dim3(W), 0, ctx->stream,
                           (const ochip_ransac_job *)ctx->scratch_dev[S_JOBS],
                           (const ochip_ransac_match *)ctx->scratch_dev[S_MATCH],
                           (const uint32_t *)ctx->scratch_dev[S_SORTED], (const uint32_t *)ctx->scratch_dev[S_EVAL], rv,
                           (double *)ctx->scratch_dev[S_COORD], (uint8_t *)ctx->scratch_dev[S_FLAGS],
                           (double *)ctx->scratch_dev[S_P], (uint64_t)T, inlier_threshold, res_dev, inl_dev, (const uint32_t *)order_dev);
    };
    static_assert(occ == 2, "the kernel's register cap is written for two wavefronts per SIMD");
    launch(ransac_homography_kernel<2>);
    ochip_prof_end(ctx, OCHIP_K_RANSAC, e0, e1);
    OCHIP_HIP(ctx, hipGetLastError());
#ifdef OCHIP_RANSAC_PHASES
    {
        unsigned long long ph[16], zero[16] = {};
        OCHIP_HIP(ctx, ochip_stream_wait(ctx, ctx->stream));
        OCHIP_HIP(ctx, hipMemcpyFromSymbol(ph, HIP_SYMBOL(g_phase), sizeof ph));
        OCHIP_HIP(ctx, hipMemcpyToSymbol(HIP_SYMBOL(g_phase), zero, sizeof zero));
        static const char *names[13] = {"prologue", "ff replay + fits", "ff walk", "sample + fit", "score (SPRT)", "fit_inliers", "score (inliers)", "total",
                                        "ff calls", "scored iterations", "inner iterations", "improvements", "plain-division trips"};
        std::fprintf(stderr, "ransac phases, %u pairs:", n_jobs);
        for (int i = 0; i < 13; i++)
            std::fprintf(stderr, " %s %.3g%s", names[i], i < 8 ? (double)ph[i] / (double)ph[7] : (double)ph[i] / n_jobs, i < 8 ? "" : "/pair");
        std::fprintf(stderr, "; total %.0f clocks per pair, slowest pair %.0f, pairs over 20 M clocks %llu, over 10 M %llu\n", (double)ph[7] / n_jobs, (double)ph[13], ph[14], ph[15]);
    }
#endif
    if (decomp_out)
    {
        size_t gd = 0;
        ochip_decomposition *dec_dev = (ochip_decomposition *)ochip_pool_get(ctx, (size_t)n_jobs * sizeof(ochip_decomposition), &gd);
        if (!dec_dev)
            return ochip_fail(ctx, OCHIP_ENOMEM, "device allocation failed (decompositions)");
        allocs.emplace_back(dec_dev, gd);
        hipLaunchKernelGGL(decompose_vote_kernel, dim3((n_jobs + 3) / 4), dim3(256), 0, ctx->stream,
                           (const ochip_ransac_job *)ctx->scratch_dev[S_JOBS], n_jobs, (const ochip_ransac_match *)ctx->scratch_dev[S_MATCH],
                           (const ochip_ransac_result *)res_dev, (const uint8_t *)inl_dev, rv, dec_dev);
        OCHIP_HIP(ctx, hipMemcpyAsync(decomp_out, dec_dev, (size_t)n_jobs * sizeof(ochip_decomposition), hipMemcpyDeviceToHost,
                                      ctx->stream));
    }
    OCHIP_HIP(ctx, hipMemcpyAsync(results, res_dev, (size_t)n_jobs * sizeof(ochip_ransac_result), hipMemcpyDeviceToHost,
                                  ctx->stream));
    if (total_matches)
        OCHIP_HIP(ctx, hipMemcpyAsync(inliers, inl_dev, (size_t)total_matches, hipMemcpyDeviceToHost, ctx->stream));
    OCHIP_HIP(ctx, ochip_stream_wait(ctx, ctx->stream));
    for (uint32_t j = 0; j < n_jobs; j++)
        ctx->ransac_hyp_corr += (uint64_t)results[j].iterations * jobs[j].n;
    return OCHIP_OK;
}
} // namespace

extern "C"
{

int ochip_edge_lists(ochip_ctx *ctx, uint32_t n_jobs, uint64_t total_matches, const uint32_t *feature_index, uint64_t n_keypoints,
                     const uint64_t *inlier_offset, uint64_t total_inliers, void *feature_match_out, void *inlier_match_out)
{
    if (!ctx)
        return OCHIP_EINVAL;
    if (n_jobs == 0)
        return OCHIP_OK;
    if (!feature_index || !inlier_offset || (total_matches && !feature_match_out) || (total_inliers && !inlier_match_out))
        return ochip_fail(ctx, OCHIP_EINVAL, "NULL argument");
    if (n_jobs != ctx->ms_pairs || n_keypoints != ctx->desc_used)
        return ochip_fail(ctx, OCHIP_ESTATE, "ochip_edge_lists must follow ochip_ransac_homography_batch_sorted of the same batch");
    OCHIP_HIP(ctx, hipSetDevice(ctx->device));
    hipStream_t st = ctx->stream;
    std::vector<std::pair<void *, size_t>> allocs;
    auto dev = [&](size_t bytes) -> void * {
        size_t got = 0;
        void *p = ochip_pool_get(ctx, std::max<size_t>(bytes, 16), &got);
        if (p)
            allocs.emplace_back(p, got);
        return p;
    };
    uint32_t *idx_dev = (uint32_t *)dev((size_t)n_keypoints * 4);
    uint64_t *off_dev = (uint64_t *)dev((size_t)n_jobs * 8);
    edge_feature_match *fm_dev = (edge_feature_match *)dev((size_t)total_matches * sizeof(edge_feature_match));
    edge_inlier_match *fmd_dev = (edge_inlier_match *)dev((size_t)total_inliers * sizeof(edge_inlier_match));
    int rc = OCHIP_OK;
    if (!idx_dev || !off_dev || !fm_dev || !fmd_dev)
        rc = ochip_fail(ctx, OCHIP_ENOMEM, "device allocation failed (edge lists)");
    if (rc == OCHIP_OK && (hipMemcpyAsync(idx_dev, feature_index, (size_t)n_keypoints * 4, hipMemcpyHostToDevice, st) != hipSuccess ||
                           hipMemcpyAsync(off_dev, inlier_offset, (size_t)n_jobs * 8, hipMemcpyHostToDevice, st) != hipSuccess))
        rc = ochip_fail(ctx, OCHIP_EHIP, "upload failed (edge lists)");
    if (rc == OCHIP_OK)
    {
        // scratch slots as ochip_ransac_homography_batch left them: 0 jobs, 1 matches, 7 results + inlier flags
        const uint8_t *inl_dev = (const uint8_t *)ctx->scratch_dev[7] + (size_t)n_jobs * sizeof(ochip_ransac_result);
        hipLaunchKernelGGL(edge_lists_kernel, dim3((n_jobs + 3) / 4), dim3(256), 0, st, (const ochip_ransac_job *)ctx->scratch_dev[0], n_jobs,
                           (const ochip_ransac_match *)ctx->scratch_dev[1], inl_dev, (const uint64_t *)ctx->img_off_dev,
                           (const double *)ctx->kp_xy_dev, (const uint32_t *)idx_dev, (const uint64_t *)off_dev, fm_dev, fmd_dev);
        if (hipGetLastError() != hipSuccess ||
            (total_matches && hipMemcpyAsync(feature_match_out, fm_dev, (size_t)total_matches * sizeof(edge_feature_match),
                                             hipMemcpyDeviceToHost, st) != hipSuccess) ||
            (total_inliers && hipMemcpyAsync(inlier_match_out, fmd_dev, (size_t)total_inliers * sizeof(edge_inlier_match),
                                             hipMemcpyDeviceToHost, st) != hipSuccess))
            rc = ochip_fail(ctx, OCHIP_EHIP, "edge lists: launch or download failed");
    }
    const hipError_t werr = ochip_stream_wait(ctx, st);
    for (auto &a : allocs)
        ochip_pool_put(ctx, a.first, a.second);
    if (rc == OCHIP_OK && werr != hipSuccess)
        rc = ochip_fail(ctx, OCHIP_EHIP, "edge lists: %s", hipGetErrorString(werr));
    return rc;
}

int ochip_refit_homography_batch(ochip_ctx *ctx, const ochip_ransac_job *jobs, uint32_t n_jobs, const ochip_ransac_match *matches,
                                 uint64_t total_matches, uint32_t rounds, double inlier_threshold, ochip_ransac_result *results,
                                 uint8_t *inliers)
{
    if (!ctx)
        return OCHIP_EINVAL;
    if (n_jobs == 0)
        return OCHIP_OK;
    if (!jobs || !results || rounds == 0 || (total_matches && (!matches || !inliers)))
        return ochip_fail(ctx, OCHIP_EINVAL, "NULL argument or zero rounds");
    if (!ctx->kp_store_ready)
        return ochip_fail(ctx, OCHIP_ESTATE, "ochip_upload_keypoints has not been called");
    OCHIP_HIP(ctx, hipSetDevice(ctx->device));
    for (uint32_t j = 0; j < n_jobs; j++)
    {
        const ochip_ransac_job &jb = jobs[j];
        if (jb.image_1 >= ctx->n_images || jb.image_2 >= ctx->n_images || !ctx->kp_set[jb.image_1] || !ctx->kp_set[jb.image_2])
            return ochip_fail(ctx, OCHIP_ESTATE, "job %u references an image without keypoints", j);
        if (jb.match_offset + jb.n > total_matches)
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
        OCHIP_HIP(ctx, hipMemcpyAsync(ctx->img_off_dev, ctx->img_off.data(), (size_t)ctx->n_images * 8, hipMemcpyHostToDevice,
                                      ctx->stream));
        OCHIP_HIP(ctx, hipMemcpyAsync(ctx->img_n_dev, ctx->img_n.data(), (size_t)ctx->n_images * 4, hipMemcpyHostToDevice,
                                      ctx->stream));
        ctx->img_tables_dirty = false;
    }
    const uint64_t T = total_matches ? total_matches : 1;
    // scratch slots as in ochip_ransac_homography_batch: 0 jobs, 1 matches, 4 coordinates, 6 LU workspaces, 7 results + flags
    const int slot[5] = {0, 1, 4, 6, 7};
    const size_t sizes[5] = {(size_t)n_jobs * sizeof(ochip_ransac_job), (size_t)T * sizeof(ochip_ransac_match), (size_t)T * 64,
                             (size_t)n_jobs * 81 * 8, (size_t)n_jobs * sizeof(ochip_ransac_result) + T};
    for (int i = 0; i < 5; i++)
    {
        int rc = ochip_ensure(ctx, &ctx->scratch_dev[slot[i]], &ctx->scratch_cap[slot[i]], sizes[i]);
        if (rc)
            return rc;
    }
    ochip_ransac_result *res_dev = (ochip_ransac_result *)ctx->scratch_dev[7];
    uint8_t *inl_dev = (uint8_t *)ctx->scratch_dev[7] + (size_t)n_jobs * sizeof(ochip_ransac_result);
    OCHIP_HIP(ctx, hipMemcpyAsync(ctx->scratch_dev[0], jobs, sizes[0], hipMemcpyHostToDevice, ctx->stream));
    if (total_matches)
    {
        OCHIP_HIP(ctx, hipMemcpyAsync(ctx->scratch_dev[1], matches, (size_t)total_matches * sizeof(ochip_ransac_match),
                                      hipMemcpyHostToDevice, ctx->stream));
        OCHIP_HIP(ctx, hipMemcpyAsync(inl_dev, inliers, (size_t)total_matches, hipMemcpyHostToDevice, ctx->stream));
    }
    rays_view rv{ctx->rays_dev, ctx->img_off_dev};
    hipLaunchKernelGGL(refit_homography_kernel, dim3(n_jobs), dim3(W), 0, ctx->stream, (const ochip_ransac_job *)ctx->scratch_dev[0],
                       (const ochip_ransac_match *)ctx->scratch_dev[1], rv, (double *)ctx->scratch_dev[4],
                       (double *)ctx->scratch_dev[6], (uint64_t)T, inlier_threshold, rounds, res_dev, inl_dev);
    OCHIP_HIP(ctx, hipGetLastError());
    OCHIP_HIP(ctx, hipMemcpyAsync(results, res_dev, (size_t)n_jobs * sizeof(ochip_ransac_result), hipMemcpyDeviceToHost, ctx->stream));
    if (total_matches)
        OCHIP_HIP(ctx, hipMemcpyAsync(inliers, inl_dev, (size_t)total_matches, hipMemcpyDeviceToHost, ctx->stream));
    OCHIP_HIP(ctx, ochip_stream_wait(ctx, ctx->stream));
    return OCHIP_OK;
}


int ochip_ransac_epipolar_batch(ochip_ctx *ctx, int model, const ochip_epipolar_job *jobs, uint32_t n_jobs, const double *corr6,
                                const uint32_t *sorted_idx, uint64_t total, const uint32_t *eval_order, uint64_t eval_total,
                                double inlier_threshold, ochip_ransac_result *results, uint8_t *inliers)
{
    static_assert(sizeof(ochip_epipolar_job) == sizeof(ochip_epipolar_job_dev), "job layout");
    if (!ctx)
        return OCHIP_EINVAL;
    if (n_jobs == 0)
        return OCHIP_OK;
    if (model != 0 && model != 1)
        return ochip_fail(ctx, OCHIP_EINVAL, "model %d: 0 = fundamental matrix, 1 = essential matrix", model);
    if (!jobs || !results || (total && (!corr6 || !sorted_idx || !inliers)) || (eval_total && !eval_order))
        return ochip_fail(ctx, OCHIP_EINVAL, "NULL argument");
    OCHIP_HIP(ctx, hipSetDevice(ctx->device));
    for (uint32_t j = 0; j < n_jobs; j++)
        if (jobs[j].corr_offset + jobs[j].n > total || jobs[j].eval_offset + jobs[j].n > eval_total)
            return ochip_fail(ctx, OCHIP_EINVAL, "job %u: offsets exceed the arrays", j);
    const uint64_t T = total ? total : 1;
    const size_t sizes[8] = {(size_t)n_jobs * sizeof(ochip_epipolar_job),
                             (size_t)T * 48,
                             (size_t)T * 4,
                             (size_t)(eval_total ? eval_total : 1) * 4,
                             (size_t)T * 64,
                             (size_t)T * 5,
                             (size_t)n_jobs * 81 * 8,
                             (size_t)n_jobs * sizeof(ochip_ransac_result) + T};
    for (int i = 0; i < 8; i++)
    {
        const int rc = ochip_ensure(ctx, &ctx->scratch_dev[i], &ctx->scratch_cap[i], sizes[i]);
        if (rc)
            return rc;
    }
    hipStream_t st = ctx->stream;
    OCHIP_HIP(ctx, hipMemcpyAsync(ctx->scratch_dev[0], jobs, sizes[0], hipMemcpyHostToDevice, st));
    if (total)
    {
        OCHIP_HIP(ctx, hipMemcpyAsync(ctx->scratch_dev[1], corr6, (size_t)total * 48, hipMemcpyHostToDevice, st));
        OCHIP_HIP(ctx, hipMemcpyAsync(ctx->scratch_dev[2], sorted_idx, (size_t)total * 4, hipMemcpyHostToDevice, st));
    }
    if (eval_total)
        OCHIP_HIP(ctx, hipMemcpyAsync(ctx->scratch_dev[3], eval_order, (size_t)eval_total * 4, hipMemcpyHostToDevice, st));
    ochip_ransac_result *res_dev = (ochip_ransac_result *)ctx->scratch_dev[7];
    uint8_t *inl_dev = (uint8_t *)ctx->scratch_dev[7] + (size_t)n_jobs * sizeof(ochip_ransac_result);
    hipEvent_t e0, e1;
    ochip_prof_begin(ctx, OCHIP_K_RANSAC, &e0, &e1);
    auto launch = [&](auto kernel) {
        hipLaunchKernelGGL(kernel, dim3(n_jobs), dim3(W), 0, st, (const ochip_epipolar_job_dev *)ctx->scratch_dev[0],
                           (const double *)ctx->scratch_dev[1], (const uint32_t *)ctx->scratch_dev[2], (const uint32_t *)ctx->scratch_dev[3],
                           (double *)ctx->scratch_dev[4], (uint8_t *)ctx->scratch_dev[5], (double *)ctx->scratch_dev[6], (uint64_t)T,
                           inlier_threshold, res_dev, inl_dev);
    };
    if (model == 0)
        launch(ransac_epipolar_kernel<8, false>);
    else
        launch(ransac_epipolar_kernel<5, true>);
    ochip_prof_end(ctx, OCHIP_K_RANSAC, e0, e1);
    OCHIP_HIP(ctx, hipGetLastError());
    OCHIP_HIP(ctx, hipMemcpyAsync(results, res_dev, (size_t)n_jobs * sizeof(ochip_ransac_result), hipMemcpyDeviceToHost, st));
    if (total)
        OCHIP_HIP(ctx, hipMemcpyAsync(inliers, inl_dev, (size_t)total, hipMemcpyDeviceToHost, st));
    OCHIP_HIP(ctx, ochip_stream_wait(ctx, st));
    return OCHIP_OK;
}

} // extern "C"
