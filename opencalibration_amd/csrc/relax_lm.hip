// libochip.so — the shared part of the relax solve (see relax_lm.hpp): Levenberg-Marquardt trust-region loop with
// Ceres' semantics and the linear solve of (J'J + D'D) y = J'r on the device.
//
// The reduced system is stored as the 64 x 64 tiles of its block envelope, lower triangle only (relax_lm.hpp: lm_matrix;
// 6 MB instead of 72 at n = 3003, 50 MB instead of 1.8 GB at the 15 004 unknowns of a 5 000-camera group) and factored in
// place by ONE launch per factorisation: chol_tiles_kernel, a left-looking tile Cholesky whose workgroups hand 64 x 64 tiles to
// each other through flags (products on v_mfma_f64_16x16x4f64, the diagonal block and its inverse in registers).  The
// augmented row carries the forward solve; the backward substitution is one workgroup walking the row envelope.
// The launch chain it replaced (chol_diag_kernel / chol_panel_kernel / chol_update_mfma_kernel, three dependent launches
// per block column) stays as the second opinion of OCHIP_TEST_HOOKS=chol_verify.
#include "relax_lm.hpp"
#include "relax_lm_back.hpp"

#include <algorithm>
#include <cmath>

using namespace ochip;

#include "relax_chol_tile.hpp"

namespace
{

// ---- dense linear algebra on the reduced system -------------------------------------------------
constexpr int NB = ochip::LM_NB;

// Wm = S A S + diag(D) tile by tile (one workgroup per stored tile), row n = gs = S g; the part of a diagonal tile above
// the diagonal and everything beyond the system's last row / column is written as zero
// The damping: D_ii^2 = diagonal[i] / radius as LevenbergMarquardtStrategy forms it (sqrt, then squared by the solver),
// computed here from the device's copy of the clamped diagonal and kept in lm_diag for the model cost change.
__global__ __launch_bounds__(256) void lm_build_kernel(lm_matrix A, const unsigned int *__restrict__ tile_ij, const double *g,
                                                       const double *scale, const double *diagonal, double radius, double *lm_diag,
                                                       lm_matrix W, double *gs, int n, unsigned int *chol_sync, int *fail_chol)
{
    // (the factorisation's claim counter, its per-tile flags and the failure flag start at zero: cleared here, one launch
    // instead of three in front of every factorisation)
    if (threadIdx.x == 0)
    {
        chol_sync[4 + blockIdx.x] = 0u;
        if (blockIdx.x == 0)
        {
            chol_sync[0] = chol_sync[1] = chol_sync[2] = chol_sync[3] = 0u;
            *fail_chol = 0;
        }
    }
    const unsigned int ij = tile_ij[blockIdx.x];
    const int I = (int)(ij & 0xFFFFu), J = (int)(ij >> 16);
    const double *a = A.tiles + ((size_t)blockIdx.x << 12);
    double *w = W.tiles + ((size_t)blockIdx.x << 12);
    for (int e = threadIdx.x; e < NB * NB; e += 256)
    {
        const int i = I * NB + (e >> 6), j = J * NB + (e & 63);
        double v = 0.0;
        if (j < n)
        {
            if (i < n)
            {
                if (j <= i)
                {
                    v = a[e] * scale[i] * scale[j];
                    if (i == j)
                    {
                        const double dd = sqrt(diagonal[i] / radius), lm = dd * dd;
                        lm_diag[i] = lm;
                        v += lm;
                        gs[i] = g[i] * scale[i];
                    }
                }
            }
            else if (i == n)
                v = g[j] * scale[j]; // augmented row n: the factorisation performs the forward solve L y = gs on it
        }
        w[e] = v;
    }
}

// the augmented row (after the factorisation: y = L^-1 gs) as a plain vector
__global__ void lm_aug_row_kernel(lm_matrix W, int n, double *out)
{
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j < n)
        out[j] = W.tiles[lm_at(W, n, j)];
}

__global__ __launch_bounds__(256) void chol_diag_kernel(lm_matrix M, int n, int k0, int nb, int *fail,
                                                        double *Linv /*[NB][NB] row-major, zero padded*/)
{
    __shared__ double colA[4][NB], rowX[4][NB]; // two pivots per step, double-buffered
    const int t = threadIdx.x, ty = t >> 4, tx = t & 15;
    double *A = M.tiles + ((size_t)M.cols[k0 / NB].first_tile << 12); // the diagonal tile
    double a[4][4], x[4][4];
#pragma unroll
    for (int p = 0; p < 4; p++)
#pragma unroll
        for (int q = 0; q < 4; q++)
        {
            const int i = ty + 16 * p, c = tx + 16 * q;
            // only the lower triangle of the input is meaningful; mirror it so that both triangles update alike
            const int lo = i > c ? i : c, hi = i > c ? c : i;
            a[p][q] = (lo < nb) ? A[lo * NB + hi] : (i == c ? 1.0 : 0.0);
            x[p][q] = (i == c) ? 1.0 : 0.0;
        }
    bool bad = false;
    chol_diag_phase<0>(a, x, colA, rowX, ty, tx, nb, bad);
    chol_diag_phase<1>(a, x, colA, rowX, ty, tx, nb, bad);
    chol_diag_phase<2>(a, x, colA, rowX, ty, tx, nb, bad);
    chol_diag_phase<3>(a, x, colA, rowX, ty, tx, nb, bad);
    if (bad)
        *fail = 1;
#pragma unroll
    for (int p = 0; p < 4; p++)
#pragma unroll
        for (int q = 0; q < 4; q++)
        {
            const int i = ty + 16 * p, c = tx + 16 * q;
            if (i < nb && c <= i)
                A[i * NB + c] = a[p][q];
            Linv[i * NB + c] = (i < nb && c <= i) ? x[p][q] : ((i == c) ? 1.0 : 0.0);
        }
}

// The rows below a diagonal block that can be non-zero: the block column's envelope (cameras further down the list
// than any camera linked to this block's cameras never get fill) and the tail (the plane unknowns, coupled to every
// camera, and the augmented row).  Kernels index this set with a logical row number.
struct row_set
{
    int begin, band_rows; // rows begin .. begin + band_rows - 1
    int tail_begin, total; // then rows tail_begin .. ; total = band_rows + tail rows
};
__device__ __forceinline__ int set_row(const row_set &s, int i)
{
    return i < s.band_rows ? s.begin + i : s.tail_begin + (i - s.band_rows);
}

// rows below the diagonal block: X = A[i, k0:k0+nb] * L_kk^{-T} = A_tile * Linv' as a 64x64x64 GEMM on the
// matrix cores (same tiling as the trailing update), in place.
__global__ __launch_bounds__(256) void chol_panel_kernel(lm_matrix M, int n, row_set rs, int k0, int nb, const double *Linv)
{
    constexpr int KC = 32;
    __shared__ double Pi[64][KC + 1], Pj[64][KC + 1];
    const int r0 = blockIdx.x * 64; // logical
    const int t = threadIdx.x, lane = t & 63, w = t >> 6;
    const int wr = (w >> 1) * 32, wc = (w & 1) * 32;
    const int lr = lane & 15, lk = lane >> 4;
    v4f64 acc[2][2];
    for (int i = 0; i < 2; i++)
        for (int j = 0; j < 2; j++)
            acc[i][j] = v4f64{0, 0, 0, 0};
    for (int m0 = 0; m0 < NB; m0 += KC)
    {
        __syncthreads();
        for (int e = t; e < 64 * KC; e += 256)
        {
            const int r = e / KC, m = e % KC;
            Pi[r][m] = (r0 + r < rs.total && m0 + m < nb) ? M.tiles[lm_at(M, set_row(rs, r0 + r), k0 + m0 + m)] : 0.0;
            Pj[r][m] = Linv[r * NB + m0 + m]; // X[i][c] = sum_m A[i][m] Linv[c][m]
        }
        __syncthreads();
#pragma unroll
        for (int kk = 0; kk < KC; kk += 4)
        {
            const double a0 = Pi[wr + lr][kk + lk], a1 = Pi[wr + 16 + lr][kk + lk];
            const double b0 = Pj[wc + lr][kk + lk], b1 = Pj[wc + 16 + lr][kk + lk];
            acc[0][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b0, acc[0][0], 0, 0, 0);
            acc[0][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b1, acc[0][1], 0, 0, 0);
            acc[1][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b0, acc[1][0], 0, 0, 0);
            acc[1][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b1, acc[1][1], 0, 0, 0);
        }
    }
    __syncthreads();
    for (int i = 0; i < 2; i++)
        for (int j = 0; j < 2; j++)
            for (int e = 0; e < 4; e++)
            {
                const int r = r0 + wr + 16 * i + 4 * e + lk, cc = wc + 16 * j + lr;
                if (r < rs.total && cc < nb)
                    M.tiles[lm_at(M, set_row(rs, r), k0 + cc)] = acc[i][j][e];
            }
}

// trailing update, lower tiles only: C[i][j] -= sum_m P[i][m] P[j][m], 64x64 tile per workgroup.
// Same trailing update on the matrix cores: v_mfma_f64_16x16x4_f64, one 32x32 sub-tile per wave
// (2x2 accumulators), operands staged through LDS in 32-deep K chunks.  This dense fp64 update of the
// reduced system is the only MFMA use on the path.
__global__ __launch_bounds__(256) void chol_update_mfma_kernel(lm_matrix M, int n, row_set rs, int k0, int nb)
{
    const int ti = blockIdx.y, tj = blockIdx.x;
    if (tj > ti)
        return;
    constexpr int KC = 32;
    __shared__ double Pi[64][KC + 1], Pj[64][KC + 1];
    const int r0 = ti * 64, c0 = tj * 64; // logical rows of the set; columns are the same set (without the augmented row)
    const int t = threadIdx.x, lane = t & 63, w = t >> 6;
    const int wr = (w >> 1) * 32, wc = (w & 1) * 32;
    const int lr = lane & 15, lk = lane >> 4;
    v4f64 acc[2][2];
    for (int i = 0; i < 2; i++)
        for (int j = 0; j < 2; j++)
            acc[i][j] = v4f64{0, 0, 0, 0};
    for (int m0 = 0; m0 < nb; m0 += KC)
    {
        const int mc = min(KC, nb - m0);
        __syncthreads();
        for (int e = t; e < 64 * KC; e += 256)
        {
            const int r = e / KC, m = e % KC;
            Pi[r][m] = (r0 + r < rs.total && m < mc) ? M.tiles[lm_at(M, set_row(rs, r0 + r), k0 + m0 + m)] : 0.0;
            Pj[r][m] = (c0 + r < rs.total && m < mc) ? M.tiles[lm_at(M, set_row(rs, c0 + r), k0 + m0 + m)] : 0.0;
        }
        __syncthreads();
#pragma unroll
        for (int kk = 0; kk < KC; kk += 4)
        {
            const double a0 = Pi[wr + lr][kk + lk], a1 = Pi[wr + 16 + lr][kk + lk];
            const double b0 = Pj[wc + lr][kk + lk], b1 = Pj[wc + 16 + lr][kk + lk];
            acc[0][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b0, acc[0][0], 0, 0, 0);
            acc[0][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b1, acc[0][1], 0, 0, 0);
            acc[1][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b0, acc[1][0], 0, 0, 0);
            acc[1][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b1, acc[1][1], 0, 0, 0);
        }
    }
    for (int i = 0; i < 2; i++)
        for (int j = 0; j < 2; j++)
            for (int e = 0; e < 4; e++)
            {
                // f64 16x16x4 result layout (measured, scripts/probe_mfma_f64.hip): D[4*reg + lane/16][lane%16]
                const int r = r0 + wr + 16 * i + 4 * e + lk, cc = c0 + wc + 16 * j + lr;
                if (r < rs.total && cc <= r)
                {
                    const int ar = set_row(rs, r), ac = set_row(rs, cc);
                    if (ac < n)
                        M.tiles[lm_at(M, ar, ac)] -= acc[i][j][e];
                }
            }
}

// ---- the whole factorisation in ONE launch ------------------------------------------------------------------------
// The chain above costs three dependent launches per 64-column block (141 at n = 3003): kernel time 55 us per block plus
// the gaps between dependent launches.  Here the factorisation is a set of 64 x 64 tiles inside the same block envelope,
// each computed ONCE by one workgroup (left-looking): tile (I, J) = A(I, J) - sum_K L(I, K) L(J, K)' over the column
// blocks K < J whose envelope holds both, accumulated in the matrix cores' registers while the operands arrive; then the
// diagonal tile is factored (the register kernel of chol_diag_kernel) and an off-diagonal tile is multiplied by the
// inverse of its column's diagonal block.  Nothing is read-modified-written in HBM.  Workgroups claim tiles from a
// list, column by column: in that order every tile follows the tiles it needs, so a claimed tile only ever waits for
// tiles that are already running and no co-residency is required.  The one exception is made by lm_system_resize when
// the dense tail is small: its rows - whose sums run along the whole factorisation and should start early - are claimed
// ahead of the band, which needs free slots for the band's workgroups and is therefore limited to an eighth of the
// device (see there).  A tile is handed over as MI355X_MICROARCH.md prescribes
// (per-XCD L2s are not coherent): write-through stores, every storing wave drained, barrier, ONE lane sets the tile's
// flag with an agent-scope store; the consumer polls that word relaxed, ONE agent-scope acquire, drain, barrier, plain loads.
typedef lm_col chol_col;
__device__ __forceinline__ int chol_tile_index(const chol_col *cols, int I, int J)
{
    return lm_tile_index(cols, I, J);
}
__device__ __forceinline__ void store_through(double *p, double v) // global_store_dwordx2 sc1: write-through, agent scope
{
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

template <int KC>
__global__ __launch_bounds__(256) void chol_tiles_kernel(double *W, int n, const chol_col *__restrict__ cols,
                                                         const int *__restrict__ kmin, const unsigned int *__restrict__ tiles,
                                                         int n_tiles, int tb, unsigned int *sync, double *linv, int *fail,
                                                         unsigned long long *timeline, const int *__restrict__ korder, int blocked_diag)
{
    // KC = 32: a step's operands go through LDS in two halves (71 KB per workgroup); KC = 64: whole (104 KB) - see the launch.
    __shared__ double T[64][65];
    __shared__ double Pi[64][KC + 1], Pj[64][KC + 1];
    __shared__ double colA[4][NB], rowX[4][NB]; // two pivots per step, double-buffered
    __shared__ int s_claim, s_ready;
    unsigned int *flags = sync + 4;
    const int t = threadIdx.x, lane = t & 63, w = t >> 6;
    const int wr = (w >> 1) * 32, wc = (w & 1) * 32;
    const int lr = lane & 15, lk = lane >> 4;
    const int n_rows = n + 1; // the augmented row carries the forward solve
    for (;;)
    {
        __syncthreads();
        if (t == 0)
            s_claim = (int)atomicAdd(&sync[0], 1u);
        __syncthreads();
        const int claim = s_claim;
        if (claim >= n_tiles)
            return;
        const unsigned int ij = tiles[claim];
        const int I = (int)(ij & 0xFFFFu);
        int J = (int)((ij >> 16) & 0x7FFFu);
        // bit 31: tile (J + 1, J) with the diagonal tile (J + 1, J + 1) behind it.  The timeline showed two hand-overs on a
        // column's critical path - diagonal tile -> the tile below it -> the next diagonal tile, 8 - 9 us each (flag,
        // acquire, operands back from memory) beside 19 us of factorisation.  The workgroup that owns the tile below the
        // diagonal therefore goes on to the next diagonal tile itself: it has summed that tile's earlier steps along with
        // its own (they read the same row of tiles), keeps its product in LDS for the last step, and starts to factor
        // without waiting for anybody.
        const bool fuse = (ij >> 31) != 0u;
        if (timeline && t == 0) // OCHIP_CHOL_TIMELINE: claimed / operands summed / factored or multiplied / published
            timeline[6 * claim] = wall_clock64();
        const int r0 = I * 64;
        int c0 = J * 64;
        int nb = min(64, n - c0); // columns of this block column
        const int k_begin = I < tb ? kmin[I] : (J >= tb ? 0 : kmin[J]);
        v4f64 acc[2][2], accd[2][2];
        for (int i = 0; i < 2; i++)
            for (int j = 0; j < 2; j++)
                acc[i][j] = accd[i][j] = v4f64{0, 0, 0, 0};
        // A tile of the tail's own columns sums over every column of the band.  When the band falls into regions (korder
        // given), it takes them in the order the regions' chains produce them - the regions' first columns, their second
        // columns, ... - instead of waiting for the whole first region before it touches the second: its steps then run
        // under the regions' chains and not after them (208 us of 843 at n = 3003).  A fixed order, whoever arrives when.
        // the tile's own entries (written by the launch before this one) do not depend on anything: requested now, they
        // arrive under the waits and sums below instead of costing a memory round trip after them
        double *wt = W + ((size_t)chol_tile_index(cols, I, J) << 12);
        double own[2][2][4];
#pragma unroll
        for (int i = 0; i < 2; i++)
#pragma unroll
            for (int j = 0; j < 2; j++)
#pragma unroll
                for (int e = 0; e < 4; e++)
                {
                    const int r = wr + 16 * i + 4 * e + lk, c = wc + 16 * j + lr;
                    own[i][j][e] = (r0 + r < n_rows && c < nb) ? wt[r * NB + c] : 0.0;
                }
        double *wt_d = fuse ? W + ((size_t)chol_tile_index(cols, I, I) << 12) : wt;
        const int nb_d = min(64, n - r0);
        double own_d[2][2][4];
        if (fuse)
        {
#pragma unroll
            for (int i = 0; i < 2; i++)
#pragma unroll
                for (int j = 0; j < 2; j++)
#pragma unroll
                    for (int e = 0; e < 4; e++)
                    {
                        const int r = wr + 16 * i + 4 * e + lk, c = wc + 16 * j + lr;
                        own_d[i][j][e] = (r0 + r < n_rows && c < nb_d) ? wt_d[r * NB + c] : 0.0;
                    }
        }
        const bool mapped = korder != nullptr && J >= tb;
        int ready = k_begin; // steps whose operands are known to be complete (the same value in every thread)
        for (int step = k_begin; step < J; step++)
        {
            const int K = mapped ? korder[step] : step;
            if (step >= ready)
            {
                if (t == 0)
                {
                    // wait for the operands of this step, take along the following steps that are complete already,
                    // ONE acquire for all of them
                    int upto = step;
                    for (;;)
                    {
                        const int Ku = mapped ? korder[upto] : upto;
                        const unsigned int *fa = flags + chol_tile_index(cols, I, Ku);
                        const unsigned int *fb = flags + chol_tile_index(cols, J, Ku);
                        const bool have = __hip_atomic_load(fa, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0 &&
                                          __hip_atomic_load(fb, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0;
                        if (have)
                        {
                            if (++upto >= J || upto - step >= 16)
                                break;
                        }
                        else if (upto > step)
                            break;
                        else
                            __builtin_amdgcn_s_sleep(2);
                    }
                    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    s_ready = upto;
                }
                __syncthreads();
                ready = s_ready; // (rewritten only after every thread has passed the barriers of the loads below)
            }
            const double *wi = W + ((size_t)chol_tile_index(cols, I, K) << 12);
            const double *wj = W + ((size_t)chol_tile_index(cols, J, K) << 12);
            for (int m0 = 0; m0 < 64; m0 += KC)
            {
                __syncthreads();
                for (int e = t; e < 64 * KC; e += 256)
                {
                    const int r = e / KC, m = e % KC;
                    Pi[r][m] = wi[r * NB + m0 + m]; // (rows beyond the augmented row are zero in every tile)
                    Pj[r][m] = wj[r * NB + m0 + m];
                }
                __syncthreads();
#pragma unroll
                for (int kk = 0; kk < KC; kk += 4)
                {
                    const double a0 = Pi[wr + lr][kk + lk], a1 = Pi[wr + 16 + lr][kk + lk];
                    const double b0 = Pj[wc + lr][kk + lk], b1 = Pj[wc + 16 + lr][kk + lk];
                    acc[0][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b0, acc[0][0], 0, 0, 0);
                    acc[0][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b1, acc[0][1], 0, 0, 0);
                    acc[1][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b0, acc[1][0], 0, 0, 0);
                    acc[1][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b1, acc[1][1], 0, 0, 0);
                    if (fuse) // the same step of the diagonal tile (I, I): L(I, K) L(I, K)'
                    {
                        const double d0 = Pi[wc + lr][kk + lk], d1 = Pi[wc + 16 + lr][kk + lk];
                        accd[0][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, d0, accd[0][0], 0, 0, 0);
                        accd[0][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, d1, accd[0][1], 0, 0, 0);
                        accd[1][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, d0, accd[1][0], 0, 0, 0);
                        accd[1][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, d1, accd[1][1], 0, 0, 0);
                    }
                }
            }
        }
        if (timeline && t == 0)
        {
            timeline[6 * claim + 1] = wall_clock64();
            timeline[6 * claim + 4] = clock64(); // shader clock: cycles the factor / product took
        }
        bool second = false; // fused: the diagonal tile (I, I) after the tile (I, J) below the previous diagonal tile
      finish_tile:
        // T = A(I, J) - acc (the tile's own entries were written by the launch before this one)
        __syncthreads();
#pragma unroll
        for (int i = 0; i < 2; i++)
#pragma unroll
            for (int j = 0; j < 2; j++)
#pragma unroll
                for (int e = 0; e < 4; e++)
                {
                    const int r = wr + 16 * i + 4 * e + lk, c = wc + 16 * j + lr;
                    T[r][c] = second ? own_d[i][j][e] - accd[i][j][e] : own[i][j][e] - acc[i][j][e];
                }
        __syncthreads();
        if (I == J)
        {
            // the 64 x 64 diagonal block and its inverse in registers (chol_diag_kernel's layout and phases)
            const int ty = t >> 4, tx = t & 15;
            double a[4][4], x[4][4];
#pragma unroll
            for (int p = 0; p < 4; p++)
#pragma unroll
                for (int q = 0; q < 4; q++)
                {
                    const int i = ty + 16 * p, c = tx + 16 * q;
                    const int lo = i > c ? i : c, hi = i > c ? c : i;
                    a[p][q] = (lo < nb) ? T[lo][hi] : (i == c ? 1.0 : 0.0);
                    x[p][q] = (i == c) ? 1.0 : 0.0;
                }
            // n not a multiple of 64: the augmented row shares this (last) tile with the diagonal block; it is a panel
            // row, kept aside and multiplied by the inverse below
            const bool has_aug = (r0 + nb == n) && nb < 64;
            if (has_aug && t < 64)
                Pi[t][KC] = t < nb ? T[nb][t] : 0.0; // (column KC: the block updates stage their panels in columns 0 .. 15)
            bool bad = false;
            if (blocked_diag)
            {
                chol_diag_panel_phase<0>(a, x, colA, rowX, ty, tx, nb, bad);
                chol_diag_block_update<0>(a, x, T, Pi, Pj, ty, tx, t);
                chol_diag_panel_phase<1>(a, x, colA, rowX, ty, tx, nb, bad);
                chol_diag_block_update<1>(a, x, T, Pi, Pj, ty, tx, t);
                chol_diag_panel_phase<2>(a, x, colA, rowX, ty, tx, nb, bad);
                chol_diag_block_update<2>(a, x, T, Pi, Pj, ty, tx, t);
                chol_diag_panel_phase<3>(a, x, colA, rowX, ty, tx, nb, bad);
            }
            else
            {
                chol_diag_phase<0>(a, x, colA, rowX, ty, tx, nb, bad);
                chol_diag_phase<1>(a, x, colA, rowX, ty, tx, nb, bad);
                chol_diag_phase<2>(a, x, colA, rowX, ty, tx, nb, bad);
                chol_diag_phase<3>(a, x, colA, rowX, ty, tx, nb, bad);
            }
            if (bad)
                *fail = 1;
            double *Li = linv + (size_t)J * NB * NB;
#pragma unroll
            for (int p = 0; p < 4; p++)
#pragma unroll
                for (int q = 0; q < 4; q++)
                {
                    const int i = ty + 16 * p, c = tx + 16 * q;
                    if (i < nb && c <= i)
                        store_through(&wt[i * NB + c], a[p][q]);
                    store_through(&Li[i * NB + c], (i < nb && c <= i) ? x[p][q] : ((i == c) ? 1.0 : 0.0));
                }
            if (has_aug)
            {
                __syncthreads();
#pragma unroll
                for (int p = 0; p < 4; p++)
#pragma unroll
                    for (int q = 0; q < 4; q++)
                    {
                        const int i = ty + 16 * p, c = tx + 16 * q;
                        T[i][c] = (i < nb && c <= i) ? x[p][q] : 0.0;
                    }
                __syncthreads();
                if (t < nb) // y[c] = sum_m aug[m] Linv[c][m]
                {
                    double sum = 0;
                    for (int m = 0; m <= t; m++)
                        sum += Pi[m][KC] * T[t][m];
                    store_through(&wt[nb * NB + t], sum); // (row n is row nb of this tile)
                }
            }
        }
        else
        {
            // X = T * Linv(J)': wait for the column's diagonal tile, then one more 64 x 64 x 64 product
            if (t == 0)
            {
                const unsigned int *fd = flags + cols[J].first_tile;
                while (__hip_atomic_load(fd, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0)
                    __builtin_amdgcn_s_sleep(2);
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
            __syncthreads();
            const double *Li = linv + (size_t)J * NB * NB;
            for (int i = 0; i < 2; i++)
                for (int j = 0; j < 2; j++)
                    acc[i][j] = v4f64{0, 0, 0, 0};
            for (int m0 = 0; m0 < 64; m0 += KC)
            {
                __syncthreads();
                for (int e = t; e < 64 * KC; e += 256)
                {
                    const int r = e / KC, m = e % KC;
                    Pi[r][m] = T[r][m0 + m];
                    Pj[r][m] = Li[r * NB + m0 + m]; // X[i][c] = sum_m T[i][m] Linv[c][m]
                }
                __syncthreads();
#pragma unroll
                for (int kk = 0; kk < KC; kk += 4)
                {
                    const double a0 = Pi[wr + lr][kk + lk], a1 = Pi[wr + 16 + lr][kk + lk];
                    const double b0 = Pj[wc + lr][kk + lk], b1 = Pj[wc + 16 + lr][kk + lk];
                    acc[0][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b0, acc[0][0], 0, 0, 0);
                    acc[0][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b1, acc[0][1], 0, 0, 0);
                    acc[1][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b0, acc[1][0], 0, 0, 0);
                    acc[1][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b1, acc[1][1], 0, 0, 0);
                }
            }
            for (int i = 0; i < 2; i++)
                for (int j = 0; j < 2; j++)
                    for (int e = 0; e < 4; e++)
                    {
                        const int r = wr + 16 * i + 4 * e + lk, c = wc + 16 * j + lr;
                        if (r0 + r < n_rows && c < nb)
                            store_through(&wt[r * NB + c], acc[i][j][e]);
                    }
            if (fuse)
            {
                // the last step of the diagonal tile out of LDS: X X' with X = L(I, J) as just computed
                __syncthreads();
                for (int i = 0; i < 2; i++)
                    for (int j = 0; j < 2; j++)
                        for (int e = 0; e < 4; e++)
                        {
                            const int r = wr + 16 * i + 4 * e + lk, c = wc + 16 * j + lr;
                            T[r][c] = (r0 + r < n_rows && c < nb) ? acc[i][j][e] : 0.0; // (T's own content went into the product)
                        }
                __syncthreads();
#pragma unroll
                for (int kk = 0; kk < 64; kk += 4)
                {
                    const double a0 = T[wr + lr][kk + lk], a1 = T[wr + 16 + lr][kk + lk];
                    const double d0 = T[wc + lr][kk + lk], d1 = T[wc + 16 + lr][kk + lk];
                    accd[0][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, d0, accd[0][0], 0, 0, 0);
                    accd[0][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, d1, accd[0][1], 0, 0, 0);
                    accd[1][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, d0, accd[1][0], 0, 0, 0);
                    accd[1][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, d1, accd[1][1], 0, 0, 0);
                }
            }
        }
        // publish: every storing wave drained, barrier, one lane sets the flag
        if (timeline && t == 0 && !second)
        {
            timeline[6 * claim + 2] = wall_clock64();
            timeline[6 * claim + 5] = clock64();
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (t == 0)
            __hip_atomic_store(flags + chol_tile_index(cols, I, J), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (timeline && t == 0)
            timeline[6 * claim + 3] = wall_clock64(); // (fused: when the diagonal tile was published)
        if (fuse && !second)
        {
            second = true;
            J = I;
            c0 = J * 64;
            nb = nb_d;
            wt = wt_d;
            goto finish_tile;
        }
    }
}

// Backward substitution L' x = y by ONE workgroup, block by block from the bottom: x_k = L_kk^-T y_k out of the stored
// inverse, then y_i -= sum_m L[k0+m][i] x[k0+m] for the columns i < k0 in which the rows of the block can be non-zero
// (first_blk[k]: the first column block whose envelope reaches row block k; everything for the tail rows).  The 94 launches this took per solve cost 1.0 ms, the
// walk over the envelope takes a tenth of that.
__global__ __launch_bounds__(LM_TG) void back_solve_kernel(lm_matrix Lm, int n, const double *Linv, double *x,
                                                          const int *first_blk, int n_blocks, const double *lm_diag,
                                                          const double *gs, double *scal)
{
    __shared__ double xb[NB];
    __shared__ double sh[LM_TG];
    const int t = threadIdx.x;
    const double *L = Lm.tiles;
    for (int i = t; i < n; i += LM_TG) // y = L^-1 gs: the augmented row
        x[i] = L[lm_at(Lm, n, i)];
    for (int k = n_blocks - 1; k >= 0; k--)
    {
        const int k0 = k * NB, nb = min(NB, n - k0);
        const double *Li = Linv + (size_t)k * NB * NB;
        __syncthreads(); // the updates of the previous block have landed
        if (t < NB)
            xb[t] = t < nb ? x[k0 + t] : 0.0;
        __syncthreads();
        double s = 0; // (L^-T y)[t] = sum_m Linv[m][t] y[m]; Linv is lower triangular with zeros above, so all 64 terms
        if (t < nb)   // can be requested up front (16 loads in flight) instead of one per loop trip
        {
#pragma unroll
            for (int m0 = 0; m0 < NB; m0 += 16)
            {
                double v[16];
#pragma unroll
                for (int j = 0; j < 16; j++)
                    v[j] = Li[(m0 + j) * NB + t];
#pragma unroll
                for (int j = 0; j < 16; j++)
                    if (m0 + j >= t && m0 + j < nb)
                        s += v[j] * xb[m0 + j];
            }
        }
        __syncthreads();
        if (t < nb)
        {
            xb[t] = s;
            x[k0 + t] = s;
        }
        __syncthreads();
        for (int i = first_blk[k] * NB + t; i < k0; i += LM_TG)
        {
            const double *Lc = L + ((size_t)lm_tile_index(Lm.cols, k, i >> 6) << 12) + (i & 63); // column i of tile (k, i / 64)
            double u = 0;
            for (int m0 = 0; m0 < nb; m0 += 16)
            {
                double v[16];
#pragma unroll
                for (int j = 0; j < 16; j++)
                    v[j] = m0 + j < nb ? Lc[(m0 + j) * NB] : 0.0;
#pragma unroll
                for (int j = 0; j < 16; j++)
                    u += v[j] * xb[m0 + j];
            }
            x[i] -= u;
        }
    }
    // model_cost_change (lm_model_change_kernel's sum, same order) on the way out: one launch less per iteration
    __syncthreads();
    double part = 0;
    for (int i = t; i < n; i += LM_TG)
        part += x[i] * gs[i] + lm_diag[i] * x[i] * x[i];
    sh[t] = part;
    __syncthreads();
    for (int s = LM_TG / 2; s > 0; s >>= 1)
    {
        if (t < s)
            sh[t] += sh[t + s];
        __syncthreads();
    }
    if (t == 0)
        scal[1] = 0.5 * sh[0];
}

// The same substitution when the band falls into regions that are coupled through the tail only (lm_envelope::
// region_begin): one workgroup per region.  Each solves the tail's blocks for itself (the same arithmetic in every
// workgroup, on a private copy - nothing is exchanged), takes the tail's contribution out of its own region's columns
// on the way, and then walks its region from the bottom; what were 47 sequential blocks at n = 3003 are 9 + 9.  The
// workgroup that finishes last adds up the regions' parts of model_cost_change.
__global__ __launch_bounds__(LM_TG) void back_solve_regions_kernel(lm_matrix Lm, int n, const double *Linv, double *x, double *work,
                                                                  const int *first_blk, int n_blocks, const int *region, int tb,
                                                                  const double *lm_diag, const double *gs, double *scal,
                                                                  unsigned int *arrived, int x_in_lds)
{
    back_solve_regions_body(Lm, n, Linv, x, work, first_blk, n_blocks, region, tb, lm_diag, gs, scal, arrived, x_in_lds, back_no_tail{});
}


// model_cost_change = -(step.gs + step' As step / 2) with step = -y, (As + D) y = gs
//                   = y.gs - (y.gs - y'D y) / 2 = (y.gs + sum D_i y_i^2) / 2 by the normal equations (no n^2 product).
__global__ __launch_bounds__(LM_TG) void lm_model_change_kernel(const double *lm_diag, const double *gs, const double *y, int n,
                                                                double *scal)
{
    __shared__ double sh[LM_TG];
    const int t = threadIdx.x;
    double part = 0;
    for (int i = t; i < n; i += LM_TG)
        part += y[i] * gs[i] + lm_diag[i] * y[i] * y[i];
    sh[t] = part;
    __syncthreads();
    for (int s = LM_TG / 2; s > 0; s >>= 1)
    {
        if (t < s)
            sh[t] += sh[t + s];
        __syncthreads();
    }
    if (t == 0)
        scal[1] = 0.5 * sh[0];
}

// diag(A) and max|g| -> scal[4] = max|g|; diag_out[i] = A_ii; scale != nullptr: diagonal[i] = clamp(A_ii scale_i^2, 1e-6, 1e32),
// what the iterations after an accepted step damp with (std::min / std::max's comparisons, NaN passes through); the
// results then go to the host block (mail.box != nullptr)
__global__ __launch_bounds__(LM_TG) void lm_diag_kernel(lm_matrix A, const double *g, double *diag_out, int n, double *scal,
                                                       const double *scale, double *diagonal, lm_mail mail)
{
    __shared__ double sh[LM_TG];
    const double gmax = lm_diag_pass<LM_TG>(A, g, diag_out, n, scale, diagonal, sh);
    if (threadIdx.x == 0)
    {
        scal[4] = gmax;
        lm_mail_post(mail);
    }
}


// ---- the projected line search of a bounds-constrained problem (what ceres::TrustRegionMinimizer::DoLineSearch does with
// an Armijo search and cubic interpolation; Ceres [3P] is not in the reference tree, this follows its documented
// behaviour): samples (step, cost, directional derivative), the polynomial through them, its minimiser on an interval.
struct ls_sample
{
    double x = 0, value = 0, slope = 0;
    bool valid = false;
};
double poly_at(const std::vector<double> &p, double x)
{
    double v = 0;
    for (double c : p)
        v = v * x + c;
    return v;
}
// polynomial (highest power first) through the samples' values and slopes: a small dense solve with full pivoting
std::vector<double> poly_through(const std::vector<ls_sample> &samples)
{
    const int m = 2 * (int)samples.size(), deg = m - 1;
    std::vector<double> A((size_t)m * m, 0.0), b(m, 0.0);
    int r = 0;
    for (const ls_sample &s : samples)
    {
        for (int j = 0; j <= deg; j++)
            A[(size_t)r * m + j] = std::pow(s.x, deg - j);
        b[r++] = s.value;
        for (int j = 0; j < deg; j++)
            A[(size_t)r * m + j] = (deg - j) * std::pow(s.x, deg - j - 1);
        b[r++] = s.slope;
    }
    std::vector<int> perm(m);
    for (int i = 0; i < m; i++)
        perm[i] = i;
    for (int k = 0; k < m; k++)
    {
        int pr = k, pc = k;
        double best = -1;
        for (int c = k; c < m; c++)
            for (int rr = k; rr < m; rr++)
                if (std::abs(A[(size_t)rr * m + c]) > best)
                {
                    best = std::abs(A[(size_t)rr * m + c]);
                    pr = rr;
                    pc = c;
                }
        if (!(best > 0))
            break;
        for (int c = 0; c < m; c++)
            std::swap(A[(size_t)k * m + c], A[(size_t)pr * m + c]);
        std::swap(b[k], b[pr]);
        for (int rr = 0; rr < m; rr++)
            std::swap(A[(size_t)rr * m + k], A[(size_t)rr * m + pc]);
        std::swap(perm[k], perm[pc]);
        for (int rr = k + 1; rr < m; rr++)
        {
            const double f = A[(size_t)rr * m + k] / A[(size_t)k * m + k];
            for (int c = k; c < m; c++)
                A[(size_t)rr * m + c] -= f * A[(size_t)k * m + c];
            b[rr] -= f * b[k];
        }
    }
    std::vector<double> y(m, 0.0), out(m, 0.0);
    for (int k = m - 1; k >= 0; k--)
    {
        double v = b[k];
        for (int c = k + 1; c < m; c++)
            v -= A[(size_t)k * m + c] * y[c];
        y[k] = A[(size_t)k * m + k] != 0 ? v / A[(size_t)k * m + k] : 0.0;
    }
    for (int k = 0; k < m; k++)
        out[perm[k]] = y[k];
    return out;
}
// candidates for the minimiser: the real parts of the roots of the derivative (closed form for a quadratic, a Newton
// polish of a dense scan above that: the derivative of the quintic through three samples is a quartic)
void derivative_root_candidates(const std::vector<double> &poly, double lo, double hi, std::vector<double> &out)
{
    std::vector<double> d;
    const int deg = (int)poly.size() - 1;
    for (int i = 0; i < deg; i++)
        d.push_back((deg - i) * poly[i]);
    while (!d.empty() && d.front() == 0.0)
        d.erase(d.begin());
    if (d.size() == 2)
        out.push_back(-d[1] / d[0]);
    else if (d.size() == 3)
    {
        const double a = d[0], b = d[1], c = d[2], D = b * b - 4 * a * c, sq = std::sqrt(std::abs(D));
        if (D >= 0)
        {
            out.push_back(b >= 0 ? (-b - sq) / (2 * a) : (2 * c) / (-b + sq));
            out.push_back(b >= 0 ? (2 * c) / (-b - sq) : (-b + sq) / (2 * a));
        }
        else
            out.push_back(-b / (2 * a));
    }
    else if (d.size() > 3)
    {
        const int N = 4096;
        double prev = poly_at(d, lo);
        for (int i = 1; i <= N; i++)
        {
            const double x = lo + (hi - lo) * i / N, v = poly_at(d, x);
            if ((prev <= 0 && v >= 0) || (prev >= 0 && v <= 0))
            {
                double a = lo + (hi - lo) * (i - 1) / N, b2 = x;
                for (int it = 0; it < 80; it++)
                {
                    const double mid = 0.5 * (a + b2), vm = poly_at(d, mid);
                    if ((poly_at(d, a) <= 0) == (vm <= 0))
                        a = mid;
                    else
                        b2 = mid;
                }
                out.push_back(0.5 * (a + b2));
            }
            prev = v;
        }
    }
}
double interpolated_step(const ls_sample &at0, const ls_sample &previous, const ls_sample &current, double lo, double hi)
{
    if (!current.valid)
        return std::min(std::max(current.x * 0.5, lo), hi);
    std::vector<ls_sample> samples{at0, current};
    if (previous.valid)
        samples.push_back(previous);
    const std::vector<double> poly = poly_through(samples);
    double best_x = 0.5 * (lo + hi), best = poly_at(poly, best_x);
    std::vector<double> cand{lo, hi};
    derivative_root_candidates(poly, lo, hi, cand);
    for (double x : cand)
        if (x >= lo && x <= hi && poly_at(poly, x) < best)
        {
            best = poly_at(poly, x);
            best_x = x;
        }
    return best_x;
}

} // namespace

namespace ochip
{

lm_system::~lm_system()
{
    if (box && ctx)
        ochip_host_free(ctx, box);
}

int lm_system_resize(lm_system *s, int n_in, const lm_envelope &env)
{
    ochip_ctx *ctx = s->ctx;
    s->env = env;
    s->n = n_in;
    s->A_clean = false; // (new layout, possibly new blocks)
    {
        const size_t need = (size_t)lm_system::BOX_VECTORS + 2 * (size_t)std::max(n_in, 1);
        if (need > s->box_cap)
        {
            if (s->box)
                ochip_host_free(ctx, s->box);
            s->box = nullptr;
            s->box_cap = 0;
            void *b = nullptr;
            if (ochip_host_alloc(ctx, need * sizeof(double), &b) != OCHIP_OK || !b)
                return ochip_fail(ctx, OCHIP_ENOMEM, "page-locked host allocation for the solve's read-backs failed (%zu doubles)", need);
            s->box = static_cast<double *>(b);
            s->box_cap = need;
            for (size_t i = 0; i < need; i++)
                s->box[i] = 0.0;
        }
    }
    if (lm_dev_upload(ctx, s->allocs, &s->first_col_dev, s->env.first_col.data(), s->env.first_col.size()) != OCHIP_OK)
        return ochip_fail(ctx, OCHIP_ENOMEM, "device allocation failed (envelope)");
    const size_t n = (size_t)std::max(n_in, 1);
    if (n > s->cap_n)
    {
        // (blocks of a smaller earlier size stay with the owner until it is destroyed)
        if (lm_dev_upload<double>(ctx, s->allocs, &s->g, nullptr, n) != OCHIP_OK ||
            lm_dev_upload<double>(ctx, s->allocs, &s->gs, nullptr, n) != OCHIP_OK ||
            lm_dev_upload<double>(ctx, s->allocs, &s->scale, nullptr, n) != OCHIP_OK ||
            lm_dev_upload<double>(ctx, s->allocs, &s->lm_diag, nullptr, n) != OCHIP_OK ||
            lm_dev_upload<double>(ctx, s->allocs, &s->diag_tmp, nullptr, n) != OCHIP_OK ||
            lm_dev_upload<double>(ctx, s->allocs, &s->diagonal, nullptr, n) != OCHIP_OK ||
            lm_dev_upload<double>(ctx, s->allocs, &s->y, nullptr, n) != OCHIP_OK)
            return ochip_fail(ctx, OCHIP_ENOMEM, "device allocation for the vectors of %zu unknowns failed", n);
        s->cap_n = n;
    }
    if (s->speculative && n > s->cap_n2)
    {
        if (lm_dev_upload<double>(ctx, s->allocs, &s->g2, nullptr, n) != OCHIP_OK ||
            lm_dev_upload<double>(ctx, s->allocs, &s->diagonal2, nullptr, n) != OCHIP_OK)
            return ochip_fail(ctx, OCHIP_ENOMEM, "device allocation for the second set of vectors of %zu unknowns failed", n);
        s->cap_n2 = n;
    }
    s->A2_clean = false;
    if (!s->scal && lm_dev_upload<double>(ctx, s->allocs, &s->scal, nullptr, 8) != OCHIP_OK)
        return OCHIP_ENOMEM;
    if (!s->fail_chol && lm_dev_upload<int>(ctx, s->allocs, &s->fail_chol, nullptr, 1) != OCHIP_OK)
        return OCHIP_ENOMEM;

    // ---- plan of the one-launch factorisation: the block envelope as tiles, their claim order
    {
        const int nn = n_in;
        const int nbc = (nn + NB - 1) / NB, nbr = (nn + 1 + NB - 1) / NB, tb = std::min(s->env.tail_begin, nn) / NB;
        constexpr bool dense = false; // (true: ignore the envelope, every tile stored - the round-2 storage)
        std::vector<chol_col> cols((size_t)std::max(nbc, 1));
        std::vector<int> kmin((size_t)std::max(nbr, 1), 0);
        int n_tiles = 0;
        for (int J = 0; J < nbc; J++)
        {
            int bend = std::max(J + 1, (std::min(dense ? nn : s->env.env_end[J], s->env.tail_begin) + NB - 1) / NB);
            if (J > 0)
                bend = std::max(bend, cols[J - 1].bend); // fill stays inside a monotone envelope
            bend = std::min(bend, nbr);
            cols[J].bend = bend;
            cols[J].tail_start = std::max(std::max(tb, bend), J + 1);
            cols[J].first_tile = n_tiles;
            cols[J].pad = 0;
            n_tiles += (bend - J) + std::max(0, nbr - cols[J].tail_start);
        }
        for (int b = 0; b < std::min(tb, nbr); b++)
        {
            int K = 0;
            while (K < nbc && cols[K].bend <= b)
                K++;
            kmin[b] = std::min(K, b);
        }
        int per_cu = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, chol_tiles_kernel<32>, 256, 0) != hipSuccess || per_cu < 1)
            per_cu = 1;
        const int slots = per_cu * ctx->prop.multiProcessorCount;
        // claim order: the tail's rows first (their sums run along the whole factorisation), then the band, each column
        // by column.  A tail tile claimed ahead of the band tiles it reads spins until workgroups further down the list
        // have produced them, i.e. it needs free slots on the device - and factorisations of other contexts run beside
        // this one (RelaxStage: up to four groups at once, the bench's pipelined relax a fifth), each with its own
        // spinning tail.  So the tail only goes first while it is a small part of the machine (an eighth of the slots:
        // five such kernels still leave three eighths to the band tiles); otherwise plain column order, in which a
        // claimed tile only ever waits for tiles claimed before it - already running - and no co-residency is needed.
        std::vector<unsigned int> order;
        order.reserve((size_t)n_tiles);
        // Fused pairs (chol_tiles_kernel): the tile (J + 1, J) is followed, in the same workgroup, by the diagonal tile
        // (J + 1, J + 1), which then has no entry of its own - where both sum over the same range of columns: inside the
        // band, and among the tail's own columns (the first tail block's diagonal tile sums over the whole band, the tile
        // beside it does not).  Bit 31 marks the pair.
        std::vector<char> fused_diag((size_t)std::max(nbc, 1) + 1, 0);
        for (int J = 0; J + 1 < nbc; J++)
        {
            const int I = J + 1;
            const bool stored = I < cols[J].bend || I >= cols[J].tail_start;
            if (stored && (I < tb || J >= tb))
                fused_diag[(size_t)I] = 1;
        }
        auto push_tile = [&](int I, int J) {
            if (I == J && fused_diag[(size_t)I])
                return;
            const unsigned int pair = (I == J + 1 && fused_diag[(size_t)I]) ? 0x80000000u : 0u;
            order.push_back((unsigned int)I | ((unsigned int)J << 16) | pair);
        };
        auto rows_of = [&](int J, bool want_tail, bool want_band) {
            for (int I = J; I < cols[J].bend; I++)
                if ((I >= tb) ? want_tail : want_band)
                    push_tile(I, J);
            for (int I = cols[J].tail_start; I < nbr; I++)
                if (want_tail)
                    push_tile(I, J);
        };
        size_t tail_tiles = 0;
        for (int J = 0; J < nbc; J++)
        {
            for (int I = J; I < cols[J].bend; I++)
                tail_tiles += I >= tb;
            tail_tiles += (size_t)std::max(0, nbr - cols[J].tail_start);
        }
        // regions of the band (a dissected camera graph): region r's columns only read columns of region r, so the chains
        // of the regions' diagonal tiles run side by side - provided their workgroups are resident together.  Claim order:
        // the band's tiles first (the critical path: every region's first column, then every region's second, ...), then
        // the tail rows' tiles in the same interleaved column order, then the tail's own columns.  Still every tile follows
        // the tiles it needs.
        std::vector<int> region_bounds; // [n_regions + 1], block indices; valid regions only
        std::vector<int> korder;
        {
            const std::vector<int> &rb = s->env.region_begin;
            const int limit = std::min(tb, nbc); // (a last region that lies inside the tail's first block is part of the tail)
            if (!dense && rb.size() > 1 && rb[0] == 0)
            {
                for (size_t r = 0; r < rb.size() && rb[r] < limit && (r == 0 || rb[r] > rb[r - 1]); r++)
                    region_bounds.push_back(rb[r]);
                region_bounds.push_back(limit);
                if (region_bounds.size() < 3)
                    region_bounds.clear();
            }
        }
        s->n_regions = region_bounds.empty() ? 1 : (int)region_bounds.size() - 1;
        if (s->n_regions > 1)
        {
            std::vector<int> band_cols;
            int longest = 0;
            for (int r = 0; r < s->n_regions; r++)
                longest = std::max(longest, region_bounds[r + 1] - region_bounds[r]);
            for (int d = 0; d < longest; d++)
                for (int r = 0; r < s->n_regions; r++)
                    if (region_bounds[r] + d < region_bounds[r + 1])
                        band_cols.push_back(region_bounds[r] + d);
            for (int J : band_cols)
                rows_of(J, false, true);
            for (int J : band_cols)
                rows_of(J, true, false);
            for (int J = region_bounds.back(); J < nbc; J++)
                rows_of(J, true, true);
            korder = band_cols; // the order in which the tail's own tiles sum over the band's columns (chol_tiles_kernel)
            for (int J = region_bounds.back(); J < nbc; J++)
                korder.push_back(J);
        }
        else if ((long)tail_tiles * 8 <= (long)slots)
        {
            for (int J = 0; J < nbc; J++)
                rows_of(J, true, false);
            for (int J = 0; J < nbc; J++)
                rows_of(J, false, true);
        }
        else
            for (int J = 0; J < nbc; J++)
                rows_of(J, true, true);
        int n_fused = 0;
        for (int I = 0; I < nbc; I++)
            n_fused += fused_diag[(size_t)I];
        if (nbr >= 32768 || (int)order.size() + n_fused != n_tiles)
            return ochip_fail(ctx, OCHIP_EINVAL, "relax: factorisation plan is inconsistent (%d tiles, %zu listed, %d fused)", n_tiles,
                              order.size(), n_fused);
        s->chol_n_tiles = n_tiles;
        s->chol_n_claims = (int)order.size();
        s->chol_nbc = nbc;
        s->chol_nbr = nbr;
        s->chol_tb = tb;
        s->chol_grid = std::max(1, std::min((int)order.size(), slots));
        s->chol_sync_bytes = (((size_t)n_tiles + 4) * 4 + 15) / 16 * 16;
        // tiles in storage order, and the matrices themselves
        std::vector<unsigned int> stored((size_t)std::max(n_tiles, 1), 0u);
        for (int J = 0; J < nbc; J++)
        {
            int t = cols[J].first_tile;
            for (int I = J; I < cols[J].bend; I++)
                stored[t++] = (unsigned int)I | ((unsigned int)J << 16);
            for (int I = cols[J].tail_start; I < nbr; I++)
                stored[t++] = (unsigned int)I | ((unsigned int)J << 16);
        }
        if ((size_t)n_tiles > s->cap_tiles)
        {
            const size_t doubles = (size_t)std::max(n_tiles, 1) * NB * NB;
            if (lm_dev_upload<double>(ctx, s->allocs, &s->A, nullptr, doubles) != OCHIP_OK ||
                lm_dev_upload<double>(ctx, s->allocs, &s->Wm, nullptr, doubles) != OCHIP_OK)
                return ochip_fail(ctx, OCHIP_ENOMEM, "device allocation for the reduced system failed (%d tiles of 32 KB, twice)", n_tiles);
            s->cap_tiles = (size_t)n_tiles;
        }
        if (s->speculative && (size_t)n_tiles > s->cap_tiles2)
        {
            if (lm_dev_upload<double>(ctx, s->allocs, &s->A2, nullptr, (size_t)std::max(n_tiles, 1) * NB * NB) != OCHIP_OK)
                return ochip_fail(ctx, OCHIP_ENOMEM, "device allocation for the candidate's reduced system failed (%d tiles of 32 KB)", n_tiles);
            s->cap_tiles2 = (size_t)n_tiles;
        }
        s->cols_host = cols;
        if ((uint64_t)nn >= ctx->relax_system_unknowns) // what the bench line reports as the relax's system memory
        {
            ctx->relax_system_unknowns = (uint64_t)nn;
            ctx->relax_system_bytes = 2ull * (uint64_t)n_tiles * NB * NB * 8;            // J'J and the factor
            ctx->relax_system_dense_bytes = ((uint64_t)nn * nn + (uint64_t)(nn + 1) * nn) * 8; // what rounds 1-2 allocated
        }
        chol_col *cols_dev = nullptr;
        if (lm_dev_upload(ctx, s->allocs, &s->tile_ij, stored.data(), stored.size()) != OCHIP_OK ||
            lm_dev_upload(ctx, s->allocs, &cols_dev, cols.data(), cols.size()) != OCHIP_OK ||
            lm_dev_upload(ctx, s->allocs, &s->chol_kmin, kmin.data(), kmin.size()) != OCHIP_OK ||
            lm_dev_upload(ctx, s->allocs, &s->chol_tiles, order.data(), std::max<size_t>(order.size(), 1)) != OCHIP_OK ||
            lm_dev_upload<unsigned int>(ctx, s->allocs, &s->chol_sync, nullptr, s->chol_sync_bytes / 4) != OCHIP_OK)
            return ochip_fail(ctx, OCHIP_ENOMEM, "device allocation failed (factorisation plan)");
        s->chol_cols = cols_dev;
        s->chol_korder = nullptr;
        if (s->n_regions > 1)
        {
            if (lm_dev_upload(ctx, s->allocs, &s->chol_korder, korder.data(), korder.size()) != OCHIP_OK)
                return ochip_fail(ctx, OCHIP_ENOMEM, "device allocation failed (factorisation plan)");
        }
        else
            region_bounds = {0, std::min(tb, nbc)}; // one band: the backward substitution's one "region"
        {
            const size_t need = (size_t)s->n_regions * ((size_t)std::max(nn, 1) + 1);
            if (lm_dev_upload(ctx, s->allocs, &s->region_dev, region_bounds.data(), region_bounds.size()) != OCHIP_OK)
                return ochip_fail(ctx, OCHIP_ENOMEM, "device allocation failed (regions of the factorisation)");
            if (need > s->back_work_cap)
            {
                if (lm_dev_upload<double>(ctx, s->allocs, &s->back_work, nullptr, need) != OCHIP_OK)
                    return ochip_fail(ctx, OCHIP_ENOMEM, "device allocation failed (backward substitution, %d regions)", s->n_regions);
                s->back_work_cap = need;
            }
        }
    }
    return OCHIP_OK;
}

int lm_download_dense(const lm_system &s, double *out)
{
    const int n = s.n;
    if (n <= 0)
        return OCHIP_OK;
    std::vector<double> tiles((size_t)s.chol_n_tiles * NB * NB);
    if (hipMemcpy(tiles.data(), s.A, tiles.size() * 8, hipMemcpyDeviceToHost) != hipSuccess)
        return ochip_fail(s.ctx, OCHIP_EHIP, "hipMemcpy failed (normal matrix)");
    std::fill(out, out + (size_t)n * n, 0.0);
    for (int J = 0; J < s.chol_nbc; J++)
    {
        const lm_col &c = s.cols_host[(size_t)J];
        int t = c.first_tile;
        auto put = [&](int I) {
            const double *a = tiles.data() + ((size_t)t << 12);
            for (int r = 0; r < NB; r++)
                for (int q = 0; q < NB; q++)
                {
                    const int i = I * NB + r, j = J * NB + q;
                    if (i < n && j <= i)
                        out[(size_t)i * n + j] = out[(size_t)j * n + i] = a[r * NB + q];
                }
            t++;
        };
        for (int I = J; I < c.bend; I++)
            put(I);
        for (int I = c.tail_start; I < s.chol_nbr; I++)
            put(I);
    }
    return OCHIP_OK;
}

void lm_launch_diag(lm_system &S, const double *scale, const int32_t *fail_ranks, int world, int with_cost, int32_t *clear_after)
{
    const lm_mail mail{S.box, S.scal, S.fail_chol, fail_ranks, world, with_cost, clear_after};
    hipLaunchKernelGGL(lm_diag_kernel, dim3(1), dim3(LM_TG), 0, S.ctx->stream, S.matA(), (const double *)S.g, S.diag_tmp, S.n, S.scal, scale,
                       S.diagonal, mail);
}

// Trust-region Levenberg-Marquardt, monotonic steps (Ceres TrustRegionMinimizer + LevenbergMarquardtStrategy semantics,
// SURVEY.md Appendix B).  The control flow runs on the host side of the library; every O(problem) operation is a kernel.
int lm_solve(lm_system &S, lm_model &M, const ochip_relax_options *opt, ochip_relax_summary *sum)
{
    ochip_ctx *ctx = S.ctx;
    hipStream_t st = ctx->stream;
    const int n = S.n;
    double *const h = S.box + lm_system::BOX_SCAL; // (page-locked: relax_lm.hpp, lm_system::box)
    const bool eliminated = M.has_eliminated();
    M.begin_solve();
    // the solver's own mail: scal[0, 8) (and the factorisation's flag) into the host block by the kernel that ends a phase
    const lm_mail mail{S.box, S.scal, S.fail_chol, nullptr, 0, 0, nullptr};
    auto grad_and_diag = [&](double *gmax, bool refresh) -> int {
        hipLaunchKernelGGL(lm_diag_kernel, dim3(1), dim3(LM_TG), 0, st, S.matA(), (const double *)S.g, S.diag_tmp, n, S.scal,
                           refresh ? (const double *)S.scale : (const double *)nullptr, S.diagonal, mail);
        OCHIP_HIP(ctx, hipGetLastError());
        OCHIP_HIP(ctx, ochip_stream_wait(ctx, st));
        *gmax = h[4];
        if (eliminated)
        {
            double extra = 0;
            const int erc = M.gradient_max_extra(&extra);
            if (erc)
                return erc;
            *gmax = std::max(*gmax, extra);
        }
        return OCHIP_OK;
    };
    auto finish_state = [&]() {
        M.launch_normalize();
        return ochip_stream_wait(ctx, st);
    };

    std::vector<double> scale(n, 1.0);
    double *const diag = S.box + lm_system::BOX_VECTORS;
    double x_cost = 0, gmax = 0;
    int erc = M.evaluate(true, 0, &x_cost);
    if (erc < 0)
        return erc;
    if (erc != 0)
    {
        sum->termination = OCHIP_RELAX_FAILURE;
        OCHIP_HIP(ctx, finish_state());
        return OCHIP_OK;
    }
    int rc = grad_and_diag(&gmax, false);
    if (rc)
        return rc;
    OCHIP_HIP(ctx, hipMemcpy(diag, S.diag_tmp, (size_t)n * 8, hipMemcpyDeviceToHost));
    for (int i = 0; i < n; i++)
        scale[i] = 1.0 / (1.0 + std::sqrt(diag[i])); // jacobi scaling, fixed from the first Jacobian
    OCHIP_HIP(ctx, hipMemcpy(S.scale, scale.data(), (size_t)n * 8, hipMemcpyHostToDevice));
    rc = grad_and_diag(&gmax, true); // (the damping diagonal of the first iterations, now that the scaling exists)
    if (rc)
        return rc;
    double x_norm = 0;
    rc = M.x_norm(&x_norm);
    if (rc)
        return rc;
    sum->initial_cost = x_cost;
    sum->iterations = 1; // iteration 0
    double radius = opt->initial_trust_region_radius, decrease_factor = 2.0;
    int invalid = 0, iter = 0;
    auto finish = [&](int term) -> int {
        sum->termination = term;
        sum->final_cost = x_cost;
        OCHIP_HIP(ctx, finish_state());
        return OCHIP_OK;
    };
    if (gmax <= opt->gradient_tolerance)
        return finish(OCHIP_RELAX_CONVERGENCE_GRADIENT);

    while (true)
    {
        if (iter >= opt->max_num_iterations)
            return finish(OCHIP_RELAX_NO_CONVERGENCE);
        if (radius <= 1e-32)
            return finish(OCHIP_RELAX_CONVERGENCE_RADIUS);
        iter++;
        sum->iterations++;
        // (the damping D^2 = clamp(diag(J'J) scale^2) / radius: the clamped diagonal is refreshed on the device with every
        // accepted point's Jacobian - lm_diag_kernel - and divided by this iteration's radius inside lm_build_kernel)
        hipEvent_t e0, e1;
        ochip_prof_begin(ctx, OCHIP_K_RELAX_SOLVE, &e0, &e1);
        if (n > 0)
            hipLaunchKernelGGL(lm_build_kernel, dim3((unsigned)S.chol_n_tiles), dim3(256), 0, st, S.matA(), (const unsigned int *)S.tile_ij,
                               (const double *)S.g, (const double *)S.scale, (const double *)S.diagonal, radius, S.lm_diag, S.matW(), S.gs,
                               n, S.chol_sync, S.fail_chol);
        else
            OCHIP_HIP(ctx, hipMemsetAsync(S.fail_chol, 0, 4, st));
        if (eliminated)
            M.launch_schur(radius, S.scale, S.matW(), n, S.fail_chol);
        {
            const size_t need = (size_t)((n + NB - 1) / NB) * NB * NB;
            if (need > S.linv_cap)
            {
                S.linv = nullptr;
                S.linv_cap = 0;
                if (lm_dev_upload<double>(ctx, S.allocs, &S.linv, nullptr, need) != OCHIP_OK)
                    return ochip_fail(ctx, OCHIP_ENOMEM, "device allocation for the diagonal-block inverses failed");
                S.linv_cap = need;
            }
        }
        constexpr bool chain = false; // (true: the launch chain per block column, the round-2 schedule; chol_verify runs it beside the tiles)
        static const bool verify = ochip_test_hook("chol_verify"); // run both on the same system and compare
        // the launch chain on (W, linv); count: add the algorithmic flops of the factorisation to the context's counter
        auto launch_chain = [&](double *Wt, double *linv, bool launch, bool count) {
            const lm_matrix W{Wt, S.chol_cols};
            for (int k0 = 0; k0 < n; k0 += NB)
            {
                const int nb = std::min(NB, n - k0);
                double *linv_k = linv + (size_t)(k0 / NB) * NB * NB;
                if (launch)
                    hipLaunchKernelGGL(chol_diag_kernel, dim3(1), dim3(256), 0, st, W, n, k0, nb, S.fail_chol, linv_k);
                // rows below the block that can be non-zero: its envelope as the plan stores it (monotone, whole row blocks),
                // then the tail (dense unknowns + augmented row)
                const int below = k0 + nb;
                const int tail_rows_begin = std::min(S.env.tail_begin, n);
                const int band_end = std::max(below, std::min(S.cols_host[(size_t)(k0 / NB)].bend * NB, tail_rows_begin));
                const int tail0 = std::max(tail_rows_begin, below);
                row_set rs{below, band_end - below, tail0, (band_end - below) + (n + 1 - tail0)};
                const int tiles = (rs.total + 63) / 64;
                if (count)
                    ctx->relax_mfma_flops += 2.0 * rs.total * nb * nb + (below < n ? 1.0 * rs.total * rs.total * nb : 0.0);
                if (!launch)
                    continue;
                hipLaunchKernelGGL(chol_panel_kernel, dim3(tiles), dim3(256), 0, st, W, n, rs, k0, nb, linv_k);
                if (below < n)
                    hipLaunchKernelGGL(chol_update_mfma_kernel, dim3(tiles, tiles), dim3(256), 0, st, W, n, rs, k0, nb);
            }
        };
        if (n == 0)
            ; // (every unknown is eliminated: nothing to factor)
        else if (chain)
            launch_chain(S.Wm, S.linv, true, true);
        else
        {
            double *Wv = nullptr, *linv_v = nullptr;
            size_t got_w = 0, got_l = 0;
            if (verify)
            {
                Wv = (double *)ochip_pool_get(ctx, S.matrix_bytes(), &got_w);
                linv_v = (double *)ochip_pool_get(ctx, (size_t)((n + NB - 1) / NB) * NB * NB * 8, &got_l);
                if (!Wv || !linv_v)
                    return ochip_fail(ctx, OCHIP_ENOMEM, "chol_verify: device allocation failed");
                OCHIP_HIP(ctx, hipMemcpyAsync(Wv, S.Wm, S.matrix_bytes(), hipMemcpyDeviceToDevice, st));
            }
            // operands through LDS in halves (71 KB per workgroup; whole, 104 KB, measured equal alone and slower beside the
            // extraction's kernels, whose workgroups leave 71 KB free on a compute unit sooner); blocked diagonal tiles
            hipLaunchKernelGGL(chol_tiles_kernel<32>, dim3((unsigned)S.chol_grid), dim3(256), 0, st, S.Wm, n, (const chol_col *)S.chol_cols,
                               (const int *)S.chol_kmin, (const unsigned int *)S.chol_tiles, S.chol_n_claims, S.chol_tb, S.chol_sync,
                               S.linv, S.fail_chol, (unsigned long long *)nullptr, (const int *)S.chol_korder, 1);
            launch_chain(S.Wm, S.linv, false, true);
            if (verify)
            {
                launch_chain(Wv, linv_v, true, false);
                std::vector<double> ya(n), yb(n);
                hipLaunchKernelGGL(lm_aug_row_kernel, dim3((n + 255) / 256), dim3(256), 0, st, S.matW(), n, S.y);
                OCHIP_HIP(ctx, hipMemcpyAsync(ya.data(), S.y, (size_t)n * 8, hipMemcpyDeviceToHost, st));
                hipLaunchKernelGGL(lm_aug_row_kernel, dim3((n + 255) / 256), dim3(256), 0, st, lm_matrix{Wv, S.chol_cols}, n, S.y);
                OCHIP_HIP(ctx, hipMemcpyAsync(yb.data(), S.y, (size_t)n * 8, hipMemcpyDeviceToHost, st));
                OCHIP_HIP(ctx, ochip_stream_wait(ctx, st));
                double worst = 0, scale_y = 0;
                bool nan = false;
                for (int i = 0; i < n; i++)
                {
                    scale_y = std::max(scale_y, std::abs(yb[i]));
                    worst = std::max(worst, std::abs(ya[i] - yb[i]));
                    nan = nan || (std::isnan(ya[i]) != std::isnan(yb[i]));
                }
                ochip_pool_put(ctx, Wv, got_w);
                ochip_pool_put(ctx, linv_v, got_l);
                const bool loud = ochip_verbose("relax");
                if (loud)
                    fprintf(stderr, "[ochip relax] factorisation check: n=%d forward solve differs by %.3g (scale %.3g)\n", n, worst, scale_y);
                // (1e-7: the two factorisations round differently - blocked against rank-1 updates, the second pivot of a step
                // from p1 p0 - and a nearly singular system amplifies that: 1.7e-9 on the 12 unknowns of a point triangulation
                // with a trust region of 1e16; a misplaced tile shows up as O(1))
                if (nan || worst > 1e-7 * std::max(scale_y, 1e-300))
                    return ochip_fail(ctx, OCHIP_EINVAL, "chol_verify: the tile factorisation and the launch chain disagree (n = %d, "
                                      "forward solve differs by %g at scale %g)", n, worst, scale_y);
            }
        }
        // row n now holds y = L^-1 gs; back-substitute L' x = y block by block
        bool candidate_launched = false;
        if (n > 0)
        {
            // (the per-region kernel also serves the single band, as one region: it is the faster walk - rows of a block
            // split over the wavefronts, x in LDS; OCHIP_BACK_SOLVE_SINGLE=1: the round-2 kernel)
            constexpr bool single = false; // (true: the round-2 single-workgroup kernel on a single band)
            static const bool x_global = ochip_test_hook("back_solve_x_global"); // test knob: x in HBM even when it fits LDS
            if (S.n_regions > 1 || (!single && S.region_dev))
            {
                const lm_model::back_args ba{S.matW(), n, S.linv, S.y, S.back_work, S.chol_kmin, (n + NB - 1) / NB, S.region_dev, S.chol_tb,
                                             S.lm_diag, S.gs, S.scal, S.chol_sync + 1, x_global ? 0 : 1, S.n_regions};
                if (M.speculates() && !M.is_constrained() && !eliminated && M.launch_back_solve_candidate(ba, S.scale))
                    candidate_launched = true;
                else
                    hipLaunchKernelGGL(back_solve_regions_kernel, dim3((unsigned)S.n_regions), dim3(LM_TG), 0, st, S.matW(), n,
                                       (const double *)S.linv, S.y, S.back_work, (const int *)S.chol_kmin, (n + NB - 1) / NB,
                                       (const int *)S.region_dev, S.chol_tb, (const double *)S.lm_diag, (const double *)S.gs, S.scal,
                                       S.chol_sync + 1, x_global ? 0 : 1);
            }
            else
                hipLaunchKernelGGL(back_solve_kernel, dim3(1), dim3(LM_TG), 0, st, S.matW(), n, (const double *)S.linv, S.y,
                                   (const int *)S.chol_kmin, (n + NB - 1) / NB, (const double *)S.lm_diag, (const double *)S.gs, S.scal);
        }
        else
            hipLaunchKernelGGL(lm_model_change_kernel, dim3(1), dim3(LM_TG), 0, st, S.lm_diag, S.gs, S.y, n, S.scal);
        if (!candidate_launched)
            M.launch_candidate(S.y, S.scale, 1.0, S.scal);
        ochip_prof_end(ctx, OCHIP_K_RELAX_SOLVE, e0, e1);
        OCHIP_HIP(ctx, hipGetLastError());
        // The candidate is evaluated right away (cost only): the step's own results - model cost change, step norms, the
        // factorisation's failure flag - come back with that evaluation's wait instead of a host round trip of their own.
        // An invalid step (rare) has then cost one evaluation whose result is ignored.
        constexpr bool separate_waits = false; // (true: a wait per read-back, the round-2 schedule)
        volatile int &cfail = *reinterpret_cast<volatile int *>(S.box + lm_system::BOX_CFAIL);
        cfail = 0;
        double cand_eval = 0;
        const bool mailed = M.mails_results();
        const bool spec = M.speculates() && !M.is_constrained() && !eliminated && S.A2 != nullptr;
        int fail_mask = 0;
        auto read_step = [&]() {
            if (mailed)
                return; // (the evaluation's last kernel posts them)
            (void)hipMemcpyAsync(h, S.scal, 64, hipMemcpyDeviceToHost, st);
            (void)hipMemcpyAsync(S.box + lm_system::BOX_CFAIL, S.fail_chol, 4, hipMemcpyDeviceToHost, st);
        };
        if (separate_waits)
        {
            read_step();
            OCHIP_HIP(ctx, ochip_stream_wait(ctx, st));
            erc = 1;
            if (!cfail && std::isfinite(h[1]) && h[1] > 0.0)
                erc = M.evaluate(false, 1, &cand_eval);
        }
        else if (spec)
        {
            // the candidate with its Jacobian, into the second set: if the step is accepted (nearly always) everything the
            // next iteration needs is there already
            erc = M.evaluate_candidate_jac(S.scale, &cand_eval, &fail_mask);
            if (erc == 0 && (fail_mask & 1))
                erc = 1;
        }
        else
        {
            M.before_wait = read_step;
            erc = M.evaluate(false, 1, &cand_eval);
            M.before_wait = nullptr;
        }
        if (erc < 0)
            return erc;
        const double model_cost_change = h[1];
        const bool valid = !cfail && std::isfinite(model_cost_change) && model_cost_change > 0.0;
        static const bool verbose = ochip_verbose("relax");
        if (verbose)
            fprintf(stderr, "[ochip relax] n=%d iter=%d cost=%.17g radius=%.6g model=%.17g step_norm=%.6g cfail=%d gmax=%.6g\n",
                    n, iter, x_cost, radius, model_cost_change, std::sqrt(h[2]), cfail, gmax);
        if (!valid)
        {
            if (++invalid >= 5)
                return finish(OCHIP_RELAX_FAILURE);
            radius *= 0.5;
            continue;
        }
        invalid = 0;
        double cand_cost = erc == 0 ? cand_eval : 1.7976931348623157e308;
        double step_norm = std::sqrt(h[2]), cand_norm = std::sqrt(h[3]);
        if (M.is_constrained())
        {
            // Projected line search along the step (bounds on the focal length): Armijo with sufficient decrease 1e-4; the
            // full step is the first trial and almost always passes, in which case nothing changes.  Otherwise the step is
            // contracted within [1e-3, 0.6] of the last trial, at most 20 trials, by the minimiser of the polynomial
            // through the trials' costs and directional derivatives (a Jacobian evaluation at every trial).
            std::vector<double> gh(n), yh(n);
            OCHIP_HIP(ctx, hipMemcpy(gh.data(), S.g, (size_t)n * 8, hipMemcpyDeviceToHost));
            OCHIP_HIP(ctx, hipMemcpy(yh.data(), S.y, (size_t)n * 8, hipMemcpyDeviceToHost));
            double gdd = 0, dmax = 0;
            for (int i = 0; i < n; i++)
            {
                const double d = -yh[i] * scale[i];
                gdd += gh[i] * d;
                dmax = std::max(dmax, std::abs(d));
            }
            if (eliminated)
            {
                double extra = 0;
                const int src = M.slope_extra(true, &extra);
                if (src)
                    return src;
                gdd += extra;
            }
            ls_sample at0, previous, current;
            at0.x = 0, at0.value = x_cost, at0.slope = gdd, at0.valid = true;
            current.x = 1.0, current.value = cand_cost, current.valid = cand_cost < 1e308;
            bool contracted = false, ok = true;
            int trials = 0;
            while (!current.valid || current.value > x_cost + 1e-4 * gdd * current.x)
            {
                if (++trials >= 20)
                {
                    ok = false;
                    break;
                }
                if (current.valid && !contracted) // the slope at the full step is needed from here on
                {
                    double c2;
                    erc = M.evaluate(true, 1, &c2);
                    if (erc < 0)
                        return erc;
                    std::vector<double> g1(n);
                    OCHIP_HIP(ctx, hipMemcpy(g1.data(), S.g, (size_t)n * 8, hipMemcpyDeviceToHost));
                    current.slope = 0;
                    for (int i = 0; i < n; i++)
                        current.slope += g1[i] * (-yh[i] * scale[i]);
                    if (eliminated)
                    {
                        double extra = 0;
                        const int src = M.slope_extra(false, &extra);
                        if (src)
                            return src;
                        current.slope += extra;
                    }
                    current.valid = erc == 0 && std::isfinite(current.slope);
                }
                contracted = true;
                const double next = interpolated_step(at0, previous, current, 1e-3 * current.x, 0.6 * current.x);
                if (next * dmax < 1e-9)
                {
                    ok = false;
                    break;
                }
                previous = current;
                M.launch_candidate(S.y, S.scale, next, S.scal);
                double c2;
                erc = M.evaluate(true, 1, &c2);
                if (erc < 0)
                    return erc;
                std::vector<double> g1(n);
                OCHIP_HIP(ctx, hipMemcpy(g1.data(), S.g, (size_t)n * 8, hipMemcpyDeviceToHost));
                current.x = next;
                current.value = c2;
                current.slope = 0;
                for (int i = 0; i < n; i++)
                    current.slope += g1[i] * (-yh[i] * scale[i]);
                if (eliminated)
                {
                    double extra = 0;
                    const int src = M.slope_extra(false, &extra);
                    if (src)
                        return src;
                    current.slope += extra;
                }
                current.valid = erc == 0 && std::isfinite(c2) && std::isfinite(current.slope);
            }
            if (contracted)
            {
                const double alpha = ok ? current.x : 1.0;
                M.launch_candidate(S.y, S.scale, alpha, S.scal);
                double c2;
                erc = M.evaluate(false, 1, &c2);
                if (erc < 0)
                    return erc;
                cand_cost = erc == 0 ? c2 : 1.7976931348623157e308;
                OCHIP_HIP(ctx, hipMemcpyAsync(h, S.scal, 64, hipMemcpyDeviceToHost, st));
                OCHIP_HIP(ctx, ochip_stream_wait(ctx, st));
                step_norm = std::sqrt(h[2]);
                cand_norm = std::sqrt(h[3]);
                // the trials overwrote J'J and J'r of the current point: restore them
                double c0;
                erc = M.evaluate(true, 0, &c0);
                if (erc < 0)
                    return erc;
            }
        }
        if (step_norm <= opt->parameter_tolerance * (x_norm + opt->parameter_tolerance))
            return finish(OCHIP_RELAX_CONVERGENCE_PARAMETER);
        const double cost_change = x_cost - cand_cost;
        if (std::abs(cost_change) <= opt->function_tolerance * x_cost)
            return finish(OCHIP_RELAX_CONVERGENCE_FUNCTION);
        const double rho = cost_change / model_cost_change;
        if (rho > 1e-3 && spec)
        {
            M.accept_swap();
            S.swap_sets();
            x_norm = cand_norm;
            x_cost = cand_eval;
            if (fail_mask & 2) // (a derivative at the accepted point is not finite: Ceres' "evaluation failed")
                return finish(OCHIP_RELAX_FAILURE);
            gmax = h[4];
            const double t = 2.0 * rho - 1.0;
            radius = radius / std::max(1.0 / 3.0, 1.0 - t * t * t);
            radius = std::min(1e16, radius);
            decrease_factor = 2.0;
            sum->successful_steps++;
            if (gmax <= opt->gradient_tolerance)
                return finish(OCHIP_RELAX_CONVERGENCE_GRADIENT);
        }
        else if (rho > 1e-3)
        {
            M.launch_accept();
            x_norm = cand_norm;
            // (the gradient norm and the diagonal of the new J'J ride on the evaluation's wait)
            auto read_gradient = [&]() {
                hipLaunchKernelGGL(lm_diag_kernel, dim3(1), dim3(LM_TG), 0, st, S.matA(), (const double *)S.g, S.diag_tmp, n, S.scal,
                                   (const double *)S.scale, S.diagonal, mail);
            };
            if (!separate_waits)
                M.before_wait = read_gradient;
            erc = M.evaluate(true, 0, &x_cost);
            M.before_wait = nullptr;
            if (erc < 0)
                return erc;
            if (erc != 0)
                return finish(OCHIP_RELAX_FAILURE);
            if (separate_waits)
            {
                read_gradient();
                OCHIP_HIP(ctx, ochip_stream_wait(ctx, st));
            }
            gmax = h[4];
            if (eliminated)
            {
                double extra = 0;
                const int xrc = M.gradient_max_extra(&extra);
                if (xrc)
                    return xrc;
                gmax = std::max(gmax, extra);
            }
            const double t = 2.0 * rho - 1.0;
            radius = radius / std::max(1.0 / 3.0, 1.0 - t * t * t);
            radius = std::min(1e16, radius);
            decrease_factor = 2.0;
            sum->successful_steps++;
            if (gmax <= opt->gradient_tolerance)
                return finish(OCHIP_RELAX_CONVERGENCE_GRADIENT);
        }
        else
        {
            radius = radius / decrease_factor;
            decrease_factor *= 2.0;
            sum->unsuccessful_steps++;
        }
    }
}

} // namespace ochip
