// The backward substitution of the relax solve over the regions of the band (relax_lm.hip: back_solve_regions_kernel), as
// a body an engine can instantiate with a tail of its own: the workgroup that finishes last sees the whole step vector and
// goes straight on to what would otherwise be the next launch (the ground-plane engine: its candidate state).
#pragma once

#include "relax_lm.hpp"

namespace ochip
{

struct back_no_tail
{
    __device__ void operator()() const
    {
    }
};

// The same substitution when the band falls into regions that are coupled through the tail only (lm_envelope::
// region_begin): one workgroup per region.  Each solves the tail's blocks for itself (the same arithmetic in every
// workgroup, on a private copy - nothing is exchanged), takes the tail's contribution out of its own region's columns
// on the way, and then walks its region from the bottom; what were 47 sequential blocks at n = 3003 are 9 + 9.  The
// workgroup that finishes last adds up the regions' parts of model_cost_change.
template <class Tail>
__device__ __forceinline__ void back_solve_regions_body(lm_matrix Lm, int n, const double *Linv, double *x, double *work,
                                                        const int *first_blk, int n_blocks, const int *region, int tb,
                                                        const double *lm_diag, const double *gs, double *scal, unsigned int *arrived,
                                                        int x_in_lds, Tail tail)
{
    constexpr int NB = LM_NB;
    typedef double v4f64 __attribute__((ext_vector_type(4)));
    // The kernel is a string of dependent trips to memory (rocprofv3: 0.21 ms for 19 blocks, ~11 us each, as four trips
    // per update of 16 rows, two for the inverse, one for x).  So: the workgroup's part of x - its region and the tail -
    // lives in LDS when it fits; a block's rows are split over the four wavefronts (16 rows each, all loads of a step in
    // flight at once), whose partial sums meet in LDS and are added in wavefront order.
    constexpr int XCAP = 4608; // (36 KB: a 5 000-camera survey's region of 36 blocks and its 1 428 tail unknowns fit)
    __shared__ double xl[XCAP];
    __shared__ double xb[NB];
    __shared__ double part[4][LM_TG];
    __shared__ double sh[LM_TG];
    __shared__ int s_last;
    const int t = threadIdx.x, r = blockIdx.x, m = gridDim.x;
    const int cq = t & 63, rq = t >> 6;
    const double *L = Lm.tiles;
    double *parts = work + (size_t)m * n;
    const int rb = region[r], re = region[r + 1];     // own column blocks [rb, re)
    const int c_lo = rb * NB, c_hi = min(re * NB, n); // own columns
    const int t_lo = min(tb * NB, n);                 // columns of the tail's blocks
    const int n_own = c_hi - c_lo, n_tail = n - t_lo;
    const bool in_lds = x_in_lds && n_own + n_tail <= XCAP; // (otherwise: this workgroup's vector in `work`)
    double *const x_own = in_lds ? xl : work + (size_t)r * n + c_lo;
    double *const x_tail = in_lds ? xl + n_own : work + (size_t)r * n + t_lo;
    auto X = [&](int i) -> double & { return i < t_lo ? x_own[i - c_lo] : x_tail[i - t_lo]; };
    for (int i = c_lo + t; i < c_hi; i += LM_TG) // y = L^-1 gs: the augmented row
        X(i) = L[lm_at(Lm, n, i)];
    for (int i = t_lo + t; i < n; i += LM_TG)
        X(i) = L[lm_at(Lm, n, i)];
    auto block_step = [&](int k, int lo0, int hi0, int lo1, int hi1) {
        const int k0 = k * NB, nb = min(NB, n - k0);
        const double *Li = Linv + (size_t)k * NB * NB;
        __syncthreads(); // the updates of the previous block have landed
        if (t < NB)
            xb[t] = t < nb ? X(k0 + t) : 0.0;
        __syncthreads();
        {
            // (L^-T y)[c] = sum_{m >= c} Linv[m][c] y[m] (rows beyond the block are rows of the identity, y is 0 there)
            double v[16], s = 0;
#pragma unroll
            for (int j = 0; j < 16; j++)
                v[j] = Li[(16 * rq + j) * NB + cq];
#pragma unroll
            for (int j = 0; j < 16; j++)
                if (16 * rq + j >= cq)
                    s += v[j] * xb[16 * rq + j];
            part[rq][cq] = s;
        }
        __syncthreads();
        if (t < NB)
        {
            const double s = ((part[0][t] + part[1][t]) + part[2][t]) + part[3][t];
            xb[t] = t < nb ? s : 0.0;
            if (t < nb)
                X(k0 + t) = s;
        }
        __syncthreads();
        // the columns in reach of the block, 256 at a time: four neighbouring columns per lane (a tile's row is contiguous,
        // every range starts and ends on a tile boundary: 32-byte loads), 16 rows per wavefront
        for (int pass = 0; pass < 2; pass++)
        {
            const int lo = pass ? lo1 : lo0, hi = pass ? hi1 : hi0;
            for (int base = lo; base < hi; base += LM_TG)
            {
                const int i = base + 4 * cq;
                double u0 = 0, u1 = 0, u2 = 0, u3 = 0;
                if (i < hi)
                {
                    const double *Lc = L + ((size_t)lm_tile_index(Lm.cols, k, i >> 6) << 12) + (i & 63);
                    v4f64 v[16];
#pragma unroll
                    for (int j = 0; j < 16; j++) // (the augmented row shares the last block's tile: rows >= nb are skipped)
                        v[j] = 16 * rq + j < nb ? *reinterpret_cast<const v4f64 *>(Lc + (16 * rq + j) * NB) : v4f64{0, 0, 0, 0};
#pragma unroll
                    for (int j = 0; j < 16; j++)
                    {
                        const double xm = xb[16 * rq + j];
                        u0 += v[j][0] * xm;
                        u1 += v[j][1] * xm;
                        u2 += v[j][2] * xm;
                        u3 += v[j][3] * xm;
                    }
                }
                part[rq][4 * cq] = u0;
                part[rq][4 * cq + 1] = u1;
                part[rq][4 * cq + 2] = u2;
                part[rq][4 * cq + 3] = u3;
                __syncthreads();
                if (base + t < hi)
                    X(base + t) -= ((part[0][t] + part[1][t]) + part[2][t]) + part[3][t];
                __syncthreads();
            }
        }
    };
    for (int k = n_blocks - 1; k >= tb; k--) // the tail's blocks: the tail's own columns below the block, and this region's
        block_step(k, t_lo, k * NB, c_lo, c_hi);
    for (int k = min(re, n_blocks) - 1; k >= rb; k--)
        block_step(k, max(first_blk[k] * NB, c_lo), k * NB, 0, 0);
    __syncthreads();
    // results and this region's part of model_cost_change (workgroup 0: the tail's as well)
    double part_sum = 0;
    for (int i = c_lo + t; i < c_hi; i += LM_TG)
    {
        const double v = X(i);
        x[i] = v;
        part_sum += v * gs[i] + lm_diag[i] * v * v;
    }
    if (r == 0)
        for (int i = t_lo + t; i < n; i += LM_TG)
        {
            const double v = X(i);
            x[i] = v;
            part_sum += v * gs[i] + lm_diag[i] * v * v;
        }
    sh[t] = part_sum;
    // (x[] / X's stores of every wavefront have left it before the barriers in front of thread 0's release and arrival)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int s = LM_TG / 2; s > 0; s >>= 1)
    {
        if (t < s)
            sh[t] += sh[t + s];
        __syncthreads();
    }
    if (t == 0)
    {
        __hip_atomic_store(&parts[r], sh[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        s_last = __hip_atomic_fetch_add(arrived, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT) == (unsigned int)(m - 1);
        if (s_last)
        {
            double sum = 0;
            for (int q = 0; q < m; q++)
                sum += __hip_atomic_load(&parts[q], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            scal[1] = 0.5 * sum;
            __hip_atomic_store(arrived, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    __syncthreads();
    if (s_last) // the workgroup that finished last: x is complete (every region's release precedes its arrival)
    {
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        tail();
    }
}

} // namespace ochip
