// libochip.so (internal) - the 64 x 64 diagonal tile of the tile Cholesky (factor and inverse in registers, 256 threads),
// shared by the one-launch factorisation of relax_lm.hip and the bootstrap chain's (relax_chain.hip).  Device code.
#pragma once

namespace
{

constexpr int CHOL_NB = 64;
typedef double v4f64 __attribute__((ext_vector_type(4)));

// Cholesky factor of the 64 x 64 diagonal block and its inverse, one workgroup.  The block lives in registers:
// thread (ty, tx) of a 16 x 16 grid owns rows ty + 16p and columns tx + 16q (cyclic, so the shrinking trailing
// matrix stays spread over all threads).  Step j: the owners of column j publish it (and the owners of row j of
// the inverse accumulator publish that) in LDS, one barrier, then every thread applies the rank-1 update to its
// register tile and the forward-elimination step  X_j /= l_jj,  X_i -= l_ij X_j  that turns the identity into
// L^-1.  The 64 steps run as 4 phases of 16 with the phase (j / 16) a compile-time constant: which register
// rows / columns are finished, live or on the pivot is then static, only the 16-row band of the pivot needs a
// runtime comparison, and finished parts cost nothing.  (This kernel is the critical path of the linear solve:
// 47 sequential launches per factorisation at n = 3003.)
template <int JB>
__device__ __forceinline__ void chol_diag_phase(double (&a)[4][4], double (&x)[4][4], double (*colA)[CHOL_NB], double (*rowX)[CHOL_NB],
                                                int ty, int tx, int nb, bool &bad)
{
    // TWO pivots per barrier: the owners publish columns j and j + 1 (and rows j, j + 1 of the inverse accumulator) as they
    // are before pivot j; every thread then derives pivot j + 1's column itself (col_{j+1} - l_j l_{j+1,j}: the update step j
    // would have applied to it) and applies both rank-1 updates, in the order and with the expressions of the
    // one-pivot-per-barrier loop, so the factor is the same to the bit with half the barriers on the critical path.
#pragma unroll 1
    for (int jt = 0; jt < 16; jt += 2)
    {
        const int j = JB * 16 + jt, buf = (jt >> 1) & 1;
        double(*c0) = colA[2 * buf], (*c1) = colA[2 * buf + 1], (*r0) = rowX[2 * buf], (*r1) = rowX[2 * buf + 1];
        if (tx == jt) // owners of column j: rows of band JB and below
#pragma unroll
            for (int p = JB; p < 4; p++)
                c0[ty + 16 * p] = a[p][JB];
        if (tx == jt + 1) // owners of column j + 1
#pragma unroll
            for (int p = JB; p < 4; p++)
                c1[ty + 16 * p] = a[p][JB];
        if (ty == jt) // owners of rows j / j + 1 of the inverse accumulator: columns up to band JB
#pragma unroll
            for (int q = 0; q <= JB; q++)
                r0[tx + 16 * q] = x[JB][q];
        if (ty == jt + 1)
#pragma unroll
            for (int q = 0; q <= JB; q++)
                r1[tx + 16 * q] = x[JB][q];
        __syncthreads();
        // ---- pivot j
        const double piv0 = c0[j];
        if (j < nb && !(piv0 > 0.0))
            bad = true;
        // 1 / sqrt(pivot): hardware estimate + two Newton steps (the factor is not on a bit-parity path)
        double rs0 = __builtin_amdgcn_rsq(piv0);
        rs0 = rs0 * (1.5 - 0.5 * piv0 * rs0 * rs0);
        rs0 = rs0 * (1.5 - 0.5 * piv0 * rs0 * rs0);
        double li0[4], lc0[4], xr0[4], li1[4], lc1[4], xr1[4];
#pragma unroll
        for (int p = JB; p < 4; p++)
        {
            const double v = c0[ty + 16 * p] * rs0;
            li0[p] = (p > JB || ty > jt) ? v : 0.0;
        }
#pragma unroll
        for (int q = JB; q < 4; q++)
        {
            const double v = c0[tx + 16 * q] * rs0;
            lc0[q] = (q > JB || tx > jt) ? v : 0.0;
        }
#pragma unroll
        for (int q = 0; q <= JB; q++)
            xr0[q] = r0[tx + 16 * q] * rs0;
        // ---- pivot j + 1: its column and its row of the inverse accumulator after pivot j's update
        const double cross = c0[j + 1] * rs0; // l_{j+1,j}
        const double piv1 = c1[j + 1] - cross * cross;
        if (j + 1 < nb && !(piv1 > 0.0))
            bad = true;
        double rs1 = __builtin_amdgcn_rsq(piv1);
        rs1 = rs1 * (1.5 - 0.5 * piv1 * rs1 * rs1);
        rs1 = rs1 * (1.5 - 0.5 * piv1 * rs1 * rs1);
        double col1_own[4]; // updated column j + 1 at this thread's rows (for its owners' final values)
#pragma unroll
        for (int p = JB; p < 4; p++)
        {
            const int r = ty + 16 * p;
            const double l0r = (p > JB || ty > jt) ? c0[r] * rs0 : 0.0;
            const double u = c1[r] - l0r * cross;
            col1_own[p] = u;
            const double v = u * rs1;
            li1[p] = (p > JB || ty > jt + 1) ? v : 0.0;
        }
#pragma unroll
        for (int q = JB; q < 4; q++)
        {
            const int c = tx + 16 * q;
            const double l0c = (q > JB || tx > jt) ? c0[c] * rs0 : 0.0;
            const double v = (c1[c] - l0c * cross) * rs1;
            lc1[q] = (q > JB || tx > jt + 1) ? v : 0.0;
        }
#pragma unroll
        for (int q = 0; q <= JB; q++)
            xr1[q] = (r1[tx + 16 * q] - cross * xr0[q]) * rs1;
        // ---- both rank-1 updates, pivot j first
#pragma unroll
        for (int p = JB; p < 4; p++)
        {
#pragma unroll
            for (int q = JB; q < 4; q++)
            {
                a[p][q] -= li0[p] * lc0[q];
                a[p][q] -= li1[p] * lc1[q];
            }
#pragma unroll
            for (int q = 0; q <= JB; q++)
            {
                x[p][q] -= li0[p] * xr0[q];
                x[p][q] -= li1[p] * xr1[q];
            }
        }
        // columns j, j + 1 become final (l below the diagonal, sqrt(pivot) = pivot * rs on it, 0 above); rows j, j + 1 of X too
        if (tx == jt)
#pragma unroll
            for (int p = JB; p < 4; p++)
            {
                const double v = c0[ty + 16 * p] * rs0;
                a[p][JB] = (p > JB || ty >= jt) ? v : 0.0;
            }
        if (tx == jt + 1)
#pragma unroll
            for (int p = JB; p < 4; p++)
                a[p][JB] = (p > JB || ty >= jt + 1) ? col1_own[p] * rs1 : 0.0;
        if (ty == jt)
#pragma unroll
            for (int q = 0; q <= JB; q++)
                x[JB][q] = xr0[q];
        if (ty == jt + 1)
#pragma unroll
            for (int q = 0; q <= JB; q++)
                x[JB][q] = xr1[q];
    }
}

// The blocked form of the same factorisation, used by chol_tiles_kernel (round 3).  The timeline of a factorisation
// (OCHIP_CHOL_TIMELINE) showed the diagonal tile at 25 us of a column's 39: 32 two-pivot steps of ~200 instructions with
// one wavefront per SIMD - instruction issue, not the barriers.  Most of those instructions were the rank-1 updates of the
// trailing matrix and of the inverse accumulator below the pivot's band.  Here a step only touches the pivot's own block
// of 16 columns (and the band's 16 rows of the inverse); after the 16 pivots of a block the rest follows as ONE rank-16
// update on the matrix cores: the panel L(rows below, 16) and the finished rows of X go through LDS, the products
// L L' (trailing blocks) and L X (rows of the inverse below the band) are formed with v_mfma_f64_16x16x4f64 and
// subtracted from the register tiles.  (Sums in a different order than the one-pivot loop: the factor is not on a
// bit-parity path; the chol_verify hook compares it with the launch chain, which keeps the unblocked phases.)
template <int JB>
__device__ __forceinline__ void chol_diag_panel_phase(double (&a)[4][4], double (&x)[4][4], double (*colA)[CHOL_NB], double (*rowX)[CHOL_NB],
                                                      int ty, int tx, int nb, bool &bad)
{
#pragma unroll 1
    for (int jt = 0; jt < 16; jt += 2)
    {
        const int j = JB * 16 + jt, buf = (jt >> 1) & 1;
        double(*c0) = colA[2 * buf], (*c1) = colA[2 * buf + 1], (*r0) = rowX[2 * buf], (*r1) = rowX[2 * buf + 1];
        if (tx == jt)
#pragma unroll
            for (int p = JB; p < 4; p++)
                c0[ty + 16 * p] = a[p][JB];
        if (tx == jt + 1)
#pragma unroll
            for (int p = JB; p < 4; p++)
                c1[ty + 16 * p] = a[p][JB];
        if (ty == jt)
#pragma unroll
            for (int q = 0; q <= JB; q++)
                r0[tx + 16 * q] = x[JB][q];
        if (ty == jt + 1)
#pragma unroll
            for (int q = 0; q <= JB; q++)
                r1[tx + 16 * q] = x[JB][q];
        __syncthreads();
        // What a step costs is latency, not instructions (56 000 cycles per tile for 32 steps whether a step has 90 or 200
        // instructions: OCHIP_CHOL_TIMELINE).  As first compiled, a step was a string of LDS round trips - every
        // `condition ? lds[..] * rs : 0` had become a branch around a read with its own wait, issued behind the Newton
        // chain of the pivot.  So: every LDS operand of the step is requested here, together and unconditionally, nothing
        // below touches LDS or branches (selects only), and the chains of dependent fp64 operations are short - Newton
        // steps as two fused operations each (e = 1 - p y y, y += y e / 2), the second pivot's 1 / sqrt from
        // d = c11 p0 - c10^2 (= p1 p0), which needs nothing of the first pivot's chain: rs1 = rsq(d) sqrt(p0).
        const double piv0 = c0[j], c10 = c0[j + 1], c11 = c1[j + 1];
        const double cc0 = c0[tx + 16 * JB], cc1 = c1[tx + 16 * JB]; // columns j, j + 1 at this thread's column of the block
        double cr0[4], cr1[4], xq0[4], xq1[4];
#pragma unroll
        for (int p = JB; p < 4; p++)
        {
            cr0[p] = c0[ty + 16 * p];
            cr1[p] = c1[ty + 16 * p];
        }
#pragma unroll
        for (int q = 0; q <= JB; q++)
        {
            xq0[q] = r0[tx + 16 * q];
            xq1[q] = r1[tx + 16 * q];
        }
        __builtin_amdgcn_sched_barrier(0);
        const double d1 = __builtin_fma(c11, piv0, -(c10 * c10));
        bad = bad || (j < nb && !(piv0 > 0.0)) || (j + 1 < nb && !(d1 > 0.0));
        double rs0 = __builtin_amdgcn_rsq(piv0), rd1 = __builtin_amdgcn_rsq(d1);
#pragma unroll
        for (int it = 0; it < 2; it++)
        {
            const double e0 = __builtin_fma(-(piv0 * rs0), rs0, 1.0), e1 = __builtin_fma(-(d1 * rd1), rd1, 1.0);
            rs0 = __builtin_fma(0.5 * rs0, e0, rs0);
            rd1 = __builtin_fma(0.5 * rd1, e1, rd1);
        }
        const double cross = c10 * rs0;        // l_{j+1,j}
        const double rs1 = rd1 * (piv0 * rs0); // 1 / sqrt(p1) = sqrt(p0) / sqrt(p1 p0)
        double li0[4], xr0[4], li1[4], xr1[4], l0full[4], col1_own[4];
#pragma unroll
        for (int p = JB; p < 4; p++)
        {
            l0full[p] = cr0[p] * rs0;
            li0[p] = (p > JB || ty > jt) ? l0full[p] : 0.0;
            col1_own[p] = cr1[p] - li0[p] * cross; // column j + 1 after pivot j's update
            li1[p] = (p > JB || ty > jt + 1) ? col1_own[p] * rs1 : 0.0;
        }
        const double l0c = (tx > jt) ? cc0 * rs0 : 0.0;
        const double l1c = (tx > jt + 1) ? (cc1 - l0c * cross) * rs1 : 0.0;
#pragma unroll
        for (int q = 0; q <= JB; q++)
        {
            xr0[q] = xq0[q] * rs0;
            xr1[q] = (xq1[q] - cross * xr0[q]) * rs1;
        }
        // the two rank-1 updates, inside the block's columns and the band's rows of X only; then columns j, j + 1 become
        // final (l below the diagonal, sqrt(pivot) = pivot * rs on it, 0 above) and rows j, j + 1 of X too
        const bool own0 = tx == jt, own1 = tx == jt + 1, row0 = ty == jt, row1 = ty == jt + 1;
#pragma unroll
        for (int p = JB; p < 4; p++)
        {
            double v = a[p][JB];
            v -= li0[p] * l0c;
            v -= li1[p] * l1c;
            const double f0 = (p > JB || ty >= jt) ? l0full[p] : 0.0;
            const double f1 = (p > JB || ty >= jt + 1) ? col1_own[p] * rs1 : 0.0;
            a[p][JB] = own0 ? f0 : (own1 ? f1 : v);
        }
#pragma unroll
        for (int q = 0; q <= JB; q++)
        {
            double v = x[JB][q];
            v -= li0[JB] * xr0[q];
            v -= li1[JB] * xr1[q];
            x[JB][q] = row0 ? xr0[q] : (row1 ? xr1[q] : v);
        }
    }
}

// after the 16 pivots of block JB: rows below the band, A(p, q) -= L(p, JB) L(q, JB)' for JB < q <= p and
// X(p, q) -= L(p, JB) X(JB, q) for q <= JB, as 16 x 16 x 16 products on the matrix cores (P: [64][>= 17] staging arrays)
template <int JB, int PITCH>
__device__ __forceinline__ void chol_diag_block_update(double (&a)[4][4], double (&x)[4][4], double (*T)[65], double (*Lp)[PITCH],
                                                       double (*XbT)[PITCH], int ty, int tx, int t)
{
    if (JB >= 3)
        return;
#pragma unroll
    for (int p = JB + 1; p < 4; p++)
        Lp[ty + 16 * p][tx] = a[p][JB]; // L(row, 16 JB + tx)
#pragma unroll
    for (int q = 0; q <= JB; q++)
        XbT[tx + 16 * q][ty] = x[JB][q]; // X(16 JB + ty, column) transposed: [column][row of the band]
    __syncthreads();
    const int w = t >> 6, lane = t & 63, lr = lane & 15, lk = lane >> 4;
    int b = 0;
#pragma unroll
    for (int p = JB + 1; p < 4; p++)
#pragma unroll
        for (int q = 0; q <= p; q++, b++)
            if ((b & 3) == w)
            {
                v4f64 acc = {0, 0, 0, 0};
#pragma unroll
                for (int kk = 0; kk < 16; kk += 4)
                {
                    const double av = Lp[16 * p + lr][kk + lk];
                    const double bv = q <= JB ? XbT[16 * q + lr][kk + lk] : Lp[16 * q + lr][kk + lk];
                    acc = __builtin_amdgcn_mfma_f64_16x16x4f64(av, bv, acc, 0, 0, 0);
                }
#pragma unroll
                for (int e = 0; e < 4; e++)
                    T[16 * p + 4 * e + lk][16 * q + lr] = acc[e];
            }
    __syncthreads();
#pragma unroll
    for (int p = JB + 1; p < 4; p++)
#pragma unroll
        for (int q = 0; q <= p; q++)
        {
            const double v = T[ty + 16 * p][tx + 16 * q];
            if (q <= JB)
                x[p][q] -= v;
            else
                a[p][q] -= v;
        }
}

} // namespace
