// libochip.so (internal) — the part of the relax solve every problem flavour shares: Levenberg-Marquardt trust-region
// loop with Ceres' semantics (ceres::TrustRegionMinimizer + LevenbergMarquardtStrategy, what
// RelaxProblem::solve -> ceres::Solver::Solve runs, src/relax/relax_problem.cpp:30-37,1404) on a
// reduced normal-equation system held in HBM as the 64 x 64 tiles of its block envelope (lower triangle only; the
// reference gives the same system to SPARSE_NORMAL_CHOLESKY, relax_problem.cpp:30-37) and factored in place (relax_lm.hip).
//
// A problem flavour (ground plane: relax.hip; mesh / intrinsics: relax_general.hip) implements lm_model: it owns the
// state, evaluates cost / J'J / J'r into the system's buffers, and applies a step to a candidate state.
#pragma once

#include "ctx.hpp"

#include <functional>
#include <utility>
#include <vector>

namespace ochip
{

constexpr int LM_NB = 64; // block size of the Cholesky factorisation (a 64 x 64 diagonal block per workgroup)
// threads of the single-workgroup kernels of a solve (reductions, the back substitution, the candidate step).  They were
// 1 024 until round 3: such a workgroup needs sixteen free wave slots on ONE compute unit, and beside the extraction's
// kernels it waited up to 17 ms for them (rocprofv3: back_solve_kernel 0.57 ms on average in the pipeline, 0.17 alone)
constexpr int LM_TG = 256;

// Block envelope of the reduced system (unknowns numbered so that coupled unknowns are close): per 64-column block the
// end of the rows that can be non-zero below it, the first row of the dense tail (unknowns coupled to everything: plane
// heights, shared intrinsics), and per block row the first column that can be non-zero (for the backward solve).
// region_begin (optional): the band's column blocks fall into regions that are not coupled to each other, only to the
// tail (a dissection of the camera graph: relax.hip, assign_tangent) - the first column block of each, ascending, [0] = 0;
// every region boundary is a multiple of 64 unknowns.  The factorisation then walks the regions' columns side by side
// and the backward substitution runs one workgroup per region.
struct lm_envelope
{
    std::vector<int> env_end, first_col, region_begin;
    int tail_begin = 0;
};

// ---- packed storage of the reduced system: the lower triangle as 64 x 64 tiles (row-major inside a tile) -----------
// Column block J owns the tiles of the row blocks J .. bend - 1 (the band: fill stays inside the monotone envelope) and
// tail_start .. (the dense tail and the augmented row n, which carries the right-hand side through the factorisation),
// stored one after the other from first_tile.  Everything outside these tiles is structurally zero and has no storage.
struct lm_col
{
    int first_tile; // index of tile (J, J); the column's tiles follow: rows J + 1 .. bend - 1, then tail_start ..
    int bend;       // end (exclusive) of the band's row blocks
    int tail_start; // first tail row block that is not already in the band
    int pad;
};
struct lm_matrix // device view
{
    double *tiles;
    const lm_col *cols;
};
__device__ __forceinline__ int lm_tile_index(const lm_col *cols, int I, int J)
{
    const lm_col c = cols[J];
    return c.first_tile + (I < c.bend ? I - J : (c.bend - J) + (I - c.tail_start));
}
// offset of entry (i, j), i >= j, which must lie inside the envelope
__device__ __forceinline__ size_t lm_at(const lm_matrix &M, int i, int j)
{
    return ((size_t)lm_tile_index(M.cols, i >> 6, j >> 6) << 12) + (size_t)(((i & 63) << 6) + (j & 63));
}

struct lm_system
{
    ochip_ctx *ctx = nullptr;
    std::vector<std::pair<void *, size_t>> *allocs = nullptr; // device blocks are recorded here (owner returns them to the pool)
    int n = 0;
    size_t cap_n = 0, linv_cap = 0, cap_tiles = 0;
    double *A = nullptr;       // J'J: packed lower triangle (lm_matrix), chol_n_tiles tiles; entries (i, j <= i), i < n
    double *g = nullptr;       // [n] J'r
    double *Wm = nullptr;      // scaled + damped system in the same tiles, row n = scaled gradient; factored in place
    double *gs = nullptr, *scale = nullptr, *lm_diag = nullptr, *diag_tmp = nullptr, *y = nullptr;
    double *diagonal = nullptr; // [n] clamp(diag(J'J) scale^2, 1e-6, 1e32) of the Jacobian of the last accepted point (lm_diag_kernel);
                                // the damping of an iteration, diagonal / radius, is formed where it is added (lm_build_kernel)
    bool A_clean = false;       // every entry of A that no engine kernel writes is zero (the layout has not changed since it was cleared)
    // Second set of everything a Jacobian evaluation produces (engines that evaluate the candidate WITH its Jacobian,
    // lm_model::speculates): the candidate's J'J, J'r and clamped diagonal land here and an accepted step swaps the sets
    bool speculative = false;   // set by the owner before lm_system_resize: allocate the second set
    double *A2 = nullptr, *g2 = nullptr, *diagonal2 = nullptr;
    bool A2_clean = false;
    size_t cap_tiles2 = 0, cap_n2 = 0;
    void swap_sets()
    {
        std::swap(A, A2);
        std::swap(g, g2);
        std::swap(diagonal, diagonal2);
        std::swap(A_clean, A2_clean);
    }
    lm_matrix matA2() const
    {
        return lm_matrix{A2, chol_cols};
    }
    double *scal = nullptr;    // [8] small results: [0] cost [1] model cost change [2] |step|^2 [3] |candidate|^2 [4] max |g|
    int *fail_chol = nullptr;
    double *linv = nullptr;    // inverses of the diagonal blocks
    int *first_col_dev = nullptr;
    lm_envelope env;
    // plan of the one-launch tile factorisation (relax_lm.hip: chol_tiles_kernel), rebuilt with the envelope
    lm_col *chol_cols = nullptr;     // per column block: first tile, band end, first tail block (device)
    std::vector<lm_col> cols_host;
    unsigned int *tile_ij = nullptr; // tiles in storage order: row block | column block << 16
    int *chol_kmin = nullptr;        // per row block: first column block whose envelope reaches it
    int *chol_korder = nullptr;      // regions only: the order in which a tile of the tail's columns takes the band's columns
    unsigned int *chol_tiles = nullptr; // tiles in claim order: row block | column block << 16
    unsigned int *chol_sync = nullptr;  // [0] claim counter, [4 + tile] done flags; zeroed before every factorisation
    int chol_n_tiles = 0, chol_nbc = 0, chol_nbr = 0, chol_tb = 0, chol_grid = 0;
    int chol_n_claims = 0; // entries of chol_tiles (a fused pair of tiles is one entry)
    size_t chol_sync_bytes = 0;
    // regions of the band (lm_envelope::region_begin): block bounds [n_regions + 1] (the last one = chol_tb), and the
    // backward substitution's private work vectors, [n_regions][n] + [n_regions] partial sums
    int n_regions = 1;
    int *region_dev = nullptr;
    double *back_work = nullptr;
    size_t back_work_cap = 0;
    // Page-locked host block the solve's read-backs go through.  Round 2: a hipMemcpyAsync to or from pageable memory is
    // staged and makes the host wait for the copy itself - eight such round trips per LM iteration.  Round 4: the kernel that
    // ends an evaluation writes the small results (cost, flags, step norms, gradient norm) into this block itself
    // (lm_mail) - a device-to-host copy is a blit kernel that queues for a compute unit like any other, ~70 us each beside
    // the extraction's kernels; an iteration had nine of them.
    // Layout (doubles): [0, 8) the solver's copy of scal, [8] the engine's cost, [9] the factorisation's failure flag (int),
    // [16, 32) the engines' failure flags (int32 per rank, at most 32 ranks), [32, 32 + n) diagonal of J'J (the solve's
    // first Jacobian only: the Jacobi scaling is computed on the host).
    double *box = nullptr;
    size_t box_cap = 0;
    static constexpr int BOX_SCAL = 0, BOX_COST = 8, BOX_CFAIL = 9, BOX_FAILS = 16, BOX_VECTORS = 32, BOX_MAX_RANKS = 32;
    ~lm_system();
    lm_matrix matA() const
    {
        return lm_matrix{A, chol_cols};
    }
    lm_matrix matW() const
    {
        return lm_matrix{Wm, chol_cols};
    }
    size_t matrix_bytes() const // of A, and of Wm
    {
        return (size_t)chol_n_tiles * LM_NB * LM_NB * sizeof(double);
    }
};

// J'J as a dense symmetric n x n matrix in host memory (tests, ochip_relaxg_evaluate)
int lm_download_dense(const lm_system &s, double *out);

// (re)size the buffers for n unknowns and take the envelope; returns OCHIP_OK or a negative code
int lm_system_resize(lm_system *s, int n, const lm_envelope &env);

// What the kernel that ends an evaluation writes to the host block (lm_system::box) when the engine mails its results:
// box[0, 8) = scal[0, 8), box[8] = scal[0] (the cost), the factorisation's failure flag, the ranks' evaluation flags.
struct lm_mail
{
    double *box = nullptr; // nullptr: nothing is mailed (the caller copies)
    const double *scal = nullptr;
    const int *fail_chol = nullptr;
    const int32_t *fail_ranks = nullptr;
    int world = 0;
    int with_cost = 0; // scal[0] is the evaluation's cost: box[8] takes it as well
    int32_t *clear_after = nullptr; // this rank's evaluation flag: zero again once it is posted (no memset in front of the next one)
};
#if defined(__HIPCC__)
// diag(A), max |g| and the clamped diagonal of the damping, by the TG threads of one workgroup (what lm_diag_kernel does;
// also the tail of an engine's last evaluation kernel).  Returns max |g| to thread 0.  sh: TG doubles of LDS.
template <int TG>
__device__ __forceinline__ double lm_diag_pass(const lm_matrix &A, const double *g, double *diag_out, int n, const double *scale,
                                               double *diagonal, double *sh)
{
    const int t = threadIdx.x;
    double m = 0;
    for (int i = t; i < n; i += TG)
    {
        const double d = A.tiles[lm_at(A, i, i)];
        if (diag_out)
            diag_out[i] = d;
        if (scale)
        {
            const double v = d * scale[i] * scale[i];
            const double lo = v < 1e-6 ? 1e-6 : v; // std::max(v, 1e-6), then std::min(.., 1e32): NaN passes through
            diagonal[i] = 1e32 < lo ? 1e32 : lo;
        }
        m = fmax(m, fabs(g[i]));
    }
    sh[t] = m;
    __syncthreads();
    for (int s = TG / 2; s > 0; s >>= 1)
    {
        if (t < s)
            sh[t] = fmax(sh[t], sh[t + s]);
        __syncthreads();
    }
    return sh[0];
}
__device__ __forceinline__ void lm_mail_post(const lm_mail &m) // one thread, after everything it mails is written
{
    if (!m.box)
        return;
    for (int i = 0; i < 8; i++)
        m.box[lm_system::BOX_SCAL + i] = m.scal[i];
    if (m.with_cost)
        m.box[lm_system::BOX_COST] = m.scal[0];
    if (m.fail_chol)
        *reinterpret_cast<int *>(m.box + lm_system::BOX_CFAIL) = *m.fail_chol;
    int32_t *f = reinterpret_cast<int32_t *>(m.box + lm_system::BOX_FAILS);
    for (int r = 0; r < m.world; r++)
        f[r] = m.fail_ranks[r];
    if (m.clear_after)
        *m.clear_after = 0;
}
#endif

struct lm_model
{
    virtual ~lm_model() = default;
    // true: evaluate() delivers scal[0, 8), the cost and both kinds of failure flags into sys.box by itself (lm_mail_post in
    // its last kernel); the solver then enqueues no copies of its own for them
    virtual bool mails_results()
    {
        return false;
    }
    // Speculative engines: the candidate is evaluated WITH its Jacobian, into the system's second set (sys.A2, g2,
    // diagonal2 = the clamped diagonal for `scale`; scal[4] = max |g2|), everything mailed: one evaluation and one wait
    // per iteration instead of two of each - nearly every step of these solves is accepted.  *fail_mask: bit 0 = a
    // residual was not finite (the candidate's cost is unusable), bit 1 = a derivative was not.  accept_swap(): the
    // candidate becomes the current state (the engine swaps its state buffers; the solver swaps the system's sets).
    virtual bool speculates()
    {
        return false;
    }
    // The backward substitution over the band's regions with the engine's candidate step as its tail (relax_lm_back.hpp:
    // the workgroup that finishes last computes the candidate state, alpha = 1): one launch less per iteration.  false:
    // not offered - the solver launches back_solve_regions_kernel and launch_candidate.
    struct back_args
    {
        lm_matrix W;
        int n;
        const double *linv;
        double *y, *work;
        const int *first_blk;
        int n_blocks;
        const int *region;
        int tb;
        const double *lm_diag, *gs;
        double *scal;
        unsigned int *arrived;
        int x_in_lds, n_regions;
    };
    virtual bool launch_back_solve_candidate(const back_args &a, const double *scale)
    {
        return false;
    }
    virtual int evaluate_candidate_jac(const double *scale, double *cost, int *fail_mask)
    {
        return OCHIP_EINVAL;
    }
    virtual void accept_swap()
    {
    }
    // Evaluate state `which` (0 = current, 1 = candidate).  with_jac: also fill sys.A (every entry (i, j <= i) of the
    // packed lower triangle, zeros included: hipMemsetAsync(sys.A, 0, sys.matrix_bytes()) first) and sys.g.  *cost = total cost.  Returns 0, 1 for a numeric failure (non-finite residual or derivative:
    // Ceres' "evaluation failed"), or a negative OCHIP_E* code for a hard error (HIP call, exchange) which ends the solve.
    virtual int evaluate(bool with_jac, int which, double *cost) = 0;
    // set by the solver around an evaluate() call: called by evaluate once everything of the evaluation is enqueued, right
    // before it waits for the stream - the solver's own reads (step quality, gradient norm, diagonal) then share that wait
    // instead of adding host round trips of their own to every iteration
    std::function<void()> before_wait;
    // enqueue: candidate = x (+) delta with delta[i] = alpha * (-y[i] * scale[i]); scal[2] = |x - candidate|^2 (ambient),
    // scal[3] = |candidate|^2 over the variable parameter blocks.  alpha = 1 except inside the projected line search of a
    // bounds-constrained problem.
    virtual void launch_candidate(const double *y, const double *scale, double alpha, double *scal) = 0;
    virtual void launch_accept() = 0;    // current = candidate
    virtual void launch_normalize() = 0; // what RelaxProblem::solve does to the state after Solve (:1410-1413)
    virtual int x_norm(double *out) = 0; // |x| over the variable parameter blocks of the current state
    virtual int num_residual_blocks() = 0;
    // bounds-constrained problems (focal length in [100, 20000], :496-497): Ceres then runs a projected line search
    // along the step before it tests it
    virtual bool is_constrained()
    {
        return false;
    }
    // ---- parameter blocks eliminated from the system by a Schur complement (relax_points.hip: the 3-D points).  The
    //      system lm_solve factors then holds the remaining unknowns only (possibly none at all); the model keeps the
    //      eliminated blocks' part of the normal equations from its last Jacobian evaluation and
    //        begin_solve            forgets the Jacobi scaling of the eliminated columns (fixed again from the solve's
    //                               first Jacobian, like the system's own),
    //        gradient_max_extra     max |g| over the eliminated columns,
    //        launch_schur           called once the scaled, damped system Wm (lower triangle and the augmented row n = the
    //                               scaled gradient) is built: subtracts the eliminated blocks' Schur term from both; the same
    //                               damping rule (clamp(diag * scale^2, 1e-6, 1e32) / radius) applies to their columns,
    //        launch_candidate       also back-substitutes the eliminated blocks, adds their share of the model cost
    //                               change to scal[1], of |step|^2 to scal[2] and of |candidate|^2 to scal[3],
    //        slope_extra            g . d over the eliminated columns for the projected line search: the gradient of the
    //                               current point (from_candidate: as launch_candidate left it) or of the Jacobian
    //                               evaluated last, d = the last full step.
    virtual void begin_solve()
    {
    }
    virtual bool has_eliminated()
    {
        return false;
    }
    virtual int gradient_max_extra(double *out)
    {
        *out = 0;
        return OCHIP_OK;
    }
    virtual void launch_schur(double radius, const double *scale, lm_matrix Wm, int n, int *fail /* set non-zero: invalid step */)
    {
    }
    virtual int slope_extra(bool from_candidate, double *out)
    {
        *out = 0;
        return OCHIP_OK;
    }
};

int lm_solve(lm_system &sys, lm_model &model, const ochip_relax_options *opt, ochip_relax_summary *sum);

// enqueue lm_diag_kernel on the system's CURRENT set: scal[4] = max |g|, diagonal = clamp(diag(A) scale^2) when scale is not
// nullptr, scal[0, 8) and the factorisation's flag mailed to the host block (engines that evaluate a candidate with its
// Jacobian through the generic route: swap_sets, evaluate(true, 1) with this as before_wait, swap_sets)
void lm_launch_diag(lm_system &sys, const double *scale, const int32_t *fail_ranks = nullptr, int world = 0, int with_cost = 0,
                    int32_t *clear_after = nullptr);

// shared by the flavours' problem_create: a device block from the context's pool, recorded in `allocs`
template <typename T>
inline int lm_dev_upload(ochip_ctx *ctx, std::vector<std::pair<void *, size_t>> *allocs, T **dst, const T *src, size_t n)
{
    size_t got = 0;
    void *d = ochip_pool_get(ctx, (n ? n : 1) * sizeof(T), &got);
    if (!d)
        return ochip_fail(ctx, OCHIP_ENOMEM, "device allocation of %zu bytes failed in relax problem", n * sizeof(T));
    allocs->emplace_back(d, got);
    if (n && src)
        if (hipMemcpy(d, src, n * sizeof(T), hipMemcpyHostToDevice) != hipSuccess)
            return ochip_fail(ctx, OCHIP_EHIP, "hipMemcpy failed in relax problem");
    *dst = (T *)d;
    return OCHIP_OK;
}

} // namespace ochip
