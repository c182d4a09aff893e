// libochip.so — the NaN bootstrap of runGroundPlane as ONE persistent launch (gfx950).
//
// Replaces the loop of src/relax/relax.cpp:52-80: the cameras of a batch arrive without an orientation; in the pose list's
// order each one takes the orientation of the pose in front of it and is relaxed - setupGroundPlaneProblem (grid filter,
// 2-ray blocks over a fresh border triangle, downward priors: src/relax/relax_problem.cpp:61-81), relaxObservedModelOnly
// (the plane's heights alone, :931-984), solve (:1390-1420) - on its own against the oriented cameras of the graph when the
// graph is more than twice the group (:61-68), else together with every pose of the group (:70-75).  Host-driven, that is a
// problem create, two Levenberg-Marquardt solves and some forty waits per camera (round 5: 0.7 ms per solve, 4.2 s per
// 1 000 images in the reference's schedule).  Here the group's cameras, edges and inlier matches are uploaded once and a
// grid of workgroups that stays resident walks the cameras: phases (set-up of a step, evaluation of the residual blocks,
// assembly of the normal equations, the scaled and damped system, its tile Cholesky) separated by grid-wide syncs whose
// last arriver runs the serial part - the trust-region logic of ceres::TrustRegionMinimizer as relax_lm.hip's lm_solve
// restates it, small systems solved on the spot - before it releases the others.  No host round trip between the first
// camera and the last.
//
// Arithmetic: relax_setup_geom.hpp (scores, cells, triangle test: the set-up kernels' expressions), relax_plane_functor.hpp
// (the cost functor and its forward-mode derivatives: the pair-record engine's), the diagonal-tile Cholesky of
// relax_chol_tile.hpp; records per (edge, 64 blocks) instead of per camera pair, a dense lower triangle of 64 x 64 tiles,
// cameras in the caller's order.  `stepped` launches run ONE phase per launch with the same code: the test hook
// OCHIP_TEST_HOOKS=chain_stepped compares the two to the bit.
//
// Every spin is bounded: a workgroup that waits longer than the limit raises the abort flag and leaves; the caller then
// continues from the last finished camera with the host loop (csrc/host/relax.cpp).
#include "ctx.hpp"
#include "dual.hpp"
#include "undistort.hpp"

#include <algorithm>
#include <cmath>
#include <cstring>
#include <map>
#include <mutex>
#include <vector>

using namespace ochip;

#include "relax_chol_tile.hpp"
#include "relax_plane_functor.hpp"
#include "relax_setup_geom.hpp"

namespace
{

constexpr int TG = 256;      // threads per workgroup: four wavefronts, one per SIMD
constexpr int WV = 64;
constexpr int NB = CHOL_NB;  // tile size of the reduced system
constexpr int REC = 56;      // doubles per (edge, 64 blocks) record: PLANE_ACC + padding
constexpr int JP = 61;       // doubles per lane of an evaluating wavefront in LDS: 54 of J, 6 of r, 1 of padding
constexpr int N_SMALL = 12;  // systems up to this size are factored by one thread in the sync's epilogue
constexpr int N_LIMIT = 1023;
constexpr int KC = 32;       // the tile factorisation stages its operands through LDS in halves
constexpr size_t LDS_EVAL = (size_t)4 * WV * JP * sizeof(double); // 124 928 bytes: the evaluation's share, and the most any phase needs
constexpr int V_WAVES = 1024; // the plane block / cost are summed per virtual wavefront (chunks i, i + 1024, ...), whatever the grid is
constexpr unsigned long long SPIN_LIMIT_TICKS = 300000000ull;    // 3 s of the 100 MHz wall clock

enum
{
    PH_INIT = 0, // camera-frame rays of every inlier match
    PH_SETUP,    // grid filter + residual-block list of the step's edges
    PH_EVAL,     // the residual blocks of one state: records per (edge, 64 blocks)
    PH_ASM,      // records -> J'J, J'r, cost
    PH_CHOL,     // tile Cholesky of the scaled + damped system (formed on the way in)
    PH_DONE
};

enum
{
    CHAIN_RUNNING = 0,
    CHAIN_FINISHED = 1,
    CHAIN_INEXACT = -1, // an edge's grid filter needs the host walk (a tie for a cell's best score, a match outside the table)
    CHAIN_ABORTED = -2  // a spin ran into its limit
};

struct chain_ctl
{
    int phase, step, which, status;
    int cur;                // the current state buffer and system set
    int mode, n_filter, cam;
    int n, n_active, nz, nbc, nbr, n_claims;
    int n_blocks, n_prior, n_live;
    int lm_first;
    int iter, invalid, iterations, successful, unsuccessful;
    int eval_state, eval_set, eval_cams;
    int fail_bits, chol_fail;
    int steps_done, solves, iterations_total, last_iterations, last_blocks, last_termination;
    int syncs;
    double radius, decrease, x_cost, x_norm, gmax, mcc, step_norm2, cand_norm2;
    double initial_cost, last_initial_cost, last_final_cost;
    double plane_xy[6];
    double z[2][3];
    double tot[10]; // the evaluation under way: the plane's block (6), its gradient (3), the cost
    // where the launch's time goes, in ticks of the 100 MHz wall clock: per phase the time from the previous sync's release
    // to the last arrival (the parallel part and the sync itself), the epilogue behind it, and how often it ran
    unsigned long long t_last, t_phase[8], t_epilogue[8], n_phase[8];
};

struct chain_bar
{
    unsigned int count, gen, abort, pad;
};

struct chain_dev
{
    uint32_t n_cams, n_edges, n_chunks, n_pairs, n_steps, n_max;
    uint64_t n_inliers;
    const double *cam_pos;       // [n_cams][3]
    double *q[2];                // [n_cams][4] the two state buffers
    double *q_commit;            // the state after the last finished step
    const uint8_t *cam_opt1;     // [n_cams] optimised in mode 1
    const ochip_plane_edge *edges;
    const ochip_plane_inlier *inliers;
    const double *models;        // [n_models][10]
    double *rays;                // [n_inliers][6]
    double *score;               // [n_inliers]
    uint8_t *keep;               // [n_inliers]
    uint32_t *blk_idx;           // per edge at blk_off[e]: the inlier indices of its residual blocks
    const uint32_t *blk_off;     // [n_edges + 1]
    uint32_t *blk_cnt;           // [n_edges]
    uint8_t *inexact;            // [n_edges]
    const uint32_t *chunk_edge, *chunk_first; // [n_chunks]
    double *rec[2];              // [n_chunks][REC]
    const uint32_t *cam_chunk_off, *cam_chunk_idx; // CSR camera -> chunk * 2 + role
    const uint32_t *pair_p, *pair_q, *pair_chunk_off, *pair_chunk_idx;
    uint8_t *cam_prior;          // [n_cams] this step's priors
    uint8_t *cam_blocks;         // [n_cams] the camera has residual blocks in this step
    int32_t *cam_t;              // [n_cams] tangent offset or -1
    double *A[2], *g[2], *diagonal[2]; // the two sets of J'J (packed lower tiles), J'r, clamped diagonal
    double *W, *linv;            // scaled + damped system (factored in place), inverses of its diagonal blocks
    double *scale, *gs, *lm_diag, *y;
    unsigned int *chol_sync;     // [0] claim counter, [4 + tile] flags
    unsigned int *chol_claims;   // tiles in claim order
    uint32_t *live;              // [n_chunks] the chunks that hold blocks in this step, ascending
    double *partials;            // [V_WAVES][10]
    chain_ctl *ctl;
    chain_bar *bar;
    const ochip_plane_chain_step *steps;
    double res, huber_a, prior_weight;
};

__device__ __forceinline__ int tile_of(int I, int J) // packed lower triangle of tiles
{
    return I * (I + 1) / 2 + J;
}
__device__ __forceinline__ size_t at(int i, int j) // entry (i, j <= i)
{
    return ((size_t)tile_of(i >> 6, j >> 6) << 12) + (size_t)(((i & 63) << 6) + (j & 63));
}
__device__ __forceinline__ int ld_ctl(const int *p)
{
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void store_through(double *p, double v)
{
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ bool nan4(const double *q)
{
    return isnan(q[0]) || isnan(q[1]) || isnan(q[2]) || isnan(q[3]);
}

// ------------------------------------------------------------------------------------------------- the grid-wide sync
// Every workgroup arrives at a monotonic counter; the last arriver runs `epilogue` (all its threads) and publishes the next
// generation, the others wait for it.  Hand-over as MI355X_MICROARCH.md prescribes for non-coherent L2s: every storing wave
// drained, workgroup barrier, ONE lane: release fence, drain, arrive; the waiting side polls relaxed, ONE acquire fence,
// workgroup barrier, plain loads.  false: aborted (a spin ran into its limit, here or in another workgroup).
template <class F> __device__ bool grid_sync(const chain_dev &D, unsigned int &gen, int *s_flag, F epilogue)
{
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0)
    {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const unsigned int old = __hip_atomic_fetch_add(&D.bar->count, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
        *s_flag = (old + 1u == (gen + 1u) * gridDim.x) ? 1 : 0;
    }
    __syncthreads();
    if (*s_flag)
    {
        if (threadIdx.x == 0)
        {
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        __syncthreads();
        epilogue();
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (threadIdx.x == 0)
        {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __hip_atomic_store(&D.bar->gen, gen + 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            *s_flag = 1;
        }
    }
    else if (threadIdx.x == 0)
    {
        const unsigned long long t0 = wall_clock64();
        int ok = 1;
        while ((int)(__hip_atomic_load(&D.bar->gen, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - (gen + 1u)) < 0)
        {
            __builtin_amdgcn_s_sleep(1);
            if (__hip_atomic_load(&D.bar->abort, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u)
            {
                ok = 0;
                break;
            }
            if (wall_clock64() - t0 > SPIN_LIMIT_TICKS)
            {
                __hip_atomic_store(&D.bar->abort, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                ok = 0;
                break;
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        *s_flag = ok;
    }
    __syncthreads();
    const bool ok = *s_flag != 0;
    __syncthreads();
    gen++;
    return ok;
}

// ------------------------------------------------------------------------------------------------------ parallel phases
__device__ void phase_init(const chain_dev &D)
{
    const int lane = threadIdx.x & 63;
    const uint32_t waves = gridDim.x * 4, gw = blockIdx.x * 4 + (threadIdx.x >> 6);
    for (uint32_t e = gw; e < D.n_edges; e += waves)
    {
        const ochip_plane_edge ed = D.edges[e];
        double ms[8], md[8];
        const double *Ma = D.models + 10 * (size_t)ed.model_a, *Mb = D.models + 10 * (size_t)ed.model_b;
        for (int k = 0; k < 8; k++)
        {
            ms[k] = Ma[k];
            md[k] = Mb[k];
        }
        for (uint32_t i = lane; i < ed.n_inliers; i += 64)
        {
            const ochip_plane_inlier m = D.inliers[ed.inlier_offset + i];
            double r1[3], r2[3];
            ochip_ud::image_to_3d(m.px1, ms, r1);
            ochip_ud::image_to_3d(m.px2, md, r2);
            double *o = D.rays + 6 * (ed.inlier_offset + i);
            o[0] = r1[0], o[1] = r1[1], o[2] = r1[2], o[3] = r2[0], o[4] = r2[1], o[5] = r2[2];
        }
    }
}

// gridFilterMatchesPerImage and the block list of the step's edges, one wavefront per edge (relax_setup.hip's two kernels
// in one pass: the blocks of an edge stay in its own slot, no scan)
__device__ void phase_setup(const chain_dev &D, unsigned char *lds)
{
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const uint32_t waves = gridDim.x * 4, gw = blockIdx.x * 4 + wv;
    unsigned long long *best_s = reinterpret_cast<unsigned long long *>(lds + (size_t)wv * WV * JP * 8);
    unsigned long long *best_d = best_s + G_MAX * G_MAX;
    unsigned int *cnt_s = reinterpret_cast<unsigned int *>(best_d + G_MAX * G_MAX), *cnt_d = cnt_s + G_MAX * G_MAX;
    const int cur = ld_ctl(&D.ctl->cur), n_filter = ld_ctl(&D.ctl->n_filter);
    const double *Q = D.q[cur];
    double tri[6];
    for (int k = 0; k < 6; k++)
        tri[k] = D.ctl->plane_xy[k];
    for (uint32_t e = gw; e < D.n_edges; e += waves)
    {
        if ((int)e >= n_filter)
        {
            if (lane == 0)
            {
                D.blk_cnt[e] = 0;
                D.inexact[e] = 0;
            }
            continue;
        }
        const ochip_plane_edge ed = D.edges[e];
        for (int c = lane; c < G_MAX * G_MAX; c += 64)
        {
            best_s[c] = 0ull;
            best_d[c] = 0ull;
            cnt_s[c] = 0u;
            cnt_d[c] = 0u;
        }
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
        __builtin_amdgcn_wave_barrier();
        double Rs[3][3], Rd[3][3], ms[8], md[8], qa[4], qb[4];
        for (int k = 0; k < 4; k++)
        {
            qa[k] = Q[4 * (size_t)ed.cam_a + k];
            qb[k] = Q[4 * (size_t)ed.cam_b + k];
        }
        to_matrix(qa, Rs);
        to_matrix(qb, Rd);
        const double *pa = D.cam_pos + 3 * (size_t)ed.cam_a, *pb = D.cam_pos + 3 * (size_t)ed.cam_b;
        const v3 so{pa[0], pa[1], pa[2]}, d_o{pb[0], pb[1], pb[2]};
        const double *Ma = D.models + 10 * (size_t)ed.model_a, *Mb = D.models + 10 * (size_t)ed.model_b;
        for (int k = 0; k < 8; k++)
        {
            ms[k] = Ma[k];
            md[k] = Mb[k];
        }
        const double cols_s = Ma[8], rows_s = Ma[9], cols_d = Mb[8], rows_d = Mb[9];
        const bool homography = (ed.flags & 1u) != 0;
        const ochip_plane_inlier *in = D.inliers + ed.inlier_offset;
        const double *rays = D.rays + 6 * ed.inlier_offset;
        double *score = D.score + ed.inlier_offset;
        unsigned char *keep = D.keep + ed.inlier_offset;
        bool bad = false;
        for (uint32_t i = lane; i < ed.n_inliers; i += 64)
        {
            const ochip_plane_inlier m = in[i];
            const double s = plane_match_score(rays + 6 * (size_t)i, rays + 6 * (size_t)i + 3, Rs, Rd, so, d_o, m, ms, md, ed.H, homography);
            score[i] = s;
            if (!(s > 0))
                continue;
            int cs, cd;
            if (!plane_match_cells(m, cols_s, rows_s, cols_d, rows_d, D.res, &cs, &cd))
            {
                bad = true;
                continue;
            }
            atomicMax(&best_s[cs], (unsigned long long)__double_as_longlong(s));
            atomicMax(&best_d[cd], (unsigned long long)__double_as_longlong(s));
        }
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
        __builtin_amdgcn_wave_barrier();
        for (uint32_t i = lane; i < ed.n_inliers; i += 64)
        {
            const ochip_plane_inlier m = in[i];
            const double s = score[i];
            unsigned char k = 0;
            int cs, cd;
            if (s > 0 && plane_match_cells(m, cols_s, rows_s, cols_d, rows_d, D.res, &cs, &cd))
            {
                const unsigned long long bits = (unsigned long long)__double_as_longlong(s);
                if (best_s[cs] == bits)
                {
                    k |= 1;
                    atomicAdd(&cnt_s[cs], 1u);
                }
                if (best_d[cd] == bits)
                {
                    k |= 2;
                    atomicAdd(&cnt_d[cd], 1u);
                }
            }
            keep[i] = k;
        }
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
        __builtin_amdgcn_wave_barrier();
        for (int c = lane; c < G_MAX * G_MAX; c += 64)
            bad = bad || cnt_s[c] > 1u || cnt_d[c] > 1u;
        // the blocks: kept matches whose rays meet over the border triangle, in match order
        const uint32_t cap = D.blk_off[e + 1] - D.blk_off[e];
        uint32_t *out = D.blk_idx + D.blk_off[e];
        unsigned int total = 0;
        const unsigned long long below = (1ull << lane) - 1ull;
        for (uint32_t i0 = 0; i0 < ed.n_inliers; i0 += 64)
        {
            const uint32_t i = i0 + lane;
            bool ok = false;
            if (i < ed.n_inliers && keep[i] != 0)
                ok = plane_block_inside(rays + 6 * (size_t)i, rays + 6 * (size_t)i + 3, qa, qb, so, d_o, tri);
            const unsigned long long mask = __ballot(ok);
            if (ok)
            {
                const unsigned int slot = total + (unsigned int)__popcll(mask & below);
                if (slot < cap)
                    out[slot] = i;
            }
            total += (unsigned int)__popcll(mask);
        }
        if (total > cap)
            bad = true;
        const bool any_bad = __any(bad) != 0;
        if (lane == 0)
        {
            D.blk_cnt[e] = any_bad ? 0u : total;
            D.inexact[e] = any_bad ? 1 : 0;
        }
    }
}

// the residual blocks of one (edge, 64 blocks) chunk: relax.hip's relax_pair_eval_kernel for one trip of a wavefront
template <bool CAMS> __device__ void eval_chunk(const chain_dev &D, const double *Q, const double *Z, const double *plane_xy, double *rec,
                                                uint32_t chunk, double (*Jl)[JP], int lane, bool &failed, bool &failed_jac, double &entry_out,
                                                double &cost_out)
{
    const uint32_t e = D.chunk_edge[chunk], first = D.chunk_first[chunk], cnt = D.blk_cnt[e];
    entry_out = 0;
    cost_out = 0;
    if (first >= cnt)
        return;
    const ochip_plane_edge &ed = D.edges[e];
    const uint32_t ca = ed.cam_a, cb = ed.cam_b, p = min(ca, cb);
    const int blocks = (int)min((uint32_t)WV, cnt - first);
    const double a2 = D.huber_a * D.huber_a;
    int ei = 0, ej = 0;
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
    double cost = 0;
    if (lane < blocks)
    {
        const uint32_t idx = D.blk_idx[D.blk_off[e] + first + lane];
        const double *rays = D.rays + 6 * (ed.inlier_offset + idx);
        const double *la = D.cam_pos + (size_t)ca * 3, *lb = D.cam_pos + (size_t)cb * 3;
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
            if (!plane_intersection_residuals<double>(in, la, lb, rays, plane_xy, r))
                failed = true;
        }
        double s = 0;
        for (int k = 0; k < 6; k++)
        {
            s += r[k] * r[k];
            if (!(r[k] - r[k] == 0.0))
                failed = true;
        }
        double sqrt_rho1 = 1.0, c = 0.5 * s;
        if (s > a2)
        {
            const double rn = sqrt(s);
            const double rho1 = fmax(2.2250738585072014e-308, D.huber_a / rn);
            sqrt_rho1 = sqrt(rho1);
            c = 0.5 * (2.0 * D.huber_a * rn - a2);
        }
        cost = c;
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
        if (CAMS)
        {
            seed_quat(qa, seeded);
            plane_intersection_residuals_mixed(seeded, qb, Z, la, lb, rays, plane_xy, rd);
            take(first_col);
            seed_quat(qb, seeded);
            plane_intersection_residuals_mixed(qa, seeded, Z, la, lb, rays, plane_xy, rd);
            take(second_col);
        }
        D3 zs[3];
        for (int k = 0; k < 3; k++)
        {
            zs[k] = D3(Z[k]);
            zs[k].v[k] = 1.0;
        }
        plane_intersection_residuals_mixed(qa, qb, zs, la, lb, rays, plane_xy, rd);
        take(6);
        for (int k = 0; k < 6; k++)
            mine[54 + k] = r[k] * sqrt_rho1;
    }
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
    __builtin_amdgcn_wave_barrier();
    const bool mine_needed = CAMS || ei >= 6; // without the cameras only the plane's entries exist
    if (lane < 54 && mine_needed)
    {
        double entry = 0;
        const int cb2 = lane < 45 ? ej : 54;
        for (int b = 0; b < blocks; b++)
        {
            const double *jb = Jl[b];
            double m = 0;
            for (int k = 0; k < 6; k++)
                m += jb[k * 9 + ei] * (lane < 45 ? jb[k * 9 + cb2] : jb[54 + k]);
            entry += m;
        }
        rec[lane] = entry;
        entry_out = entry;
    }
    for (int off = 32; off >= 1; off >>= 1)
        cost += __shfl_xor(cost, off);
    if (lane == 0)
        rec[54] = cost;
    cost_out = cost;
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

__device__ void phase_eval(const chain_dev &D, unsigned char *lds)
{
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const uint32_t waves = gridDim.x * 4, gw = blockIdx.x * 4 + wv;
    const int state = ld_ctl(&D.ctl->eval_state), set = ld_ctl(&D.ctl->eval_set), cams = ld_ctl(&D.ctl->eval_cams);
    const int nbc = ld_ctl(&D.ctl->nbc);
    // the target set's J'J starts from zero (the assembly writes the entries the step's blocks reach, nothing else)
    {
        const size_t doubles = (size_t)tile_of(nbc, 0) << 12;
        double *A = D.A[set];
        for (size_t i = (size_t)blockIdx.x * TG + threadIdx.x; i < doubles; i += (size_t)gridDim.x * TG)
            A[i] = 0.0;
    }
    double(*Jl)[JP] = reinterpret_cast<double(*)[JP]>(lds + (size_t)wv * WV * JP * 8);
    const double *Q = D.q[state];
    double Z[3], xy[6];
    for (int k = 0; k < 3; k++)
        Z[k] = D.ctl->z[state][k];
    for (int k = 0; k < 6; k++)
        xy[k] = D.ctl->plane_xy[k];
    bool failed = false, failed_jac = false;
    const uint32_t n_live = (uint32_t)ld_ctl(&D.ctl->n_live);
    // a virtual wavefront takes the live chunks vw, vw + V_WAVES, ... and leaves the sums of their plane entries, plane
    // gradient and cost: the serial part adds the V_WAVES sums in a fixed tree, the result does not depend on the grid
    for (uint32_t vw = gw; vw < min((uint32_t)V_WAVES, n_live); vw += waves)
    {
        double acc = 0, cost_acc = 0;
        for (uint32_t i = vw; i < n_live; i += V_WAVES)
        {
            const uint32_t c = D.live[i];
            double *rec = D.rec[set] + (size_t)c * REC;
            double entry, cost;
            if (cams)
                eval_chunk<true>(D, Q, Z, xy, rec, c, Jl, lane, failed, failed_jac, entry, cost);
            else
                eval_chunk<false>(D, Q, Z, xy, rec, c, Jl, lane, failed, failed_jac, entry, cost);
            acc += entry;
            cost_acc += cost;
        }
        double *o = D.partials + (size_t)vw * 10;
        if (lane >= 39 && lane < 45) // plane_tri(6 + i, 6 + j): entries 39 .. 44
            o[lane - 39] = acc;
        if (lane >= 51 && lane < 54) // the plane's gradient: entries 45 + 6 ..
            o[6 + lane - 51] = acc;
        if (lane == 0)
            o[9] = cost_acc;
    }
    const int fail_bits = (__ballot(failed) ? 1 : 0) | (__ballot(failed_jac) ? 2 : 0);
    if (fail_bits && lane == 0)
        atomicOr(&D.ctl->fail_bits, fail_bits);
}

__device__ __forceinline__ bool chunk_live(const chain_dev &D, uint32_t chunk)
{
    return D.chunk_first[chunk] < D.blk_cnt[D.chunk_edge[chunk]];
}

// records -> J'J and J'r of the cameras (a wavefront per camera, a thread per camera pair) and this workgroup's share of
// the plane's block, gradient and the cost
__device__ void phase_asm(const chain_dev &D, unsigned char *lds)
{
    const int t = threadIdx.x, lane = t & 63, wv = t >> 6;
    const uint32_t waves = gridDim.x * 4, gw = blockIdx.x * 4 + wv;
    const int set = ld_ctl(&D.ctl->eval_set), state = ld_ctl(&D.ctl->eval_state), cams = ld_ctl(&D.ctl->eval_cams);
    const int nz = ld_ctl(&D.ctl->nz), zt = 3 * ld_ctl(&D.ctl->n_active);
    double *A = D.A[set], *g = D.g[set];
    const double *R = D.rec[set];
    const double *Q = D.q[state];
    if (cams)
    {
        for (uint32_t c = gw; c < D.n_cams; c += waves)
        {
            const int tc = D.cam_t[c];
            if (tc < 0)
                continue;
            double v[18]; // D (6), CZ (9), G (3)
            for (int k = 0; k < 18; k++)
                v[k] = 0;
            for (uint32_t e = D.cam_chunk_off[c] + lane; e < D.cam_chunk_off[c + 1]; e += WV)
            {
                const uint32_t idx = D.cam_chunk_idx[e];
                if (!chunk_live(D, idx >> 1))
                    continue;
                const double *a = R + (size_t)(idx >> 1) * REC;
                const int o = (idx & 1) ? 3 : 0;
                int k = 0;
                for (int i = 0; i < 3; i++)
                    for (int j = i; j < 3; j++)
                        v[k++] += a[plane_tri(o + i, o + j)];
                for (int i = 0; i < 3; i++)
                    for (int j = 0; j < 3; j++)
                        v[6 + i * 3 + j] += a[plane_tri(o + i, 6 + j)];
                for (int i = 0; i < 3; i++)
                    v[15 + i] += a[45 + o + i];
            }
            for (int off = 32; off >= 1; off >>= 1)
                for (int k = 0; k < 18; k++)
                    v[k] += __shfl_xor(v[k], off);
            if (lane != 0)
                continue;
            double *Dd = v, *CZ = v + 6, *G = v + 15;
            if (D.cam_prior[c])
            {
                double r, j3[3];
                downward_prior(Q + (size_t)c * 4, D.prior_weight, &r, j3);
                int k = 0;
                for (int i = 0; i < 3; i++)
                {
                    for (int j = i; j < 3; j++)
                        Dd[k++] += j3[i] * j3[j];
                    G[i] += j3[i] * r;
                }
            }
            int k = 0;
            for (int i = 0; i < 3; i++)
                for (int j = i; j < 3; j++)
                    A[at(tc + j, tc + i)] = Dd[k++];
            for (int i = 0; i < 3; i++)
            {
                g[tc + i] = G[i];
                if (nz)
                    for (int j = 0; j < 3; j++)
                        A[at(zt + j, tc + i)] = CZ[i * 3 + j];
            }
        }
        for (uint32_t pr = blockIdx.x * TG + t; pr < D.n_pairs; pr += gridDim.x * TG)
        {
            const int tp = D.cam_t[D.pair_p[pr]], tq = D.cam_t[D.pair_q[pr]];
            if (tp < 0 || tq < 0)
                continue;
            double s[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
            bool any = false;
            for (uint32_t e = D.pair_chunk_off[pr]; e < D.pair_chunk_off[pr + 1]; e++)
            {
                const uint32_t c = D.pair_chunk_idx[e];
                if (!chunk_live(D, c))
                    continue;
                any = true;
                const double *a = R + (size_t)c * REC;
                for (int i = 0; i < 3; i++)
                    for (int j = 0; j < 3; j++)
                        s[i * 3 + j] += a[plane_tri(i, 3 + j)];
            }
            if (!any)
                continue;
            for (int i = 0; i < 3; i++)
                for (int j = 0; j < 3; j++)
                {
                    if (tp > tq)
                        A[at(tp + i, tq + j)] = s[i * 3 + j];
                    else
                        A[at(tq + j, tp + i)] = s[i * 3 + j];
                }
        }
    }
}

// relax_lm.hip's chol_tiles_kernel on the dense lower triangle: every tile computed once (left-looking), tiles claimed in
// column order, handed over by write-through stores + an agent-scope flag; tile (J + 1, J) goes on to the diagonal tile
// (J + 1, J + 1) in the same workgroup (bit 31 of its claim)
__device__ void phase_chol(const chain_dev &D, unsigned char *lds)
{
    double(*T)[65] = reinterpret_cast<double(*)[65]>(lds);
    double(*Pi)[KC + 1] = reinterpret_cast<double(*)[KC + 1]>(lds + sizeof(double) * 64 * 65);
    double(*Pj)[KC + 1] = Pi + 64;
    double(*colA)[NB] = reinterpret_cast<double(*)[NB]>(reinterpret_cast<unsigned char *>(Pj + 64));
    double(*rowX)[NB] = colA + 4;
    int *s_claim = reinterpret_cast<int *>(rowX + 4), *s_ready = s_claim + 1;
    const int n = ld_ctl(&D.ctl->n), n_claims = ld_ctl(&D.ctl->n_claims), cur = ld_ctl(&D.ctl->cur);
    double *W = D.W;
    const double *A = D.A[cur];
    // entry (r, c) of tile (I, J) of the system the factorisation starts from: S A S + the damping on the diagonal, row n =
    // S g, zero elsewhere (what a build pass wrote into W before round 6's fold: the same expressions)
    auto start_value = [&](int I, int J, int r, int c) -> double {
        const int i = I * NB + r, j = J * NB + c;
        if (j >= n)
            return 0.0;
        if (i < n)
        {
            if (j > i)
                return 0.0;
            double v = A[((size_t)tile_of(I, J) << 12) + (size_t)(r * NB + c)] * D.scale[i] * D.scale[j];
            if (i == j)
                v += D.lm_diag[i];
            return v;
        }
        return i == n ? D.gs[j] : 0.0;
    };
    unsigned int *sync = D.chol_sync, *flags = sync + 4;
    const int t = threadIdx.x, lane = t & 63, w = t >> 6;
    const int wr = (w >> 1) * 32, wc = (w & 1) * 32;
    const int lr = lane & 15, lk = lane >> 4;
    const int n_rows = n + 1;
    for (;;)
    {
        __syncthreads();
        if (t == 0)
            *s_claim = (int)atomicAdd(&sync[0], 1u);
        __syncthreads();
        const int claim = *s_claim;
        if (claim >= n_claims)
            return;
        const unsigned int ij = D.chol_claims[claim];
        const int I = (int)(ij & 0xFFFFu);
        int J = (int)((ij >> 16) & 0x7FFFu);
        const bool fuse = (ij >> 31) != 0u;
        const int r0 = I * 64;
        int c0 = J * 64;
        int nb = min(64, n - c0);
        v4f64 acc[2][2], accd[2][2];
        for (int i = 0; i < 2; i++)
            for (int j = 0; j < 2; j++)
                acc[i][j] = accd[i][j] = v4f64{0, 0, 0, 0};
        double *wt = W + ((size_t)tile_of(I, J) << 12);
        double own[2][2][4];
#pragma unroll
        for (int i = 0; i < 2; i++)
#pragma unroll
            for (int j = 0; j < 2; j++)
#pragma unroll
                for (int e = 0; e < 4; e++)
                {
                    const int r = wr + 16 * i + 4 * e + lk, c = wc + 16 * j + lr;
                    own[i][j][e] = (r0 + r < n_rows && c < nb) ? start_value(I, J, r, c) : 0.0;
                }
        double *wt_d = fuse ? W + ((size_t)tile_of(I, I) << 12) : wt;
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
                        own_d[i][j][e] = (r0 + r < n_rows && c < nb_d) ? start_value(I, I, r, c) : 0.0;
                    }
        }
        int ready = 0;
        bool gave_up = false;
        for (int step = 0; step < J; step++)
        {
            const int K = step;
            if (step >= ready)
            {
                if (t == 0)
                {
                    int upto = step;
                    const unsigned long long t0 = wall_clock64();
                    for (;;)
                    {
                        const unsigned int *fa = flags + tile_of(I, upto);
                        const unsigned int *fb = flags + tile_of(J, upto);
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
                        {
                            __builtin_amdgcn_s_sleep(2);
                            if (wall_clock64() - t0 > SPIN_LIMIT_TICKS)
                            {
                                __hip_atomic_store(&D.bar->abort, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                                upto = -1;
                                break;
                            }
                        }
                    }
                    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    *s_ready = upto;
                }
                __syncthreads();
                ready = *s_ready;
                if (ready < 0)
                {
                    gave_up = true;
                    break;
                }
            }
            const double *wi = W + ((size_t)tile_of(I, K) << 12);
            const double *wj = W + ((size_t)tile_of(J, K) << 12);
            for (int m0 = 0; m0 < 64; m0 += KC)
            {
                __syncthreads();
                for (int e = t; e < 64 * KC; e += 256)
                {
                    const int r = e / KC, m = e % KC;
                    Pi[r][m] = wi[r * NB + m0 + m];
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
                    if (fuse)
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
        if (gave_up)
            return;
        bool second = false;
    finish_tile:
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
            const bool has_aug = (r0 + nb == n) && nb < 64;
            if (has_aug && t < 64)
                Pi[t][KC] = t < nb ? T[nb][t] : 0.0;
            bool bad = false;
            chol_diag_panel_phase<0>(a, x, colA, rowX, ty, tx, nb, bad);
            chol_diag_block_update<0>(a, x, T, Pi, Pj, ty, tx, t);
            chol_diag_panel_phase<1>(a, x, colA, rowX, ty, tx, nb, bad);
            chol_diag_block_update<1>(a, x, T, Pi, Pj, ty, tx, t);
            chol_diag_panel_phase<2>(a, x, colA, rowX, ty, tx, nb, bad);
            chol_diag_block_update<2>(a, x, T, Pi, Pj, ty, tx, t);
            chol_diag_panel_phase<3>(a, x, colA, rowX, ty, tx, nb, bad);
            if (bad)
                D.ctl->chol_fail = 1;
            double *Li = D.linv + (size_t)J * NB * NB;
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
                if (t < nb)
                {
                    double sum = 0;
                    for (int m = 0; m <= t; m++)
                        sum += Pi[m][KC] * T[t][m];
                    store_through(&wt[nb * NB + t], sum);
                }
            }
        }
        else
        {
            if (t == 0)
            {
                const unsigned int *fd = flags + tile_of(J, J);
                const unsigned long long t0 = wall_clock64();
                int ok = 1;
                while (__hip_atomic_load(fd, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0)
                {
                    __builtin_amdgcn_s_sleep(2);
                    if (wall_clock64() - t0 > SPIN_LIMIT_TICKS)
                    {
                        __hip_atomic_store(&D.bar->abort, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        ok = 0;
                        break;
                    }
                }
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                *s_ready = ok;
            }
            __syncthreads();
            if (*s_ready == 0)
                return;
            const double *Li = D.linv + (size_t)J * NB * NB;
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
                    Pj[r][m] = Li[r * NB + m0 + m];
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
                __syncthreads();
                for (int i = 0; i < 2; i++)
                    for (int j = 0; j < 2; j++)
                        for (int e = 0; e < 4; e++)
                        {
                            const int r = wr + 16 * i + 4 * e + lk, c = wc + 16 * j + lr;
                            T[r][c] = (r0 + r < n_rows && c < nb) ? acc[i][j][e] : 0.0;
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
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (t == 0)
            __hip_atomic_store(flags + tile_of(I, J), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
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

// -------------------------------------------------------------------------------------- the serial part (one workgroup)
// Everything below runs in the epilogue of a sync, by the TG threads of the workgroup that arrived last; C is that
// workgroup's copy of the control block in LDS (written back before the sync releases the others).  The parts hand each
// other on through an action code (no recursion: a finished solve may start the next solve, the next step, ... inside the
// same epilogue); every decision is taken from values all threads read alike, so the control flow is uniform.
enum
{
    ACT_NONE = 0,
    ACT_BEGIN_STEP,
    ACT_AFTER_SETUP,
    ACT_BEGIN_SOLVE,
    ACT_FINISH,
    ACT_AFTER_SOLVE,
    ACT_AFTER_EVAL,
    ACT_EVAL_SERIAL,
    ACT_AFTER_ASM,
    ACT_START_ITERATION,
    ACT_AFTER_CHOL,
    ACT_CANDIDATE
};

struct epi
{
    const chain_dev &D;
    chain_ctl &C;
    double *sh; // [TG] reductions
    double *xs; // [N_LIMIT + 1] the step
    int t;
    // arguments of the next action
    int arg_step = 0, arg_which = 0, arg_termination = 0;
    bool arg_count_cost = false;

    __device__ bool optimised(uint32_t c) const
    {
        return C.mode == 0 ? (int)c == C.cam : D.cam_opt1[c] != 0;
    }
    __device__ double reduce_sum(double v)
    {
        __syncthreads();
        sh[t] = v;
        __syncthreads();
        for (int s = TG / 2; s > 0; s >>= 1)
        {
            if (t < s)
                sh[t] += sh[t + s];
            __syncthreads();
        }
        const double r = sh[0];
        __syncthreads();
        return r;
    }
    __device__ double reduce_max(double v)
    {
        __syncthreads();
        sh[t] = v;
        __syncthreads();
        for (int s = TG / 2; s > 0; s >>= 1)
        {
            if (t < s)
                sh[t] = fmax(sh[t], sh[t + s]);
            __syncthreads();
        }
        const double r = sh[0];
        __syncthreads();
        return r;
    }
    __device__ int finish_with(int termination, bool count_cost)
    {
        arg_termination = termination;
        arg_count_cost = count_cost;
        return ACT_FINISH;
    }

    // node.orientation = previous; the step's triangle and start heights
    __device__ int begin_step()
    {
        const int s = arg_step;
        __syncthreads();
        if (s >= (int)D.n_steps)
        {
            if (t == 0)
            {
                C.phase = PH_DONE;
                C.status = CHAIN_FINISHED;
            }
            __syncthreads();
            return ACT_NONE;
        }
        if (t == 0)
        {
            const ochip_plane_chain_step &st = D.steps[s];
            C.step = s;
            C.cam = (int)st.cam;
            C.mode = (int)st.mode;
            C.n_filter = (int)st.n_filter;
            if (st.mode != 2u) // (mode 2: the group's own solve behind the bootstrap, relax.cpp:81-84 - nobody takes an orientation)
            {
                double *q = D.q[C.cur] + 4 * (size_t)st.cam;
                for (int k = 0; k < 4; k++)
                    q[k] = st.prev_cam >= 0 ? D.q[C.cur][4 * (size_t)st.prev_cam + k] : st.prev_q[k];
            }
            for (int k = 0; k < 6; k++)
                C.plane_xy[k] = st.tri_xy[k];
            for (int k = 0; k < 3; k++)
                C.z[C.cur][k] = st.z0;
            C.phase = PH_SETUP;
        }
        __syncthreads();
        if (C.n_filter > 0)
            return ACT_NONE;
        // no edge takes part (the usual step once the graph is more than twice the group: the first edge of the whitelist
        // already touches a camera without an orientation, relax_problem.cpp:251-252): nothing to filter, no blocks - the
        // camera's prior is all there is, and the whole step runs right here, without a phase and without a sync
        const double *Q = D.q[C.cur];
        double priors = 0;
        for (uint32_t c = t; c < D.n_cams; c += TG)
        {
            const bool pr = optimised(c) && !nan4(Q + 4 * (size_t)c);
            D.cam_prior[c] = pr ? 1 : 0;
            D.cam_blocks[c] = 0;
            priors += pr ? 1.0 : 0.0;
        }
        for (uint32_t e = t; e < D.n_edges; e += TG)
            D.blk_cnt[e] = 0;
        const double n_prior = reduce_sum(priors);
        if (t == 0)
        {
            C.n_blocks = 0;
            C.n_live = 0;
            C.n_prior = (int)n_prior;
        }
        __syncthreads();
        arg_which = 0;
        return ACT_BEGIN_SOLVE;
    }

    // an evaluation without residual blocks (the priors alone) of a small system: what PH_EVAL and PH_ASM would leave - the
    // priors' 3 x 3 blocks, gradients and cost, zeros elsewhere - written by this workgroup
    __device__ int eval_serial()
    {
        __syncthreads();
        const int set = C.eval_set, n = C.n;
        double *A = D.A[set], *g = D.g[set];
        const double *Q = D.q[C.eval_state];
        for (int e = t; e < n * n; e += TG)
            if (e % n <= e / n)
                A[at(e / n, e % n)] = 0.0;
        for (int i = t; i < n; i += TG)
            g[i] = 0.0;
        __syncthreads();
        double pc = 0;
        for (uint32_t c = t; c < D.n_cams; c += TG)
        {
            const int tc = D.cam_t[c];
            if (tc < 0 || !D.cam_prior[c])
                continue;
            double r, j3[3];
            downward_prior(Q + (size_t)c * 4, D.prior_weight, &r, j3);
            double Dd[6] = {0, 0, 0, 0, 0, 0}, G[3] = {0, 0, 0}; // (the sums the camera's gather starts from)
            int k = 0;
            for (int i = 0; i < 3; i++)
            {
                for (int j = i; j < 3; j++)
                    Dd[k++] += j3[i] * j3[j];
                G[i] += j3[i] * r;
            }
            k = 0;
            for (int i = 0; i < 3; i++)
                for (int j = i; j < 3; j++)
                    A[at(tc + j, tc + i)] = Dd[k++];
            for (int i = 0; i < 3; i++)
                g[tc + i] = G[i];
            pc += 0.5 * r * r;
        }
        const double prior_cost = reduce_sum(pc);
        if (t == 0)
        {
            for (int q = 0; q < 9; q++)
                C.tot[q] = 0.0;
            C.tot[9] = 0.0 + prior_cost;
        }
        __syncthreads();
        return ACT_AFTER_ASM;
    }

    // after PH_SETUP: block count, priors, which cameras have blocks
    __device__ int after_setup()
    {
        double blocks = 0, bad = 0;
        for (uint32_t e = t; e < D.n_edges; e += TG)
        {
            blocks += (double)D.blk_cnt[e];
            bad += D.inexact[e] ? 1.0 : 0.0;
        }
        const double n_blocks = reduce_sum(blocks), n_bad = reduce_sum(bad);
        if (n_bad > 0)
        {
            if (t == 0)
            {
                C.status = CHAIN_INEXACT;
                C.phase = PH_DONE;
            }
            __syncthreads();
            return ACT_NONE;
        }
        // the chunks that hold blocks, ascending: every thread takes a run of chunks, counts, an exclusive scan places them
        {
            const uint32_t per = (D.n_chunks + TG - 1) / TG, lo = min(D.n_chunks, t * per), hi = min(D.n_chunks, lo + per);
            uint32_t mine = 0;
            for (uint32_t c = lo; c < hi; c++)
                mine += chunk_live(D, c) ? 1u : 0u;
            uint32_t *cnt = reinterpret_cast<uint32_t *>(xs);
            __syncthreads();
            cnt[t] = mine;
            __syncthreads();
            for (int off = 1; off < TG; off <<= 1)
            {
                const uint32_t o = t >= off ? cnt[t - off] : 0u;
                __syncthreads();
                cnt[t] += o;
                __syncthreads();
            }
            uint32_t at_ = cnt[t] - mine;
            const uint32_t total = cnt[TG - 1];
            for (uint32_t c = lo; c < hi; c++)
                if (chunk_live(D, c))
                    D.live[at_++] = c;
            __syncthreads();
            if (t == 0)
                C.n_live = (int)total;
        }
        double priors = 0;
        const double *Q = D.q[C.cur];
        for (uint32_t c = t; c < D.n_cams; c += TG)
        {
            const bool pr = optimised(c) && !nan4(Q + 4 * (size_t)c);
            D.cam_prior[c] = pr ? 1 : 0;
            priors += pr ? 1.0 : 0.0;
            bool has = false;
            for (uint32_t e = D.cam_chunk_off[c]; e < D.cam_chunk_off[c + 1] && !has; e++)
                has = chunk_live(D, D.cam_chunk_idx[e] >> 1);
            D.cam_blocks[c] = has ? 1 : 0;
        }
        const double n_prior = reduce_sum(priors);
        if (t == 0)
        {
            C.n_blocks = (int)n_blocks;
            C.n_prior = (int)n_prior;
        }
        __syncthreads();
        arg_which = 0;
        return ACT_BEGIN_SOLVE;
    }

    __device__ int begin_solve()
    {
        const int which = arg_which;
        __syncthreads();
        if (t == 0)
            C.which = which;
        __syncthreads();
        if (C.n_blocks == 0 && C.n_prior == 0)
            return ACT_AFTER_SOLVE; // RelaxProblem::solve returns before Solve (:1398-1402)
        // tangent offsets of the active cameras, in camera order: a run of cameras per thread, an exclusive scan
        {
            const uint32_t per = (D.n_cams + TG - 1) / TG, lo = min(D.n_cams, t * per), hi = min(D.n_cams, lo + per);
            uint32_t mine = 0;
            for (uint32_t c = lo; c < hi; c++)
                mine += (which == 1 && optimised(c) && (D.cam_blocks[c] || D.cam_prior[c])) ? 1u : 0u;
            uint32_t *cnt = reinterpret_cast<uint32_t *>(xs);
            __syncthreads();
            cnt[t] = mine;
            __syncthreads();
            for (int off = 1; off < TG; off <<= 1)
            {
                const uint32_t o = t >= off ? cnt[t - off] : 0u;
                __syncthreads();
                cnt[t] += o;
                __syncthreads();
            }
            uint32_t k = cnt[t] - mine;
            const uint32_t total = cnt[TG - 1];
            for (uint32_t c = lo; c < hi; c++)
            {
                const bool active = which == 1 && optimised(c) && (D.cam_blocks[c] || D.cam_prior[c]);
                D.cam_t[c] = active ? (int32_t)(3 * k++) : -1;
            }
            __syncthreads();
            if (t == 0)
                C.n_active = (int)total;
        }
        __syncthreads();
        if (t == 0)
        {
            C.solves++;
            C.last_blocks = C.n_blocks + C.n_prior;
            C.nz = C.n_blocks > 0 ? 3 : 0;
            C.n = 3 * C.n_active + C.nz;
            C.nbc = (C.n + NB - 1) / NB;
            C.nbr = (C.n + 1 + NB - 1) / NB;
            C.iterations = 0;
            C.successful = C.unsuccessful = 0;
            C.initial_cost = C.x_cost = 0;
            if (C.n > N_SMALL)
            {
                // claim order of the factorisation: column by column; tile (J + 1, J) takes the diagonal tile behind it along
                int nc = 0;
                for (int J = 0; J < C.nbc; J++)
                    for (int I = J; I < C.nbr; I++)
                    {
                        if (I == J && J > 0)
                            continue;
                        const unsigned int pair = (I == J + 1 && I < C.nbc) ? 0x80000000u : 0u;
                        D.chol_claims[nc++] = (unsigned int)I | ((unsigned int)J << 16) | pair;
                    }
                C.n_claims = nc;
            }
        }
        __syncthreads();
        if (C.n == 0)
            return finish_with(OCHIP_RELAX_NO_PARAMETERS, false);
        if (t == 0)
        {
            C.lm_first = 1;
            C.iter = 0;
            C.invalid = 0;
            C.radius = 1.0; // initial_trust_region_radius, relax_problem.cpp:36
            C.decrease = 2.0;
            C.eval_state = C.cur;
            C.eval_set = C.cur;
            C.eval_cams = which;
            C.fail_bits = 0;
            C.phase = PH_EVAL;
        }
        __syncthreads();
        return (C.n_live == 0 && C.n <= N_SMALL) ? ACT_EVAL_SERIAL : ACT_NONE;
    }

    // p.second->orientation.normalize() (:1410-1413), the solve's figures, on to the next solve or step
    __device__ int finish()
    {
        __syncthreads();
        double *Q = D.q[C.cur];
        for (uint32_t c = t; c < D.n_cams; c += TG)
            if (optimised(c))
            {
                double *q = Q + 4 * (size_t)c;
                const double nrm = sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
                for (int k = 0; k < 4; k++)
                    q[k] = q[k] / nrm;
            }
        if (t == 0)
        {
            C.last_termination = arg_termination;
            C.iterations_total += C.iterations;
            C.last_iterations = C.iterations;
            C.last_initial_cost = C.initial_cost;
            C.last_final_cost = arg_count_cost ? C.x_cost : 0.0;
        }
        __syncthreads();
        return ACT_AFTER_SOLVE;
    }

    __device__ int after_solve()
    {
        __syncthreads();
        if (C.which == 0)
        {
            arg_which = 1;
            return ACT_BEGIN_SOLVE;
        }
        // the step is done: the state is what relax() would have written back
        const double *Q = D.q[C.cur];
        for (uint32_t i = t; i < D.n_cams * 4; i += TG)
            D.q_commit[i] = Q[i];
        const int next = C.step + 1;
        __syncthreads();
        if (t == 0)
            C.steps_done = next;
        __syncthreads();
        arg_step = next;
        return ACT_BEGIN_STEP;
    }

    // after PH_EVAL: the virtual wavefronts' sums in a fixed tree (4 per thread, 16 threads' sums, 16 of those), the priors'
    // cost; without camera derivatives (the plane's heights alone) nothing is left to assemble
    __device__ int after_eval()
    {
        double *s1 = xs, *s2 = xs + TG * 10; // [TG][10], [10][16]
        {
            double v[10];
            for (int q = 0; q < 10; q++)
                v[q] = 0;
            for (int r = 0; r < V_WAVES / TG; r++)
                if (t + r * TG < C.n_live) // (a virtual wavefront without chunks leaves nothing)
                {
                    const double *p = D.partials + (size_t)(t + r * TG) * 10;
                    for (int q = 0; q < 10; q++)
                        v[q] += p[q];
                }
            for (int q = 0; q < 10; q++)
                s1[t * 10 + q] = v[q];
        }
        __syncthreads();
        if (t < 160)
        {
            const int q = t >> 4, part = t & 15;
            double sum = 0;
            for (int k = 0; k < 16; k++)
                sum += s1[(part * 16 + k) * 10 + q];
            s2[q * 16 + part] = sum;
        }
        __syncthreads();
        if (t < 10)
        {
            double sum = 0;
            for (int k = 0; k < 16; k++)
                sum += s2[t * 16 + k];
            sh[t] = sum;
        }
        __syncthreads();
        const double blocks_total[10] = {sh[0], sh[1], sh[2], sh[3], sh[4], sh[5], sh[6], sh[7], sh[8], sh[9]};
        // priors of constant cameras are fixed cost (not in the reduced program)
        double pc = 0;
        const double *Q = D.q[C.eval_state];
        for (uint32_t c = t; c < D.n_cams; c += TG)
            if (D.cam_prior[c] && D.cam_t[c] >= 0)
            {
                double r, j3[3];
                downward_prior(Q + (size_t)c * 4, D.prior_weight, &r, j3);
                pc += 0.5 * r * r;
            }
        const double prior_cost = reduce_sum(pc);
        if (t == 0)
        {
            for (int q = 0; q < 9; q++)
                C.tot[q] = blocks_total[q];
            C.tot[9] = blocks_total[9] + prior_cost;
        }
        __syncthreads();
        if (C.eval_cams)
        {
            if (t == 0)
                C.phase = PH_ASM;
            __syncthreads();
            return ACT_NONE;
        }
        return ACT_AFTER_ASM;
    }

    // after PH_ASM: the groups' partial sums in group order, then the trust-region logic
    __device__ int after_asm()
    {
        const int set = C.eval_set;
        double *A = D.A[set], *g = D.g[set];
        __syncthreads();
        if (t < 10)
            sh[t] = C.tot[t];
        __syncthreads();
        const double cost = sh[9];
        if (t == 0 && C.nz)
        {
            const int zt = 3 * C.n_active;
            int k = 0;
            for (int i = 0; i < 3; i++)
                for (int j = i; j < 3; j++)
                    A[at(zt + j, zt + i)] = sh[k++];
            for (int i = 0; i < 3; i++)
                g[zt + i] = sh[6 + i];
        }
        const int fail_bits = C.fail_bits, n = C.n, first = C.lm_first;
        __syncthreads();
        if (t == 0)
            C.fail_bits = 0;
        if (first)
        {
            if (fail_bits)
                return finish_with(OCHIP_RELAX_FAILURE, false);
            // jacobi scaling, fixed from the first Jacobian; the damping's clamped diagonal; max |g|
            double m = 0;
            for (int i = t; i < n; i += TG)
            {
                const double d = A[at(i, i)];
                const double s = 1.0 / (1.0 + sqrt(d));
                D.scale[i] = s;
                const double v = d * s * s;
                const double lo = v < 1e-6 ? 1e-6 : v;
                D.diagonal[set][i] = 1e32 < lo ? 1e32 : lo;
                m = fmax(m, fabs(g[i]));
            }
            const double gmax = reduce_max(m);
            double xn_part = 0;
            {
                const double *Q = D.q[C.cur];
                for (uint32_t c = t; c < D.n_cams; c += TG)
                    if (D.cam_t[c] >= 0)
                        for (int k = 0; k < 4; k++)
                            xn_part += Q[c * 4 + k] * Q[c * 4 + k];
                if (t < 3 && C.nz)
                    xn_part += C.z[C.cur][t] * C.z[C.cur][t];
            }
            const double xn = reduce_sum(xn_part);
            if (t == 0)
            {
                C.x_norm = sqrt(xn);
                C.x_cost = cost;
                C.initial_cost = cost;
                C.gmax = gmax;
                C.iterations = 1;
                C.lm_first = 0;
            }
            __syncthreads();
            if (gmax <= 1e-10)
                return finish_with(OCHIP_RELAX_CONVERGENCE_GRADIENT, true);
            return ACT_START_ITERATION;
        }
        // the candidate was evaluated with its Jacobian into the other set: clamped diagonal and max |g| of that set
        double m = 0;
        for (int i = t; i < n; i += TG)
        {
            const double d = A[at(i, i)];
            const double v = d * D.scale[i] * D.scale[i];
            const double lo = v < 1e-6 ? 1e-6 : v;
            D.diagonal[set][i] = 1e32 < lo ? 1e32 : lo;
            m = fmax(m, fabs(g[i]));
        }
        const double gmax2 = reduce_max(m);
        const double mcc = C.mcc, x_cost = C.x_cost, x_norm = C.x_norm, radius = C.radius, decrease = C.decrease;
        const double step_norm = sqrt(C.step_norm2), cand_norm = sqrt(C.cand_norm2);
        const int chol_fail = C.chol_fail, invalid = C.invalid;
        __syncthreads();
        const bool valid = !chol_fail && isfinite(mcc) && mcc > 0.0;
        if (!valid)
        {
            if (t == 0)
            {
                C.invalid = invalid + 1;
                C.radius = radius * 0.5;
            }
            __syncthreads();
            if (invalid + 1 >= 5)
                return finish_with(OCHIP_RELAX_FAILURE, true);
            return ACT_START_ITERATION;
        }
        const double cand_cost = (fail_bits & 1) ? 1.7976931348623157e308 : cost;
        if (t == 0)
            C.invalid = 0;
        __syncthreads();
        if (step_norm <= 1e-8 * (x_norm + 1e-8))
            return finish_with(OCHIP_RELAX_CONVERGENCE_PARAMETER, true);
        const double cost_change = x_cost - cand_cost;
        if (fabs(cost_change) <= 1e-6 * x_cost)
            return finish_with(OCHIP_RELAX_CONVERGENCE_FUNCTION, true);
        const double rho = cost_change / mcc;
        if (rho > 1e-3)
        {
            if (t == 0)
            {
                C.cur ^= 1; // the candidate is the state, its set the system
                C.x_norm = cand_norm;
                C.x_cost = cost;
                C.gmax = gmax2;
                const double tt = 2.0 * rho - 1.0;
                C.radius = fmin(1e16, radius / fmax(1.0 / 3.0, 1.0 - tt * tt * tt));
                C.decrease = 2.0;
                C.successful++;
            }
            __syncthreads();
            if (fail_bits & 2)
                return finish_with(OCHIP_RELAX_FAILURE, true);
            if (gmax2 <= 1e-10)
                return finish_with(OCHIP_RELAX_CONVERGENCE_GRADIENT, true);
            return ACT_START_ITERATION;
        }
        if (t == 0)
        {
            C.radius = radius / decrease;
            C.decrease = decrease * 2.0;
            C.unsuccessful++;
        }
        __syncthreads();
        return ACT_START_ITERATION;
    }

    __device__ int start_iteration()
    {
        __syncthreads();
        const int iter = C.iter, n = C.n, cur = C.cur;
        const double radius = C.radius;
        __syncthreads();
        if (iter >= 100) // max_num_iterations, relax_problem.cpp:32
            return finish_with(OCHIP_RELAX_NO_CONVERGENCE, true);
        if (radius <= 1e-32)
            return finish_with(OCHIP_RELAX_CONVERGENCE_RADIUS, true);
        if (t == 0)
        {
            C.iter = iter + 1;
            C.iterations++;
            if (n > N_SMALL)
            {
                C.phase = PH_CHOL;
                C.chol_fail = 0;
            }
        }
        if (n > N_SMALL)
        {
            // what the factorisation starts from is formed tile by tile on its way in (phase_chol: start_value); here the
            // damping D_ii^2 = diagonal_i / radius as LevenbergMarquardtStrategy forms it (sqrt, then squared), the scaled
            // gradient, and the claim counter and tile flags back to zero
            const double *g = D.g[cur], *diagonal = D.diagonal[cur];
            for (int i = t; i < n; i += TG)
            {
                const double dd = sqrt(diagonal[i] / radius);
                D.lm_diag[i] = dd * dd;
                D.gs[i] = g[i] * D.scale[i];
            }
            const int tiles = tile_of(C.nbr, 0) + 4;
            for (int i = t; i < tiles; i += TG)
                D.chol_sync[i] = 0u;
            __syncthreads();
            return ACT_NONE;
        }
        // a small system: scaled, damped, factored and solved by one thread (its entries come in one trip to memory)
        double *sm = xs + N_LIMIT + 1; // [n][n] of A, then g, diagonal, scale
        {
            const double *A = D.A[cur], *g = D.g[cur], *diagonal = D.diagonal[cur];
            for (int e = t; e < n * n; e += TG)
                if (e % n <= e / n)
                    sm[e] = A[at(e / n, e % n)];
            for (int i = t; i < n; i += TG)
            {
                sm[n * n + i] = g[i];
                sm[n * n + n + i] = diagonal[i];
                sm[n * n + 2 * n + i] = D.scale[i];
            }
        }
        __syncthreads();
        if (t == 0)
        {
            const double *g = sm + n * n, *diagonal = g + n, *scale = diagonal + n;
            double Wm[N_SMALL][N_SMALL], gs[N_SMALL], lm[N_SMALL], y[N_SMALL];
            bool bad = false;
            for (int i = 0; i < n; i++)
            {
                for (int j = 0; j <= i; j++)
                    Wm[i][j] = sm[i * n + j] * scale[i] * scale[j];
                const double dd = sqrt(diagonal[i] / radius);
                lm[i] = dd * dd;
                Wm[i][i] += lm[i];
                gs[i] = g[i] * scale[i];
            }
            for (int j = 0; j < n; j++)
            {
                double d = Wm[j][j];
                for (int k = 0; k < j; k++)
                    d -= Wm[j][k] * Wm[j][k];
                if (!(d > 0.0))
                    bad = true;
                const double l = sqrt(d);
                Wm[j][j] = l;
                for (int i = j + 1; i < n; i++)
                {
                    double v = Wm[i][j];
                    for (int k = 0; k < j; k++)
                        v -= Wm[i][k] * Wm[j][k];
                    Wm[i][j] = v / l;
                }
            }
            for (int i = 0; i < n; i++)
            {
                double v = gs[i];
                for (int k = 0; k < i; k++)
                    v -= Wm[i][k] * y[k];
                y[i] = v / Wm[i][i];
            }
            for (int i = n - 1; i >= 0; i--)
            {
                double v = y[i];
                for (int k = i + 1; k < n; k++)
                    v -= Wm[k][i] * y[k];
                y[i] = v / Wm[i][i];
            }
            double part = 0;
            for (int i = 0; i < n; i++)
            {
                xs[i] = y[i];
                part += y[i] * gs[i] + lm[i] * y[i] * y[i];
            }
            C.mcc = 0.5 * part;
            C.chol_fail = bad ? 1 : 0;
        }
        __syncthreads();
        return ACT_CANDIDATE;
    }

    // after PH_CHOL: y = L^-1 gs is the augmented row; L' x = y block by block from the bottom (relax_lm.hip's
    // back_solve_kernel on the dense triangle), then the model's cost change
    __device__ int after_chol()
    {
        const int n = C.n, nbc = C.nbc;
        const double *L = D.W;
        double *xb = sh; // [64]
        __syncthreads();
        for (int i = t; i < n; i += TG)
            xs[i] = L[at(n, i)];
        for (int k = nbc - 1; k >= 0; k--)
        {
            const int k0 = k * NB, nb = min(NB, n - k0);
            const double *Li = D.linv + (size_t)k * NB * NB;
            __syncthreads();
            if (t < NB)
                xb[t] = t < nb ? xs[k0 + t] : 0.0;
            __syncthreads();
            double s = 0;
            if (t < nb)
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
                xs[k0 + t] = s;
            }
            __syncthreads();
            for (int i = t; i < k0; i += TG)
            {
                const double *Lc = L + ((size_t)tile_of(k, i >> 6) << 12) + (i & 63);
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
                xs[i] -= u;
            }
        }
        __syncthreads();
        double part = 0;
        for (int i = t; i < n; i += TG)
            part += xs[i] * D.gs[i] + D.lm_diag[i] * xs[i] * xs[i];
        const double total = reduce_sum(part);
        if (t == 0)
            C.mcc = 0.5 * total;
        __syncthreads();
        return ACT_CANDIDATE;
    }

    // candidate = x (+) (-y scale): relax.hip's plane_candidate_body; then its evaluation, with the Jacobian, into the other set
    __device__ int candidate()
    {
        __syncthreads();
        const int cur = C.cur, zt = 3 * C.n_active, nz = C.nz;
        const double *Q = D.q[cur];
        double *O = D.q[cur ^ 1];
        double sn = 0, xn = 0;
        for (uint32_t c = t; c < D.n_cams; c += TG)
        {
            const int tc = D.cam_t[c];
            const double *q = Q + (size_t)c * 4;
            double *o = O + (size_t)c * 4;
            if (tc < 0)
            {
                for (int k = 0; k < 4; k++)
                    o[k] = q[k];
                continue;
            }
            double d[3];
            for (int k = 0; k < 3; k++)
                d[k] = 1.0 * (-xs[tc + k] * D.scale[tc + k]);
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
        double z1 = 0;
        if (t < 3)
        {
            const double z0 = C.z[cur][t];
            z1 = nz ? z0 + 1.0 * (-xs[zt + t] * D.scale[zt + t]) : z0;
            if (nz)
            {
                sn += (z0 - z1) * (z0 - z1);
                xn += z1 * z1;
            }
        }
        __syncthreads();
        if (t < 3)
            C.z[cur ^ 1][t] = z1;
        const double sn_all = reduce_sum(sn), xn_all = reduce_sum(xn);
        if (t == 0)
        {
            C.step_norm2 = sn_all;
            C.cand_norm2 = xn_all;
            C.eval_state = cur ^ 1;
            C.eval_set = cur ^ 1;
            C.eval_cams = C.which;
            C.phase = PH_EVAL;
        }
        __syncthreads();
        return (C.n_live == 0 && C.n <= N_SMALL) ? ACT_EVAL_SERIAL : ACT_NONE;
    }

    __device__ void advance(int phase)
    {
        int act = ACT_NONE;
        switch (phase)
        {
        case PH_INIT:
            arg_step = 0;
            act = ACT_BEGIN_STEP;
            break;
        case PH_SETUP:
            act = ACT_AFTER_SETUP;
            break;
        case PH_EVAL:
            act = ACT_AFTER_EVAL;
            break;
        case PH_ASM:
            act = ACT_AFTER_ASM;
            break;
        case PH_CHOL:
            act = ACT_AFTER_CHOL;
            break;
        default:
            break;
        }
        while (act != ACT_NONE)
        {
            switch (act)
            {
            case ACT_BEGIN_STEP:
                act = begin_step();
                break;
            case ACT_AFTER_SETUP:
                act = after_setup();
                break;
            case ACT_BEGIN_SOLVE:
                act = begin_solve();
                break;
            case ACT_FINISH:
                act = finish();
                break;
            case ACT_AFTER_SOLVE:
                act = after_solve();
                break;
            case ACT_AFTER_EVAL:
                act = after_eval();
                break;
            case ACT_EVAL_SERIAL:
                act = eval_serial();
                break;
            case ACT_AFTER_ASM:
                act = after_asm();
                break;
            case ACT_START_ITERATION:
                act = start_iteration();
                break;
            case ACT_AFTER_CHOL:
                act = after_chol();
                break;
            case ACT_CANDIDATE:
                act = candidate();
                break;
            default:
                act = ACT_NONE;
                break;
            }
        }
        __syncthreads();
    }
};

__global__ __launch_bounds__(TG) void plane_chain_kernel(chain_dev D, int stepped)
{
    extern __shared__ __align__(16) unsigned char lds[];
    __shared__ chain_ctl s_ctl;
    __shared__ int s_flag;
    __shared__ double s_sh[TG];
    unsigned int gen = __hip_atomic_load(&D.bar->gen, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    for (;;)
    {
        const int phase = ld_ctl(&D.ctl->phase);
        if (phase == PH_DONE)
            return;
        switch (phase)
        {
        case PH_INIT:
            phase_init(D);
            break;
        case PH_SETUP:
            phase_setup(D, lds);
            break;
        case PH_EVAL:
            phase_eval(D, lds);
            break;
        case PH_ASM:
            phase_asm(D, lds);
            break;
        case PH_CHOL:
            phase_chol(D, lds);
            break;
        default:
            break;
        }
        const bool ok = grid_sync(D, gen, &s_flag, [&]() {
            // the control block through LDS: the serial logic reads and writes it hundreds of times
            int *src = reinterpret_cast<int *>(D.ctl), *dst = reinterpret_cast<int *>(&s_ctl);
            for (int i = threadIdx.x; i < (int)(sizeof(chain_ctl) / 4); i += TG)
                dst[i] = __hip_atomic_load(src + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __syncthreads();
            const unsigned long long t_in = wall_clock64();
            if (threadIdx.x == 0)
            {
                s_ctl.syncs++;
                if (s_ctl.t_last)
                    s_ctl.t_phase[phase] += t_in - s_ctl.t_last;
                s_ctl.n_phase[phase]++;
            }
            __syncthreads();
            epi E{D, s_ctl, s_sh, reinterpret_cast<double *>(lds), (int)threadIdx.x};
            E.advance(phase);
            __syncthreads();
            if (threadIdx.x == 0)
            {
                const unsigned long long t_out = wall_clock64();
                s_ctl.t_epilogue[phase] += t_out - t_in;
                s_ctl.t_last = t_out;
            }
            __syncthreads();
            for (int i = threadIdx.x; i < (int)(sizeof(chain_ctl) / 4); i += TG)
                src[i] = dst[i];
        });
        if (!ok || stepped)
            return;
    }
}

std::mutex g_chain_mutex; // one chain at a time per process: two resident grids can starve each other of compute units

} // namespace

// ------------------------------------------------------------------------------------------------------------ the C ABI
struct ochip_plane_chain
{
    ochip_ctx *ctx = nullptr;
    chain_dev dev{};
    std::vector<std::pair<void *, size_t>> allocs;
    unsigned int grid = 1;
    uint32_t n_cams = 0, n_steps = 0;
    chain_ctl *ctl_host = nullptr; // page-locked: the control block, then the abort flag (16 bytes), then the state [n_cams][4]
    unsigned int *abort_host() const
    {
        return reinterpret_cast<unsigned int *>(ctl_host + 1);
    }
    double *q_host() const
    {
        return reinterpret_cast<double *>(reinterpret_cast<unsigned char *>(ctl_host + 1) + 16);
    }
    hipError_t read_back(hipStream_t st, bool with_state) const // enqueue; the caller waits
    {
        hipError_t e = hipMemcpyAsync(ctl_host, dev.ctl, sizeof(chain_ctl), hipMemcpyDeviceToHost, st);
        if (e == hipSuccess)
            e = hipMemcpyAsync(abort_host(), &dev.bar->abort, 4, hipMemcpyDeviceToHost, st);
        if (e == hipSuccess && with_state)
            e = hipMemcpyAsync(q_host(), dev.q_commit, (size_t)n_cams * 32, hipMemcpyDeviceToHost, st);
        return e;
    }
};

extern "C"
{

void ochip_plane_chain_destroy(ochip_plane_chain *c)
{
    if (!c)
        return;
    (void)hipSetDevice(c->ctx->device);
    (void)ochip_stream_wait(c->ctx, c->ctx->stream);
    for (auto &a : c->allocs)
        ochip_pool_put(c->ctx, a.first, a.second);
    if (c->ctl_host)
        ochip_host_free(c->ctx, c->ctl_host);
    delete c;
}

int ochip_plane_chain_create(ochip_ctx *ctx, const ochip_plane_edge *edges, uint32_t n_edges, const ochip_plane_inlier *inliers,
                             uint64_t n_inliers, const double *cam_pos, const double *cam_q, const uint8_t *cam_optimize, uint32_t n_cams,
                             const double *models10, uint32_t n_models, double grid_fraction, double huber_a, double prior_weight,
                             const ochip_plane_chain_step *steps, uint32_t n_steps, ochip_plane_chain **out)
{
    if (!ctx || !out)
        return OCHIP_EINVAL;
    *out = nullptr;
    if ((n_edges && !edges) || (n_inliers && !inliers) || !n_cams || !cam_pos || !cam_q || !cam_optimize || (n_models && !models10) ||
        (n_steps && !steps) || !(grid_fraction > 0))
        return ochip_fail(ctx, OCHIP_EINVAL, "ochip_plane_chain_create: bad argument");
    if (!(1.0 / grid_fraction < (double)(G_MAX - 1)))
        return ochip_fail(ctx, OCHIP_EINVAL, "ochip_plane_chain_create: a grid of %g per cell is finer than the device's cell tables", grid_fraction);
    uint32_t n_opt = 0;
    for (uint32_t c = 0; c < n_cams; c++)
        n_opt += cam_optimize[c] ? 1 : 0;
    const uint32_t n_max = 3 * std::max(n_opt, 1u) + 3;
    if (n_max > (uint32_t)N_LIMIT)
        return ochip_fail(ctx, OCHIP_EINVAL, "ochip_plane_chain_create: %u optimised cameras are more than the chain's dense system takes (%d unknowns)",
                          n_opt, N_LIMIT);
    for (uint32_t e = 0; e < n_edges; e++)
        if (edges[e].cam_a >= n_cams || edges[e].cam_b >= n_cams || edges[e].cam_a == edges[e].cam_b || edges[e].model_a >= n_models ||
            edges[e].model_b >= n_models || edges[e].inlier_offset + edges[e].n_inliers > n_inliers)
            return ochip_fail(ctx, OCHIP_EINVAL, "ochip_plane_chain_create: edge %u is out of range (or links a camera to itself)", e);
    for (uint32_t s = 0; s < n_steps; s++)
        if (steps[s].cam >= n_cams || steps[s].mode > 2u || steps[s].n_filter > n_edges || steps[s].prev_cam >= (int32_t)n_cams)
            return ochip_fail(ctx, OCHIP_EINVAL, "ochip_plane_chain_create: step %u is out of range", s);
    OCHIP_HIP(ctx, hipSetDevice(ctx->device));
    hipStream_t st = ctx->stream;

    // ---- the static tables: block slots, chunks, camera -> chunks, camera pairs -> chunks
    const uint32_t cells = (uint32_t)std::ceil(1.0 / grid_fraction);
    const uint32_t cap_edge = 2 * cells * cells; // a kept match holds the best score of a cell of one of the two images
    std::vector<uint32_t> blk_off(n_edges + 1, 0), chunk_edge, chunk_first;
    for (uint32_t e = 0; e < n_edges; e++)
    {
        const uint32_t cap = std::min(edges[e].n_inliers, cap_edge);
        blk_off[e + 1] = blk_off[e] + cap;
        for (uint32_t f = 0; f < cap; f += WV)
        {
            chunk_edge.push_back(e);
            chunk_first.push_back(f);
        }
    }
    const uint32_t n_chunks = (uint32_t)chunk_edge.size();
    std::vector<uint32_t> cam_off(n_cams + 1, 0), cam_idx(2 * (size_t)n_chunks);
    for (uint32_t c = 0; c < n_chunks; c++)
    {
        cam_off[edges[chunk_edge[c]].cam_a + 1]++;
        cam_off[edges[chunk_edge[c]].cam_b + 1]++;
    }
    for (uint32_t c = 0; c < n_cams; c++)
        cam_off[c + 1] += cam_off[c];
    {
        std::vector<uint32_t> fill(cam_off.begin(), cam_off.end() - 1);
        for (uint32_t c = 0; c < n_chunks; c++)
        {
            const ochip_plane_edge &ed = edges[chunk_edge[c]];
            const uint32_t p = std::min(ed.cam_a, ed.cam_b), q = std::max(ed.cam_a, ed.cam_b);
            cam_idx[fill[p]++] = c * 2;
            cam_idx[fill[q]++] = c * 2 + 1;
        }
    }
    std::map<uint64_t, std::vector<uint32_t>> pairs; // (p << 32 | q) -> chunks, in chunk order
    for (uint32_t c = 0; c < n_chunks; c++)
    {
        const ochip_plane_edge &ed = edges[chunk_edge[c]];
        const uint64_t p = std::min(ed.cam_a, ed.cam_b), q = std::max(ed.cam_a, ed.cam_b);
        pairs[(p << 32) | q].push_back(c);
    }
    std::vector<uint32_t> pair_p, pair_q, pair_off{0}, pair_idx;
    for (const auto &kv : pairs)
    {
        pair_p.push_back((uint32_t)(kv.first >> 32));
        pair_q.push_back((uint32_t)(kv.first & 0xFFFFFFFFu));
        pair_idx.insert(pair_idx.end(), kv.second.begin(), kv.second.end());
        pair_off.push_back((uint32_t)pair_idx.size());
    }

    auto *c = new (std::nothrow) ochip_plane_chain();
    if (!c)
        return ochip_fail(ctx, OCHIP_ENOMEM, "host allocation failed");
    c->ctx = ctx;
    c->n_cams = n_cams;
    c->n_steps = n_steps;
    int rc = OCHIP_OK;
    auto dev = [&](size_t bytes) -> void * {
        size_t got = 0;
        void *p = ochip_pool_get(ctx, bytes ? bytes : 16, &got);
        if (!p)
        {
            if (rc == OCHIP_OK)
                rc = ochip_fail(ctx, OCHIP_ENOMEM, "ochip_plane_chain_create: device allocation of %zu bytes failed", bytes);
            return nullptr;
        }
        c->allocs.emplace_back(p, got);
        return p;
    };
    // every input in ONE page-locked block, one copy
    struct piece
    {
        const void *src;
        size_t bytes;
        void **dst;
    };
    chain_dev &D = c->dev;
    std::vector<piece> pieces = {
        {edges, (size_t)n_edges * sizeof(ochip_plane_edge), (void **)&D.edges},
        {inliers, (size_t)n_inliers * sizeof(ochip_plane_inlier), (void **)&D.inliers},
        {cam_pos, (size_t)n_cams * 24, (void **)&D.cam_pos},
        {cam_q, (size_t)n_cams * 32, (void **)&D.q[0]},
        {cam_optimize, (size_t)n_cams, (void **)&D.cam_opt1},
        {models10, (size_t)n_models * 80, (void **)&D.models},
        {blk_off.data(), blk_off.size() * 4, (void **)&D.blk_off},
        {chunk_edge.data(), chunk_edge.size() * 4, (void **)&D.chunk_edge},
        {chunk_first.data(), chunk_first.size() * 4, (void **)&D.chunk_first},
        {cam_off.data(), cam_off.size() * 4, (void **)&D.cam_chunk_off},
        {cam_idx.data(), cam_idx.size() * 4, (void **)&D.cam_chunk_idx},
        {pair_p.data(), pair_p.size() * 4, (void **)&D.pair_p},
        {pair_q.data(), pair_q.size() * 4, (void **)&D.pair_q},
        {pair_off.data(), pair_off.size() * 4, (void **)&D.pair_chunk_off},
        {pair_idx.data(), pair_idx.size() * 4, (void **)&D.pair_chunk_idx},
        {steps, (size_t)n_steps * sizeof(ochip_plane_chain_step), (void **)&D.steps},
    };
    size_t total = 0;
    std::vector<size_t> offs;
    for (const piece &p : pieces)
    {
        offs.push_back(total);
        total += (p.bytes + 255) / 256 * 256;
    }
    void *staging = nullptr;
    unsigned char *blob = (unsigned char *)dev(total);
    if (rc == OCHIP_OK && ochip_host_alloc(ctx, std::max<size_t>(total, 256), &staging) != OCHIP_OK)
        rc = OCHIP_ENOMEM;
    if (rc == OCHIP_OK)
    {
        for (size_t i = 0; i < pieces.size(); i++)
        {
            if (pieces[i].bytes)
                std::memcpy((unsigned char *)staging + offs[i], pieces[i].src, pieces[i].bytes);
            *pieces[i].dst = blob + offs[i];
        }
        if (hipMemcpyAsync(blob, staging, total, hipMemcpyHostToDevice, st) != hipSuccess)
            rc = ochip_fail(ctx, OCHIP_EHIP, "ochip_plane_chain_create: upload failed");
    }
    const int nbr_max = (int)(n_max + 1 + NB - 1) / NB;
    const size_t tiles = (size_t)nbr_max * (nbr_max + 1) / 2, tile_doubles = tiles << 12;
    D.n_cams = n_cams;
    D.n_edges = n_edges;
    D.n_chunks = n_chunks;
    D.n_pairs = (uint32_t)pair_p.size();
    D.n_steps = n_steps;
    D.n_max = n_max;
    D.n_inliers = n_inliers;
    D.res = grid_fraction;
    D.huber_a = huber_a;
    D.prior_weight = prior_weight;
    D.q[1] = (double *)dev((size_t)n_cams * 32);
    D.q_commit = (double *)dev((size_t)n_cams * 32);
    D.rays = (double *)dev((size_t)n_inliers * 48);
    D.score = (double *)dev((size_t)n_inliers * 8);
    D.keep = (uint8_t *)dev((size_t)n_inliers);
    D.blk_idx = (uint32_t *)dev((size_t)blk_off[n_edges] * 4);
    D.blk_cnt = (uint32_t *)dev((size_t)n_edges * 4);
    D.inexact = (uint8_t *)dev((size_t)n_edges);
    D.rec[0] = (double *)dev((size_t)n_chunks * REC * 8);
    D.rec[1] = (double *)dev((size_t)n_chunks * REC * 8);
    D.cam_prior = (uint8_t *)dev(n_cams);
    D.cam_blocks = (uint8_t *)dev(n_cams);
    D.cam_t = (int32_t *)dev((size_t)n_cams * 4);
    for (int s = 0; s < 2; s++)
    {
        D.A[s] = (double *)dev(tile_doubles * 8);
        D.g[s] = (double *)dev((size_t)(n_max + 1) * 8);
        D.diagonal[s] = (double *)dev((size_t)(n_max + 1) * 8);
    }
    D.W = (double *)dev(tile_doubles * 8);
    D.linv = (double *)dev((size_t)nbr_max * NB * NB * 8);
    D.scale = (double *)dev((size_t)(n_max + 1) * 8);
    D.gs = (double *)dev((size_t)(n_max + 1) * 8);
    D.lm_diag = (double *)dev((size_t)(n_max + 1) * 8);
    D.y = (double *)dev((size_t)(n_max + 1) * 8);
    D.chol_sync = (unsigned int *)dev((tiles + 4) * 4);
    D.chol_claims = (unsigned int *)dev((tiles + 4) * 4);
    // the grid: what the evaluation can use (a wavefront per chunk), at most half of the compute units (the other half, and
    // the other seven wave slots of every SIMD, stay with the extraction and the link stage of the batches behind this
    // one); OCHIP_CHAIN_WORKGROUPS overrides (processes that share a device in the tests)
    int per_cu = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, plane_chain_kernel, TG, LDS_EVAL) != hipSuccess || per_cu < 1)
        per_cu = 1;
    unsigned int grid = (unsigned int)std::max(1, ctx->prop.multiProcessorCount / 2);
    if (const char *env = std::getenv("OCHIP_CHAIN_WORKGROUPS"))
        grid = (unsigned int)std::max(1, std::min(std::atoi(env), ctx->prop.multiProcessorCount * per_cu));
    {
        // a wavefront per chunk of the edges that ever take part: a batch whose cameras only have their priors (the usual
        // case once the graph is more than twice the group, relax.cpp:61-68) runs on ONE workgroup
        uint32_t most_edges = 0, used_chunks = 0;
        for (uint32_t s = 0; s < n_steps; s++)
            most_edges = std::max(most_edges, steps[s].n_filter);
        for (uint32_t c = 0; c < n_chunks; c++)
            used_chunks += chunk_edge[c] < most_edges ? 1u : 0u;
        grid = std::min(grid, std::max(1u, (used_chunks + 3) / 4));
    }
    c->grid = grid;
    D.partials = (double *)dev((size_t)V_WAVES * 10 * 8);
    D.live = (uint32_t *)dev((size_t)n_chunks * 4);
    D.ctl = (chain_ctl *)dev(sizeof(chain_ctl));
    D.bar = (chain_bar *)dev(sizeof(chain_bar));
    if (rc == OCHIP_OK && ochip_host_alloc(ctx, sizeof(chain_ctl) + 16 + (size_t)n_cams * 32, (void **)&c->ctl_host) != OCHIP_OK)
        rc = OCHIP_ENOMEM;
    if (rc == OCHIP_OK)
    {
        chain_ctl init{};
        init.phase = PH_INIT;
        init.status = CHAIN_RUNNING;
        *c->ctl_host = init;
        hipError_t e = hipMemcpyAsync(D.ctl, c->ctl_host, sizeof(chain_ctl), hipMemcpyHostToDevice, st);
        if (e == hipSuccess)
            e = hipMemsetAsync(D.bar, 0, sizeof(chain_bar), st);
        if (e == hipSuccess) // (rows of a last, partial row block that no factorisation writes are read as operands: finite, once)
            e = hipMemsetAsync(D.W, 0, tile_doubles * 8, st);
        if (e == hipSuccess)
            e = hipMemcpyAsync(D.q[1], D.q[0], (size_t)n_cams * 32, hipMemcpyDeviceToDevice, st);
        if (e == hipSuccess)
            e = hipMemcpyAsync(D.q_commit, D.q[0], (size_t)n_cams * 32, hipMemcpyDeviceToDevice, st);
        if (e == hipSuccess)
            e = hipFuncSetAttribute(reinterpret_cast<const void *>(plane_chain_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS_EVAL);
        if (e == hipSuccess)
            e = ochip_stream_wait(ctx, st); // (the staging block goes back to the pool)
        if (e != hipSuccess)
            rc = ochip_fail(ctx, OCHIP_EHIP, "ochip_plane_chain_create: %s", hipGetErrorString(e));
    }
    if (staging)
        ochip_host_free(ctx, staging);
    if (rc != OCHIP_OK)
    {
        ochip_plane_chain_destroy(c);
        return rc;
    }
    *out = c;
    return OCHIP_OK;
}

int ochip_plane_chain_run(ochip_plane_chain *c, int stepped, double *cam_q_out, ochip_plane_chain_result *result)
{
    if (!c || !cam_q_out || !result)
        return OCHIP_EINVAL;
    ochip_ctx *ctx = c->ctx;
    OCHIP_HIP(ctx, hipSetDevice(ctx->device));
    hipStream_t st = ctx->stream;
    std::memset(result, 0, sizeof *result);
    {
        std::lock_guard<std::mutex> lk(g_chain_mutex);
        if (!stepped)
        {
            hipLaunchKernelGGL(plane_chain_kernel, dim3(c->grid), dim3(TG), LDS_EVAL, st, c->dev, 0);
            OCHIP_HIP(ctx, hipGetLastError());
            OCHIP_HIP(ctx, c->read_back(st, true));
            OCHIP_HIP(ctx, ochip_stream_wait(ctx, st));
        }
        else
        {
            // the test route: one phase per launch, the control block read back in between
            for (long launches = 0;; launches++)
            {
                hipLaunchKernelGGL(plane_chain_kernel, dim3(c->grid), dim3(TG), LDS_EVAL, st, c->dev, 1);
                OCHIP_HIP(ctx, hipGetLastError());
                OCHIP_HIP(ctx, c->read_back(st, false));
                OCHIP_HIP(ctx, ochip_stream_wait(ctx, st));
                if (c->ctl_host->phase == PH_DONE || *c->abort_host() != 0u)
                    break;
                if (launches > 50000000)
                    return ochip_fail(ctx, OCHIP_EHIP, "ochip_plane_chain_run: the stepped chain does not end");
            }
            OCHIP_HIP(ctx, c->read_back(st, true));
            OCHIP_HIP(ctx, ochip_stream_wait(ctx, st));
        }
        if (*c->abort_host() != 0u)
            c->ctl_host->status = CHAIN_ABORTED;
    }
    const chain_ctl &C = *c->ctl_host;
    std::memcpy(cam_q_out, c->q_host(), (size_t)c->n_cams * 32);
    result->steps_done = (uint32_t)C.steps_done;
    result->status = C.status == CHAIN_FINISHED ? 0 : (C.status == CHAIN_INEXACT ? 1 : 2);
    result->solves = C.solves;
    result->iterations_total = C.iterations_total;
    result->last_iterations = C.last_iterations;
    result->last_residual_blocks = C.last_blocks;
    result->last_initial_cost = C.last_initial_cost;
    result->last_final_cost = C.last_final_cost;
    result->grid_syncs = C.syncs;
    result->workgroups = c->grid;
    for (int i = 0; i < 3; i++)
        result->plane_z[i] = C.z[C.cur][i];
    if (ochip_verbose("relax"))
    {
        static const char *names[7] = {"rays", "set-up", "evaluate", "assemble", "factor", "-", "-"};
        for (int p = 0; p < PH_DONE; p++)
            if (C.n_phase[p])
                fprintf(stderr, "[relax chain]   %-9s %7llu times, parallel part + sync %9.3f ms (%6.1f us each), serial part behind it %9.3f ms (%6.1f us each)\n",
                        names[p], C.n_phase[p], C.t_phase[p] * 1e-5, C.t_phase[p] * 1e-2 / (double)C.n_phase[p], C.t_epilogue[p] * 1e-5,
                        C.t_epilogue[p] * 1e-2 / (double)C.n_phase[p]);
    }
    if (C.status == CHAIN_RUNNING)
        return ochip_fail(ctx, OCHIP_EHIP, "ochip_plane_chain_run: the chain left its launch unfinished (phase %d, step %d)", C.phase, C.step);
    return OCHIP_OK;
}

} // extern "C"
