// ORACLE — test infrastructure only (see oracle.hpp).
// Flat C entry points so tests/ (ctypes) and bench.py's cpu_baseline leg can drive the restatement.
#include "oracle.hpp"
#include "akaze.hpp"

#include <algorithm>
#include <chrono>
#include <cstring>
#include <numeric>
#include <random>

#ifdef _OPENMP
#include <omp.h>
#endif

using namespace oracle;

static std::vector<feature_2d> make_features(const double *loc, const float *strength, const uint64_t *desc, size_t n)
{
    std::vector<feature_2d> f(n);
    for (size_t i = 0; i < n; i++)
    {
        if (loc)
        {
            f[i].location[0] = loc[2 * i];
            f[i].location[1] = loc[2 * i + 1];
        }
        if (strength)
            f[i].strength = strength[i];
        if (desc)
        {
            // std::bitset<486> on LP64 libstdc++ is 8 little-endian 64-bit words
            static_assert(sizeof(f[i].descriptor) == 64, "bitset<486> layout");
            uint64_t w[8];
            std::memcpy(w, desc + 8 * i, 64);
            w[7] &= (uint64_t(1) << (486 - 448)) - 1; // bits 486..511 are always zero in a bitset<486>
            std::memcpy((void *)&f[i].descriptor, w, 64);
        }
    }
    return f;
}

static camera_model make_model(const double *m) // f, ppx, ppy, k1, k2, k3, p1, p2, cols, rows
{
    camera_model c;
    c.focal_length_pixels = m[0];
    c.principle_point[0] = m[1];
    c.principle_point[1] = m[2];
    c.radial_distortion[0] = m[3];
    c.radial_distortion[1] = m[4];
    c.radial_distortion[2] = m[5];
    c.tangential_distortion[0] = m[6];
    c.tangential_distortion[1] = m[7];
    c.pixels_cols = (size_t)m[8];
    c.pixels_rows = (size_t)m[9];
    return c;
}

static std::vector<correspondence> make_corrs(const double *corr, size_t M)
{
    std::vector<correspondence> c(M);
    for (size_t i = 0; i < M; i++)
    {
        c[i].measurement1 = Vec3{corr[7 * i], corr[7 * i + 1], corr[7 * i + 2]};
        c[i].measurement2 = Vec3{corr[7 * i + 3], corr[7 * i + 4], corr[7 * i + 5]};
        c[i].quality = corr[7 * i + 6];
    }
    return c;
}

extern "C"
{

// libstdc++ behaviour the reference depends on (SURVEY.md §8c golden values)
void oc_libstdcxx_selfcheck(uint64_t *out /*17*/)
{
    std::default_random_engine g(42);
    out[0] = g();
    std::default_random_engine g2(42);
    std::vector<size_t> v(10);
    std::iota(v.begin(), v.end(), 0);
    std::shuffle(v.begin(), v.end(), g2);
    for (int i = 0; i < 10; i++)
        out[1 + i] = v[i];
    std::uniform_int_distribution<size_t> d(0, 9);
    for (int i = 0; i < 6; i++)
        out[11 + i] = d(g2);
}

size_t oc_subsample(const double *loc, const float *strength, size_t n, double spacing, size_t count, uint64_t *out)
{
    auto f = make_features(loc, strength, nullptr, n);
    auto idx = spatially_subsample_feature_indices(f, spacing, count);
    for (size_t i = 0; i < idx.size(); i++)
        out[i] = idx[i];
    return idx.size();
}

size_t oc_match(const uint64_t *desc1, size_t n1, const uint64_t *desc2, size_t n2, const uint64_t *idx1, size_t n_idx1,
                const uint64_t *idx2, size_t n_idx2, uint64_t *out_i1, uint64_t *out_i2, double *out_dist)
{
    auto f1 = make_features(nullptr, nullptr, desc1, n1);
    auto f2 = make_features(nullptr, nullptr, desc2, n2);
    std::vector<size_t> i1(idx1, idx1 + n_idx1), i2(idx2, idx2 + n_idx2);
    auto m = match_features_subset(f1, f2, i1, i2);
    for (size_t i = 0; i < m.size(); i++)
    {
        out_i1[i] = m[i].feature_index_1;
        out_i2[i] = m[i].feature_index_2;
        out_dist[i] = m[i].distance;
    }
    return m.size();
}

void oc_image_to_3d(const double *px, size_t n, const double *model, double *rays)
{
    const camera_model c = make_model(model);
    for (size_t i = 0; i < n; i++)
    {
        const Vec3 r = image_to_3d(px + 2 * i, c);
        rays[3 * i] = r.x;
        rays[3 * i + 1] = r.y;
        rays[3 * i + 2] = r.z;
    }
}

void oc_image_from_3d(const double *rays, size_t n, const double *model, double *px)
{
    const camera_model c = make_model(model);
    for (size_t i = 0; i < n; i++)
    {
        const Vec2 p = image_from_3d(Vec3{rays[3 * i], rays[3 * i + 1], rays[3 * i + 2]}, c);
        px[2 * i] = p.x;
        px[2 * i + 1] = p.y;
    }
}

void oc_homography_fit4(const double *corr, size_t M, const uint64_t *idx4, double *H, double *Hinv)
{
    auto c = make_corrs(corr, M);
    homography_model h;
    h.fit(c, {idx4[0], idx4[1], idx4[2], idx4[3]});
    std::memcpy(H, h.homography.m, 72);
    std::memcpy(Hinv, h.homography_inverse.m, 72);
}

void oc_homography_fit_inliers(const double *corr, size_t M, const uint8_t *inl, double *H, double *Hinv)
{
    auto c = make_corrs(corr, M);
    std::vector<bool> in(M);
    for (size_t i = 0; i < M; i++)
        in[i] = inl[i] != 0;
    homography_model h;
    h.fitInliers(c, in);
    std::memcpy(H, h.homography.m, 72);
    std::memcpy(Hinv, h.homography_inverse.m, 72);
}

double oc_homography_evaluate(const double *corr, size_t M, const double *H, const double *Hinv, uint8_t *inl,
                              double *errors)
{
    auto c = make_corrs(corr, M);
    homography_model h;
    std::memcpy(h.homography.m, H, 72);
    std::memcpy(h.homography_inverse.m, Hinv, 72);
    std::vector<bool> in;
    const double s = h.evaluate(c, in);
    for (size_t i = 0; i < M; i++)
    {
        inl[i] = in[i];
        if (errors)
            errors[i] = h.error(c[i]);
    }
    return s;
}

// returns score; H (9, row-major), inliers (M); optional trace of the minimal samples
double oc_ransac_homography(const double *corr, size_t M, double *H, uint8_t *inl, uint64_t *trace_samples,
                            size_t max_trace, uint64_t *iterations)
{
    auto c = make_corrs(corr, M);
    homography_model h;
    std::vector<bool> in;
    ransac_trace tr;
    const double s = ransac(c, h, in, &tr);
    std::memcpy(H, h.homography.m, 72);
    for (size_t i = 0; i < in.size(); i++)
        inl[i] = in[i];
    if (iterations)
    {
        iterations[0] = tr.iterations;
        iterations[1] = tr.improvements;
    }
    if (trace_samples)
        for (size_t i = 0; i < tr.samples.size() && i < max_trace; i++)
            for (int j = 0; j < 4; j++)
                trace_samples[4 * i + j] = tr.samples[i][j];
    return s;
}

// poses: 4 x {qx,qy,qz,qw, tx,ty,tz, score}
int oc_homography_decompose(const double *H, const double *corr, size_t M, const uint8_t *inl, double *poses)
{
    auto c = make_corrs(corr, M);
    std::vector<bool> in(M);
    for (size_t i = 0; i < M; i++)
        in[i] = inl[i] != 0;
    homography_model h;
    std::memcpy(h.homography.m, H, 72);
    h.homography_inverse = inverse3(h.homography);
    std::array<decomposed_pose, 4> p;
    const bool ok = h.decompose(c, in, p);
    for (int i = 0; i < 4; i++)
    {
        double *o = poses + 8 * i;
        o[0] = p[i].orientation.x;
        o[1] = p[i].orientation.y;
        o[2] = p[i].orientation.z;
        o[3] = p[i].orientation.w;
        o[4] = p[i].position.x;
        o[5] = p[i].position.y;
        o[6] = p[i].position.z;
        o[7] = p[i].score;
    }
    return ok ? 1 : 0;
}

// One directed pair of link_stage.cpp:75-112.  Output buffers sized for n_idx1 matches.
// summary: [n_matches, n_inliers, can_decompose, accepted(edge has inlier list), ransac_score, ransac iterations,
//           ransac improvements]
void oc_link_pair(const double *loc1, const uint64_t *desc1, size_t n1, const uint64_t *idx1, size_t n_idx1,
                  const double *loc2, const uint64_t *desc2, size_t n2, const uint64_t *idx2, size_t n_idx2,
                  const double *model1, const double *model2, uint64_t *m_i1, uint64_t *m_i2, double *m_dist,
                  uint8_t *inl, double *H, double *poses, double *summary)
{
    auto f1 = make_features(loc1, nullptr, desc1, n1);
    auto f2 = make_features(loc2, nullptr, desc2, n2);
    std::vector<size_t> i1(idx1, idx1 + n_idx1), i2(idx2, idx2 + n_idx2);
    camera_relations r = link_pair(f1, f2, i1, i2, make_model(model1), make_model(model2));
    // matches are only kept in relations when accepted; re-run match for the full list (deterministic)
    auto m = match_features_subset(f1, f2, i1, i2);
    for (size_t i = 0; i < m.size(); i++)
    {
        m_i1[i] = m[i].feature_index_1;
        m_i2[i] = m[i].feature_index_2;
        m_dist[i] = m[i].distance;
        inl[i] = r.coarse_inliers[i];
    }
    std::memcpy(H, r.ransac_relation.m, 72);
    for (int i = 0; i < 4; i++)
    {
        double *o = poses + 8 * i;
        const auto &p = r.relative_poses[i];
        o[0] = p.orientation.x, o[1] = p.orientation.y, o[2] = p.orientation.z, o[3] = p.orientation.w;
        o[4] = p.position.x, o[5] = p.position.y, o[6] = p.position.z, o[7] = p.score;
    }
    summary[0] = (double)m.size();
    summary[1] = (double)std::count(r.coarse_inliers.begin(), r.coarse_inliers.end(), true);
    summary[2] = r.can_decompose;
    summary[3] = !r.inlier_matches.empty();
    summary[4] = r.ransac_score;
    summary[5] = (double)r.ransac_iterations;
    summary[6] = (double)r.ransac_improvements;
}

// The body of RelaxGroup::finalize's edge loop (src/relax/relax_group.cpp:142-176) for one edge: matches (i1, i2 into
// the images' features) with their previous inlier flags in `inl` (in/out); H = the re-fitted homography;
// summary: [n_inliers, can_decompose, accepted (inlier list kept)].
void oc_refit_edge(const double *loc1, size_t n1, const double *loc2, size_t n2, const double *model1, const double *model2,
                   const uint64_t *m_i1, const uint64_t *m_i2, const double *m_dist, size_t M, uint8_t *inl, double *H,
                   double *poses, double *summary)
{
    auto f1 = make_features(loc1, nullptr, nullptr, n1);
    auto f2 = make_features(loc2, nullptr, nullptr, n2);
    std::vector<feature_match> matches(M);
    for (size_t i = 0; i < M; i++)
        matches[i] = feature_match{(size_t)m_i1[i], (size_t)m_i2[i], m_dist[i]};
    const std::vector<correspondence> correspondences = distort_keypoints(f1, f2, matches, make_model(model1), make_model(model2));
    std::vector<bool> inliers(correspondences.size(), false);
    for (size_t i = 0; i < M; i++)
        if (inl[i])
            inliers[i] = true; // :152-155 (flags instead of the old inlier list's match_index values)
    homography_model h;
    for (int i = 0; i < 3; i++) // :158-162
    {
        h.fitInliers(correspondences, inliers);
        h.evaluate(correspondences, inliers);
    }
    std::memcpy(H, h.homography.m, 72);
    std::array<decomposed_pose, 4> rel;
    const bool can_decompose = h.decompose(correspondences, inliers, rel);
    const size_t num_inliers = std::count(inliers.begin(), inliers.end(), true);
    for (size_t i = 0; i < M; i++)
        inl[i] = inliers[i];
    for (int i = 0; i < 4; i++)
    {
        double *o = poses + 8 * i;
        const auto &p = rel[i];
        o[0] = p.orientation.x, o[1] = p.orientation.y, o[2] = p.orientation.z, o[3] = p.orientation.w;
        o[4] = p.position.x, o[5] = p.position.y, o[6] = p.position.z, o[7] = p.score;
    }
    summary[0] = (double)num_inliers;
    summary[1] = can_decompose;
    summary[2] = can_decompose && num_inliers > homography_model::MINIMUM_POINTS * 1.5;
}

// ---------------------------------------------------------------------------------------------
// CPU baseline driver (bench.py cpu_baseline leg): the reference's scheduling, i.e. one closure per
// directed pair under `#pragma omp parallel for schedule(dynamic,1)` (pipeline.cpp:42-49), with the
// destination subset recomputed inside each closure exactly as link_stage.cpp:80-81 does
// ("faithful") or cached per image ("cached").  Features of image i: loc[off[i]..off[i+1]).
// seconds_out: [total wall, sum match, sum undistort, sum ransac] (bucket names of link_stage.cpp:77,86,90)
void oc_link_batch_cpu(const double *loc, const float *strength, const uint64_t *desc, const uint64_t *off,
                       size_t n_images, const uint64_t *num_sparse, const double *model, const uint32_t *pairs,
                       size_t n_pairs, int faithful, int threads, uint64_t *out_counts /*n_pairs x 2*/,
                       double *out_H /*n_pairs x 9*/, double *seconds_out)
{
    std::vector<std::vector<feature_2d>> feats(n_images);
    for (size_t i = 0; i < n_images; i++)
        feats[i] = make_features(loc + 2 * off[i], strength + off[i], desc + 8 * off[i], off[i + 1] - off[i]);
    const camera_model cm = make_model(model);
#ifdef _OPENMP
    if (threads > 0)
        omp_set_num_threads(threads);
#endif
    using clk = std::chrono::steady_clock;
    auto secs = [](clk::time_point a, clk::time_point b) { return std::chrono::duration<double>(b - a).count(); };
    const auto t0 = clk::now();
    std::vector<std::vector<size_t>> subset(n_images);
    // the source subset is computed once per source image (link_stage.cpp:63-65)
#pragma omp parallel for schedule(dynamic, 1)
    for (size_t i = 0; i < n_images; i++)
        subset[i] = spatially_subsample_feature_indices(feats[i], 40.0, num_sparse[i]);
    double t_match = 0, t_und = 0, t_ransac = 0;
#pragma omp parallel for schedule(dynamic, 1) reduction(+ : t_match, t_und, t_ransac)
    for (size_t p = 0; p < n_pairs; p++)
    {
        const uint32_t a = pairs[2 * p], b = pairs[2 * p + 1];
        const auto ta = clk::now();
        std::vector<size_t> idx2_local;
        if (faithful)
            idx2_local = spatially_subsample_feature_indices(feats[b], 40.0, num_sparse[b]);
        const std::vector<size_t> &idx2 = faithful ? idx2_local : subset[b];
        auto matches = match_features_subset(feats[a], feats[b], subset[a], idx2);
        const auto tb = clk::now();
        auto corr = distort_keypoints(feats[a], feats[b], matches, cm, cm);
        const auto tc = clk::now();
        homography_model h;
        std::vector<bool> inl;
        ransac(corr, h, inl);
        std::array<decomposed_pose, 4> poses;
        h.decompose(corr, inl, poses);
        const auto td = clk::now();
        t_match += secs(ta, tb);
        t_und += secs(tb, tc);
        t_ransac += secs(tc, td);
        out_counts[2 * p] = matches.size();
        out_counts[2 * p + 1] = std::count(inl.begin(), inl.end(), true);
        std::memcpy(out_H + 9 * p, h.homography.m, 72);
    }
    seconds_out[0] = secs(t0, clk::now());
    seconds_out[1] = t_match;
    seconds_out[2] = t_und;
    seconds_out[3] = t_ransac;
}

int oc_num_threads()
{
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

// the restated sinf / cosf against THIS process's libm over every float of [lo_bits, hi_bits] (bit patterns): the number of
// arguments where either differs, the first such argument in *first_bad
uint64_t oc_libm_sincosf_mismatches(uint32_t lo_bits, uint32_t hi_bits, uint32_t *first_bad)
{
    uint64_t bad = 0;
#pragma omp parallel for reduction(+ : bad) schedule(static)
    for (int64_t b = (int64_t)lo_bits; b <= (int64_t)hi_bits; b++)
    {
        const uint32_t u = (uint32_t)b;
        float x;
        std::memcpy(&x, &u, 4);
        const float s0 = sinf(x), c0 = cosf(x), s1 = oracle::akaze::libm_sinf(x), c1 = oracle::akaze::libm_cosf(x);
        if (std::memcmp(&s0, &s1, 4) != 0 || std::memcmp(&c0, &c1, 4) != 0)
        {
            bad++;
#pragma omp critical
            if (first_bad && (*first_bad == 0 || u < *first_bad))
                *first_bad = u;
        }
    }
    return bad;
}
void oc_akaze_float_functions(float y, float x, float *atan2_deg, float *sin_x, float *cos_x)
{
    *atan2_deg = oracle::akaze::cv_fast_atan2_deg(y, x);
    *sin_x = oracle::akaze::libm_sinf(x);
    *cos_x = oracle::akaze::libm_cosf(x);
}
void oc_akaze_subpixel_solve(float Dxx, float Dxy, float Dyy, float Dx, float Dy, float *dxy2)
{
    oracle::akaze::subpixel_solve(Dxx, Dxy, Dyy, Dx, Dy, dxy2, dxy2 + 1);
}

// FED step sizes of one diffusion cycle (fed_tau_by_process_time, fed.cpp of AKAZE [3P]); returns their number
size_t oc_fed_tau(float T, int M, float tau_max, int reordering, float *out, size_t cap)
{
    std::vector<float> tau;
    oracle::akaze::fed_tau_by_process_time(T, M, tau_max, reordering != 0, tau);
    for (size_t i = 0; i < tau.size() && i < cap; i++)
        out[i] = tau[i];
    return tau.size();
}

} // extern "C"

// ---- a9: fundamental / essential matrix models (oracle/epipolar.cpp).  model: 0 = fundamental, 1 = essential;
//      rays6: n x {measurement1, measurement2}; M9 row-major.
namespace
{
std::vector<oracle::correspondence> epipolar_corrs(const double *rays6, const double *quality, size_t n)
{
    std::vector<oracle::correspondence> c(n);
    for (size_t i = 0; i < n; i++)
    {
        c[i].measurement1 = {rays6[6 * i], rays6[6 * i + 1], rays6[6 * i + 2]};
        c[i].measurement2 = {rays6[6 * i + 3], rays6[6 * i + 4], rays6[6 * i + 5]};
        c[i].quality = quality ? quality[i] : 0.0;
    }
    return c;
}
void mat_out(const oracle::Mat3 &M, double *M9)
{
    for (int r = 0; r < 3; r++)
        for (int c = 0; c < 3; c++)
            M9[3 * r + c] = M(r, c);
}
oracle::Mat3 mat_in(const double *M9)
{
    oracle::Mat3 M;
    for (int r = 0; r < 3; r++)
        for (int c = 0; c < 3; c++)
            M(r, c) = M9[3 * r + c];
    return M;
}
} // namespace

extern "C" double oc_ransac_epipolar(int model, const double *rays6, const double *quality, size_t n, double threshold, double *M9,
                                     uint8_t *inliers_out, size_t *iterations)
{
    const auto corrs = epipolar_corrs(rays6, quality, n);
    std::vector<bool> inl;
    double score;
    if (model == 0)
    {
        oracle::fundamental_matrix_model m;
        if (threshold > 0)
            m.inlier_threshold = threshold;
        score = oracle::ransac(corrs, m, inl, iterations);
        mat_out(m.fundamental_matrix, M9);
    }
    else
    {
        oracle::essential_matrix_model m;
        if (threshold > 0)
            m.inlier_threshold = threshold;
        score = oracle::ransac(corrs, m, inl, iterations);
        mat_out(m.essential_matrix, M9);
    }
    for (size_t i = 0; i < inl.size(); i++)
        inliers_out[i] = inl[i];
    return score;
}

extern "C" void oc_epipolar_fit_inliers(int model, const double *rays6, size_t n, const uint8_t *inliers, double *M9)
{
    const auto corrs = epipolar_corrs(rays6, nullptr, n);
    std::vector<bool> inl(inliers, inliers + n);
    if (model == 0)
    {
        oracle::fundamental_matrix_model m;
        m.fitInliers(corrs, inl);
        mat_out(m.fundamental_matrix, M9);
    }
    else
    {
        oracle::essential_matrix_model m;
        m.fitInliers(corrs, inl);
        mat_out(m.essential_matrix, M9);
    }
}

extern "C" double oc_epipolar_evaluate(const double *M9, const double *rays6, size_t n, double threshold, uint8_t *inliers_out,
                                       double *errors_out)
{
    const auto corrs = epipolar_corrs(rays6, nullptr, n);
    oracle::fundamental_matrix_model m; // both models share error() and evaluate()
    m.fundamental_matrix = mat_in(M9);
    if (threshold > 0)
        m.inlier_threshold = threshold;
    std::vector<bool> inl;
    const double score = m.evaluate(corrs, inl);
    for (size_t i = 0; i < n; i++)
    {
        inliers_out[i] = inl[i];
        if (errors_out)
            errors_out[i] = m.error(corrs[i]);
    }
    return score;
}

extern "C" int oc_essential_decompose(const double *E9, double *poses28)
{
    oracle::essential_matrix_model m;
    m.essential_matrix = mat_in(E9);
    std::array<oracle::decomposed_pose, 4> poses;
    const bool ok = m.decompose({}, {}, poses);
    for (int i = 0; i < 4; i++)
    {
        const double v[7] = {poses[i].orientation.x, poses[i].orientation.y, poses[i].orientation.z, poses[i].orientation.w,
                             poses[i].position.x,    poses[i].position.y,    poses[i].position.z};
        std::copy(v, v + 7, poses28 + 7 * i);
    }
    return ok ? 1 : 0;
}

extern "C" void oc_jacobi_svd(const double *A, int n, double *U, double *S, double *V)
{
    oracle::jacobi_svd_square(A, n, U, S, V);
}
