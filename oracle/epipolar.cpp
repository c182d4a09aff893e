// TEST INFRASTRUCTURE (oracle): the fundamental- and essential-matrix RANSAC models of the reference (SURVEY.md section 8
// row a9), which its pipeline never calls (link_stage.cpp:91-98 instantiates the homography model only) but its unit
// tests and benchmarks do.  Restated from
//   src/model_inliers/fundamental_matrix_model.cpp:12-217, src/model_inliers/essential_matrix_model.cpp:12-155,
//   src/model_inliers/ransac.cpp:53-257 (the loop, here once more as a template over the model: has_check_degeneracy is
//   true for the fundamental matrix, neither model has checkSampleDegeneracy),
// with Eigen::JacobiSVD restated from its published algorithm (two-sided Jacobi on the scaled square matrix, sweeps
// over p > q until every off-diagonal pair is below precision * max |diagonal|, singular values made positive and
// sorted in decreasing order).  Eigen itself is not in the image, so the restatement is pinned against the reference's
// own unit tests (tests/test_oracle_epipolar.py restates test/test_ransac_unit.cpp:58-300) at their tolerances; a
// singular vector's sign and, in a degenerate null space (the essential model fits 9 unknowns to 5 points), the vector
// itself are whatever the rotation sequence leaves, as in Eigen.
#include "oracle.hpp"

#include <algorithm>
#include <cmath>
#include <limits>
#include <numeric>
#include <random>

namespace oracle
{
namespace
{
// ---- Eigen::JacobiSVD<Matrix<double, n, n>> (ComputeFullU | ComputeFullV), row-major storage -------------------------
struct jrot
{
    double c, s;
};

// JacobiRotation::makeJacobi(x, y, z) for the real symmetric 2 x 2 block [x y; y z]
static jrot make_jacobi(double x, double y, double z)
{
    const double deno = 2.0 * std::fabs(y);
    if (deno < std::numeric_limits<double>::min())
        return jrot{1.0, 0.0};
    const double tau = (x - z) / deno;
    const double w = std::sqrt(tau * tau + 1.0);
    const double t = tau > 0 ? 1.0 / (tau + w) : 1.0 / (tau - w);
    const double sign_t = t > 0 ? 1.0 : -1.0;
    const double n = 1.0 / std::sqrt(t * t + 1.0);
    return jrot{n, -sign_t * (y / std::fabs(y)) * std::fabs(t) * n};
}

// rows p, q of M (n columns): M.applyOnTheLeft(p, q, j)  ->  x' = c x + s y, y' = -s x + c y  (with j.adjoint())
static void apply_left(std::vector<double> &M, int n, int p, int q, jrot j)
{
    for (int k = 0; k < n; k++)
    {
        const double x = M[p * n + k], y = M[q * n + k];
        M[p * n + k] = j.c * x + j.s * y;
        M[q * n + k] = -j.s * x + j.c * y;
    }
}
// columns p, q: M.applyOnTheRight(p, q, j)  ->  x' = c x - s y, y' = s x + c y
static void apply_right(std::vector<double> &M, int n, int p, int q, jrot j)
{
    for (int k = 0; k < n; k++)
    {
        const double x = M[k * n + p], y = M[k * n + q];
        M[k * n + p] = j.c * x - j.s * y;
        M[k * n + q] = j.s * x + j.c * y;
    }
}

// A = U diag(S) V^T, S decreasing.  A, U, V: n x n row-major.
static void jacobi_svd(const std::vector<double> &A, int n, std::vector<double> &U, std::vector<double> &S,
                       std::vector<double> &V)
{
    const double precision = 2.0 * std::numeric_limits<double>::epsilon();
    const double consider_as_zero = std::numeric_limits<double>::min();
    double scale = 0;
    for (double v : A)
        scale = std::max(scale, std::fabs(v));
    if (scale == 0)
        scale = 1;
    std::vector<double> W(A);
    for (double &v : W)
        v /= scale;
    U.assign((size_t)n * n, 0.0);
    V.assign((size_t)n * n, 0.0);
    for (int i = 0; i < n; i++)
        U[i * n + i] = V[i * n + i] = 1.0;
    double max_diag = 0;
    for (int i = 0; i < n; i++)
        max_diag = std::max(max_diag, std::fabs(W[i * n + i]));
    bool finished = false;
    while (!finished)
    {
        finished = true;
        for (int p = 1; p < n; p++)
            for (int q = 0; q < p; q++)
            {
                const double threshold = std::max(consider_as_zero, precision * max_diag);
                if (std::fabs(W[p * n + q]) > threshold || std::fabs(W[q * n + p]) > threshold)
                {
                    finished = false;
                    // real_2x2_jacobi_svd: first a rotation that makes the block symmetric, then makeJacobi
                    double m00 = W[p * n + p], m01 = W[p * n + q], m10 = W[q * n + p], m11 = W[q * n + q];
                    const double t = m00 + m11, d = m10 - m01;
                    jrot rot1;
                    if (std::fabs(d) < std::numeric_limits<double>::min())
                        rot1 = jrot{1.0, 0.0};
                    else
                    {
                        const double u = t / d, tmp = std::sqrt(1.0 + u * u);
                        rot1 = jrot{u / tmp, 1.0 / tmp};
                    }
                    // m.applyOnTheLeft(0, 1, rot1)
                    const double n00 = rot1.c * m00 + rot1.s * m10, n01 = rot1.c * m01 + rot1.s * m11;
                    const double n11 = -rot1.s * m01 + rot1.c * m11;
                    const jrot j_right = make_jacobi(n00, n01, n11);
                    // j_left = rot1 * j_right.transpose()
                    const jrot jt{j_right.c, -j_right.s};
                    const jrot j_left{rot1.c * jt.c - rot1.s * jt.s, rot1.c * jt.s + rot1.s * jt.c};
                    apply_left(W, n, p, q, j_left);
                    apply_right(U, n, p, q, jrot{j_left.c, -j_left.s}); // j_left.transpose()
                    apply_right(W, n, p, q, j_right);
                    apply_right(V, n, p, q, j_right);
                    max_diag = std::max(max_diag, std::max(std::fabs(W[p * n + p]), std::fabs(W[q * n + q])));
                }
            }
    }
    S.assign(n, 0.0);
    for (int i = 0; i < n; i++)
    {
        const double a = W[i * n + i];
        S[i] = std::fabs(a);
        if (a < 0)
            for (int k = 0; k < n; k++)
                U[k * n + i] = -U[k * n + i];
        S[i] *= scale;
    }
    for (int i = 0; i < n; i++) // selection sort by decreasing singular value, columns of U and V follow
    {
        int pos = i;
        for (int k = i + 1; k < n; k++)
            if (S[k] > S[pos])
                pos = k;
        if (S[pos] == 0)
            break;
        if (pos != i)
        {
            std::swap(S[i], S[pos]);
            for (int k = 0; k < n; k++)
            {
                std::swap(U[k * n + i], U[k * n + pos]);
                std::swap(V[k * n + i], V[k * n + pos]);
            }
        }
    }
}

static Mat3 svd_recompose(const std::vector<double> &U, const double s[3], const std::vector<double> &V)
{
    Mat3 out;
    for (int r = 0; r < 3; r++)
        for (int c = 0; c < 3; c++)
        {
            double v = 0;
            for (int k = 0; k < 3; k++)
                v += U[r * 3 + k] * s[k] * V[c * 3 + k];
            out(r, c) = v;
        }
    return out;
}

static std::vector<double> to_vec(const Mat3 &M)
{
    std::vector<double> v(9);
    for (int r = 0; r < 3; r++)
        for (int c = 0; c < 3; c++)
            v[r * 3 + c] = M(r, c);
    return v;
}

// calculateFundamentalMatrix / calculateEssentialMatrix: null vector of A^T A, then the rank / singular value constraint
static Mat3 null_vector_matrix(const std::vector<std::array<double, 9>> &rows)
{
    std::vector<double> AtA(81, 0.0);
    for (int i = 0; i < 9; i++)
        for (int j = 0; j < 9; j++)
        {
            double v = 0;
            for (const auto &r : rows)
                v += r[i] * r[j];
            AtA[i * 9 + j] = v;
        }
    std::vector<double> U, S, V;
    jacobi_svd(AtA, 9, U, S, V);
    Mat3 M;
    for (int r = 0; r < 3; r++)
        for (int c = 0; c < 3; c++)
            M(r, c) = V[(r * 3 + c) * 9 + 8];
    return M;
}

static std::array<double, 9> epipolar_row(const correspondence &c)
{
    const Vec2 p1 = hnormalized(c.measurement1), p2 = hnormalized(c.measurement2);
    const double x = p1.x, y = p1.y, x_ = p2.x, y_ = p2.y;
    return {x * x_, x * y_, x, y * x_, y * y_, y, x_, y_, 1.0};
}

static Mat3 enforce_rank2(const Mat3 &M, bool equal_singular_values)
{
    std::vector<double> U, S, V;
    jacobi_svd(to_vec(M), 3, U, S, V);
    double s[3] = {S[0], S[1], 0.0};
    if (equal_singular_values)
        s[0] = s[1] = (S[0] + S[1]) / 2.0;
    return svd_recompose(U, s, V);
}

static double sampson_error(const Mat3 &F, const correspondence &cor)
{
    const Vec3 x1 = cor.measurement1 / cor.measurement1.z, x2 = cor.measurement2 / cor.measurement2.z;
    const Vec3 Fx1 = mul(F, x1), Ftx2 = mul(transpose(F), x2);
    const double x2tFx1 = dot(x2, Fx1);
    const double denom = Fx1.x * Fx1.x + Fx1.y * Fx1.y + Ftx2.x * Ftx2.x + Ftx2.y * Ftx2.y;
    if (denom < 1e-20)
        return std::numeric_limits<double>::max();
    return std::sqrt((x2tFx1 * x2tFx1) / denom);
}

template <class Model> double evaluate_model(Model &m, const std::vector<correspondence> &corrs, std::vector<bool> &inliers)
{
    inliers.resize(corrs.size());
    double total_score = 0;
    for (size_t i = 0; i < corrs.size(); i++)
    {
        const double e = m.error(corrs[i]);
        if (e < m.inlier_threshold)
        {
            inliers[i] = true;
            const double ratio = e / m.inlier_threshold;
            total_score += 1.0 - ratio * ratio;
        }
        else
            inliers[i] = false;
    }
    return total_score;
}
} // namespace

void jacobi_svd_square(const double *A, int n, double *U, double *S, double *V)
{
    std::vector<double> a(A, A + (size_t)n * n), u, s, v;
    jacobi_svd(a, n, u, s, v);
    std::copy(u.begin(), u.end(), U);
    std::copy(s.begin(), s.end(), S);
    std::copy(v.begin(), v.end(), V);
}

// ---- fundamental_matrix_model (fundamental_matrix_model.cpp) ---------------------------------------------------------
fundamental_matrix_model::fundamental_matrix_model()
{
    for (int r = 0; r < 3; r++)
        for (int c = 0; c < 3; c++)
            fundamental_matrix(r, c) = NAN;
}

void fundamental_matrix_model::fit(const std::vector<correspondence> &corrs, const std::array<size_t, 8> &initial_indices)
{
    std::vector<std::array<double, 9>> rows;
    for (size_t i : initial_indices)
        rows.push_back(epipolar_row(corrs[i]));
    fundamental_matrix = enforce_rank2(null_vector_matrix(rows), false);
}

void fundamental_matrix_model::fitInliers(const std::vector<correspondence> &corrs, const std::vector<bool> &inliers)
{
    const size_t num_inliers = std::count(inliers.begin(), inliers.end(), true);
    if (num_inliers < MINIMUM_POINTS)
        return;
    std::vector<std::array<double, 9>> rows;
    for (size_t i = 0; i < corrs.size(); i++)
        if (inliers[i])
            rows.push_back(epipolar_row(corrs[i]));
    fundamental_matrix = enforce_rank2(null_vector_matrix(rows), false);
}

double fundamental_matrix_model::error(const correspondence &cor)
{
    return sampson_error(fundamental_matrix, cor);
}

double fundamental_matrix_model::evaluate(const std::vector<correspondence> &corrs, std::vector<bool> &inliers)
{
    return evaluate_model(*this, corrs, inliers);
}

// DEGENSAC (:125-215): if the F inliers are dominated by a plane, F = [e']_x H with the epipole from the off-plane points
void fundamental_matrix_model::checkDegeneracy(const std::vector<correspondence> &corrs, std::vector<bool> &inliers)
{
    std::vector<size_t> f_inlier_idx;
    for (size_t i = 0; i < inliers.size(); i++)
        if (inliers[i])
            f_inlier_idx.push_back(i);
    if (f_inlier_idx.size() < homography_model::MINIMUM_POINTS)
        return;
    homography_model h_model;
    h_model.inlier_threshold = inlier_threshold * 2;
    std::array<size_t, 4> h_indices;
    for (size_t i = 0; i < 4; i++)
        h_indices[i] = f_inlier_idx[i];
    h_model.fit(corrs, h_indices);
    std::vector<bool> h_inliers(corrs.size(), false);
    size_t h_inlier_count = 0;
    for (size_t idx : f_inlier_idx)
        if (h_model.error(corrs[idx]) < h_model.inlier_threshold)
        {
            h_inliers[idx] = true;
            h_inlier_count++;
        }
    const double h_ratio = static_cast<double>(h_inlier_count) / f_inlier_idx.size();
    if (h_ratio < 0.7)
        return;
    h_model.fitInliers(corrs, h_inliers);
    std::vector<size_t> non_h_idx;
    for (size_t idx : f_inlier_idx)
    {
        if (h_model.error(corrs[idx]) < h_model.inlier_threshold)
            h_inliers[idx] = true;
        else
        {
            h_inliers[idx] = false;
            non_h_idx.push_back(idx);
        }
    }
    if (non_h_idx.size() < 2)
        return;
    // epipole: right singular vector of the smallest singular value of the rows (x2 x H x1)^T, i.e. the eigenvector of
    // the smallest eigenvalue of their 3 x 3 Gram matrix (Eigen runs its Jacobi sweeps on the R factor of the rows'
    // QR decomposition, whose Gram matrix is the same)
    std::vector<double> G(9, 0.0);
    for (size_t i : non_h_idx)
    {
        const Vec3 x1 = corrs[i].measurement1 / corrs[i].measurement1.z, x2 = corrs[i].measurement2 / corrs[i].measurement2.z;
        const Vec3 r = cross(x2, mul(h_model.homography, x1));
        const double rr[3] = {r.x, r.y, r.z};
        for (int a = 0; a < 3; a++)
            for (int b = 0; b < 3; b++)
                G[a * 3 + b] += rr[a] * rr[b];
    }
    std::vector<double> U, S, V;
    jacobi_svd(G, 3, U, S, V);
    const Vec3 epipole{V[0 * 3 + 2], V[1 * 3 + 2], V[2 * 3 + 2]};
    Mat3 e_cross;
    e_cross(0, 0) = 0, e_cross(0, 1) = -epipole.z, e_cross(0, 2) = epipole.y;
    e_cross(1, 0) = epipole.z, e_cross(1, 1) = 0, e_cross(1, 2) = -epipole.x;
    e_cross(2, 0) = -epipole.y, e_cross(2, 1) = epipole.x, e_cross(2, 2) = 0;
    const Mat3 F_candidate = enforce_rank2(mul(e_cross, h_model.homography), false);
    const Mat3 old_F = fundamental_matrix;
    std::vector<bool> old_inliers = inliers;
    fundamental_matrix = F_candidate;
    const double candidate_score = evaluate(corrs, inliers);
    fundamental_matrix = old_F;
    const double original_score = evaluate(corrs, old_inliers);
    if (candidate_score > original_score)
        fundamental_matrix = F_candidate;
    else
        inliers = old_inliers;
}

// ---- essential_matrix_model (essential_matrix_model.cpp) -------------------------------------------------------------
essential_matrix_model::essential_matrix_model()
{
    for (int r = 0; r < 3; r++)
        for (int c = 0; c < 3; c++)
            essential_matrix(r, c) = NAN;
}

void essential_matrix_model::fit(const std::vector<correspondence> &corrs, const std::array<size_t, 5> &initial_indices)
{
    std::vector<std::array<double, 9>> rows;
    for (size_t i : initial_indices)
        rows.push_back(epipolar_row(corrs[i]));
    essential_matrix = enforce_rank2(null_vector_matrix(rows), true);
}

void essential_matrix_model::fitInliers(const std::vector<correspondence> &corrs, const std::vector<bool> &inliers)
{
    const size_t num_inliers = std::count(inliers.begin(), inliers.end(), true);
    if (num_inliers < MINIMUM_POINTS)
        return;
    std::vector<std::array<double, 9>> rows;
    for (size_t i = 0; i < corrs.size(); i++)
        if (inliers[i])
            rows.push_back(epipolar_row(corrs[i]));
    essential_matrix = enforce_rank2(null_vector_matrix(rows), true);
}

double essential_matrix_model::error(const correspondence &cor)
{
    return sampson_error(essential_matrix, cor);
}

double essential_matrix_model::evaluate(const std::vector<correspondence> &corrs, std::vector<bool> &inliers)
{
    return evaluate_model(*this, corrs, inliers);
}

static Quat quat_from_rotation(const Mat3 &m) // Eigen::Quaterniond(Matrix3d)
{
    Quat q;
    double t = m(0, 0) + m(1, 1) + m(2, 2);
    if (t > 0)
    {
        t = std::sqrt(t + 1.0);
        q.w = 0.5 * t;
        t = 0.5 / t;
        q.x = (m(2, 1) - m(1, 2)) * t;
        q.y = (m(0, 2) - m(2, 0)) * t;
        q.z = (m(1, 0) - m(0, 1)) * t;
        return q;
    }
    int i = 0;
    if (m(1, 1) > m(0, 0))
        i = 1;
    if (m(2, 2) > m(i, i))
        i = 2;
    const int j = (i + 1) % 3, k = (j + 1) % 3;
    t = std::sqrt(m(i, i) - m(j, j) - m(k, k) + 1.0);
    double v[3];
    v[i] = 0.5 * t;
    t = 0.5 / t;
    q.w = (m(k, j) - m(j, k)) * t;
    v[j] = (m(j, i) + m(i, j)) * t;
    v[k] = (m(k, i) + m(i, k)) * t;
    q.x = v[0], q.y = v[1], q.z = v[2];
    return q;
}

bool essential_matrix_model::decompose(const std::vector<correspondence> &, const std::vector<bool> &,
                                       std::array<decomposed_pose, 4> &poses)
{
    std::vector<double> U, S, V;
    jacobi_svd(to_vec(essential_matrix), 3, U, S, V);
    Mat3 Um, Vt, Wm, Wt;
    for (int r = 0; r < 3; r++)
        for (int c = 0; c < 3; c++)
        {
            Um(r, c) = U[r * 3 + c];
            Vt(r, c) = V[c * 3 + r];
            Wm(r, c) = Wt(r, c) = 0;
        }
    Wm(0, 1) = -1, Wm(1, 0) = 1, Wm(2, 2) = 1;
    Wt = transpose(Wm);
    Mat3 R1 = mul(mul(Um, Wm), Vt), R2 = mul(mul(Um, Wt), Vt);
    auto negate = [](Mat3 &M) {
        for (int r = 0; r < 3; r++)
            for (int c = 0; c < 3; c++)
                M(r, c) = -M(r, c);
    };
    if (det(R1) < 0)
        negate(R1);
    if (det(R2) < 0)
        negate(R2);
    const Vec3 t{Um(0, 2), Um(1, 2), Um(2, 2)};
    poses[0].orientation = quat_from_rotation(R1);
    poses[0].position = t;
    poses[1].orientation = quat_from_rotation(R1);
    poses[1].position = t * -1.0;
    poses[2].orientation = quat_from_rotation(R2);
    poses[2].position = t;
    poses[3].orientation = quat_from_rotation(R2);
    poses[3].position = t * -1.0;
    return true;
}

// ---- ransac<Model> (ransac.cpp:53-257) for the two models above -------------------------------------------------------
namespace
{
template <int n> double fast_pow(double d);
template <> inline double fast_pow<5>(double d)
{
    const double t = d * d;
    return t * t * d;
}
template <> inline double fast_pow<8>(double d)
{
    double t = d * d;
    t = t * t;
    return t * t;
}
template <class Model> struct has_degeneracy_check
{
    static constexpr bool value = false;
};
template <> struct has_degeneracy_check<fundamental_matrix_model>
{
    static constexpr bool value = true;
};

template <class Model>
double ransac_model(const std::vector<correspondence> &matches, Model &model, std::vector<bool> &inliers, size_t *iterations_out)
{
    constexpr size_t K = Model::MINIMUM_POINTS;
    const size_t MIN_ITERATIONS = 20, MAX_ITERATIONS = 10000, MAX_INNER_ITERATIONS = 5;
    const double PROBABILITY = 0.999;
    const double log_1m_p = std::log(1 - PROBABILITY);
    inliers.resize(matches.size());
    std::fill(inliers.begin(), inliers.end(), false);
    if (iterations_out)
        *iterations_out = 0;
    if (matches.size() < K)
        return 0;
    bool has_quality = false;
    for (const auto &m : matches)
        if (m.quality != 0)
        {
            has_quality = true;
            break;
        }
    std::vector<size_t> sorted_idx;
    if (has_quality)
    {
        sorted_idx.resize(matches.size());
        std::iota(sorted_idx.begin(), sorted_idx.end(), 0);
        std::sort(sorted_idx.begin(), sorted_idx.end(),
                  [&matches](size_t a, size_t b) { return matches[a].quality < matches[b].quality; });
    }
    std::vector<size_t> eval_order(matches.size());
    std::iota(eval_order.begin(), eval_order.end(), 0);
    Model best_model{};
    double best_score = 0;
    std::default_random_engine generator(42);
    size_t prosac_n = has_quality ? K : matches.size();
    auto map_idx = [&sorted_idx, has_quality](size_t i) -> size_t { return has_quality ? sorted_idx[i] : i; };
    auto random_k_from_n = [&generator, &map_idx](size_t pool) {
        std::array<size_t, K> indices;
        std::uniform_int_distribution<size_t> dist(0, pool - 1);
        for (size_t j = 0; j < K; j++)
        {
            size_t candidate;
            bool unique;
            do
            {
                candidate = dist(generator);
                unique = true;
                for (size_t k = 0; k < j; k++)
                    if (indices[k] == map_idx(candidate))
                    {
                        unique = false;
                        break;
                    }
            } while (!unique);
            indices[j] = map_idx(candidate);
        }
        return indices;
    };
    auto prosac_sample = [&generator, &sorted_idx](size_t pool) {
        std::array<size_t, K> indices;
        indices[0] = sorted_idx[pool - 1];
        std::uniform_int_distribution<size_t> dist(0, pool - 2);
        for (size_t j = 1; j < K; j++)
        {
            size_t candidate;
            bool unique;
            do
            {
                candidate = dist(generator);
                unique = true;
                for (size_t k = 0; k < j; k++)
                    if (indices[k] == sorted_idx[candidate])
                    {
                        unique = false;
                        break;
                    }
            } while (!unique);
            indices[j] = sorted_idx[candidate];
        }
        return indices;
    };
    size_t probability_iterations = MAX_ITERATIONS;
    std::shuffle(eval_order.begin(), eval_order.end(), generator);
    std::vector<bool> candidate_inliers(matches.size(), false);
    size_t i = 0;
    for (; i < probability_iterations; i++)
    {
        if (has_quality && prosac_n < matches.size() && i > 0 && i % 10 == 0)
            prosac_n++;
        std::array<size_t, K> initial_indices;
        if (has_quality && prosac_n < matches.size() && prosac_n > K)
            initial_indices = prosac_sample(prosac_n);
        else
            initial_indices = random_k_from_n(has_quality ? prosac_n : matches.size());
        model.fit(matches, initial_indices);
        double score = 0;
        size_t checked = 0;
        bool rejected = false;
        std::fill(candidate_inliers.begin(), candidate_inliers.end(), false);
        for (size_t idx : eval_order)
        {
            const double e = model.error(matches[idx]);
            if (e < model.inlier_threshold)
            {
                candidate_inliers[idx] = true;
                const double ratio = e / model.inlier_threshold;
                score += 1.0 - ratio * ratio;
            }
            checked++;
            if (checked > 20 && best_score > 0 && score < best_score * static_cast<double>(checked) / matches.size() * 0.6)
            {
                rejected = true;
                break;
            }
        }
        if (rejected)
            continue;
        if (score > best_score)
        {
            best_model = model;
            best_score = score;
            inliers = candidate_inliers;
            if constexpr (has_degeneracy_check<Model>::value)
            {
                model.checkDegeneracy(matches, inliers);
                const double degen_score = model.evaluate(matches, inliers);
                if (degen_score > best_score)
                {
                    best_model = model;
                    best_score = degen_score;
                }
            }
            model.fitInliers(matches, inliers);
            double inlier_score = model.evaluate(matches, inliers);
            if (inlier_score > best_score)
            {
                best_model = model;
                best_score = inlier_score;
                for (size_t j = 1; j < MAX_INNER_ITERATIONS; j++)
                {
                    model.fitInliers(matches, inliers);
                    inlier_score = model.evaluate(matches, inliers);
                    if (inlier_score > best_score)
                    {
                        best_model = model;
                        best_score = inlier_score;
                    }
                    else
                        break;
                }
            }
            const double omega = best_score / matches.size();
            const double omega_n = fast_pow<(int)K>(omega);
            const double log_1m_omega_n = std::log(1 - omega_n);
            probability_iterations =
                std::max(MIN_ITERATIONS, std::min(MAX_ITERATIONS, static_cast<size_t>(log_1m_p / log_1m_omega_n)));
        }
    }
    if (iterations_out)
        *iterations_out = i;
    model = best_model;
    return model.evaluate(matches, inliers) / matches.size();
}
} // namespace

double ransac(const std::vector<correspondence> &matches, fundamental_matrix_model &model, std::vector<bool> &inliers,
              size_t *iterations)
{
    return ransac_model(matches, model, inliers, iterations);
}
double ransac(const std::vector<correspondence> &matches, essential_matrix_model &model, std::vector<bool> &inliers,
              size_t *iterations)
{
    return ransac_model(matches, model, inliers, iterations);
}

} // namespace oracle
