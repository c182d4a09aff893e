// ORACLE — test infrastructure only (see oracle.hpp).
// Restates src/model_inliers/ransac.cpp:53-282 and src/model_inliers/homography_model.cpp:14-185.
#include "oracle.hpp"

#include <algorithm>
#include <numeric>
#include <random>

namespace oracle
{

// ------------------------------------------------------------------------------ homography_model
homography_model::homography_model() // homography_model.cpp:14-17
{
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++)
            homography.m[i][j] = homography_inverse.m[i][j] = NAN;
}

static inline void dlt_rows(const correspondence &c, double *r0, double *r1) // homography_model.cpp:26-35,61-70
{
    const Vec2 p1 = hnormalized(c.measurement1);
    const double x = p1.x, y = p1.y;
    const Vec2 p2 = hnormalized(c.measurement2);
    const double x_ = p2.x, y_ = p2.y;
    const double a[9] = {-x, -y, -1, 0, 0, 0, x * x_, y * x_, x_};
    const double b[9] = {0, 0, 0, -x, -y, -1, x * y_, y * y_, y_};
    for (int i = 0; i < 9; i++)
    {
        r0[i] = a[i];
        r1[i] = b[i];
    }
}

static void set_from_solution(homography_model &m, const double H_[9]) // homography_model.cpp:45-49,82-86
{
    for (int r = 0; r < 3; r++)
        for (int c = 0; c < 3; c++)
            m.homography.m[r][c] = H_[r * 3 + c];
    const double s = m.homography.m[2][2];
    for (int r = 0; r < 3; r++)
        for (int c = 0; c < 3; c++)
            m.homography.m[r][c] /= s;
    m.homography_inverse = inverse3(m.homography);
}

void homography_model::fit(const std::vector<correspondence> &corrs, const std::array<size_t, 4> &initial_indices)
{
    double P[81];
    for (size_t i = 0; i < 4; i++)
        dlt_rows(corrs[initial_indices[i]], P + (i * 2) * 9, P + (i * 2 + 1) * 9);
    for (int j = 0; j < 9; j++)
        P[8 * 9 + j] = 0;
    P[8 * 9 + 8] = 1;
    double rhs[9] = {0, 0, 0, 0, 0, 0, 0, 0, 1};
    double H_[9];
    full_piv_lu_solve9(P, 9, rhs, H_);
    set_from_solution(*this, H_);
}

void homography_model::fitInliers(const std::vector<correspondence> &corrs, const std::vector<bool> &inliers)
{
    const size_t num_inliers = std::count(inliers.begin(), inliers.end(), true);
    const size_t rows = num_inliers * 2 + 1;
    std::vector<double> P(rows * 9);
    for (size_t i = 0, j = 0; i < corrs.size(); i++)
        if (inliers[i])
        {
            dlt_rows(corrs[i], P.data() + (j * 2) * 9, P.data() + (j * 2 + 1) * 9);
            j++;
        }
    for (int j = 0; j < 9; j++)
        P[(rows - 1) * 9 + j] = 0;
    P[(rows - 1) * 9 + 8] = 1;
    std::vector<double> rhs(rows, 0.0);
    rhs[rows - 1] = 1;
    double H_[9];
    full_piv_lu_solve9(P.data(), rows, rhs.data(), H_);
    set_from_solution(*this, H_);
}

double homography_model::error(const correspondence &corr) // homography_model.cpp:89-97
{
    const Vec3 m1 = corr.measurement1 / corr.measurement1.z;
    const Vec3 m2 = corr.measurement2 / corr.measurement2.z;
    const Vec2 f = hnormalized(mul(homography, m1));
    const Vec2 b = hnormalized(mul(homography_inverse, m2));
    const double fx = f.x - m2.x, fy = f.y - m2.y;
    const double bx = b.x - m1.x, by = b.y - m1.y;
    const double fwd = fx * fx + fy * fy;
    const double bwd = bx * bx + by * by;
    return std::sqrt((fwd + bwd) / 2.0);
}

double homography_model::evaluate(const std::vector<correspondence> &corrs, std::vector<bool> &inliers)
{
    inliers.resize(corrs.size());
    double total_score = 0;
    for (size_t i = 0; i < corrs.size(); i++)
    {
        const double e = error(corrs[i]);
        if (e < inlier_threshold)
        {
            inliers[i] = true;
            const double ratio = e / inlier_threshold;
            total_score += 1.0 - ratio * ratio;
        }
        else
            inliers[i] = false;
    }
    return total_score;
}

bool homography_model::checkSampleDegeneracy(const std::vector<correspondence> &corrs,
                                             const std::array<size_t, 4> &indices)
{
    Vec2 pts[4];
    for (size_t i = 0; i < 4; i++)
        pts[i] = hnormalized(corrs[indices[i]].measurement1);
    for (int i = 0; i < 4; i++)
        for (int j = i + 1; j < 4; j++)
            for (int k = j + 1; k < 4; k++)
            {
                const double v1x = pts[j].x - pts[i].x, v1y = pts[j].y - pts[i].y;
                const double v2x = pts[k].x - pts[i].x, v2y = pts[k].y - pts[i].y;
                if (std::abs(v1x * v2y - v1y * v2x) < 1e-10)
                    return true;
            }
    return false;
}

// Eigen/src/Geometry/Quaternion.h quaternionbase_assign_impl<Other,3,3> (Shoemake 1987) [3P]
Quat quaternion_from_matrix(const Mat3 &mat)
{
    double q[4]; // x y z w
    double t = mat.m[0][0] + mat.m[1][1] + mat.m[2][2];
    if (t > 0)
    {
        t = std::sqrt(t + 1.0);
        q[3] = 0.5 * t;
        t = 0.5 / t;
        q[0] = (mat.m[2][1] - mat.m[1][2]) * t;
        q[1] = (mat.m[0][2] - mat.m[2][0]) * t;
        q[2] = (mat.m[1][0] - mat.m[0][1]) * t;
    }
    else
    {
        int i = 0;
        if (mat.m[1][1] > mat.m[0][0])
            i = 1;
        if (mat.m[2][2] > mat.m[i][i])
            i = 2;
        const int j = (i + 1) % 3, k = (j + 1) % 3;
        t = std::sqrt(mat.m[i][i] - mat.m[j][j] - mat.m[k][k] + 1.0);
        q[i] = 0.5 * t;
        t = 0.5 / t;
        q[3] = (mat.m[k][j] - mat.m[j][k]) * t;
        q[j] = (mat.m[j][i] + mat.m[i][j]) * t;
        q[k] = (mat.m[k][i] + mat.m[i][k]) * t;
    }
    return Quat{q[0], q[1], q[2], q[3]};
}

// ------------------------------------------------------------------- cv::decomposeHomographyMat [3P]
// OpenCV 4.x modules/calib3d/src/homography_decomp.cpp (HomographyDecompInria), K = I.
// Restated from the published algorithm (E. Malis, M. Vargas, "Deeper understanding of the homography
// decomposition for vision-based control", INRIA RR-6303).  Pinned by restated
// test/test_ransac_unit.cpp:22-52,114-176 at their tolerances (1e-14 / 1e-7).
static void sym3_eigenvalues_jacobi(const Mat3 &S, double ev[3])
{
    double a[3][3];
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++)
            a[i][j] = S.m[i][j];
    for (int sweep = 0; sweep < 60; sweep++)
    {
        const double off = std::abs(a[0][1]) + std::abs(a[0][2]) + std::abs(a[1][2]);
        if (off == 0.0)
            break;
        for (int p = 0; p < 2; p++)
            for (int q = p + 1; q < 3; q++)
            {
                if (a[p][q] == 0.0)
                    continue;
                const double theta = (a[q][q] - a[p][p]) / (2.0 * a[p][q]);
                const double t = (theta >= 0 ? 1.0 : -1.0) / (std::abs(theta) + std::sqrt(theta * theta + 1.0));
                const double c = 1.0 / std::sqrt(t * t + 1.0), s = t * c;
                for (int k = 0; k < 3; k++)
                {
                    const double akp = a[k][p], akq = a[k][q];
                    a[k][p] = c * akp - s * akq;
                    a[k][q] = s * akp + c * akq;
                }
                for (int k = 0; k < 3; k++)
                {
                    const double apk = a[p][k], aqk = a[q][k];
                    a[p][k] = c * apk - s * aqk;
                    a[q][k] = s * apk + c * aqk;
                }
            }
    }
    ev[0] = a[0][0];
    ev[1] = a[1][1];
    ev[2] = a[2][2];
    std::sort(ev, ev + 3, [](double x, double y) { return x > y; });
}

static inline double opposite_of_minor(const Mat3 &M, int row, int col)
{
    const int x1 = col == 0 ? 1 : 0, x2 = col == 2 ? 1 : 2;
    const int y1 = row == 0 ? 1 : 0, y2 = row == 2 ? 1 : 2;
    return M.m[y1][x2] * M.m[y2][x1] - M.m[y1][x1] * M.m[y2][x2];
}
static inline int signd(double x)
{
    return x >= 0 ? 1 : -1;
}

static Mat3 find_rmat_from_tstar_n(const Mat3 &Hn, const Vec3 &tstar, const Vec3 &n, double v)
{
    Mat3 M = identity3();
    const double ts[3] = {tstar.x, tstar.y, tstar.z}, nn[3] = {n.x, n.y, n.z};
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++)
            M.m[i][j] -= (2 / v) * ts[i] * nn[j];
    Mat3 R = mul(Hn, M);
    if (det(R) < 0)
        for (int i = 0; i < 3; i++)
            for (int j = 0; j < 3; j++)
                R.m[i][j] *= -1;
    return R;
}

size_t decompose_homography_mat(const Mat3 &H, cam_motion out[4])
{
    // normalize with K = I, then removeScale(): divide by the middle singular value
    double ev[3];
    sym3_eigenvalues_jacobi(mul(transpose(H), H), ev);
    const double sv1 = std::sqrt(ev[1] > 0 ? ev[1] : 0.0);
    Mat3 Hn;
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++)
            Hn.m[i][j] = H.m[i][j] * (1.0 / sv1);

    const double epsilon = 0.001;
    Mat3 S = mul(transpose(Hn), Hn);
    S.m[0][0] -= 1.0;
    S.m[1][1] -= 1.0;
    S.m[2][2] -= 1.0;

    double ninf = 0; // cv::norm(S, NORM_INF) on a Matx = max |element|
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++)
            ninf = std::max(ninf, std::abs(S.m[i][j]));
    if (ninf < epsilon)
    {
        out[0].R = Hn;
        out[0].t = Vec3{0, 0, 0};
        out[0].n = Vec3{0, 0, 0};
        return 1;
    }

    const double M00 = opposite_of_minor(S, 0, 0), M11 = opposite_of_minor(S, 1, 1), M22 = opposite_of_minor(S, 2, 2);
    const double rtM00 = std::sqrt(M00), rtM11 = std::sqrt(M11), rtM22 = std::sqrt(M22);
    const double M01 = opposite_of_minor(S, 0, 1), M12 = opposite_of_minor(S, 1, 2), M02 = opposite_of_minor(S, 0, 2);
    const int e12 = signd(M12), e02 = signd(M02), e01 = signd(M01);
    const double nS00 = std::abs(S.m[0][0]), nS11 = std::abs(S.m[1][1]), nS22 = std::abs(S.m[2][2]);

    int indx = 0;
    if (nS00 < nS11)
    {
        indx = 1;
        if (nS11 < nS22)
            indx = 2;
    }
    else if (nS00 < nS22)
        indx = 2;

    double npa[3], npb[3];
    switch (indx)
    {
    case 0:
        npa[0] = S.m[0][0], npb[0] = S.m[0][0];
        npa[1] = S.m[0][1] + rtM22, npb[1] = S.m[0][1] - rtM22;
        npa[2] = S.m[0][2] + e12 * rtM11, npb[2] = S.m[0][2] - e12 * rtM11;
        break;
    case 1:
        npa[0] = S.m[0][1] + rtM22, npb[0] = S.m[0][1] - rtM22;
        npa[1] = S.m[1][1], npb[1] = S.m[1][1];
        npa[2] = S.m[1][2] - e02 * rtM00, npb[2] = S.m[1][2] + e02 * rtM00;
        break;
    default:
        npa[0] = S.m[0][2] + e01 * rtM11, npb[0] = S.m[0][2] - e01 * rtM11;
        npa[1] = S.m[1][2] + rtM00, npb[1] = S.m[1][2] - rtM00;
        npa[2] = S.m[2][2], npb[2] = S.m[2][2];
        break;
    }

    const double traceS = S.m[0][0] + S.m[1][1] + S.m[2][2];
    // OpenCV really calls sqrtf() here (single precision), restated as such
    const double v = 2.0 * (double)sqrtf((float)(1 + traceS - M00 - M11 - M22));
    const double ESii = signd(S.m[indx][indx]);
    const double r_2 = 2 + traceS + v;
    const double nt_2 = 2 + traceS - v;
    const double r = std::sqrt(r_2);
    const double n_t = std::sqrt(nt_2);

    const double na_n = std::sqrt(npa[0] * npa[0] + npa[1] * npa[1] + npa[2] * npa[2]);
    const double nb_n = std::sqrt(npb[0] * npb[0] + npb[1] * npb[1] + npb[2] * npb[2]);
    const Vec3 na{npa[0] / na_n, npa[1] / na_n, npa[2] / na_n};
    const Vec3 nb{npb[0] / nb_n, npb[1] / nb_n, npb[2] / nb_n};

    const double half_nt = 0.5 * n_t;
    const double esii_t_r = ESii * r;
    const Vec3 ta_star = (nb * esii_t_r - na * n_t) * half_nt;
    const Vec3 tb_star = (na * esii_t_r - nb * n_t) * half_nt;

    const Mat3 Ra = find_rmat_from_tstar_n(Hn, ta_star, na, v);
    const Vec3 ta = mul(Ra, ta_star);
    const Mat3 Rb = find_rmat_from_tstar_n(Hn, tb_star, nb, v);
    const Vec3 tb = mul(Rb, tb_star);

    out[0] = cam_motion{Ra, ta, na};
    out[1] = cam_motion{Ra, ta * -1.0, na * -1.0};
    out[2] = cam_motion{Rb, tb, nb};
    out[3] = cam_motion{Rb, tb * -1.0, nb * -1.0};
    return 4;
}

bool homography_model::decompose(const std::vector<correspondence> &corrs, const std::vector<bool> &inliers,
                                 std::array<decomposed_pose, 4> &poses) // homography_model.cpp:138-185
{
    cam_motion motions[4];
    const size_t solutions = decompose_homography_mat(homography, motions);

    for (size_t i = 0; i < solutions; i++)
    {
        const Mat3 &R = motions[i].R;
        const Vec3 &T = motions[i].t;
        const Vec3 &N = motions[i].n;
        poses[i].score = 0;
        for (size_t j = 0; j < corrs.size(); j++)
        {
            if (!inliers[j])
                continue;
            const double dot1 = dot(N, corrs[j].measurement1);
            const double dot2 = dot(mul(R, N), corrs[j].measurement2);
            if (dot1 >= 0 && dot2 >= 0)
                poses[i].score++;
        }
        poses[i].orientation = quaternion_from_matrix(R);
        poses[i].position = T;
    }
    for (size_t i = solutions; i < poses.size(); i++)
        poses[i].score = -1;
    // same libstdc++ std::stable_sort, same (non-strict) comparator as the reference
    std::stable_sort(poses.begin(), poses.end(),
                     [](const decomposed_pose &p1, const decomposed_pose &p2) { return p1.score >= p2.score; });
    return poses[0].score > 0;
}

// ------------------------------------------------------------------------------ ransac.cpp:32-51
static inline double fast_pow4(double d)
{
    double t = d * d;
    return t * t;
}

// ransac.cpp:53-257, Model = homography_model (the only instantiation the pipeline calls,
// link_stage.cpp:91-93).  has_check_sample_degeneracy = true, has_check_degeneracy = false.
double ransac(const std::vector<correspondence> &matches, homography_model &model, std::vector<bool> &inliers,
              ransac_trace *trace)
{
    using Model = homography_model;
    const size_t MIN_ITERATIONS = 20;
    const size_t MAX_ITERATIONS = 10000;
    const size_t MAX_INNER_ITERATIONS = 5;
    const double PROBABILITY = 0.999;

    const double log_1m_p = std::log(1 - PROBABILITY);

    inliers.resize(matches.size());
    std::fill(inliers.begin(), inliers.end(), false);

    if (matches.size() < Model::MINIMUM_POINTS)
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

    size_t prosac_n = has_quality ? Model::MINIMUM_POINTS : matches.size();

    auto map_idx = [&sorted_idx, has_quality](size_t i) -> size_t { return has_quality ? sorted_idx[i] : i; };

    auto random_k_from_n = [&generator, &map_idx](size_t pool) {
        std::array<size_t, Model::MINIMUM_POINTS> indices;
        std::uniform_int_distribution<size_t> dist(0, pool - 1);
        for (size_t j = 0; j < Model::MINIMUM_POINTS; j++)
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
        std::array<size_t, Model::MINIMUM_POINTS> indices;
        indices[0] = sorted_idx[pool - 1];
        std::uniform_int_distribution<size_t> dist(0, pool - 2);
        for (size_t j = 1; j < Model::MINIMUM_POINTS; j++)
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

    for (size_t i = 0; i < probability_iterations; i++)
    {
        if (trace)
            trace->iterations = i + 1;
        if (has_quality && prosac_n < matches.size() && i > 0 && i % 10 == 0)
            prosac_n++;

        std::array<size_t, Model::MINIMUM_POINTS> initial_indices;
        if (has_quality && prosac_n < matches.size() && prosac_n > Model::MINIMUM_POINTS)
            initial_indices = prosac_sample(prosac_n);
        else
            initial_indices = random_k_from_n(has_quality ? prosac_n : matches.size());
        if (trace)
            trace->samples.push_back(initial_indices);

        if (Model::checkSampleDegeneracy(matches, initial_indices))
            continue;

        model.fit(matches, initial_indices);

        double score = 0;
        size_t checked = 0;
        bool rejected = false;
        std::fill(candidate_inliers.begin(), candidate_inliers.end(), false);
        for (size_t idx : eval_order)
        {
            double e = model.error(matches[idx]);
            if (e < model.inlier_threshold)
            {
                candidate_inliers[idx] = true;
                double ratio = e / model.inlier_threshold;
                score += 1.0 - ratio * ratio;
            }
            checked++;
            if (checked > 20 && best_score > 0 &&
                score < best_score * static_cast<double>(checked) / matches.size() * 0.6)
            {
                rejected = true;
                break;
            }
        }
        if (rejected)
            continue;

        if (score > best_score)
        {
            if (trace)
                trace->improvements++;
            best_model = model;
            best_score = score;
            inliers = candidate_inliers;

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

            double omega = best_score / matches.size();
            double omega_n = fast_pow4(omega);
            double log_1m_omega_n = std::log(1 - omega_n);
            probability_iterations =
                std::max(MIN_ITERATIONS, std::min(MAX_ITERATIONS, static_cast<size_t>(log_1m_p / log_1m_omega_n)));
        }
    }

    model = best_model;
    return model.evaluate(matches, inliers) / matches.size();
}

void assembleInliers(const std::vector<feature_match> &matches, const std::vector<bool> &inliers,
                     const std::vector<feature_2d> &source_features, const std::vector<feature_2d> &dest_features,
                     std::vector<feature_match_denormalized> &inlier_list)
{
    inlier_list.reserve(std::count(inliers.begin(), inliers.end(), true));
    for (size_t i = 0; i < matches.size(); i++)
        if (inliers[i])
        {
            feature_match_denormalized fmd;
            fmd.pixel_1[0] = source_features[matches[i].feature_index_1].location[0];
            fmd.pixel_1[1] = source_features[matches[i].feature_index_1].location[1];
            fmd.pixel_2[0] = dest_features[matches[i].feature_index_2].location[0];
            fmd.pixel_2[1] = dest_features[matches[i].feature_index_2].location[1];
            fmd.feature_index_1 = matches[i].feature_index_1;
            fmd.feature_index_2 = matches[i].feature_index_2;
            fmd.match_index = i;
            inlier_list.push_back(fmd);
        }
}

} // namespace oracle
