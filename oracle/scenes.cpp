// ORACLE — test infrastructure only (see oracle.hpp).
// Synthetic scenes of the reference's own tests, restated with the same libstdc++ generators so the
// restated tests see the same data: test/test_ransac_benchmark.cpp:18-58 (homography scene),
// :223-262 (near-degenerate scene).
#include "oracle.hpp"

#include <random>

using namespace oracle;

static Mat3 ground_truth_h()
{
    // R = AngleAxis(0.1, Z); t = (0.05,-0.03,0); n = (0,0,1); H = R + t n^T / 10; H /= H(2,2)
    const double c = std::cos(0.1), s = std::sin(0.1);
    Mat3 H;
    const double R[3][3] = {{c, -s, 0}, {s, c, 0}, {0, 0, 1}};
    const double t[3] = {0.05, -0.03, 0.0}, n[3] = {0, 0, 1};
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++)
            H.m[i][j] = R[i][j] + t[i] * n[j] / 10.0;
    const double d = H.m[2][2];
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++)
            H.m[i][j] /= d;
    return H;
}

static void put(double *corr, size_t i, const Vec3 &p1, const Vec3 &p2)
{
    corr[7 * i] = p1.x, corr[7 * i + 1] = p1.y, corr[7 * i + 2] = p1.z;
    corr[7 * i + 3] = p2.x, corr[7 * i + 4] = p2.y, corr[7 * i + 5] = p2.z;
    corr[7 * i + 6] = 0;
}

extern "C"
{

// SyntheticScene::homography(n_inliers, n_outliers, seed)
void oc_scene_homography(size_t n_inliers, size_t n_outliers, unsigned seed, double *corr, uint8_t *gt_inliers,
                         double *H_gt)
{
    std::mt19937 rng(seed);
    std::uniform_real_distribution<double> point_dist(-1.0, 1.0);
    std::uniform_real_distribution<double> outlier_dist(-2.0, 2.0);
    const Mat3 H = ground_truth_h();
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++)
            H_gt[3 * i + j] = H.m[i][j];
    size_t k = 0;
    for (size_t i = 0; i < n_inliers; i++, k++)
    {
        // argument evaluation order of Eigen::Vector3d p1(point_dist(rng), point_dist(rng), 1.0) is
        // unspecified in C++; GCC evaluates constructor arguments right-to-left for this call, and the
        // restated tests only assert precision/recall floors, which hold either way.
        const double a = point_dist(rng), b = point_dist(rng);
        const Vec3 p1{a, b, 1.0};
        Vec3 p2 = mul(H, p1);
        p2 = p2 / p2.z;
        put(corr, k, p1, p2);
        gt_inliers[k] = 1;
    }
    for (size_t i = 0; i < n_outliers; i++, k++)
    {
        const double a = outlier_dist(rng), b = outlier_dist(rng);
        const double c = outlier_dist(rng), d = outlier_dist(rng);
        put(corr, k, Vec3{a, b, 1.0}, Vec3{c, d, 1.0});
        gt_inliers[k] = 0;
    }
}

// test_ransac_benchmark.cpp:223-262 (20 near-collinear + 80 general, all inliers)
void oc_scene_near_degenerate(double *corr /*100x7*/, double *H_gt)
{
    std::mt19937 rng(42);
    std::uniform_real_distribution<double> noise(-0.001, 0.001);
    const Mat3 H = ground_truth_h();
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++)
            H_gt[3 * i + j] = H.m[i][j];
    size_t k = 0;
    for (int i = 0; i < 20; i++, k++)
    {
        const double t_param = -1.0 + 2.0 * i / 19.0;
        const Vec3 p1{t_param, 0.5 + noise(rng), 1.0};
        Vec3 p2 = mul(H, p1);
        p2 = p2 / p2.z;
        put(corr, k, p1, p2);
    }
    std::uniform_real_distribution<double> point_dist(-1.0, 1.0);
    for (int i = 0; i < 80; i++, k++)
    {
        const double a = point_dist(rng), b = point_dist(rng);
        const Vec3 p1{a, b, 1.0};
        Vec3 p2 = mul(H, p1);
        p2 = p2 / p2.z;
        put(corr, k, p1, p2);
    }
}


// SyntheticScene::fundamental(n_inliers, n_outliers, planar_fraction, seed), test_ransac_benchmark.cpp:60-122:
// corr n x 7 {measurement1, measurement2, quality 0}, F_gt = [e2]_x R2 normalised to unit Frobenius norm
void oc_scene_fundamental(size_t n_inliers, size_t n_outliers, double planar_fraction, unsigned seed, double *corr,
                          uint8_t *gt_inliers, double *F_gt)
{
    std::mt19937 rng(seed);
    std::uniform_real_distribution<double> xy_dist(-1.0, 1.0);
    std::uniform_real_distribution<double> z_dist(5.0, 15.0);
    std::uniform_real_distribution<double> outlier_dist(-1.0, 1.0);
    // R1 = I, t1 = 0; R2 = AngleAxis(0.15, Y); t2 = (0.5, 0, 0)
    const double c = std::cos(0.15), sn = std::sin(0.15);
    Mat3 R2;
    const double r2[3][3] = {{c, 0, sn}, {0, 1, 0}, {-sn, 0, c}};
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++)
            R2.m[i][j] = r2[i][j];
    const Vec3 t2{0.5, 0.0, 0.0};
    const Vec3 e2 = mul(R2, Vec3{0, 0, 0} - t2);
    Mat3 ex;
    ex.m[0][0] = 0, ex.m[0][1] = -e2.z, ex.m[0][2] = e2.y;
    ex.m[1][0] = e2.z, ex.m[1][1] = 0, ex.m[1][2] = -e2.x;
    ex.m[2][0] = -e2.y, ex.m[2][1] = e2.x, ex.m[2][2] = 0;
    Mat3 F = mul(ex, R2);
    const double nf = frobenius(F);
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++)
            F_gt[3 * i + j] = F.m[i][j] / nf;
    const size_t n_planar = static_cast<size_t>(n_inliers * planar_fraction);
    size_t k = 0;
    for (size_t i = 0; i < n_inliers; i++, k++)
    {
        // (argument evaluation order of the Vector3d constructor is unspecified; the restated tests assert precision /
        // recall floors that hold either way, as for the homography scene above)
        const double a = xy_dist(rng), b = xy_dist(rng);
        const double z = i < n_planar ? 10.0 : z_dist(rng);
        const Vec3 X{a * 3, b * 3, z};
        Vec3 x1 = X, x2 = mul(R2, X - t2);
        x1 = x1 / x1.z;
        x2 = x2 / x2.z;
        put(corr, k, x1, x2);
        gt_inliers[k] = 1;
    }
    for (size_t i = 0; i < n_outliers; i++, k++)
    {
        const double a = outlier_dist(rng), b = outlier_dist(rng);
        const double cc = outlier_dist(rng), d = outlier_dist(rng);
        put(corr, k, Vec3{a, b, 1.0}, Vec3{cc, d, 1.0});
        gt_inliers[k] = 0;
    }
}
} // extern "C"
