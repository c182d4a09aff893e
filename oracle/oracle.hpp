// ORACLE — TEST INFRASTRUCTURE ONLY.
//
// CPU restatement of the opencalibration hot path (extract -> match -> model_inliers -> relax), used
// only as the checker for the MI355X implementation in opencalibration_amd/.  Nothing under
// opencalibration_amd/ may include, link or call anything in this directory; only tests/,
// __graft_entry__.smoke() and bench.py's cpu_baseline leg do.
//
// The real reference cannot be compiled here (Eigen, Ceres, OpenCV, GDAL, spdlog, GTest absent; see
// DESIGN.md "Oracle"), so every function restates the reference source it cites (paths relative to
// /root/reference) and is pinned by restated versions of the reference's own synthetic unit tests
// (tests/test_oracle_*.py).  Where the arithmetic lives in a third-party library (Eigen evaluation
// order, OpenCV decomposeHomographyMat, Ceres) the published algorithm is restated and the pin is the
// reference test's tolerance, not bit-exactness: those spots are marked [3P].
//
// libstdc++ is on the result path of the reference (std::sort with ties, std::shuffle,
// std::default_random_engine, std::uniform_int_distribution) and this oracle calls the very same
// functions from the very same libstdc++ (GCC 11), which is why it is C++ and not C.
#pragma once

#include <array>
#include <bitset>
#include <cmath>
#include <cstddef>
#include <cstdint>
#include <limits>
#include <vector>

namespace oracle
{

// ---------------------------------------------------------------- small fixed-size linear algebra
// Restates the handful of Eigen operations the reference uses.  Evaluation order is plain
// left-to-right with separate multiply and add (the reference is built for generic x86-64: no FMA,
// CMakeLists.txt:59,68-74).  [3P] Eigen's exact unrolling order cannot be checked here.
struct Vec2
{
    double x = 0, y = 0;
};
struct Vec3
{
    double x = 0, y = 0, z = 0;
};
struct Mat3
{
    double m[3][3];
    double &operator()(int r, int c)
    {
        return m[r][c];
    }
    double operator()(int r, int c) const
    {
        return m[r][c];
    }
};

inline Vec3 operator-(const Vec3 &a, const Vec3 &b)
{
    return {a.x - b.x, a.y - b.y, a.z - b.z};
}
inline Vec3 operator+(const Vec3 &a, const Vec3 &b)
{
    return {a.x + b.x, a.y + b.y, a.z + b.z};
}
inline Vec3 operator*(const Vec3 &a, double s)
{
    return {a.x * s, a.y * s, a.z * s};
}
inline Vec3 operator/(const Vec3 &a, double s)
{
    return {a.x / s, a.y / s, a.z / s};
}
inline double dot(const Vec3 &a, const Vec3 &b)
{
    return a.x * b.x + a.y * b.y + a.z * b.z;
}
inline Vec3 cross(const Vec3 &a, const Vec3 &b)
{
    return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x};
}
inline double norm(const Vec3 &a)
{
    return std::sqrt(dot(a, a));
}
inline Vec3 normalized(const Vec3 &a)
{
    // Eigen normalized(): z = squaredNorm(); if (z > 0) return *this / sqrt(z)
    double z = dot(a, a);
    if (z > 0)
        return a / std::sqrt(z);
    return a;
}
inline Vec2 hnormalized(const Vec3 &a)
{
    return {a.x / a.z, a.y / a.z};
}
inline Vec3 mul(const Mat3 &A, const Vec3 &v)
{
    return {A.m[0][0] * v.x + A.m[0][1] * v.y + A.m[0][2] * v.z, A.m[1][0] * v.x + A.m[1][1] * v.y + A.m[1][2] * v.z,
            A.m[2][0] * v.x + A.m[2][1] * v.y + A.m[2][2] * v.z};
}
inline Mat3 mul(const Mat3 &A, const Mat3 &B)
{
    Mat3 C;
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++)
            C.m[i][j] = A.m[i][0] * B.m[0][j] + A.m[i][1] * B.m[1][j] + A.m[i][2] * B.m[2][j];
    return C;
}
inline Mat3 transpose(const Mat3 &A)
{
    Mat3 C;
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++)
            C.m[i][j] = A.m[j][i];
    return C;
}
inline double det(const Mat3 &A)
{
    return A.m[0][0] * (A.m[1][1] * A.m[2][2] - A.m[1][2] * A.m[2][1]) -
           A.m[0][1] * (A.m[1][0] * A.m[2][2] - A.m[1][2] * A.m[2][0]) +
           A.m[0][2] * (A.m[1][0] * A.m[2][1] - A.m[1][1] * A.m[2][0]);
}
Mat3 inverse3(const Mat3 &A); // Eigen compute_inverse_size3 (cofactors * 1/det)
Mat3 identity3();
double frobenius(const Mat3 &A);

// Eigen FullPivLU<...>::solve restated for a (rows x 9) system, column-major scan order for the
// pivot search with strict '>' (first maximum wins), rank threshold eps*diagonalSize*maxpivot.
// A is row-major rows x 9 and is destroyed.  homography_model.cpp:44,81 call sites.
void full_piv_lu_solve9(double *A, size_t rows, const double *rhs, double out[9]);

// ---------------------------------------------------------------- types (include/opencalibration/types)
struct feature_2d // feature_2d.hpp:9-21
{
    static constexpr int DESCRIPTOR_BITS = 486;
    double location[2] = {NAN, NAN};
    float strength = 0;
    std::bitset<DESCRIPTOR_BITS> descriptor;
};
static_assert(sizeof(feature_2d) == 88, "feature_2d layout differs from the reference's 88-byte record");

struct feature_match // feature_match.hpp:11-23
{
    size_t feature_index_1, feature_index_2;
    double distance;
};
struct feature_match_denormalized // feature_match.hpp:26-39
{
    double pixel_1[2], pixel_2[2];
    size_t feature_index_1, feature_index_2, match_index;
};
struct correspondence // correspondence.hpp:8-13
{
    Vec3 measurement1, measurement2;
    double quality = 0;
};
struct camera_model // camera_model.hpp:22-60 (PLANAR projection only)
{
    size_t pixels_rows = 0, pixels_cols = 0;
    double focal_length_pixels = 0;
    double principle_point[2] = {0, 0};
    double radial_distortion[3] = {0, 0, 0};
    double tangential_distortion[2] = {0, 0};
};
struct Quat // Eigen::Quaterniond coefficient order x,y,z,w
{
    double x = NAN, y = NAN, z = NAN, w = NAN;
};
struct decomposed_pose // decomposed_pose.hpp:7-21
{
    Quat orientation;
    Vec3 position{NAN, NAN, NAN};
    int score = 0;
};

// ---------------------------------------------------------------- match (src/match/match_features.cpp)
std::vector<size_t> spatially_subsample_feature_indices(const std::vector<feature_2d> &features, double spacing_pixels,
                                                        size_t count = 0);
std::vector<feature_match> match_features_subset(const std::vector<feature_2d> &set_1,
                                                 const std::vector<feature_2d> &set_2,
                                                 const std::vector<size_t> &indices_1,
                                                 const std::vector<size_t> &indices_2);

// ---------------------------------------------------------------- distort (src/distort/distort_keypoints.cpp)
Vec3 image_to_3d(const double keypoint[2], const camera_model &model);
Vec2 image_from_3d(const Vec3 &ray, const camera_model &model); // forward model, distort_keypoints.hpp:44-66
std::vector<correspondence> distort_keypoints(const std::vector<feature_2d> &f1, const std::vector<feature_2d> &f2,
                                              const std::vector<feature_match> &matches, const camera_model &m1,
                                              const camera_model &m2);

// ---------------------------------------------------------------- model_inliers
struct homography_model // homography_model.hpp:16-37, homography_model.cpp
{
    static constexpr size_t MINIMUM_POINTS = 4;
    homography_model();
    void fit(const std::vector<correspondence> &corrs, const std::array<size_t, 4> &initial_indices);
    void fitInliers(const std::vector<correspondence> &corrs, const std::vector<bool> &inliers);
    double evaluate(const std::vector<correspondence> &corrs, std::vector<bool> &inliers);
    bool decompose(const std::vector<correspondence> &corrs, const std::vector<bool> &inliers,
                   std::array<decomposed_pose, 4> &poses);
    double error(const correspondence &cor);
    static bool checkSampleDegeneracy(const std::vector<correspondence> &corrs, const std::array<size_t, 4> &indices);

    double inlier_threshold = 0.005;
    Mat3 homography, homography_inverse;
};

// fundamental_matrix_model.hpp:14-31 / essential_matrix_model.hpp:14-33 (oracle/epipolar.cpp): the two RANSAC models the
// reference's tests exercise and its pipeline never calls
struct fundamental_matrix_model
{
    fundamental_matrix_model();
    static constexpr size_t MINIMUM_POINTS = 8;
    void fit(const std::vector<correspondence> &corrs, const std::array<size_t, MINIMUM_POINTS> &initial_indices);
    void fitInliers(const std::vector<correspondence> &corrs, const std::vector<bool> &inliers);
    double evaluate(const std::vector<correspondence> &corrs, std::vector<bool> &inliers);
    double error(const correspondence &cor);
    void checkDegeneracy(const std::vector<correspondence> &corrs, std::vector<bool> &inliers);
    double inlier_threshold = 0.01;
    Mat3 fundamental_matrix;
};
struct essential_matrix_model
{
    essential_matrix_model();
    static constexpr size_t MINIMUM_POINTS = 5;
    void fit(const std::vector<correspondence> &corrs, const std::array<size_t, MINIMUM_POINTS> &initial_indices);
    void fitInliers(const std::vector<correspondence> &corrs, const std::vector<bool> &inliers);
    double evaluate(const std::vector<correspondence> &corrs, std::vector<bool> &inliers);
    double error(const correspondence &cor);
    bool decompose(const std::vector<correspondence> &corrs, const std::vector<bool> &inliers, std::array<decomposed_pose, 4> &poses);
    double inlier_threshold = 0.01;
    Mat3 essential_matrix;
};
double ransac(const std::vector<correspondence> &matches, fundamental_matrix_model &model, std::vector<bool> &inliers,
              size_t *iterations = nullptr); // ransac.cpp:53-257, with checkDegeneracy
double ransac(const std::vector<correspondence> &matches, essential_matrix_model &model, std::vector<bool> &inliers,
              size_t *iterations = nullptr);
// Eigen::JacobiSVD of a square matrix as restated in epipolar.cpp: A = U diag(S) V^T (row-major n x n), S decreasing
void jacobi_svd_square(const double *A, int n, double *U, double *S, double *V);

struct ransac_trace // optional instrumentation for golden vectors (not in the reference)
{
    std::vector<std::array<size_t, 4>> samples; // minimal sample of every iteration, in order
    size_t iterations = 0;                      // loop trips executed
    size_t improvements = 0;
};
double ransac(const std::vector<correspondence> &matches, homography_model &model, std::vector<bool> &inliers,
              ransac_trace *trace = nullptr); // ransac.cpp:53-257
void assembleInliers(const std::vector<feature_match> &matches, const std::vector<bool> &inliers,
                     const std::vector<feature_2d> &source_features, const std::vector<feature_2d> &dest_features,
                     std::vector<feature_match_denormalized> &inlier_list); // ransac.cpp:263-282

// [3P] cv::decomposeHomographyMat(H, I) restated (OpenCV 4.x calib3d homography_decomp.cpp,
// Malis & Vargas INRIA RR-6303 analytical method).  Returns number of solutions (1 or 4).
struct cam_motion
{
    Mat3 R;
    Vec3 t, n;
};
size_t decompose_homography_mat(const Mat3 &H, cam_motion out[4]);
Quat quaternion_from_matrix(const Mat3 &R); // Eigen::Quaterniond(Matrix3d) [3P]

// ---------------------------------------------------------------- link stage, one directed pair
struct camera_relations // camera_relations.hpp:13-35
{
    std::vector<feature_match_denormalized> inlier_matches;
    std::vector<feature_match> matches;
    Mat3 ransac_relation;
    std::array<decomposed_pose, 4> relative_poses;
    // extras for parity checks (not in the reference struct)
    std::vector<bool> coarse_inliers;
    size_t num_coarse_matches = 0;
    bool can_decompose = false;
    double ransac_score = 0;
    size_t ransac_iterations = 0, ransac_improvements = 0;
};
// link_stage.cpp:75-112 body of the per-pair closure.  idx1/idx2 are the 40 px subsets (the reference
// recomputes idx2 per pair, link_stage.cpp:80-81; the result only depends on the image).
camera_relations link_pair(const std::vector<feature_2d> &f1, const std::vector<feature_2d> &f2,
                           const std::vector<size_t> &idx1, const std::vector<size_t> &idx2, const camera_model &m1,
                           const camera_model &m2);

} // namespace oracle
