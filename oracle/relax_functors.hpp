// ORACLE — test infrastructure only (see oracle.hpp).
// Cost functors of include/opencalibration/relax/relax_cost_function.hpp and the geometry templates of
// include/opencalibration/geometry/intersection.hpp, restated on a scalar type T (double or Jet<N>).
#pragma once

#include "jet.hpp"
#include "oracle.hpp"

#include <algorithm>
#include <cmath>
#include <array>
#include <limits>
#include <vector>

namespace oracle
{

template <typename T> struct V3
{
    T x, y, z;
};
template <typename T> inline V3<T> operator+(const V3<T> &a, const V3<T> &b)
{
    return {a.x + b.x, a.y + b.y, a.z + b.z};
}
template <typename T> inline V3<T> operator-(const V3<T> &a, const V3<T> &b)
{
    return {a.x - b.x, a.y - b.y, a.z - b.z};
}
template <typename T> inline V3<T> operator*(const V3<T> &a, const T &s)
{
    return {a.x * s, a.y * s, a.z * s};
}
template <typename T> inline V3<T> operator/(const V3<T> &a, const T &s)
{
    return {a.x / s, a.y / s, a.z / s};
}
template <typename T> inline T dot3(const V3<T> &a, const V3<T> &b)
{
    return a.x * b.x + a.y * b.y + a.z * b.z;
}
template <typename T> inline V3<T> cross3(const V3<T> &a, const V3<T> &b)
{
    return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x};
}
template <typename T> inline T norm3(const V3<T> &a)
{
    using std::sqrt;
    return sqrt(dot3(a, a));
}
template <typename T> inline V3<T> normalized3(const V3<T> &a)
{
    const T z = dot3(a, a);
    if (z > T(0.0))
    {
        using std::sqrt;
        return a / sqrt(z);
    }
    return a;
}
// Eigen QuaternionBase::_transformVector: v + w*(2 q×v) + q×(2 q×v), q stored x,y,z,w (not normalised)
template <typename T> inline V3<T> quat_rotate(const T *q, const V3<T> &v)
{
    const V3<T> qv{q[0], q[1], q[2]};
    V3<T> uv = cross3(qv, v);
    uv = uv + uv;
    return v + uv * q[3] + cross3(qv, uv);
}

// relax_cost_function.hpp:16-19
template <typename T> inline T angleBetweenUnitVectors(const V3<T> &n1, const V3<T> &n2)
{
    using std::acos;
    T d = dot3(n1, n2);
    const T lo = T(-1 + 1e-12), hi = T(1 - 1e-12);
    // std::clamp(v, lo, hi): (v < lo) ? lo : (hi < v) ? hi : v
    const T c = (d < lo) ? lo : ((hi < d) ? hi : d);
    return acos(c);
}

// relax_cost_function.hpp:21-49
struct PointsDownwardsPrior
{
    explicit PointsDownwardsPrior(double weight) : _weight(weight)
    {
    }
    template <typename T> bool operator()(const T *rotation1, T *residuals) const
    {
        const V3<T> cam_center{T(0.0), T(0.0), T(1.0)};
        const V3<T> down{T(0.0), T(0.0), T(-1.0)};
        const V3<T> rotated = quat_rotate(rotation1, cam_center);
        residuals[0] = T(_weight) * angleBetweenUnitVectors<T>(rotated, down);
        return true;
    }
    double _weight;
};

constexpr int ROBUST_CENTROID_MAX_POINTS = 5;

// relax_cost_function.hpp:73-117
template <typename T> V3<T> robustCentroid(const V3<T> *points, int n, T huber_threshold)
{
    constexpr int MAX_STAGES = 3;
    V3<T> centroid{T(0.0), T(0.0), T(0.0)};
    for (int i = 0; i < n; i++)
        centroid = centroid + points[i];
    centroid = centroid / T(double(n));

    T weights[ROBUST_CENTROID_MAX_POINTS];
    for (int i = 0; i < n; i++)
        weights[i] = T(1.0);

    for (int stage = 0; stage < MAX_STAGES; stage++)
    {
        T total_w = T(0.0);
        T min_w = T(std::numeric_limits<double>::max());
        T max_w = T(0.0);
        for (int i = 0; i < n; i++)
        {
            T err = norm3(points[i] - centroid);
            T w = T(1.0) / (err + T(1e-8));
            if (err > huber_threshold)
                w = w * (huber_threshold / err);
            weights[i] = w;
            total_w = total_w + w;
            if (w < min_w)
                min_w = w;
            if (w > max_w)
                max_w = w;
        }
        V3<T> weighted_sum{T(0.0), T(0.0), T(0.0)};
        for (int i = 0; i < n; i++)
            weighted_sum = weighted_sum + points[i] * weights[i];
        centroid = weighted_sum / total_w;
        if (min_w > max_w * T(0.5))
            break;
    }
    return centroid;
}

// geometry/intersection.hpp:26-47
template <typename T> struct plane_norm_offset
{
    V3<T> norm, offset;
};
template <typename T> inline plane_norm_offset<T> cornerPlane2normOffsetPlane(const V3<T> corner[3])
{
    plane_norm_offset<T> out;
    out.offset = corner[0];
    out.norm = normalized3(cross3(corner[0] - corner[1], corner[0] - corner[2]));
    return out;
}
template <typename T>
inline bool rayPlaneIntersection(const V3<T> &dir, const V3<T> &offset, const plane_norm_offset<T> &p, V3<T> &out)
{
    const T denom = dot3(p.norm, dir);
    using std::abs;
    if (abs(denom) < T(1e-9))
    {
        out = {T(NAN), T(NAN), T(NAN)};
        return false;
    }
    const T t = (dot3(p.norm, p.offset) - dot3(offset, p.norm)) / denom;
    out = offset + dir * t;
    return true;
}

// relax_cost_function.hpp:601-656, N = 2 (PlaneIntersectionAngleCost :658-684)
struct PlaneIntersectionAngleCost
{
    double camera_loc[2][3], camera_ray[2][3], plane_point[3][2];
    template <typename T>
    bool operator()(const T *rotation0, const T *rotation1, const T *z0, const T *z1, const T *z2, T *residuals) const
    {
        constexpr int N = 2;
        const T *rotations[2] = {rotation0, rotation1};
        const T plane_z[3] = {*z0, *z1, *z2};
        V3<T> corner[3];
        for (int i = 0; i < 3; i++)
            corner[i] = {T(plane_point[i][0]), T(plane_point[i][1]), plane_z[i]};
        const plane_norm_offset<T> pno = cornerPlane2normOffsetPlane(corner);

        V3<T> intersection[N];
        bool all_valid = true;
        T avg_dist = T(0.0);
        for (int i = 0; i < N; i++)
        {
            const V3<T> ray_cam{T(camera_ray[i][0]), T(camera_ray[i][1]), T(camera_ray[i][2])};
            const V3<T> dir = quat_rotate(rotations[i], ray_cam);
            const V3<T> off{T(camera_loc[i][0]), T(camera_loc[i][1]), T(camera_loc[i][2])};
            all_valid &= rayPlaneIntersection(dir, off, pno, intersection[i]);
            avg_dist = avg_dist + norm3(intersection[i] - off);
        }
        avg_dist = avg_dist / T(double(N));
        const T huber_threshold = avg_dist * T(0.01);
        const V3<T> centroid = robustCentroid(intersection, N, huber_threshold);
        for (int i = 0; i < N; i++)
        {
            const V3<T> r = (intersection[i] - centroid) / avg_dist;
            residuals[i * 3 + 0] = r.x;
            residuals[i * 3 + 1] = r.y;
            residuals[i * 3 + 2] = r.z;
        }
        return all_valid;
    }
};


// relax_cost_function.hpp:51-69
struct DifferenceCost
{
    explicit DifferenceCost(double weight) : _weight(weight)
    {
    }
    template <typename T> bool operator()(const T *val1, const T *val2, T *residual) const
    {
        residual[0] = T(_weight) * (val1[0] - val2[0]);
        return true;
    }
    double _weight;
};

// relax_cost_function.hpp:119-155
struct AdjacentTriangleNormalCost
{
    double xyA[2], xyB[2], xyC[2], xyD[2], weight;
    template <typename T> bool operator()(const T *zA, const T *zB, const T *zC, const T *zD, T *residuals) const
    {
        const V3<T> A{T(xyA[0]), T(xyA[1]), *zA}, B{T(xyB[0]), T(xyB[1]), *zB}, C{T(xyC[0]), T(xyC[1]), *zC},
            D{T(xyD[0]), T(xyD[1]), *zD};
        const V3<T> AB = B - A;
        const V3<T> n1 = normalized3(cross3(AB, C - A));
        const V3<T> n2 = normalized3(cross3(AB, D - A));
        residuals[0] = T(weight) * angleBetweenUnitVectors<T>(n1, n2);
        return true;
    }
};

// relax_cost_function.hpp:157-185
struct DistortionMonotonicityCost
{
    double r_max, weight;
    template <typename T> bool operator()(const T *radial, T *residuals) const
    {
        for (int i = 0; i < 10; i++)
        {
            const T r = T(r_max * (i + 1.0) / 10);
            const T r2 = r * r;
            const T r4 = r2 * r2;
            const T r6 = r4 * r2;
            const T deriv = T(1.0) + T(3.0) * radial[0] * r2 + T(5.0) * radial[1] * r4 + T(7.0) * radial[2] * r6;
            residuals[i] = deriv < T(0.0) ? T(weight) * (-deriv) : T(0.0);
        }
        return true;
    }
};

// distort_keypoints.hpp:26-42 on a scalar type, all coefficients of type T
template <typename T> inline void distortProjectedRayT(const T p[2], const T radial[3], const T tangential[2], T out[2])
{
    T r2[3];
    r2[0] = p[0] * p[0] + p[1] * p[1];
    for (int i = 1; i < 3; i++)
        r2[i] = r2[i - 1] * r2[0];
    const T radial_dot = radial[0] * r2[0] + radial[1] * r2[1] + radial[2] * r2[2];
    const T prod = p[0] * p[1];
    for (int i = 0; i < 2; i++)
        out[i] = (T(1.0) + radial_dot) * p[i] + T(2.0) * prod * tangential[i] +
                 tangential[1 - i] * (r2[0] + T(2.0) * p[i] * p[i]);
}

// InverseDifferentiableCameraModel<T> (camera_model.hpp:22-66) and image_to_3d on it (distort_keypoints.hpp:97-116):
// pixel -> ray in closed form
template <typename T> struct inverse_model_t
{
    T focal_length_pixels;
    T principle_point[2];
    T radial_distortion[3];
    T tangential_distortion[2];
};
template <typename T> inline V3<T> image_to_3d_inverse(const T keypoint[2], const inverse_model_t<T> &m)
{
    const T unprojected[2] = {(keypoint[0] - m.principle_point[0]) / m.focal_length_pixels,
                              (keypoint[1] - m.principle_point[1]) / m.focal_length_pixels};
    T und[2];
    distortProjectedRayT<T>(unprojected, m.radial_distortion, m.tangential_distortion, und);
    return normalized3(V3<T>{und[0], und[1], T(1.0)});
}

// relax_cost_function.hpp:601-656 (fixed intrinsics) and :501-566 (FocalRadial: rays recomputed from the pixels
// through the shared inverse model) for N = 2..5 rays
template <int N> struct MultiRayCost
{
    double camera_loc[N][3], camera_ray[N][3], camera_pixel[N][2], plane_point[3][2];
    double shared_tangential[2] = {0, 0}; // the part of sharedModel that is not a parameter block

    template <typename T>
    bool finish(const V3<T> dir[N], const T *z0, const T *z1, const T *z2, T *residuals) const
    {
        const T plane_z[3] = {*z0, *z1, *z2};
        V3<T> corner[3];
        for (int i = 0; i < 3; i++)
            corner[i] = {T(plane_point[i][0]), T(plane_point[i][1]), plane_z[i]};
        const plane_norm_offset<T> pno = cornerPlane2normOffsetPlane(corner);
        V3<T> intersection[N];
        bool all_valid = true;
        T avg_dist = T(0.0);
        for (int i = 0; i < N; i++)
        {
            const V3<T> off{T(camera_loc[i][0]), T(camera_loc[i][1]), T(camera_loc[i][2])};
            all_valid &= rayPlaneIntersection(dir[i], off, pno, intersection[i]);
            avg_dist = avg_dist + norm3(intersection[i] - off);
        }
        avg_dist = avg_dist / T(double(N));
        const T huber_threshold = avg_dist * T(0.01);
        const V3<T> centroid = robustCentroid(intersection, N, huber_threshold);
        for (int i = 0; i < N; i++)
        {
            const V3<T> r = (intersection[i] - centroid) / avg_dist;
            residuals[i * 3 + 0] = r.x;
            residuals[i * 3 + 1] = r.y;
            residuals[i * 3 + 2] = r.z;
        }
        return all_valid;
    }
    template <typename T>
    bool computeResiduals(const T *const *rotations, const T *z0, const T *z1, const T *z2, T *residuals) const
    {
        V3<T> dir[N];
        for (int i = 0; i < N; i++)
            dir[i] = quat_rotate(rotations[i], V3<T>{T(camera_ray[i][0]), T(camera_ray[i][1]), T(camera_ray[i][2])});
        return finish<T>(dir, z0, z1, z2, residuals);
    }
    template <typename T>
    bool computeResidualsFocalRadial(const T *const *rotations, const T *z0, const T *z1, const T *z2, const T *focal,
                                     const T *principal, const T *radial, T *residuals) const
    {
        inverse_model_t<T> model;
        model.focal_length_pixels = *focal;
        model.principle_point[0] = principal[0], model.principle_point[1] = principal[1];
        for (int i = 0; i < 3; i++)
            model.radial_distortion[i] = radial[i];
        model.tangential_distortion[0] = T(shared_tangential[0]), model.tangential_distortion[1] = T(shared_tangential[1]);
        V3<T> dir[N];
        for (int i = 0; i < N; i++)
        {
            const T px[2] = {T(camera_pixel[i][0]), T(camera_pixel[i][1])};
            dir[i] = quat_rotate(rotations[i], image_to_3d_inverse<T>(px, model));
        }
        return finish<T>(dir, z0, z1, z2, residuals);
    }
};
// parameter orders of the wrappers: 2-ray (:658-684, :568-599): r0 r1 z0 z1 z2 [f pp k];
// N-ray (:686-735, :737-790): z0 z1 z2 [f pp k] r0 .. rN-1
struct TwoRayFocalRadial
{
    MultiRayCost<2> impl;
    template <typename T>
    bool operator()(const T *r0, const T *r1, const T *z0, const T *z1, const T *z2, const T *f, const T *pp, const T *k,
                    T *res) const
    {
        const T *rot[2] = {r0, r1};
        return impl.computeResidualsFocalRadial<T>(rot, z0, z1, z2, f, pp, k, res);
    }
};
template <int N> struct NRay
{
    MultiRayCost<N> impl;
    template <typename T, typename... R> bool operator()(const T *z0, const T *z1, const T *z2, const R *...rest) const
    {
        // rest = r0 .. rN-1, residuals
        const T *all[N + 1] = {rest...};
        return impl.template computeResiduals<T>(all, z0, z1, z2, const_cast<T *>(all[N]));
    }
};
template <int N> struct NRayFocalRadial
{
    MultiRayCost<N> impl;
    template <typename T, typename... R>
    bool operator()(const T *z0, const T *z1, const T *z2, const T *f, const T *pp, const T *k, const R *...rest) const
    {
        const T *all[N + 1] = {rest...};
        return impl.template computeResidualsFocalRadial<T>(all, z0, z1, z2, f, pp, k, const_cast<T *>(all[N]));
    }
};

// ---- the flavours only the reference's tests reach (relax.cpp:14-42 relative orientation, :104-115 3-D points) ----------
template <typename T> inline void quat_inverse(const T *q, T *out) // Eigen QuaternionBase::inverse(): conjugate / squaredNorm
{
    const T n2 = q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3];
    if (n2 > T(0.0))
    {
        out[0] = -q[0] / n2, out[1] = -q[1] / n2, out[2] = -q[2] / n2, out[3] = q[3] / n2;
    }
    else
        out[0] = out[1] = out[2] = out[3] = T(0.0);
}
template <typename T> inline void quat_product(const T *a, const T *b, T *out) // Eigen quaternion product a * b (x y z w)
{
    out[3] = a[3] * b[3] - a[0] * b[0] - a[1] * b[1] - a[2] * b[2];
    out[0] = a[3] * b[0] + a[0] * b[3] + a[1] * b[2] - a[2] * b[1];
    out[1] = a[3] * b[1] + a[1] * b[3] + a[2] * b[0] - a[0] * b[2];
    out[2] = a[3] * b[2] + a[2] * b[3] + a[0] * b[1] - a[1] * b[0];
}
template <typename T> inline T quat_angle(const T *q) // Eigen::AngleAxis<T>(q).angle()
{
    using std::abs;
    using std::atan2;
    using std::sqrt;
    const T n = sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2]);
    if (n != T(0.0))
        return T(2.0) * atan2(n, abs(q[3]));
    return T(0.0);
}

// relax_cost_function.hpp:187-251
struct DecomposedRotationCost
{
    DecomposedRotationCost(const Quat &relative_rotation, const Vec3 &relative_translation, const Vec3 &translation1,
                           const Vec3 &translation2, int score)
    {
        const Vec3 d = translation2 - translation1;
        has_translation = dot(d, d) > 1e-9 && dot(relative_translation, relative_translation) > 1e-9;
        const double qn = std::sqrt(relative_rotation.x * relative_rotation.x + relative_rotation.y * relative_rotation.y +
                                    relative_rotation.z * relative_rotation.z + relative_rotation.w * relative_rotation.w);
        rel[0] = relative_rotation.x / qn, rel[1] = relative_rotation.y / qn, rel[2] = relative_rotation.z / qn, rel[3] = relative_rotation.w / qn;
        const Vec3 td = normalized(d), rd = normalized(relative_translation);
        tdir[0] = td.x, tdir[1] = td.y, tdir[2] = td.z;
        rtdir[0] = rd.x, rtdir[1] = rd.y, rtdir[2] = rd.z;
        weight = std::sqrt(score / 8.);
    }
    template <typename T> bool operator()(const T *rotation1, const T *rotation2, T *residuals) const
    {
        T res[3];
        T inv1[4], inv2[4];
        quat_inverse(rotation1, inv1);
        quat_inverse(rotation2, inv2);
        if (has_translation)
        {
            const V3<T> t21 = quat_rotate(inv1, V3<T>{T(tdir[0]), T(tdir[1]), T(tdir[2])});
            res[0] = angleBetweenUnitVectors<T>(t21, V3<T>{T(rtdir[0]), T(rtdir[1]), T(rtdir[2])});
            // rotation2.inverse() * (relative_rotation * -translation_direction).cast<T>(): the inner product in doubles
            const V3<double> rt = quat_rotate<double>(rel, V3<double>{-tdir[0], -tdir[1], -tdir[2]});
            const V3<T> t12 = quat_rotate(inv2, V3<T>{T(rt.x), T(rt.y), T(rt.z)});
            res[1] = angleBetweenUnitVectors<T>(t12, V3<T>{T(-rtdir[0]), T(-rtdir[1]), T(-rtdir[2])});
        }
        else
            res[0] = res[1] = T(M_PI);
        T r21[4], relT[4] = {T(rel[0]), T(rel[1]), T(rel[2]), T(rel[3])}, prod[4];
        quat_product(rotation1, inv2, r21);
        quat_product(relT, r21, prod);
        res[2] = quat_angle(prod);
        for (int i = 0; i < 3; i++)
            residuals[i] = T(weight) * res[i];
        return true;
    }
    bool has_translation;
    double rel[4], tdir[3], rtdir[3], weight;
};

// relax_cost_function.hpp:253-307: the decomposition with the smallest residual norm among those scoring > max / 4
struct MultiDecomposedRotationCost
{
    MultiDecomposedRotationCost(const std::array<decomposed_pose, 4> &poses, const Vec3 &translation1, const Vec3 &translation2)
    {
        int max_score = 0;
        for (const auto &pose : poses)
            if (pose.score > max_score)
                max_score = pose.score;
        for (const auto &pose : poses)
            if (pose.score > 0.25 * max_score)
                decompose.emplace_back(pose.orientation, pose.position, translation1, translation2, pose.score);
    }
    template <typename T> bool operator()(const T *rotation1, const T *rotation2, T *residuals) const
    {
        using std::isfinite;
        T lowest_res_norm(std::numeric_limits<double>::infinity());
        T lowest_res[3] = {T(NAN), T(NAN), T(NAN)};
        for (const auto &d : decompose)
        {
            T res[3];
            if (!d(rotation1, rotation2, res))
                continue;
            const bool finite = isfinite(res[0]) && isfinite(res[1]) && isfinite(res[2]);
            const T n2 = res[0] * res[0] + res[1] * res[1] + res[2] * res[2];
            if (finite && n2 < lowest_res_norm)
            {
                lowest_res_norm = n2;
                for (int i = 0; i < 3; i++)
                    lowest_res[i] = res[i];
            }
        }
        for (int i = 0; i < 3; i++)
            residuals[i] = lowest_res[i];
        return isfinite(lowest_res_norm);
    }
    std::vector<DecomposedRotationCost> decompose;
};

// image_from_3d<T>(ray, model) (distort_keypoints.hpp:44-66) with every intrinsic of type T
template <typename T>
inline void image_from_3d_T(const V3<T> &ray, const T &focal, const T pp[2], const T radial[3], const T tangential[2], T pixel[2])
{
    const T min_z = T(1e-3);
    const T cz = (ray.z < min_z) ? min_z : ray.z;
    const T projected[2] = {ray.x / cz, ray.y / cz};
    T distorted[2];
    distortProjectedRayT<T>(projected, radial, tangential, distorted);
    pixel[0] = distorted[0] * focal + pp[0];
    pixel[1] = distorted[1] * focal + pp[1];
}

// relax_cost_function.hpp:309-500: reprojection error of a 3-D point; which intrinsics are parameters is the variant
struct PixelErrorCost
{
    Vec3 loc;
    camera_model model;
    double pixel[2];
    template <typename T>
    bool eval(const T *rotation, const T *point, const T *focal, const T *principal, const T *radial, const T *tangential, T *residuals) const
    {
        T inv[4];
        quat_inverse(rotation, inv);
        const V3<T> ray = quat_rotate(inv, V3<T>{point[0] - T(loc.x), point[1] - T(loc.y), point[2] - T(loc.z)});
        const T f = focal ? *focal : T(model.focal_length_pixels);
        const T pp[2] = {principal ? principal[0] : T(model.principle_point[0]), principal ? principal[1] : T(model.principle_point[1])};
        const T k[3] = {radial ? radial[0] : T(model.radial_distortion[0]), radial ? radial[1] : T(model.radial_distortion[1]),
                        radial ? radial[2] : T(model.radial_distortion[2])};
        const T tg[2] = {tangential ? tangential[0] : T(model.tangential_distortion[0]),
                         tangential ? tangential[1] : T(model.tangential_distortion[1])};
        T px[2];
        image_from_3d_T<T>(ray, f, pp, k, tg, px);
        residuals[0] = px[0] - T(pixel[0]);
        residuals[1] = px[1] - T(pixel[1]);
        return true;
    }
};
struct PixelErrorCost_Orientation : PixelErrorCost // :309-345
{
    template <typename T> bool operator()(const T *rotation, const T *point, T *residuals) const
    {
        return eval<T>(rotation, point, nullptr, nullptr, nullptr, nullptr, residuals);
    }
};
struct PixelErrorCost_OrientationFocal : PixelErrorCost // :347-393
{
    template <typename T> bool operator()(const T *rotation, const T *point, const T *focal, const T *principal, T *residuals) const
    {
        return eval<T>(rotation, point, focal, principal, nullptr, nullptr, residuals);
    }
};
struct PixelErrorCost_OrientationFocalRadial : PixelErrorCost // :395-445
{
    template <typename T>
    bool operator()(const T *rotation, const T *point, const T *focal, const T *principal, const T *radial, T *residuals) const
    {
        return eval<T>(rotation, point, focal, principal, radial, nullptr, residuals);
    }
};
struct PixelErrorCost_OrientationFocalRadialTangential : PixelErrorCost // :447-500
{
    template <typename T>
    bool operator()(const T *rotation, const T *point, const T *focal, const T *principal, const T *radial, const T *tangential,
                    T *residuals) const
    {
        return eval<T>(rotation, point, focal, principal, radial, tangential, residuals);
    }
};

} // namespace oracle
