// ORACLE — test infrastructure only (see oracle.hpp).
// Cost functors of include/opencalibration/relax/relax_cost_function.hpp and the geometry templates of
// include/opencalibration/geometry/intersection.hpp, restated on a scalar type T (double or Jet<N>).
#pragma once

#include "jet.hpp"

#include <algorithm>
#include <cmath>
#include <limits>

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

} // namespace oracle
