// libochip.so (internal) - the geometry and the score of gridFilterMatchesPerImage / addRayTriangleMeasurementCost
// (src/relax/relax_problem.cpp:234-309,388-560) as device functions, shared by the set-up kernels (relax_setup.hip) and the
// bootstrap chain (relax_chain.hip): the expressions of csrc/host/relax_util.hpp in the same order, fp64, no contraction.
#pragma once

#include <hip/hip_runtime.h>

#include "../../include/ochip.h"

namespace
{

struct v3
{
    double x, y, z;
};
__device__ __forceinline__ v3 sub(const v3 &a, const v3 &b)
{
    return {a.x - b.x, a.y - b.y, a.z - b.z};
}
__device__ __forceinline__ v3 add(const v3 &a, const v3 &b)
{
    return {a.x + b.x, a.y + b.y, a.z + b.z};
}
__device__ __forceinline__ v3 mul(const v3 &a, double s)
{
    return {a.x * s, a.y * s, a.z * s};
}
__device__ __forceinline__ double dot(const v3 &a, const v3 &b)
{
    return a.x * b.x + a.y * b.y + a.z * b.z;
}
__device__ __forceinline__ v3 cross(const v3 &a, const v3 &b)
{
    return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x};
}
// Eigen::Quaternion::toRotationMatrix
__device__ __forceinline__ void to_matrix(const double *q, double R[3][3])
{
    const double tx = 2 * q[0], ty = 2 * q[1], tz = 2 * q[2];
    const double twx = tx * q[3], twy = ty * q[3], twz = tz * q[3];
    const double txx = tx * q[0], txy = ty * q[0], txz = tz * q[0];
    const double tyy = ty * q[1], tyz = tz * q[1], tzz = tz * q[2];
    R[0][0] = 1 - (tyy + tzz), R[0][1] = txy - twz, R[0][2] = txz + twy;
    R[1][0] = txy + twz, R[1][1] = 1 - (txx + tzz), R[1][2] = tyz - twx;
    R[2][0] = txz - twy, R[2][1] = tyz + twx, R[2][2] = 1 - (txx + tyy);
}
__device__ __forceinline__ v3 apply(const double R[3][3], const v3 &v)
{
    return {R[0][0] * v.x + R[0][1] * v.y + R[0][2] * v.z, R[1][0] * v.x + R[1][1] * v.y + R[1][2] * v.z,
            R[2][0] * v.x + R[2][1] * v.y + R[2][2] * v.z};
}
// Eigen QuaternionBase::_transformVector
__device__ __forceinline__ v3 rotate(const double *q, const v3 &v)
{
    const v3 qv{q[0], q[1], q[2]};
    v3 uv = cross(qv, v);
    uv = add(uv, uv);
    return add(add(v, mul(uv, q[3])), cross(qv, uv));
}
// src/geometry/intersection.cpp:116-143: midpoint of closest approach, signed squared gap
__device__ __forceinline__ void ray_intersection(const v3 &d1, const v3 &o1, const v3 &d2, const v3 &o2, v3 *mid, double *err)
{
    const double nan = __longlong_as_double(0x7ff8000000000000ll);
    *mid = {nan, nan, nan};
    *err = nan;
    const double n11 = dot(d1, d1), n12 = dot(d1, d2), n22 = dot(d2, d2);
    const double denom = n11 * n22 - n12 * n12;
    if (fabs(denom) > 1e-9)
    {
        const v3 off = sub(o1, o2);
        const double od1 = dot(off, d1), od2 = dot(off, d2);
        const double t = (n12 * od2 - n22 * od1) / denom;
        const double s = (n11 * od2 - n12 * od1) / denom;
        const v3 p1 = add(o1, mul(d1, t)), p2 = add(o2, mul(d2, s));
        *mid = mul(add(p1, p2), 0.5);
        const v3 g = sub(p1, p2);
        *err = dot(g, g) * (t >= 0 && s >= 0 ? 1 : -1);
    }
}

constexpr int G_MAX = 16; // cells per axis the per-wave tables cover (1 / 0.15 < 7; finer grids than 1 / 15 go to the host)

// the score of one inlier match (:262-289) from its two camera-frame rays
__device__ __forceinline__ double plane_match_score(const double r1[3], const double r2[3], const double Rs[3][3], const double Rd[3][3],
                                                    const v3 &so, const v3 &d_o, const ochip_plane_inlier &m, const double ms[8],
                                                    const double md[8], const double *H, bool homography)
{
    const v3 sd = apply(Rs, v3{r1[0], r1[1], r1[2]}), dd = apply(Rd, v3{r2[0], r2[1], r2[2]});
    v3 mid;
    double gap;
    ray_intersection(sd, so, dd, d_o, &mid, &gap);
    const double intersection_score = gap < 0 ? 0. : 1. / (1. + gap);
    const double cos_angle = dot(sd, dd);
    const double angle_score = 1.0 - cos_angle * cos_angle;
    const double descriptor_score = m.descriptor_score;
    double ransac_score = 1.0;
    if (homography)
    {
        const double sx = (m.px1[0] - ms[1]) / ms[0];
        const double sy = (m.px1[1] - ms[2]) / ms[0];
        const double dx = (m.px2[0] - md[1]) / md[0];
        const double dy = (m.px2[1] - md[2]) / md[0];
        const double hx = H[0] * sx + H[1] * sy + H[2] * 1.0, hy = H[3] * sx + H[4] * sy + H[5] * 1.0,
                     hz = H[6] * sx + H[7] * sy + H[8] * 1.0;
        const double ex = dx - hx / hz, ey = dy - hy / hz;
        ransac_score = 1.0 / (1.0 + sqrt(ex * ex + ey * ey));
    }
    return intersection_score * angle_score * descriptor_score * ransac_score;
}

// the cells of a match in the two images' grids (GridFilter::addMeasurement); false: outside the G_MAX x G_MAX table
__device__ __forceinline__ bool plane_match_cells(const ochip_plane_inlier &m, double cols_s, double rows_s, double cols_d, double rows_d,
                                                  double res, int *cs, int *cd)
{
    const int cx = (int)floor(m.px1[0] / cols_s / res), cy = (int)floor(m.px1[1] / rows_s / res);
    const int dx = (int)floor(m.px2[0] / cols_d / res), dy = (int)floor(m.px2[1] / rows_d / res);
    *cs = cx * G_MAX + cy;
    *cd = dx * G_MAX + dy;
    return !(cx < 0 || cy < 0 || cx >= G_MAX || cy >= G_MAX || dx < 0 || dy < 0 || dx >= G_MAX || dy >= G_MAX);
}

// a kept match becomes a residual block if its rays' closest approach lies over the border triangle (:452-466;
// MeshIntersectionSearcher::triangleIntersect on the single triangle: inside unless anticlockwise of an edge)
__device__ __forceinline__ bool plane_block_inside(const double r1[3], const double r2[3], const double *qa, const double *qb, const v3 &so,
                                                   const v3 &d_o, const double *tri)
{
    v3 mid;
    double gap;
    ray_intersection(rotate(qa, v3{r1[0], r1[1], r1[2]}), so, rotate(qb, v3{r2[0], r2[1], r2[2]}), d_o, &mid, &gap);
    if (isnan(mid.x) || isnan(mid.y))
        return false;
    bool ok = true;
    for (int k = 0; k < 3; k++)
    {
        const double *b = tri + 2 * k, *c = tri + 2 * ((k + 1) % 3);
        if ((b[0] - mid.x) * (c[1] - mid.y) - (b[1] - mid.y) * (c[0] - mid.x) < 0)
            ok = false;
    }
    return ok;
}

} // namespace
