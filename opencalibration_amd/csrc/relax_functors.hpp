// libochip.so (internal) — the relax cost functors on a scalar type T (double or Dual<3>/Dual<4>), device side.
// Restated from include/opencalibration/relax/relax_cost_function.hpp (line numbers below) and
// include/opencalibration/geometry/intersection.hpp:26-47; same operation order as the CPU oracle's restatement.
#pragma once

#include "dual.hpp"

namespace ochip
{

template <typename T> __device__ __forceinline__ Vec3T<T> gquat_rotate(const T *q, const Vec3T<T> &v)
{
    // Eigen QuaternionBase::_transformVector
    const Vec3T<T> qv{q[0], q[1], q[2]};
    Vec3T<T> uv = cross(qv, v);
    uv = uv + uv;
    return v + scale(uv, q[3]) + cross(qv, uv);
}

// tangent seed of the EigenQuaternionManifold at q: d(q_delta * q)/d delta (ceres manifold.cc, Order XYZW)
template <typename P> __device__ __forceinline__ void gseed_quat(const double *q, Dual<3, P> *out)
{
    const double x = q[0], y = q[1], z = q[2], w = q[3];
    const double pj[4][3] = {{w, z, -y}, {-z, w, x}, {y, -x, w}, {-x, -y, -z}};
    for (int k = 0; k < 4; k++)
    {
        out[k] = Dual<3, P>(q[k]);
        for (int c = 0; c < 3; c++)
            out[k].v[c] = (P)pj[k][c];
    }
}

template <typename T> __device__ __forceinline__ Vec3T<T> gnormalized(const Vec3T<T> &a)
{
    const T zz = dot(a, a);
    if (value_of(zz) > 0.0)
        return divide(a, dsqrt(zz));
    return a;
}

// relax_cost_function.hpp:16-19
template <typename T> __device__ __forceinline__ T gangle_between_unit_vectors(const Vec3T<T> &n1, const Vec3T<T> &n2)
{
    T d = dot(n1, n2);
    const double lo = -1 + 1e-12, hi = 1 - 1e-12;
    if (value_of(d) < lo)
        d = T(lo);
    else if (hi < value_of(d))
        d = T(hi);
    return dacos(d);
}

// distort_keypoints.hpp:26-42 + :97-116: pixel -> camera-frame unit ray through the INVERSE lens model
// m = {f, ppx, ppy, k1, k2, k3, p1, p2}
template <typename T> __device__ __forceinline__ Vec3T<T> gimage_to_3d_inverse(const double *px, const T *m)
{
    const T u[2] = {(T(px[0]) - m[1]) / m[0], (T(px[1]) - m[2]) / m[0]};
    T r2[3];
    r2[0] = u[0] * u[0] + u[1] * u[1];
    r2[1] = r2[0] * r2[0];
    r2[2] = r2[1] * r2[0];
    const T radial_dot = m[3] * r2[0] + m[4] * r2[1] + m[5] * r2[2];
    const T prod = u[0] * u[1];
    T und[2];
    for (int i = 0; i < 2; i++)
        und[i] = (T(1.0) + radial_dot) * u[i] + T(2.0) * prod * m[6 + i] + m[6 + (1 - i)] * (r2[0] + T(2.0) * u[i] * u[i]);
    return gnormalized(Vec3T<T>{und[0], und[1], T(1.0)});
}

// MultiRayPlaneIntersectionAngleCost<N>::computeResiduals (:601-656; the FocalRadial variant :501-566 differs only in
// where the camera-frame rays come from).  q: N quaternions, ray: N camera-frame rays, loc: N camera positions,
// txy: the triangle's three corners' x,y, z: their heights.  res: 3N.  Returns all_valid.
template <typename T, int N>
__device__ bool gmulti_ray_residuals(const T (*q)[4], const Vec3T<T> *ray, const double (*loc)[3], const double *txy,
                                     const T *z, T *res)
{
    Vec3T<T> corner[3];
    for (int i = 0; i < 3; i++)
        corner[i] = {T(txy[2 * i]), T(txy[2 * i + 1]), z[i]};
    // cornerPlane2normOffsetPlane (intersection.hpp:26-32)
    const Vec3T<T> nrm = gnormalized(cross(corner[0] - corner[1], corner[0] - corner[2]));
    const Vec3T<T> offset = corner[0];
    Vec3T<T> isect[N];
    bool all_valid = true;
    T avg_dist = T(0.0);
    for (int i = 0; i < N; i++)
    {
        const Vec3T<T> dir = gquat_rotate(q[i], ray[i]);
        const Vec3T<T> off{T(loc[i][0]), T(loc[i][1]), T(loc[i][2])};
        // rayPlaneIntersection (intersection.hpp:34-47)
        const T denom = dot(nrm, dir);
        if (fabs(value_of(denom)) < 1e-9)
        {
            all_valid = false;
            isect[i] = {T(NAN), T(NAN), T(NAN)};
        }
        else
        {
            const T t = (dot(nrm, offset) - dot(off, nrm)) / denom;
            isect[i] = off + scale(dir, t);
        }
        avg_dist = avg_dist + norm(isect[i] - off);
    }
    avg_dist = avg_dist / T(double(N));
    const T huber_threshold = avg_dist * T(0.01);
    // robustCentroid (:73-117)
    Vec3T<T> centroid = isect[0];
    for (int i = 1; i < N; i++)
        centroid = centroid + isect[i];
    centroid = divide(centroid, T(double(N)));
    for (int stage = 0; stage < 3; stage++)
    {
        T total_w = T(0.0), w[N];
        double min_w = 1.7976931348623157e308, max_w = 0.0;
        for (int i = 0; i < N; i++)
        {
            const T err = norm(isect[i] - centroid);
            T wi = T(1.0) / (err + T(1e-8));
            if (value_of(err) > value_of(huber_threshold))
                wi = wi * (huber_threshold / err);
            w[i] = wi;
            total_w = total_w + wi;
            if (value_of(wi) < min_w)
                min_w = value_of(wi);
            if (value_of(wi) > max_w)
                max_w = value_of(wi);
        }
        Vec3T<T> ws = scale(isect[0], w[0]);
        for (int i = 1; i < N; i++)
            ws = ws + scale(isect[i], w[i]);
        centroid = divide(ws, total_w);
        if (min_w > max_w * 0.5)
            break;
    }
    for (int i = 0; i < N; i++)
    {
        const Vec3T<T> r = divide(isect[i] - centroid, avg_dist);
        res[3 * i] = r.x;
        res[3 * i + 1] = r.y;
        res[3 * i + 2] = r.z;
    }
    return all_valid;
}

// PointsDownwardsPrior (:21-49): residual and tangent Jacobian
__device__ inline void gdownward_prior(const double *q, double weight, double *res, double *jac3)
{
    Dual<3> qd[4];
    gseed_quat(q, qd);
    const Vec3T<Dual<3>> cam_center{Dual<3>(0.0), Dual<3>(0.0), Dual<3>(1.0)};
    const Vec3T<Dual<3>> rot = gquat_rotate(qd, cam_center);
    Dual<3> d = Dual<3>(0.0) * rot.x + Dual<3>(0.0) * rot.y + Dual<3>(-1.0) * rot.z;
    const double lo = -1 + 1e-12, hi = 1 - 1e-12;
    if (d.a < lo)
        d = Dual<3>(lo);
    else if (hi < d.a)
        d = Dual<3>(hi);
    const Dual<3> ang = Dual<3>(weight) * dacos(d);
    *res = ang.a;
    for (int c = 0; c < 3; c++)
        jac3[c] = ang.v[c];
}

// Eigen QuaternionBase::inverse(): conjugate / squared norm (the zero quaternion stays zero)
template <typename T> __device__ __forceinline__ void gquat_inverse(const T *q, T *out)
{
    const T n2 = q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3];
    if (value_of(n2) > 0.0)
    {
        out[0] = (T(0.0) - q[0]) / n2;
        out[1] = (T(0.0) - q[1]) / n2;
        out[2] = (T(0.0) - q[2]) / n2;
        out[3] = q[3] / n2;
    }
    else
        out[0] = out[1] = out[2] = out[3] = T(0.0);
}
template <typename T> __device__ __forceinline__ void gquat_product(const T *a, const T *b, T *out) // a * b, x y z w
{
    out[3] = a[3] * b[3] - a[0] * b[0] - a[1] * b[1] - a[2] * b[2];
    out[0] = a[3] * b[0] + a[0] * b[3] + a[1] * b[2] - a[2] * b[1];
    out[1] = a[3] * b[1] + a[1] * b[3] + a[2] * b[0] - a[0] * b[2];
    out[2] = a[3] * b[2] + a[2] * b[3] + a[0] * b[1] - a[1] * b[0];
}

// MultiDecomposedRotationCost (:253-307) over DecomposedRotationCost (:188-251): the relative pose of two cameras against
// up to four homography decompositions {relative rotation q (4), relative translation t (3), score}; of the decompositions
// scoring more than a quarter of the best, the one with the smallest residual norm is the block's residual.  pos1, pos2:
// the cameras' positions.  Returns false when no decomposition gives finite residuals (Ceres: evaluation failed).
template <typename T>
__device__ bool gmulti_decomposed_rotation(const T *q1, const T *q2, const double *poses32, const double *pos1, const double *pos2, T *res)
{
    double max_score = 0;
    for (int k = 0; k < 4; k++)
        max_score = fmax(max_score, (double)(int)poses32[8 * k + 7]);
    const double dvec[3] = {pos2[0] - pos1[0], pos2[1] - pos1[1], pos2[2] - pos1[2]};
    const double d2 = dvec[0] * dvec[0] + dvec[1] * dvec[1] + dvec[2] * dvec[2];
    double lowest = 1.7976931348623157e308 * 2; // +inf
    bool any = false;
    T inv1[4], inv2[4], r21[4];
    gquat_inverse(q1, inv1);
    gquat_inverse(q2, inv2);
    gquat_product(q1, inv2, r21);
    for (int k = 0; k < 4; k++)
    {
        const double *ps = poses32 + 8 * k;
        const int score = (int)ps[7];
        if (!((double)score > 0.25 * max_score))
            continue;
        const double t2n = ps[4] * ps[4] + ps[5] * ps[5] + ps[6] * ps[6];
        const bool has_translation = d2 > 1e-9 && t2n > 1e-9;
        const double qn = sqrt(ps[0] * ps[0] + ps[1] * ps[1] + ps[2] * ps[2] + ps[3] * ps[3]);
        const double rel[4] = {ps[0] / qn, ps[1] / qn, ps[2] / qn, ps[3] / qn};
        // Eigen's normalized(): divide by the norm when it is positive
        double tdir[3] = {dvec[0], dvec[1], dvec[2]}, rtdir[3] = {ps[4], ps[5], ps[6]};
        if (d2 > 0)
            for (int i = 0; i < 3; i++)
                tdir[i] /= sqrt(d2);
        if (t2n > 0)
            for (int i = 0; i < 3; i++)
                rtdir[i] /= sqrt(t2n);
        const double weight = sqrt(score / 8.);
        T r[3];
        if (has_translation)
        {
            const Vec3T<T> t21 = gquat_rotate(inv1, Vec3T<T>{T(tdir[0]), T(tdir[1]), T(tdir[2])});
            r[0] = gangle_between_unit_vectors<T>(t21, Vec3T<T>{T(rtdir[0]), T(rtdir[1]), T(rtdir[2])});
            // (relative_rotation * -translation_direction) is a product of doubles, cast afterwards
            const double relq[4] = {rel[0], rel[1], rel[2], rel[3]};
            const Vec3T<double> rt = gquat_rotate<double>(relq, Vec3T<double>{-tdir[0], -tdir[1], -tdir[2]});
            const Vec3T<T> t12 = gquat_rotate(inv2, Vec3T<T>{T(rt.x), T(rt.y), T(rt.z)});
            r[1] = gangle_between_unit_vectors<T>(t12, Vec3T<T>{T(-rtdir[0]), T(-rtdir[1]), T(-rtdir[2])});
        }
        else
            r[0] = r[1] = T(3.14159265358979323846);
        const T relT[4] = {T(rel[0]), T(rel[1]), T(rel[2]), T(rel[3])};
        T prod[4];
        gquat_product(relT, r21, prod);
        // Eigen::AngleAxis<T>(q).angle(): 2 atan2(|vec|, |w|), 0 for a zero vector part
        const T n = dsqrt(prod[0] * prod[0] + prod[1] * prod[1] + prod[2] * prod[2]);
        r[2] = value_of(n) != 0.0 ? T(2.0) * datan2(n, dabs(prod[3])) : T(0.0);
        double n2 = 0;
        bool finite = true;
        for (int i = 0; i < 3; i++)
        {
            r[i] = T(weight) * r[i];
            const double v = value_of(r[i]);
            finite = finite && (v - v == 0.0);
            n2 += v * v;
        }
        if (finite && n2 < lowest)
        {
            lowest = n2;
            any = true;
            for (int i = 0; i < 3; i++)
                res[i] = r[i];
        }
    }
    return any;
}

// AdjacentTriangleNormalCost (:119-155): xy = A B C D corners' x,y (8), z = their heights (4)
template <typename T> __device__ inline T gadjacent_triangle_normal(const double *xy, const T *z, double weight)
{
    const Vec3T<T> A{T(xy[0]), T(xy[1]), z[0]}, B{T(xy[2]), T(xy[3]), z[1]}, C{T(xy[4]), T(xy[5]), z[2]}, D{T(xy[6]), T(xy[7]), z[3]};
    const Vec3T<T> AB = B - A;
    const Vec3T<T> n1 = gnormalized(cross(AB, C - A));
    const Vec3T<T> n2 = gnormalized(cross(AB, D - A));
    return T(weight) * gangle_between_unit_vectors<T>(n1, n2);
}

} // namespace ochip
