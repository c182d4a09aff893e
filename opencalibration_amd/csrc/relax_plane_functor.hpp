// libochip.so (internal) - the 2-ray plane-intersection cost functor and the downward prior of the ground-plane relax
// (include/opencalibration/relax/relax_cost_function.hpp:21-49,601-684), shared by the pair-record engine (relax.hip) and
// the bootstrap chain (relax_chain.hip).  Device code; include after dual.hpp inside `using namespace ochip`.
#pragma once

#include "dual.hpp"

namespace
{

constexpr int PLANE_ACC = 55; // 45 upper-triangular entries of the 9x9 [p|q|z] block + 9 gradient + cost

__host__ __device__ inline int plane_tri(int i, int j) // i <= j, 9x9 upper triangle
{
    return i * 9 - i * (i - 1) / 2 + (j - i);
}

// ---- the cost functor (relax_cost_function.hpp:601-656 with N = 2), T = double or Dual<3>
template <typename T> struct functor_io
{
    T qa[4], qb[4], z[3];
};

template <typename T> __device__ __forceinline__ Vec3T<T> quat_rotate(const T *q, const Vec3T<T> &v)
{
    // Eigen QuaternionBase::_transformVector
    const Vec3T<T> qv{q[0], q[1], q[2]};
    Vec3T<T> uv = cross(qv, v);
    uv = uv + uv;
    return v + scale(uv, q[3]) + cross(qv, uv);
}

template <typename T>
__device__ bool plane_intersection_residuals(const functor_io<T> &in, const double *loc_a, const double *loc_b,
                                             const double *rays, const double *plane_xy, T *res)
{
    Vec3T<T> corner[3];
    for (int i = 0; i < 3; i++)
        corner[i] = {T(plane_xy[2 * i]), T(plane_xy[2 * i + 1]), in.z[i]};
    // cornerPlane2normOffsetPlane (intersection.hpp:26-32)
    Vec3T<T> nrm = cross(corner[0] - corner[1], corner[0] - corner[2]);
    {
        const T zz = dot(nrm, nrm);
        if (value_of(zz) > 0.0)
            nrm = divide(nrm, dsqrt(zz));
    }
    const Vec3T<T> offset = corner[0];
    Vec3T<T> isect[2];
    bool all_valid = true;
    T avg_dist = T(0.0);
    for (int i = 0; i < 2; i++)
    {
        const double *l = i == 0 ? loc_a : loc_b;
        const T *q = i == 0 ? in.qa : in.qb;
        const Vec3T<T> ray_cam{T(rays[3 * i]), T(rays[3 * i + 1]), T(rays[3 * i + 2])};
        const Vec3T<T> dir = quat_rotate(q, ray_cam);
        const Vec3T<T> off{T(l[0]), T(l[1]), T(l[2])};
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
    avg_dist = avg_dist / T(2.0);
    const T huber_threshold = avg_dist * T(0.01);
    // robustCentroid (relax_cost_function.hpp:73-117), n = 2
    Vec3T<T> centroid = divide(isect[0] + isect[1], T(2.0));
    for (int stage = 0; stage < 3; stage++)
    {
        T total_w = T(0.0), w[2];
        double min_w = 1.7976931348623157e308, max_w = 0.0;
        for (int i = 0; i < 2; i++)
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
        const Vec3T<T> ws = scale(isect[0], w[0]) + scale(isect[1], w[1]);
        centroid = divide(ws, total_w);
        if (min_w > max_w * 0.5)
            break;
    }
    for (int i = 0; i < 2; i++)
    {
        const Vec3T<T> r = divide(isect[i] - centroid, avg_dist);
        res[3 * i] = r.x;
        res[3 * i + 1] = r.y;
        res[3 * i + 2] = r.z;
    }
    return all_valid;
}

// ---- the same functor with mixed argument types (round 3).  A Jacobian pass seeds ONE of the three parameter blocks
// (camera a's tangent, camera b's, the plane heights); run on Dual<3> throughout, the other two blocks and everything
// that only depends on them - a whole ray with its rotation, or the plane's normal - carried three zero partials through
// every operation: a third of the kernel's arithmetic (it is bound by fp64 issue: 9 000 wavefronts of ~18 000
// instructions).  Here the unseeded blocks are plain doubles and an expression becomes a Dual where a seeded value first
// enters it.  Same operations in the same order on the values; a partial that used to be computed as 0 * x + y is now y.
typedef Dual<3> D3;
OCHIP_HD D3 operator+(const D3 &f, double g)
{
    D3 h = f;
    h.a = f.a + g;
    return h;
}
OCHIP_HD D3 operator+(double f, const D3 &g)
{
    D3 h = g;
    h.a = f + g.a;
    return h;
}
OCHIP_HD D3 operator-(const D3 &f, double g)
{
    D3 h = f;
    h.a = f.a - g;
    return h;
}
OCHIP_HD D3 operator-(double f, const D3 &g)
{
    D3 h;
    h.a = f - g.a;
    for (int i = 0; i < 3; i++)
        h.v[i] = -g.v[i];
    return h;
}
OCHIP_HD D3 operator*(const D3 &f, double g)
{
    D3 h;
    h.a = f.a * g;
    for (int i = 0; i < 3; i++)
        h.v[i] = f.v[i] * g;
    return h;
}
OCHIP_HD D3 operator*(double f, const D3 &g)
{
    D3 h;
    h.a = f * g.a;
    for (int i = 0; i < 3; i++)
        h.v[i] = f * g.v[i];
    return h;
}
OCHIP_HD D3 operator/(const D3 &f, double g)
{
    D3 h;
    const double ginv = 1.0 / g;
    h.a = f.a * ginv;
    for (int i = 0; i < 3; i++)
        h.v[i] = f.v[i] * ginv;
    return h;
}
OCHIP_HD D3 operator/(double f, const D3 &g)
{
    D3 h;
    const double ginv = 1.0 / g.a;
    const double fg = f * ginv;
    h.a = fg;
    for (int i = 0; i < 3; i++)
        h.v[i] = (-(fg * g.v[i])) * ginv;
    return h;
}
template <typename A, typename B> OCHIP_HD auto vadd(const Vec3T<A> &a, const Vec3T<B> &b)
{
    using R = decltype(a.x + b.x);
    return Vec3T<R>{a.x + b.x, a.y + b.y, a.z + b.z};
}
template <typename A, typename B> OCHIP_HD auto vsub(const Vec3T<A> &a, const Vec3T<B> &b)
{
    using R = decltype(a.x - b.x);
    return Vec3T<R>{a.x - b.x, a.y - b.y, a.z - b.z};
}
template <typename A, typename B> OCHIP_HD auto vscale(const Vec3T<A> &a, const B &s)
{
    using R = decltype(a.x * s);
    return Vec3T<R>{a.x * s, a.y * s, a.z * s};
}
template <typename A, typename B> OCHIP_HD auto vdivide(const Vec3T<A> &a, const B &s)
{
    using R = decltype(a.x / s);
    return Vec3T<R>{a.x / s, a.y / s, a.z / s};
}
template <typename A, typename B> OCHIP_HD auto vdot(const Vec3T<A> &a, const Vec3T<B> &b)
{
    return a.x * b.x + a.y * b.y + a.z * b.z;
}
template <typename A, typename B> OCHIP_HD auto vcross(const Vec3T<A> &a, const Vec3T<B> &b)
{
    using R = decltype(a.y * b.z - a.z * b.y);
    return Vec3T<R>{a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x};
}
template <typename A> OCHIP_HD A vnorm(const Vec3T<A> &a)
{
    return dsqrt(vdot(a, a));
}
template <typename TQ> OCHIP_HD Vec3T<TQ> quat_rotate_mixed(const TQ *q, const Vec3T<double> &v)
{
    const Vec3T<TQ> qv{q[0], q[1], q[2]};
    Vec3T<TQ> uv = vcross(qv, v);
    uv = vadd(uv, uv);
    return vadd(vadd(v, vscale(uv, q[3])), vcross(qv, uv));
}

// TA, TB, TZ: double or Dual<3> (at most one of them a Dual); the residuals come out in the widest of them
template <typename TA, typename TB, typename TZ, typename R>
__device__ bool plane_intersection_residuals_mixed(const TA *qa, const TB *qb, const TZ *z, const double *loc_a, const double *loc_b,
                                                   const double *rays, const double *plane_xy, R *res)
{
    // cornerPlane2normOffsetPlane: corners (x, y) are constants, the heights z are the parameters
    const Vec3T<TZ> e1{TZ(plane_xy[0] - plane_xy[2]), TZ(plane_xy[1] - plane_xy[3]), z[0] - z[1]};
    const Vec3T<TZ> e2{TZ(plane_xy[0] - plane_xy[4]), TZ(plane_xy[1] - plane_xy[5]), z[0] - z[2]};
    Vec3T<TZ> nrm = vcross(e1, e2);
    {
        const TZ zz = vdot(nrm, nrm);
        if (value_of(zz) > 0.0)
            nrm = vdivide(nrm, dsqrt(zz));
    }
    const Vec3T<TZ> offset{TZ(plane_xy[0]), TZ(plane_xy[1]), z[0]};
    const TZ plane_d = vdot(nrm, offset);
    bool all_valid = true;
    const Vec3T<double> off_a{loc_a[0], loc_a[1], loc_a[2]}, off_b{loc_b[0], loc_b[1], loc_b[2]};
    const Vec3T<TA> dir_a = quat_rotate_mixed(qa, Vec3T<double>{rays[0], rays[1], rays[2]});
    const Vec3T<TB> dir_b = quat_rotate_mixed(qb, Vec3T<double>{rays[3], rays[4], rays[5]});
    using RA = decltype(vdot(nrm, dir_a));
    using RB = decltype(vdot(nrm, dir_b));
    Vec3T<RA> isect_a;
    Vec3T<RB> isect_b;
    {
        const RA denom = vdot(nrm, dir_a);
        if (fabs(value_of(denom)) < 1e-9)
        {
            all_valid = false;
            isect_a = {RA(NAN), RA(NAN), RA(NAN)};
        }
        else
        {
            const RA t = (plane_d - vdot(off_a, nrm)) / denom;
            isect_a = vadd(off_a, vscale(dir_a, t));
        }
    }
    {
        const RB denom = vdot(nrm, dir_b);
        if (fabs(value_of(denom)) < 1e-9)
        {
            all_valid = false;
            isect_b = {RB(NAN), RB(NAN), RB(NAN)};
        }
        else
        {
            const RB t = (plane_d - vdot(off_b, nrm)) / denom;
            isect_b = vadd(off_b, vscale(dir_b, t));
        }
    }
    R avg_dist = (vnorm(vsub(isect_a, off_a)) + vnorm(vsub(isect_b, off_b))) / 2.0;
    const R huber_threshold = avg_dist * 0.01;
    // robustCentroid (relax_cost_function.hpp:73-117), n = 2
    Vec3T<R> centroid = vdivide(vadd(isect_a, isect_b), 2.0);
    for (int stage = 0; stage < 3; stage++)
    {
        const R err_a = vnorm(vsub(isect_a, centroid)), err_b = vnorm(vsub(isect_b, centroid));
        R wa = 1.0 / (err_a + 1e-8), wb = 1.0 / (err_b + 1e-8);
        if (value_of(err_a) > value_of(huber_threshold))
            wa = wa * (huber_threshold / err_a);
        if (value_of(err_b) > value_of(huber_threshold))
            wb = wb * (huber_threshold / err_b);
        const R total_w = wa + wb;
        const double min_w = fmin(value_of(wa), value_of(wb)), max_w = fmax(value_of(wa), value_of(wb));
        centroid = vdivide(vadd(vscale(isect_a, wa), vscale(isect_b, wb)), total_w);
        if (min_w > max_w * 0.5)
            break;
    }
    const Vec3T<R> ra = vdivide(vsub(isect_a, centroid), avg_dist), rb = vdivide(vsub(isect_b, centroid), avg_dist);
    res[0] = ra.x, res[1] = ra.y, res[2] = ra.z;
    res[3] = rb.x, res[4] = rb.y, res[5] = rb.z;
    return all_valid;
}

// tangent seed of the EigenQuaternionManifold at q: d(q_delta * q)/d delta (ceres manifold.cc, Order XYZW)
__device__ __forceinline__ void seed_quat(const double *q, Dual<3> *out)
{
    const double x = q[0], y = q[1], z = q[2], w = q[3];
    const double pj[4][3] = {{w, z, -y}, {-z, w, x}, {y, -x, w}, {-x, -y, -z}};
    for (int k = 0; k < 4; k++)
    {
        out[k] = Dual<3>(q[k]);
        for (int c = 0; c < 3; c++)
            out[k].v[c] = pj[k][c];
    }
}

// downward prior of one camera (relax_cost_function.hpp:21-49): residual and tangent Jacobian
__device__ void downward_prior(const double *q, double weight, double *res, double *jac3)
{
    Dual<3> qd[4];
    seed_quat(q, qd);
    const Vec3T<Dual<3>> cam_center{Dual<3>(0.0), Dual<3>(0.0), Dual<3>(1.0)};
    const Vec3T<Dual<3>> rot = quat_rotate(qd, cam_center);
    // angleBetweenUnitVectors(rot, (0,0,-1)) with the clamp of relax_cost_function.hpp:16-19
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

} // namespace
