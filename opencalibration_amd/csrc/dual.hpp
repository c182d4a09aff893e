// Forward-mode dual numbers for the device relax kernels (value + N partials) and a tiny 3-vector on
// them.  Same derivative formulas as ceres::Jet (the reference evaluates its cost functors through
// ceres::AutoDiffCostFunction, src/relax/autodiff_cost_function.cpp:8-123).
#pragma once

#include <hip/hip_runtime.h>

namespace ochip
{

// P: the type the partials are kept and propagated in.  double everywhere on the path; float only for the C5 sweep of
// Jacobian precision (OCHIP_TEST_HOOKS=jacobian_fp32, relax_general.hip) - values stay fp64 either way.
template <int N, typename P = double> struct Dual
{
    double a;
    P v[N];
    __host__ __device__ Dual() : a(0)
    {
        for (int i = 0; i < N; i++)
            v[i] = 0;
    }
    __host__ __device__ Dual(double s) : a(s)
    {
        for (int i = 0; i < N; i++)
            v[i] = 0;
    }
};

#define OCHIP_HD __host__ __device__ __forceinline__

template <int N, typename P> OCHIP_HD Dual<N, P> operator+(const Dual<N, P> &f, const Dual<N, P> &g)
{
    Dual<N, P> h;
    h.a = f.a + g.a;
    for (int i = 0; i < N; i++)
        h.v[i] = f.v[i] + g.v[i];
    return h;
}
template <int N, typename P> OCHIP_HD Dual<N, P> operator-(const Dual<N, P> &f, const Dual<N, P> &g)
{
    Dual<N, P> h;
    h.a = f.a - g.a;
    for (int i = 0; i < N; i++)
        h.v[i] = f.v[i] - g.v[i];
    return h;
}
template <int N, typename P> OCHIP_HD Dual<N, P> operator*(const Dual<N, P> &f, const Dual<N, P> &g)
{
    Dual<N, P> h;
    h.a = f.a * g.a;
    for (int i = 0; i < N; i++)
        h.v[i] = (P)f.a * g.v[i] + f.v[i] * (P)g.a;
    return h;
}
template <int N, typename P> OCHIP_HD Dual<N, P> operator/(const Dual<N, P> &f, const Dual<N, P> &g)
{
    Dual<N, P> h;
    const double ginv = 1.0 / g.a;
    const double fg = f.a * ginv;
    h.a = fg;
    for (int i = 0; i < N; i++)
        h.v[i] = (f.v[i] - (P)fg * g.v[i]) * (P)ginv;
    return h;
}
template <int N, typename P> OCHIP_HD bool operator<(const Dual<N, P> &f, const Dual<N, P> &g)
{
    return f.a < g.a;
}
template <int N, typename P> OCHIP_HD bool operator>(const Dual<N, P> &f, const Dual<N, P> &g)
{
    return f.a > g.a;
}
template <int N, typename P> OCHIP_HD Dual<N, P> dsqrt(const Dual<N, P> &f)
{
    Dual<N, P> h;
    h.a = sqrt(f.a);
    const double d = 1.0 / (2.0 * h.a);
    for (int i = 0; i < N; i++)
        h.v[i] = (P)d * f.v[i];
    return h;
}
template <int N, typename P> OCHIP_HD Dual<N, P> dabs(const Dual<N, P> &f)
{
    Dual<N, P> h;
    h.a = fabs(f.a);
    const double s = copysign(1.0, f.a);
    for (int i = 0; i < N; i++)
        h.v[i] = (P)s * f.v[i];
    return h;
}
template <int N, typename P> OCHIP_HD Dual<N, P> dacos(const Dual<N, P> &f)
{
    Dual<N, P> h;
    h.a = acos(f.a);
    const double d = -1.0 / sqrt(1.0 - f.a * f.a);
    for (int i = 0; i < N; i++)
        h.v[i] = (P)d * f.v[i];
    return h;
}
template <int N, typename P> OCHIP_HD Dual<N, P> datan2(const Dual<N, P> &y, const Dual<N, P> &x)
{
    Dual<N, P> h;
    h.a = atan2(y.a, x.a);
    const double d = 1.0 / (x.a * x.a + y.a * y.a);
    for (int i = 0; i < N; i++)
        h.v[i] = (P)(d * x.a) * y.v[i] - (P)(d * y.a) * x.v[i];
    return h;
}
OCHIP_HD double datan2(double y, double x)
{
    return atan2(y, x);
}
OCHIP_HD double dsqrt(double x)
{
    return sqrt(x);
}
OCHIP_HD double dabs(double x)
{
    return fabs(x);
}
OCHIP_HD double dacos(double x)
{
    return acos(x);
}
OCHIP_HD double value_of(double x)
{
    return x;
}
template <int N, typename P> OCHIP_HD double value_of(const Dual<N, P> &x)
{
    return x.a;
}

template <typename T> struct Vec3T
{
    T x, y, z;
};
template <typename T> OCHIP_HD Vec3T<T> operator+(const Vec3T<T> &a, const Vec3T<T> &b)
{
    return {a.x + b.x, a.y + b.y, a.z + b.z};
}
template <typename T> OCHIP_HD Vec3T<T> operator-(const Vec3T<T> &a, const Vec3T<T> &b)
{
    return {a.x - b.x, a.y - b.y, a.z - b.z};
}
template <typename T> OCHIP_HD Vec3T<T> scale(const Vec3T<T> &a, const T &s)
{
    return {a.x * s, a.y * s, a.z * s};
}
template <typename T> OCHIP_HD Vec3T<T> divide(const Vec3T<T> &a, const T &s)
{
    return {a.x / s, a.y / s, a.z / s};
}
template <typename T> OCHIP_HD T dot(const Vec3T<T> &a, const Vec3T<T> &b)
{
    return a.x * b.x + a.y * b.y + a.z * b.z;
}
template <typename T> OCHIP_HD Vec3T<T> cross(const Vec3T<T> &a, const Vec3T<T> &b)
{
    return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x};
}
template <typename T> OCHIP_HD T norm(const Vec3T<T> &a)
{
    return dsqrt(dot(a, a));
}

} // namespace ochip
