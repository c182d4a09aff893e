// ORACLE — test infrastructure only (see oracle.hpp).
// Forward-mode dual numbers restating ceres::Jet<double,N> (ceres/jet.h) [3P]: value + N partials,
// with the same derivative formulas.  Used by the restated TinySolver autodiff and relax functors.
#pragma once

#include <cmath>

namespace oracle
{

template <int N> struct Jet
{
    double a = 0;
    double v[N];
    Jet()
    {
        for (int i = 0; i < N; i++)
            v[i] = 0;
    }
    Jet(double s) : a(s) // NOLINT: implicit like ceres::Jet(const T&)
    {
        for (int i = 0; i < N; i++)
            v[i] = 0;
    }
    Jet(double s, int k) : a(s)
    {
        for (int i = 0; i < N; i++)
            v[i] = 0;
        v[k] = 1.0;
    }
};

template <int N> inline Jet<N> operator+(const Jet<N> &f, const Jet<N> &g)
{
    Jet<N> h;
    h.a = f.a + g.a;
    for (int i = 0; i < N; i++)
        h.v[i] = f.v[i] + g.v[i];
    return h;
}
template <int N> inline Jet<N> operator-(const Jet<N> &f, const Jet<N> &g)
{
    Jet<N> h;
    h.a = f.a - g.a;
    for (int i = 0; i < N; i++)
        h.v[i] = f.v[i] - g.v[i];
    return h;
}
template <int N> inline Jet<N> operator-(const Jet<N> &f)
{
    Jet<N> h;
    h.a = -f.a;
    for (int i = 0; i < N; i++)
        h.v[i] = -f.v[i];
    return h;
}
template <int N> inline Jet<N> operator*(const Jet<N> &f, const Jet<N> &g)
{
    Jet<N> h;
    h.a = f.a * g.a;
    for (int i = 0; i < N; i++)
        h.v[i] = f.a * g.v[i] + f.v[i] * g.a;
    return h;
}
template <int N> inline Jet<N> operator/(const Jet<N> &f, const Jet<N> &g)
{
    // ceres: g_a_inverse = 1/g.a; f_a_by_g_a = f.a * g_a_inverse; v = (f.v - f_a_by_g_a * g.v) * g_a_inverse
    Jet<N> h;
    const double ginv = 1.0 / g.a;
    const double fg = f.a * ginv;
    h.a = fg;
    for (int i = 0; i < N; i++)
        h.v[i] = (f.v[i] - fg * g.v[i]) * ginv;
    return h;
}
template <int N> inline Jet<N> operator+(const Jet<N> &f, double s)
{
    Jet<N> h = f;
    h.a = f.a + s;
    return h;
}
template <int N> inline Jet<N> operator+(double s, const Jet<N> &f)
{
    return f + s;
}
template <int N> inline Jet<N> operator-(const Jet<N> &f, double s)
{
    Jet<N> h = f;
    h.a = f.a - s;
    return h;
}
template <int N> inline Jet<N> operator-(double s, const Jet<N> &f)
{
    Jet<N> h;
    h.a = s - f.a;
    for (int i = 0; i < N; i++)
        h.v[i] = -f.v[i];
    return h;
}
template <int N> inline Jet<N> operator*(const Jet<N> &f, double s)
{
    Jet<N> h;
    h.a = f.a * s;
    for (int i = 0; i < N; i++)
        h.v[i] = f.v[i] * s;
    return h;
}
template <int N> inline Jet<N> operator*(double s, const Jet<N> &f)
{
    return f * s;
}
template <int N> inline Jet<N> operator/(const Jet<N> &f, double s)
{
    const double sinv = 1.0 / s;
    Jet<N> h;
    h.a = f.a * sinv;
    for (int i = 0; i < N; i++)
        h.v[i] = f.v[i] * sinv;
    return h;
}
template <int N> inline Jet<N> operator/(double s, const Jet<N> &g)
{
    Jet<N> h;
    const double ginv = 1.0 / g.a;
    h.a = s * ginv;
    const double m = -s / (g.a * g.a);
    for (int i = 0; i < N; i++)
        h.v[i] = g.v[i] * m;
    return h;
}
template <int N> inline Jet<N> &operator+=(Jet<N> &f, const Jet<N> &g)
{
    f = f + g;
    return f;
}
template <int N> inline Jet<N> &operator-=(Jet<N> &f, const Jet<N> &g)
{
    f = f - g;
    return f;
}
template <int N> inline Jet<N> &operator*=(Jet<N> &f, const Jet<N> &g)
{
    f = f * g;
    return f;
}
template <int N> inline Jet<N> &operator/=(Jet<N> &f, const Jet<N> &g)
{
    f = f / g;
    return f;
}
template <int N> inline Jet<N> &operator*=(Jet<N> &f, double s)
{
    f = f * s;
    return f;
}
template <int N> inline Jet<N> &operator/=(Jet<N> &f, double s)
{
    f = f / s;
    return f;
}

#define ORACLE_JET_CMP(op)                                                                                             \
    template <int N> inline bool operator op(const Jet<N> &f, const Jet<N> &g)                                         \
    {                                                                                                                  \
        return f.a op g.a;                                                                                             \
    }                                                                                                                  \
    template <int N> inline bool operator op(const Jet<N> &f, double g)                                                \
    {                                                                                                                  \
        return f.a op g;                                                                                               \
    }                                                                                                                  \
    template <int N> inline bool operator op(double f, const Jet<N> &g)                                                \
    {                                                                                                                  \
        return f op g.a;                                                                                               \
    }
ORACLE_JET_CMP(<)
ORACLE_JET_CMP(<=)
ORACLE_JET_CMP(>)
ORACLE_JET_CMP(>=)
ORACLE_JET_CMP(==)
ORACLE_JET_CMP(!=)
#undef ORACLE_JET_CMP

template <int N> inline Jet<N> scaled(const Jet<N> &f, double val, double dscale)
{
    Jet<N> h;
    h.a = val;
    for (int i = 0; i < N; i++)
        h.v[i] = dscale * f.v[i];
    return h;
}
template <int N> inline Jet<N> sqrt(const Jet<N> &f)
{
    const double t = std::sqrt(f.a);
    return scaled(f, t, 1.0 / (2.0 * t));
}
template <int N> inline Jet<N> abs(const Jet<N> &f)
{
    // ceres: Jet(abs(f.a), copysign(1, f.a) * f.v)
    return scaled(f, std::abs(f.a), std::copysign(1.0, f.a));
}
template <int N> inline Jet<N> acos(const Jet<N> &f)
{
    return scaled(f, std::acos(f.a), -1.0 / std::sqrt(1.0 - f.a * f.a));
}
template <int N> inline Jet<N> asin(const Jet<N> &f)
{
    return scaled(f, std::asin(f.a), 1.0 / std::sqrt(1.0 - f.a * f.a));
}
template <int N> inline Jet<N> sin(const Jet<N> &f)
{
    return scaled(f, std::sin(f.a), std::cos(f.a));
}
template <int N> inline Jet<N> cos(const Jet<N> &f)
{
    return scaled(f, std::cos(f.a), -std::sin(f.a));
}
template <int N> inline Jet<N> atan2(const Jet<N> &g, const Jet<N> &f)
{
    // ceres: tmp = 1/(f.a^2 + g.a^2); Jet(atan2(g.a, f.a), tmp * (-g.a * f.v + f.a * g.v))
    const double tmp = 1.0 / (f.a * f.a + g.a * g.a);
    Jet<N> h;
    h.a = std::atan2(g.a, f.a);
    for (int i = 0; i < N; i++)
        h.v[i] = tmp * (-g.a * f.v[i] + f.a * g.v[i]);
    return h;
}
template <int N> inline bool isfinite(const Jet<N> &f)
{
    // ceres::isfinite(Jet): value and all partials finite
    if (!std::isfinite(f.a))
        return false;
    for (int i = 0; i < N; i++)
        if (!std::isfinite(f.v[i]))
            return false;
    return true;
}
template <int N> inline bool isnan(const Jet<N> &f)
{
    if (std::isnan(f.a))
        return true;
    for (int i = 0; i < N; i++)
        if (std::isnan(f.v[i]))
            return true;
    return false;
}
inline double value_of(double x)
{
    return x;
}
template <int N> inline double value_of(const Jet<N> &f)
{
    return f.a;
}

} // namespace oracle
