// Pixel -> unit bearing vector of a camera with radial / tangential lens distortion, shared by the device
// kernels (hipcc) and the host classes (g++), both built with -ffp-contract=off so the two agree to the bit.
//
// Restates image_to_3d (src/distort/distort_keypoints.cpp:68-103): (px - pp) / f, then - only if a
// distortion coefficient is non-zero - the inverse of distortProjectedRay
// (include/opencalibration/distort/distort_keypoints.hpp:26-42) found with
// ceres::TinySolver<TinySolverAutoDiffFunction<DistortionFunctor, 2, 2>> [third party, ceres/tiny_solver.h]:
// Levenberg-Marquardt with Jacobi scaling from the first Jacobian, a 2x2 LDLT solve and Nielsen's u/v damping
// update, at most 10 iterations, parameter tolerance 1e-2 / (|pp| + f) (:77-90); finally
// homogeneous().normalized().
#pragma once

#include <cmath>

#if defined(__HIPCC__)
#define OCHIP_UD __host__ __device__ inline
#else
#define OCHIP_UD inline
#endif

namespace ochip_ud
{

struct D2 // value + 2 partials, the arithmetic of ceres::Jet<double, 2>
{
    double a, v0, v1;
};
OCHIP_UD D2 lift(const D2 *, double s)
{
    return D2{s, 0.0, 0.0};
}
OCHIP_UD double lift(const double *, double s)
{
    return s;
}
OCHIP_UD D2 operator+(const D2 &f, const D2 &g)
{
    return D2{f.a + g.a, f.v0 + g.v0, f.v1 + g.v1};
}
OCHIP_UD D2 operator-(const D2 &f, const D2 &g)
{
    return D2{f.a - g.a, f.v0 - g.v0, f.v1 - g.v1};
}
OCHIP_UD D2 operator*(const D2 &f, const D2 &g)
{
    return D2{f.a * g.a, f.a * g.v0 + f.v0 * g.a, f.a * g.v1 + f.v1 * g.a};
}

// distort_keypoints.hpp:26-42
template <typename T> OCHIP_UD void distort_projected_ray(const T rp[2], const double radial[3], const double tangential[2], T out[2])
{
    const T *tag = nullptr;
    T r2[3];
    r2[0] = rp[0] * rp[0] + rp[1] * rp[1];
    for (int i = 1; i < 3; i++)
        r2[i] = r2[i - 1] * r2[0];
    const T radial_dot = lift(tag, radial[0]) * r2[0] + lift(tag, radial[1]) * r2[1] + lift(tag, radial[2]) * r2[2];
    const T prod = rp[0] * rp[1];
    for (int i = 0; i < 2; i++)
        out[i] = (lift(tag, 1.0) + radial_dot) * rp[i] + lift(tag, 2.0) * prod * lift(tag, tangential[i]) +
                 lift(tag, tangential[1 - i]) * (r2[0] + lift(tag, 2.0) * rp[i] * rp[i]);
}

struct tiny_state
{
    double J[2][2], r[2], jac_scale[2], jtj[2][2], g[2], cost, gmax;
};

OCHIP_UD void tiny_update(tiny_state &s, const double *xx, const double target[2], const double radial[3],
                          const double tangential[2], bool first)
{
    const D2 p[2] = {D2{xx[0], 1.0, 0.0}, D2{xx[1], 0.0, 1.0}};
    D2 d[2];
    distort_projected_ray<D2>(p, radial, tangential, d);
    for (int i = 0; i < 2; i++)
    {
        const D2 res = D2{target[i], 0.0, 0.0} - d[i];
        s.r[i] = -res.a; // residuals_ = -residuals_
        s.J[i][0] = res.v0;
        s.J[i][1] = res.v1;
    }
    if (first)
        for (int c = 0; c < 2; c++)
            s.jac_scale[c] = 1.0 / (1.0 + sqrt(s.J[0][c] * s.J[0][c] + s.J[1][c] * s.J[1][c]));
    for (int i = 0; i < 2; i++)
        for (int c = 0; c < 2; c++)
            s.J[i][c] *= s.jac_scale[c];
    for (int a = 0; a < 2; a++)
    {
        for (int b = 0; b < 2; b++)
            s.jtj[a][b] = s.J[0][a] * s.J[0][b] + s.J[1][a] * s.J[1][b];
        s.g[a] = s.J[0][a] * s.r[0] + s.J[1][a] * s.r[1];
    }
    s.gmax = fmax(fabs(s.g[0]), fabs(s.g[1]));
    s.cost = (s.r[0] * s.r[0] + s.r[1] * s.r[1]) / 2;
}

// x enters as the initial guess (the distorted normalised point) and leaves as the undistorted one
OCHIP_UD void tiny_solve_distortion(const double target[2], const double radial[3], const double tangential[2],
                                    double parameter_tolerance_opt, double x[2])
{
    const double gradient_tolerance = parameter_tolerance_opt * 1e-2;
    const double function_tolerance = 1e-6;
    const double cost_threshold = 1e-16;
    const int max_num_iterations = 10;
    const double initial_trust_region_radius = 1e4;
    tiny_state s;
    s.jac_scale[0] = s.jac_scale[1] = 1.0;
    tiny_update(s, x, target, radial, tangential, true);
    if (s.gmax < gradient_tolerance || s.cost < cost_threshold)
        return;
    double u = 1.0 / initial_trust_region_radius, v = 2;
    for (int iterations = 1; iterations < max_num_iterations; iterations++)
    {
        double A[2][2] = {{s.jtj[0][0], s.jtj[0][1]}, {s.jtj[1][0], s.jtj[1][1]}};
        for (int i = 0; i < 2; i++)
        {
            const double d = sqrt(u * fmin(fmax(s.jtj[i][i], 1e-6), 1e32));
            A[i][i] += d * d;
        }
        // 2x2 LDLT (pivot on the larger diagonal) solve A * step = g
        double step[2];
        {
            const int p = A[1][1] > A[0][0] ? 1 : 0, q = 1 - p;
            const double d0 = A[p][p], l = A[q][p] / d0, d1 = A[q][q] - l * A[q][p];
            const double y0 = s.g[p], y1 = s.g[q] - l * y0;
            const double z1 = y1 / d1, z0 = y0 / d0 - l * z1;
            step[p] = z0;
            step[q] = z1;
        }
        const double dx[2] = {s.jac_scale[0] * step[0], s.jac_scale[1] * step[1]};
        const double xnorm = sqrt(x[0] * x[0] + x[1] * x[1]);
        const double ptol = parameter_tolerance_opt * (xnorm + parameter_tolerance_opt);
        if (sqrt(dx[0] * dx[0] + dx[1] * dx[1]) < ptol)
            break;
        const double xn[2] = {x[0] + dx[0], x[1] + dx[1]};
        double dn[2];
        distort_projected_ray<double>(xn, radial, tangential, dn);
        const double fn[2] = {target[0] - dn[0], target[1] - dn[1]};
        const double cost_change = 2 * s.cost - (fn[0] * fn[0] + fn[1] * fn[1]);
        const double t0 = 2 * s.g[0] - (s.jtj[0][0] * step[0] + s.jtj[0][1] * step[1]);
        const double t1 = 2 * s.g[1] - (s.jtj[1][0] * step[0] + s.jtj[1][1] * step[1]);
        const double model_cost_change = step[0] * t0 + step[1] * t1;
        const double rho = cost_change / model_cost_change;
        if (rho > 0)
        {
            x[0] = xn[0];
            x[1] = xn[1];
            if (fabs(cost_change) < function_tolerance)
                break;
            tiny_update(s, x, target, radial, tangential, false);
            if (s.gmax < gradient_tolerance || s.cost < cost_threshold)
                break;
            const double tmp = 2 * rho - 1;
            u = u * fmax(1 / 3., 1 - tmp * tmp * tmp);
            v = 2;
        }
        else
        {
            if (fabs(cost_change) < function_tolerance)
                break;
            u *= v;
            v *= 2;
        }
    }
}

// model8 = {f, ppx, ppy, k1, k2, k3, p1, p2}
OCHIP_UD void image_to_3d(const double keypoint[2], const double model8[8], double ray[3])
{
    const double unprojected[2] = {(keypoint[0] - model8[1]) / model8[0], (keypoint[1] - model8[2]) / model8[0]};
    double und[2] = {unprojected[0], unprojected[1]};
    if (model8[3] != 0 || model8[4] != 0 || model8[5] != 0 || model8[6] != 0 || model8[7] != 0)
    {
        const double ppn = sqrt(model8[1] * model8[1] + model8[2] * model8[2]);
        tiny_solve_distortion(unprojected, model8 + 3, model8 + 6, 1e-2 / (ppn + model8[0]), und);
    }
    // Eigen's homogeneous().normalized(): divide by the norm if it is positive
    const double z = und[0] * und[0] + und[1] * und[1] + 1.0 * 1.0;
    ray[0] = und[0];
    ray[1] = und[1];
    ray[2] = 1.0;
    if (z > 0)
    {
        const double n = sqrt(z);
        ray[0] = und[0] / n;
        ray[1] = und[1] / n;
        ray[2] = 1.0 / n;
    }
}

} // namespace ochip_ud
