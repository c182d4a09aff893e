// ORACLE — test infrastructure only (see oracle.hpp).
// Restates src/distort/distort_keypoints.cpp:48-137 and include/opencalibration/distort/distort_keypoints.hpp:26-66.
#include "jet.hpp"
#include "oracle.hpp"

namespace oracle
{

// distort_keypoints.hpp:26-42
template <typename T>
static void distortProjectedRay(const T ray_projected[2], const double radial[3], const double tangential[2], T out[2])
{
    T r2[3];
    r2[0] = ray_projected[0] * ray_projected[0] + ray_projected[1] * ray_projected[1];
    for (int i = 1; i < 3; i++)
        r2[i] = r2[i - 1] * r2[0];
    const T radial_dot = T(radial[0]) * r2[0] + T(radial[1]) * r2[1] + T(radial[2]) * r2[2];
    const T prod = ray_projected[0] * ray_projected[1];
    for (int i = 0; i < 2; i++)
    {
        out[i] = (T(1.0) + radial_dot) * ray_projected[i] + T(2.0) * prod * T(tangential[i]) +
                 T(tangential[1 - i]) * (r2[0] + T(2.0) * ray_projected[i] * ray_projected[i]);
    }
}

// ceres::TinySolver<TinySolverAutoDiffFunction<DistortionFunctor,2,2>> restated [3P] (ceres/tiny_solver.h):
// Levenberg-Marquardt with Jacobi scaling from the first Jacobian, LDLT normal-equation solve,
// Nielsen's u/v damping update.  Residual = target - distort(x).  distort_keypoints.cpp:11-44,77-90.
static void tiny_solve_distortion(const double target[2], const double radial[3], const double tangential[2],
                                  double parameter_tolerance_opt, double x[2])
{
    const double gradient_tolerance = parameter_tolerance_opt * 1e-2;
    const double function_tolerance = 1e-6;
    const double cost_threshold = 1e-16;
    const int max_num_iterations = 10;
    const double initial_trust_region_radius = 1e4;

    double J[2][2], r[2], jac_scale[2] = {1, 1}, jtj[2][2], g[2], cost = 0, gmax = 0;
    int iterations = 0;

    auto update = [&](const double *xx) {
        Jet<2> p[2] = {Jet<2>(xx[0], 0), Jet<2>(xx[1], 1)}, d[2];
        distortProjectedRay<Jet<2>>(p, radial, tangential, d);
        for (int i = 0; i < 2; i++)
        {
            const Jet<2> res = Jet<2>(target[i]) - d[i];
            r[i] = -res.a; // residuals_ = -residuals_
            J[i][0] = res.v[0];
            J[i][1] = res.v[1];
        }
        if (iterations == 0)
            for (int c = 0; c < 2; c++)
                jac_scale[c] = 1.0 / (1.0 + std::sqrt(J[0][c] * J[0][c] + J[1][c] * J[1][c]));
        for (int i = 0; i < 2; i++)
            for (int c = 0; c < 2; c++)
                J[i][c] *= jac_scale[c];
        for (int a = 0; a < 2; a++)
        {
            for (int b = 0; b < 2; b++)
                jtj[a][b] = J[0][a] * J[0][b] + J[1][a] * J[1][b];
            g[a] = J[0][a] * r[0] + J[1][a] * r[1];
        }
        gmax = std::max(std::abs(g[0]), std::abs(g[1]));
        cost = (r[0] * r[0] + r[1] * r[1]) / 2;
    };

    update(x);
    if (gmax < gradient_tolerance || cost < cost_threshold)
        return;

    double u = 1.0 / initial_trust_region_radius, v = 2;
    for (iterations = 1; iterations < max_num_iterations; iterations++)
    {
        double A[2][2] = {{jtj[0][0], jtj[0][1]}, {jtj[1][0], jtj[1][1]}};
        for (int i = 0; i < 2; i++)
        {
            const double d = std::sqrt(u * std::min(std::max(jtj[i][i], 1e-6), 1e32));
            A[i][i] += d * d;
        }
        // 2x2 LDLT (pivot on the larger diagonal) solve A * step = g
        double step[2];
        {
            const int p = A[1][1] > A[0][0] ? 1 : 0, q = 1 - p;
            const double d0 = A[p][p], l = A[q][p] / d0, d1 = A[q][q] - l * A[q][p];
            const double y0 = g[p], y1 = g[q] - l * y0;
            const double z1 = y1 / d1, z0 = y0 / d0 - l * z1;
            step[p] = z0;
            step[q] = z1;
        }
        const double dx[2] = {jac_scale[0] * step[0], jac_scale[1] * step[1]};
        const double xnorm = std::sqrt(x[0] * x[0] + x[1] * x[1]);
        const double ptol = parameter_tolerance_opt * (xnorm + parameter_tolerance_opt);
        if (std::sqrt(dx[0] * dx[0] + dx[1] * dx[1]) < ptol)
            break;
        const double xn[2] = {x[0] + dx[0], x[1] + dx[1]};
        double dn[2];
        distortProjectedRay<double>(xn, radial, tangential, dn);
        const double fn[2] = {target[0] - dn[0], target[1] - dn[1]};
        const double cost_change = 2 * cost - (fn[0] * fn[0] + fn[1] * fn[1]);
        const double t0 = 2 * g[0] - (jtj[0][0] * step[0] + jtj[0][1] * step[1]);
        const double t1 = 2 * g[1] - (jtj[1][0] * step[0] + jtj[1][1] * step[1]);
        const double model_cost_change = step[0] * t0 + step[1] * t1;
        const double rho = cost_change / model_cost_change;
        if (rho > 0)
        {
            x[0] = xn[0];
            x[1] = xn[1];
            if (std::abs(cost_change) < function_tolerance)
                break;
            update(x);
            if (gmax < gradient_tolerance || cost < cost_threshold)
                break;
            const double tmp = 2 * rho - 1;
            u = u * std::max(1 / 3., 1 - tmp * tmp * tmp);
            v = 2;
        }
        else
        {
            if (std::abs(cost_change) < function_tolerance)
                break;
            u *= v;
            v *= 2;
        }
    }
}

static inline bool has_distortion(const camera_model &m)
{
    return m.radial_distortion[0] != 0 || m.radial_distortion[1] != 0 || m.radial_distortion[2] != 0 ||
           m.tangential_distortion[0] != 0 || m.tangential_distortion[1] != 0;
}

Vec3 image_to_3d(const double keypoint[2], const camera_model &model) // distort_keypoints.cpp:68-103
{
    const double unprojected[2] = {(keypoint[0] - model.principle_point[0]) / model.focal_length_pixels,
                                   (keypoint[1] - model.principle_point[1]) / model.focal_length_pixels};
    double und[2] = {unprojected[0], unprojected[1]};
    if (has_distortion(model))
    {
        const double ppn = std::sqrt(model.principle_point[0] * model.principle_point[0] +
                                     model.principle_point[1] * model.principle_point[1]);
        tiny_solve_distortion(unprojected, model.radial_distortion, model.tangential_distortion,
                              1e-2 / (ppn + model.focal_length_pixels), und);
    }
    return normalized(Vec3{und[0], und[1], 1.0});
}

Vec2 image_from_3d(const Vec3 &ray, const camera_model &model) // distort_keypoints.hpp:44-66 (forward model)
{
    const double z = ray.z;
    const double min_z = 1e-3;
    const double clamped_z = (z < min_z) ? min_z : z;
    const double rp[2] = {ray.x / clamped_z, ray.y / clamped_z};
    double rd[2];
    distortProjectedRay<double>(rp, model.radial_distortion, model.tangential_distortion, rd);
    return Vec2{rd[0] * model.focal_length_pixels + model.principle_point[0],
                rd[1] * model.focal_length_pixels + model.principle_point[1]};
}

std::vector<correspondence> distort_keypoints(const std::vector<feature_2d> &features1,
                                              const std::vector<feature_2d> &features2,
                                              const std::vector<feature_match> &matches, const camera_model &model1,
                                              const camera_model &model2) // distort_keypoints.cpp:48-66
{
    std::vector<correspondence> distorted;
    distorted.reserve(matches.size());
    for (const feature_match &m : matches)
    {
        correspondence cor;
        cor.measurement1 = image_to_3d(features1[m.feature_index_1].location, model1);
        cor.measurement2 = image_to_3d(features2[m.feature_index_2].location, model2);
        cor.quality = m.distance;
        distorted.push_back(cor);
    }
    return distorted;
}

} // namespace oracle
