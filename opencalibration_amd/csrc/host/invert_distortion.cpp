#include "invert_distortion.hpp"

#include "ransac.hpp" // image_to_3d of the forward model

#include <algorithm>
#include <cmath>
#include <limits>
#include <vector>

namespace opencalibration_amd
{

namespace
{

// distortProjectedRay (distort_keypoints.hpp:26-42) and its derivative with respect to the five coefficients
// (k1 k2 k3 p1 p2): the distorted point is linear in them
void distort(const double p[2], const double k[3], const double t[2], double out[2], double dk[2][5])
{
    double r2[3];
    r2[0] = p[0] * p[0] + p[1] * p[1];
    r2[1] = r2[0] * r2[0];
    r2[2] = r2[1] * r2[0];
    const double radial = k[0] * r2[0] + k[1] * r2[1] + k[2] * r2[2];
    const double prod = p[0] * p[1];
    for (int i = 0; i < 2; i++)
    {
        out[i] = (1.0 + radial) * p[i] + 2.0 * prod * t[i] + t[1 - i] * (r2[0] + 2.0 * p[i] * p[i]);
        if (dk)
        {
            for (int c = 0; c < 3; c++)
                dk[i][c] = r2[c] * p[i];
            dk[i][3 + i] = 2.0 * prod;
            dk[i][3 + (1 - i)] = r2[0] + 2.0 * p[i] * p[i];
        }
    }
}

// Levenberg-Marquardt on five unknowns with the control flow of ceres::TinySolver [third party, ceres/tiny_solver.h:
// Jacobi scaling fixed by the first Jacobian, (J'J + u diag) step = J'r by LDL', gain ratio, Nielsen's update of u],
// default options.  eval(x, residuals, jacobian or null): residuals m, jacobian m x 5 row-major.
template <typename Eval> void tiny_solver5(Eval &&eval, int m, double x[5])
{
    constexpr int N = 5;
    std::vector<double> r(m), J((size_t)m * N), trial(m);
    double col_scale[N], H[N][N], g[N], cost = 0, gmax = 0;
    bool first = true;
    auto linearise = [&]() {
        eval(x, r.data(), J.data());
        for (int i = 0; i < m; i++)
            r[i] = -r[i];
        if (first)
        {
            for (int c = 0; c < N; c++)
            {
                double s = 0;
                for (int i = 0; i < m; i++)
                    s += J[(size_t)i * N + c] * J[(size_t)i * N + c];
                col_scale[c] = 1.0 / (1.0 + std::sqrt(s));
            }
            first = false;
        }
        for (int i = 0; i < m; i++)
            for (int c = 0; c < N; c++)
                J[(size_t)i * N + c] *= col_scale[c];
        gmax = 0;
        for (int a = 0; a < N; a++)
        {
            for (int b = 0; b < N; b++)
            {
                double s = 0;
                for (int i = 0; i < m; i++)
                    s += J[(size_t)i * N + a] * J[(size_t)i * N + b];
                H[a][b] = s;
            }
            double s = 0;
            for (int i = 0; i < m; i++)
                s += J[(size_t)i * N + a] * r[i];
            g[a] = s;
            gmax = std::max(gmax, std::abs(s));
        }
        cost = 0;
        for (int i = 0; i < m; i++)
            cost += r[i] * r[i];
        cost /= 2;
    };
    linearise();
    const double eps = std::numeric_limits<double>::epsilon();
    if (gmax < 1e-10 || cost < eps)
        return;
    double u = 1.0 / 1e4, v = 2;
    for (int it = 1; it < 50; it++)
    {
        double A[N][N], L[N][N] = {}, D[N], y[N], step[N];
        for (int a = 0; a < N; a++)
            for (int b = 0; b < N; b++)
                A[a][b] = H[a][b];
        for (int a = 0; a < N; a++)
        {
            const double d = std::sqrt(u * std::min(std::max(H[a][a], 1e-6), 1e32));
            A[a][a] += d * d;
        }
        for (int j = 0; j < N; j++)
        {
            double d = A[j][j];
            for (int k = 0; k < j; k++)
                d -= L[j][k] * L[j][k] * D[k];
            D[j] = d;
            for (int i = j + 1; i < N; i++)
            {
                double s = A[i][j];
                for (int k = 0; k < j; k++)
                    s -= L[i][k] * L[j][k] * D[k];
                L[i][j] = s / d;
            }
        }
        for (int i = 0; i < N; i++)
        {
            y[i] = g[i];
            for (int k = 0; k < i; k++)
                y[i] -= L[i][k] * y[k];
        }
        for (int i = N - 1; i >= 0; i--)
        {
            step[i] = y[i] / D[i];
            for (int k = i + 1; k < N; k++)
                step[i] -= L[k][i] * step[k];
        }
        double dx[N], xn[N], dx2 = 0, x2 = 0;
        for (int i = 0; i < N; i++)
        {
            dx[i] = col_scale[i] * step[i];
            dx2 += dx[i] * dx[i];
            x2 += x[i] * x[i];
            xn[i] = x[i] + dx[i];
        }
        if (std::sqrt(dx2) < 1e-8 * (std::sqrt(x2) + 1e-8))
            break;
        eval(xn, trial.data(), nullptr);
        double f2 = 0;
        for (int i = 0; i < m; i++)
            f2 += trial[i] * trial[i];
        const double cost_change = 2 * cost - f2;
        double model_change = 0;
        for (int a = 0; a < N; a++)
        {
            double t = 2 * g[a];
            for (int b = 0; b < N; b++)
                t -= H[a][b] * step[b];
            model_change += step[a] * t;
        }
        const double rho = cost_change / model_change;
        if (rho > 0)
        {
            for (int i = 0; i < N; i++)
                x[i] = xn[i];
            if (std::abs(cost_change) < 1e-6)
                break;
            linearise();
            if (gmax < 1e-10 || cost < eps)
                break;
            const double t = 2 * rho - 1;
            u *= std::max(1 / 3., 1 - t * t * t);
            v = 2;
        }
        else
        {
            if (std::abs(cost_change) < 1e-6)
                break;
            u *= v;
            v *= 2;
        }
    }
}

} // namespace

void image_to_3d(const double keypoint[2], const InverseCameraModel &m, double ray[3])
{
    const double u[2] = {(keypoint[0] - m.principle_point[0]) / m.focal_length_pixels,
                         (keypoint[1] - m.principle_point[1]) / m.focal_length_pixels};
    double und[2];
    distort(u, m.radial_distortion, m.tangential_distortion, und, nullptr);
    const double z = und[0] * und[0] + und[1] * und[1] + 1.0;
    const double l = std::sqrt(z);
    ray[0] = und[0] / l;
    ray[1] = und[1] / l;
    ray[2] = 1.0 / l;
}

void image_from_3d(const double ray[3], const CameraModel &m, double pixel[2])
{
    const double z = ray[2] < 1e-3 ? 1e-3 : ray[2];
    const double p[2] = {ray[0] / z, ray[1] / z};
    double d[2];
    distort(p, m.radial_distortion, m.tangential_distortion, d, nullptr);
    pixel[0] = d[0] * m.focal_length_pixels + m.principle_point[0];
    pixel[1] = d[1] * m.focal_length_pixels + m.principle_point[1];
}

InverseCameraModel convertModel(const CameraModel &standard)
{
    InverseCameraModel inv;
    static_cast<CameraModel &>(inv) = standard;
    for (double &k : inv.radial_distortion)
        k *= -1; // the first guess of the reference, overwritten by the fit below
    inv.tangential_distortion[0] = inv.tangential_distortion[1] = 0;
    const size_t si = standard.pixels_cols / 20, sj = standard.pixels_rows / 20;
    if (si == 0 || sj == 0)
        return inv;
    struct sample
    {
        double ray[3], u[2]; // the training ray and the normalised pixel that the forward model maps it to
    };
    std::vector<sample> samples;
    for (size_t i = 0; i < standard.pixels_cols; i += si)
        for (size_t j = 0; j < standard.pixels_rows; j += sj)
        {
            const double p[2] = {(double)i, (double)j};
            sample s;
            image_to_3d(p, standard, s.ray);
            double p2[2];
            image_from_3d(s.ray, standard, p2);
            if (std::isnan(s.ray[0]) || std::isnan(s.ray[1]) || std::isnan(s.ray[2]))
                continue;
            s.u[0] = (p2[0] - inv.principle_point[0]) / inv.focal_length_pixels;
            s.u[1] = (p2[1] - inv.principle_point[1]) / inv.focal_length_pixels;
            samples.push_back(s);
        }
    // residual of sample s: normalize(undistort(u_s; params), 1) - ray_s
    auto eval = [&](const double *x, double *res, double *jac) {
        for (size_t c = 0; c < samples.size(); c++)
        {
            const sample &s = samples[c];
            double und[2], dk[2][5] = {};
            distort(s.u, x, x + 3, und, dk);
            const double l = std::sqrt(und[0] * und[0] + und[1] * und[1] + 1.0);
            const double n[3] = {und[0] / l, und[1] / l, 1.0 / l};
            for (int k = 0; k < 3; k++)
                res[3 * c + k] = n[k] - s.ray[k];
            if (jac)
                for (int k = 0; k < 3; k++)
                    for (int q = 0; q < 5; q++)
                    {
                        // d n_k / d und_a = (delta_ka - n_k n_a) / l for a = x, y
                        double d = 0;
                        for (int a = 0; a < 2; a++)
                            d += ((k == a ? 1.0 : 0.0) - n[k] * n[a]) / l * dk[a][q];
                        jac[(3 * c + k) * 5 + q] = d;
                    }
        }
    };
    double params[5] = {0, 0, 0, 0, 0};
    if (!samples.empty())
        tiny_solver5(eval, (int)samples.size() * 3, params);
    for (int i = 0; i < 3; i++)
        inv.radial_distortion[i] = params[i];
    inv.tangential_distortion[0] = params[3];
    inv.tangential_distortion[1] = params[4];
    return inv;
}

CameraModel convertModel(const InverseCameraModel &inv, size_t id)
{
    CameraModel standard = inv;
    standard.id = id;
    for (double &k : standard.radial_distortion)
        k *= -1;
    standard.tangential_distortion[0] = standard.tangential_distortion[1] = 0;
    const size_t si = inv.pixels_cols / 20, sj = inv.pixels_rows / 20;
    if (si == 0 || sj == 0)
        return standard;
    struct sample
    {
        double rp[2], target[2]; // projected training ray, its pixel
    };
    std::vector<sample> samples;
    for (size_t i = 0; i < inv.pixels_cols; i += si)
        for (size_t j = 0; j < inv.pixels_rows; j += sj)
        {
            const double p[2] = {(double)i, (double)j};
            double ray[3];
            image_to_3d(p, inv, ray);
            if (std::isnan(ray[0]) || std::isnan(ray[1]) || std::isnan(ray[2]))
                continue;
            const double z = ray[2] < 1e-3 ? 1e-3 : ray[2];
            samples.push_back(sample{{ray[0] / z, ray[1] / z}, {p[0], p[1]}});
        }
    const double f = standard.focal_length_pixels;
    auto eval = [&](const double *x, double *res, double *jac) {
        for (size_t c = 0; c < samples.size(); c++)
        {
            double d[2], dk[2][5] = {};
            distort(samples[c].rp, x, x + 3, d, dk);
            for (int k = 0; k < 2; k++)
            {
                res[2 * c + k] = d[k] * f + standard.principle_point[k] - samples[c].target[k];
                if (jac)
                    for (int q = 0; q < 5; q++)
                        jac[(2 * c + k) * 5 + q] = dk[k][q] * f;
            }
        }
    };
    double params[5] = {0, 0, 0, 0, 0};
    if (!samples.empty())
        tiny_solver5(eval, (int)samples.size() * 2, params);
    for (int i = 0; i < 3; i++)
        standard.radial_distortion[i] = params[i];
    standard.tangential_distortion[0] = params[3];
    standard.tangential_distortion[1] = params[4];
    return standard;
}

} // namespace opencalibration_amd
