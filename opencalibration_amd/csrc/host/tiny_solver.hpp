// Levenberg-Marquardt on N unknowns with the control flow of ceres::TinySolver [third party, ceres/tiny_solver.h: Jacobi
// scaling fixed by the first Jacobian, (J'J + u diag) step = J'r by LDL', gain ratio, Nielsen's update of u] and its
// options.  eval(x, residuals, jacobian or null): residuals m, jacobian m x N row-major.  Returns the final cost.
#pragma once

#include <algorithm>
#include <cmath>
#include <limits>
#include <vector>

namespace opencalibration_amd
{

struct tiny_solver_options // ceres::TinySolver::Options and its defaults
{
    double gradient_tolerance = 1e-10, parameter_tolerance = 1e-8, function_tolerance = 1e-6;
    double cost_threshold = std::numeric_limits<double>::epsilon(), initial_trust_region_radius = 1e4;
    int max_num_iterations = 50;
};

template <int N, typename Eval> double tiny_solver_n(Eval &&eval, int m, double *x, const tiny_solver_options &o)
{
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
    if (gmax < o.gradient_tolerance || cost < o.cost_threshold)
        return cost;
    double u = 1.0 / o.initial_trust_region_radius, v = 2;
    for (int it = 1; it < o.max_num_iterations; it++)
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
        if (std::sqrt(dx2) < o.parameter_tolerance * (std::sqrt(x2) + o.parameter_tolerance))
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
            if (std::abs(cost_change) < o.function_tolerance)
                break;
            linearise();
            if (gmax < o.gradient_tolerance || cost < o.cost_threshold)
                break;
            const double tmp = 2 * rho - 1;
            u = u * std::max(1 / 3., 1 - tmp * tmp * tmp);
            v = 2;
        }
        else
        {
            if (std::abs(cost_change) < o.function_tolerance)
                break;
            u *= v;
            v *= 2;
        }
    }
    return cost;
}

} // namespace opencalibration_amd
