// ORACLE — test infrastructure only.  See mini_ceres.hpp for what is restated and from where. [3P]
#include "mini_ceres.hpp"

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <limits>

namespace oracle
{
namespace mc
{

int Problem::AddParameterBlock(double *p, int size)
{
    auto it = index.find(p);
    if (it != index.end())
        return it->second;
    ParameterBlock b;
    b.data = p;
    b.size = size;
    blocks.push_back(b);
    index.emplace(p, (int)blocks.size() - 1);
    return (int)blocks.size() - 1;
}

void Problem::AddResidualBlock(CostFunction *cost, const LossFunction *loss, const std::vector<double *> &params)
{
    ResidualBlock rb;
    rb.cost.reset(cost);
    rb.loss = loss;
    for (size_t i = 0; i < params.size(); i++)
        rb.blocks.push_back(AddParameterBlock(params[i], cost->block_sizes[i]));
    residuals.push_back(std::move(rb));
}

void Problem::SetManifold(double *p, Manifold m)
{
    blocks[index.at(p)].manifold = m;
}
void Problem::SetSubsetManifold(double *p, const std::vector<int> &constant_parameters)
{
    ParameterBlock &b = blocks[index.at(p)];
    b.manifold = Manifold::SUBSET;
    b.subset_constant = constant_parameters;
}
void Problem::SetParameterLowerBound(double *p, int idx, double v)
{
    ParameterBlock &b = blocks[index.at(p)];
    if (b.lower.empty())
        b.lower.assign(b.size, -std::numeric_limits<double>::max());
    b.lower[idx] = v;
}
void Problem::SetParameterUpperBound(double *p, int idx, double v)
{
    ParameterBlock &b = blocks[index.at(p)];
    if (b.upper.empty())
        b.upper.assign(b.size, std::numeric_limits<double>::max());
    b.upper[idx] = v;
}
void Problem::SetParameterBlockConstant(double *p)
{
    blocks[index.at(p)].constant = true;
}
void Problem::SetParameterBlockVariable(double *p)
{
    blocks[index.at(p)].constant = false;
}
bool Problem::IsParameterBlockConstant(double *p) const
{
    return blocks[index.at(p)].constant;
}
std::vector<double *> Problem::GetParameterBlocks() const
{
    std::vector<double *> out;
    for (const auto &b : blocks)
        out.push_back(b.data);
    return out;
}

namespace
{

// EigenQuaternionManifold (ceres/manifold.cc, Order XYZW): plus = q_delta * x, q_delta = (sin|d|/|d| d, cos|d|)
void quat_plus(const double *x, const double *delta, double *out)
{
    const double n2 = delta[0] * delta[0] + delta[1] * delta[1] + delta[2] * delta[2];
    const double n = std::sqrt(n2);
    if (n == 0.0)
    {
        for (int i = 0; i < 4; i++)
            out[i] = x[i];
        return;
    }
    const double s = std::sin(n) / n;
    const double dx = s * delta[0], dy = s * delta[1], dz = s * delta[2], dw = std::cos(n);
    const double qx = x[0], qy = x[1], qz = x[2], qw = x[3];
    // Eigen quaternion product (a = q_delta, b = x)
    out[3] = dw * qw - dx * qx - dy * qy - dz * qz;
    out[0] = dw * qx + dx * qw + dy * qz - dz * qy;
    out[1] = dw * qy + dy * qw + dz * qx - dx * qz;
    out[2] = dw * qz + dz * qw + dx * qy - dy * qx;
}
void quat_plus_jacobian(const double *x, double *J /*4x3 row-major*/)
{
    const double qx = x[0], qy = x[1], qz = x[2], qw = x[3];
    const double v[12] = {qw, qz, -qy, -qz, qw, qx, qy, -qx, qw, -qx, -qy, -qz};
    for (int i = 0; i < 12; i++)
        J[i] = v[i];
}

struct Program
{
    Problem *problem;
    std::vector<int> var_blocks;          // indices of non-constant parameter blocks, program order
    std::vector<int> tangent_offset;      // per problem block (-1 if constant)
    std::vector<int> ambient_offset;      // per problem block (-1 if constant)
    std::vector<int> active_residuals;    // residual blocks with at least one variable block
    int num_tangent = 0, num_ambient = 0, num_res = 0;
    double fixed_cost = 0;
};

struct Evaluation
{
    double cost = 0;
    std::vector<double> residuals;               // stacked, corrected
    std::vector<std::vector<double>> jac;        // per active residual block: rows x (sum tangent of its var blocks)
};

// Evaluate one residual block at the parameter values in `state` (ambient vector for variable blocks).
bool eval_block(const Program &prog, const ResidualBlock &rb, const std::vector<double> &state, bool want_jac,
                double *cost, std::vector<double> &res, std::vector<double> *jac_tangent)
{
    const Problem &P = *prog.problem;
    const int nb = (int)rb.blocks.size();
    const int nr = rb.cost->num_residuals;
    std::vector<const double *> params(nb);
    for (int i = 0; i < nb; i++)
    {
        const int b = rb.blocks[i];
        params[i] = prog.ambient_offset[b] >= 0 ? &state[prog.ambient_offset[b]] : P.blocks[b].data;
    }
    res.assign(nr, 0.0);
    std::vector<std::vector<double>> jac_store(nb);
    std::vector<double *> jac_ptr(nb, nullptr);
    if (want_jac)
        for (int i = 0; i < nb; i++)
            if (prog.ambient_offset[rb.blocks[i]] >= 0)
            {
                jac_store[i].assign((size_t)nr * P.blocks[rb.blocks[i]].size, 0.0);
                jac_ptr[i] = jac_store[i].data();
            }
    if (!rb.cost->Evaluate(params.data(), res.data(), want_jac ? jac_ptr.data() : nullptr))
        return false;
    for (double r : res)
        if (!std::isfinite(r))
            return false;
    double sq = 0;
    for (double r : res)
        sq += r * r;

    // ambient -> tangent
    int tcols = 0;
    if (want_jac)
    {
        for (int i = 0; i < nb; i++)
            if (jac_ptr[i])
                tcols += P.blocks[rb.blocks[i]].tangent_size();
        jac_tangent->assign((size_t)nr * tcols, 0.0);
        int c0 = 0;
        for (int i = 0; i < nb; i++)
        {
            if (!jac_ptr[i])
                continue;
            const ParameterBlock &pb = P.blocks[rb.blocks[i]];
            if (pb.manifold == Manifold::EIGEN_QUATERNION)
            {
                double PJ[12];
                quat_plus_jacobian(params[i], PJ);
                for (int r = 0; r < nr; r++)
                    for (int c = 0; c < 3; c++)
                    {
                        double s = 0;
                        for (int k = 0; k < 4; k++)
                            s += jac_ptr[i][r * 4 + k] * PJ[k * 3 + c];
                        (*jac_tangent)[(size_t)r * tcols + c0 + c] = s;
                    }
                c0 += 3;
            }
            else if (pb.manifold == Manifold::SUBSET)
            {
                int t = 0;
                for (int c = 0; c < pb.size; c++)
                {
                    if (pb.is_subset_constant(c))
                        continue;
                    for (int r = 0; r < nr; r++)
                        (*jac_tangent)[(size_t)r * tcols + c0 + t] = jac_ptr[i][r * pb.size + c];
                    t++;
                }
                c0 += t;
            }
            else
            {
                for (int r = 0; r < nr; r++)
                    for (int c = 0; c < pb.size; c++)
                        (*jac_tangent)[(size_t)r * tcols + c0 + c] = jac_ptr[i][r * pb.size + c];
                c0 += pb.size;
            }
        }
        for (double v : *jac_tangent)
            if (!std::isfinite(v))
                return false;
    }

    if (!rb.loss)
    {
        *cost = 0.5 * sq;
        return true;
    }
    double rho[3];
    rb.loss->Evaluate(sq, rho);
    *cost = 0.5 * rho[0];
    // Corrector (ceres/corrector.cc)
    const double sqrt_rho1 = std::sqrt(rho[1]);
    double residual_scaling, alpha_sq_norm;
    if (sq == 0.0 || rho[2] <= 0.0)
    {
        residual_scaling = sqrt_rho1;
        alpha_sq_norm = 0.0;
    }
    else
    {
        const double D = 1.0 + 2.0 * sq * rho[2] / rho[1];
        const double alpha = 1.0 - std::sqrt(D);
        residual_scaling = sqrt_rho1 / (1 - alpha);
        alpha_sq_norm = alpha / sq;
    }
    if (want_jac)
    {
        if (alpha_sq_norm == 0.0)
            for (double &v : *jac_tangent)
                v *= sqrt_rho1;
        else
            for (int c = 0; c < tcols; c++)
            {
                double rtj = 0;
                for (int r = 0; r < nr; r++)
                    rtj += (*jac_tangent)[(size_t)r * tcols + c] * res[r];
                for (int r = 0; r < nr; r++)
                    (*jac_tangent)[(size_t)r * tcols + c] =
                        sqrt_rho1 * ((*jac_tangent)[(size_t)r * tcols + c] - alpha_sq_norm * res[r] * rtj);
            }
    }
    for (double &r : res)
        r *= residual_scaling;
    return true;
}

bool evaluate(const Program &prog, const std::vector<double> &state, bool want_jac, Evaluation *ev)
{
    ev->cost = 0;
    ev->residuals.clear();
    if (want_jac)
        ev->jac.assign(prog.active_residuals.size(), {});
    std::vector<double> res;
    for (size_t k = 0; k < prog.active_residuals.size(); k++)
    {
        const ResidualBlock &rb = prog.problem->residuals[prog.active_residuals[k]];
        double c = 0;
        if (!eval_block(prog, rb, state, want_jac, &c, res, want_jac ? &ev->jac[k] : nullptr))
            return false;
        ev->cost += c;
        ev->residuals.insert(ev->residuals.end(), res.begin(), res.end());
    }
    return true;
}

// Cholesky solve of A x = b (A symmetric positive definite, row-major n x n, destroyed).  Rows are walked inside their
// profile only (first[i] = first non-zero column of row i; fill stays inside a row's profile), which skips exact zeros
// and nothing else: the arithmetic on the non-zero entries is that of the plain dense factorisation.
bool cholesky_solve(std::vector<double> &A, std::vector<double> &b, int n)
{
    std::vector<int> first(n);
    for (int i = 0; i < n; i++)
    {
        int f = 0;
        while (f < i && A[(size_t)i * n + f] == 0.0)
            f++;
        first[i] = f;
    }
    for (int j = 0; j < n; j++)
    {
        double d = A[(size_t)j * n + j];
        for (int k = first[j]; k < j; k++)
            d -= A[(size_t)j * n + k] * A[(size_t)j * n + k];
        if (!(d > 0) || !std::isfinite(d))
            return false;
        d = std::sqrt(d);
        A[(size_t)j * n + j] = d;
        for (int i = j + 1; i < n; i++)
        {
            if (first[i] > j)
                continue;
            double s = A[(size_t)i * n + j];
            for (int k = std::max(first[i], first[j]); k < j; k++)
                s -= A[(size_t)i * n + k] * A[(size_t)j * n + k];
            A[(size_t)i * n + j] = s / d;
        }
    }
    for (int i = 0; i < n; i++)
    {
        double s = b[i];
        for (int k = first[i]; k < i; k++)
            s -= A[(size_t)i * n + k] * b[k];
        b[i] = s / A[(size_t)i * n + i];
    }
    for (int i = n - 1; i >= 0; i--)
    {
        // column sweep: x_i is final, remove it from the rows above inside row i's profile
        b[i] = b[i] / A[(size_t)i * n + i];
        for (int k = first[i]; k < i; k++)
            b[k] -= A[(size_t)i * n + k] * b[i];
    }
    return true;
}


// ---- ceres/polynomial.cc restated [3P]: interpolating polynomial through samples (value and, where known, gradient),
// its minimum on an interval.  Coefficients highest power first.
struct FunctionSample
{
    double x = 0, value = 0, gradient = 0;
    bool value_is_valid = false, gradient_is_valid = false;
};
double eval_poly(const std::vector<double> &p, double x)
{
    double v = 0;
    for (double c : p)
        v = v * x + c;
    return v;
}
std::vector<double> find_interpolating_polynomial(const std::vector<FunctionSample> &samples)
{
    int num_constraints = 0;
    for (const auto &s : samples)
        num_constraints += (s.value_is_valid ? 1 : 0) + (s.gradient_is_valid ? 1 : 0);
    const int degree = num_constraints - 1;
    const int m = num_constraints;
    std::vector<double> lhs((size_t)m * m, 0.0), rhs(m, 0.0);
    int row = 0;
    for (const auto &s : samples)
    {
        if (s.value_is_valid)
        {
            for (int j = 0; j <= degree; ++j)
                lhs[(size_t)row * m + j] = std::pow(s.x, degree - j);
            rhs[row++] = s.value;
        }
        if (s.gradient_is_valid)
        {
            for (int j = 0; j < degree; ++j)
                lhs[(size_t)row * m + j] = (degree - j) * std::pow(s.x, degree - j - 1);
            rhs[row++] = s.gradient;
        }
    }
    // lhs.fullPivLu().solve(rhs)
    std::vector<int> colperm(m);
    for (int i = 0; i < m; i++)
        colperm[i] = i;
    for (int k = 0; k < m; k++)
    {
        int pr = k, pc = k;
        double best = -1;
        for (int c = k; c < m; c++)
            for (int r = k; r < m; r++)
                if (std::abs(lhs[(size_t)r * m + c]) > best)
                {
                    best = std::abs(lhs[(size_t)r * m + c]);
                    pr = r;
                    pc = c;
                }
        if (best <= 0)
            break;
        for (int c = 0; c < m; c++)
            std::swap(lhs[(size_t)k * m + c], lhs[(size_t)pr * m + c]);
        std::swap(rhs[k], rhs[pr]);
        for (int r = 0; r < m; r++)
            std::swap(lhs[(size_t)r * m + k], lhs[(size_t)r * m + pc]);
        std::swap(colperm[k], colperm[pc]);
        for (int r = k + 1; r < m; r++)
        {
            const double f = lhs[(size_t)r * m + k] / lhs[(size_t)k * m + k];
            for (int c = k; c < m; c++)
                lhs[(size_t)r * m + c] -= f * lhs[(size_t)k * m + c];
            rhs[r] -= f * rhs[k];
        }
    }
    std::vector<double> y(m, 0.0), out(m, 0.0);
    for (int k = m - 1; k >= 0; k--)
    {
        double v = rhs[k];
        for (int c = k + 1; c < m; c++)
            v -= lhs[(size_t)k * m + c] * y[c];
        y[k] = lhs[(size_t)k * m + k] != 0 ? v / lhs[(size_t)k * m + k] : 0.0;
    }
    for (int k = 0; k < m; k++)
        out[colperm[k]] = y[k];
    return out;
}
// real parts of the roots of a polynomial (FindPolynomialRoots with a null imaginary output): closed forms up to degree
// 2 as in polynomial.cc, Durand-Kerner above (polynomial.cc takes the eigenvalues of the companion matrix)
std::vector<double> polynomial_root_real_parts(std::vector<double> p)
{
    while (!p.empty() && p.front() == 0.0)
        p.erase(p.begin());
    const int degree = (int)p.size() - 1;
    std::vector<double> out;
    if (degree <= 0)
        return out;
    if (degree == 1)
    {
        out.push_back(-p[1] / p[0]);
        return out;
    }
    if (degree == 2)
    {
        const double a = p[0], b = p[1], c = p[2];
        const double D = b * b - 4 * a * c, sqrt_D = std::sqrt(std::abs(D));
        if (D >= 0)
        {
            if (b >= 0)
            {
                out.push_back((-b - sqrt_D) / (2.0 * a));
                out.push_back((2.0 * c) / (-b - sqrt_D));
            }
            else
            {
                out.push_back((2.0 * c) / (-b + sqrt_D));
                out.push_back((-b + sqrt_D) / (2.0 * a));
            }
        }
        else
        {
            out.push_back(-b / (2.0 * a));
            out.push_back(-b / (2.0 * a));
        }
        return out;
    }
    std::vector<std::pair<double, double>> z(degree); // complex roots (re, im)
    for (int i = 0; i < degree; i++)
    {
        const double ang = 2.0 * M_PI * i / degree + 0.4, rad = 1.0 + std::abs(p.back() / p[0]);
        z[i] = {rad * std::cos(ang), rad * std::sin(ang)};
    }
    auto cmul = [](std::pair<double, double> a, std::pair<double, double> b) {
        return std::make_pair(a.first * b.first - a.second * b.second, a.first * b.second + a.second * b.first);
    };
    auto cdiv = [](std::pair<double, double> a, std::pair<double, double> b) {
        const double d = b.first * b.first + b.second * b.second;
        return std::make_pair((a.first * b.first + a.second * b.second) / d, (a.second * b.first - a.first * b.second) / d);
    };
    for (int it = 0; it < 500; it++)
    {
        double change = 0;
        for (int i = 0; i < degree; i++)
        {
            std::pair<double, double> v{p[0], 0.0};
            for (int k = 1; k <= degree; k++)
            {
                v = cmul(v, z[i]);
                v.first += p[k];
            }
            std::pair<double, double> den{p[0], 0.0};
            for (int j = 0; j < degree; j++)
                if (j != i)
                    den = cmul(den, {z[i].first - z[j].first, z[i].second - z[j].second});
            const auto d = cdiv(v, den);
            z[i].first -= d.first;
            z[i].second -= d.second;
            change = std::max(change, std::abs(d.first) + std::abs(d.second));
        }
        if (change < 1e-15)
            break;
    }
    for (const auto &r : z)
        out.push_back(r.first);
    return out;
}
void minimize_polynomial(const std::vector<double> &poly, double x_min, double x_max, double *optimal_x, double *optimal_value)
{
    *optimal_x = (x_min + x_max) / 2.0;
    *optimal_value = eval_poly(poly, *optimal_x);
    const double vmin = eval_poly(poly, x_min);
    if (vmin < *optimal_value)
    {
        *optimal_value = vmin;
        *optimal_x = x_min;
    }
    const double vmax = eval_poly(poly, x_max);
    if (vmax < *optimal_value)
    {
        *optimal_value = vmax;
        *optimal_x = x_max;
    }
    if (poly.size() <= 2)
        return;
    std::vector<double> deriv;
    const int degree = (int)poly.size() - 1;
    for (int i = 0; i < degree; i++)
        deriv.push_back((degree - i) * poly[i]);
    for (double root : polynomial_root_real_parts(deriv))
    {
        if (root < x_min || root > x_max)
            continue;
        const double v = eval_poly(poly, root);
        if (v < *optimal_value)
        {
            *optimal_value = v;
            *optimal_x = root;
        }
    }
}
// LineSearch::InterpolatingPolynomialMinimizingStepSize (ceres/line_search.cc), CUBIC interpolation
double interpolating_step(const FunctionSample &lowerbound, const FunctionSample &previous, const FunctionSample &current,
                          double min_step_size, double max_step_size)
{
    if (!current.value_is_valid || std::max(min_step_size, max_step_size) <= 0) // bisection when the value is not usable
        return std::min(std::max(current.x * 0.5, min_step_size), max_step_size);
    std::vector<FunctionSample> samples{lowerbound, current};
    if (previous.value_is_valid)
        samples.push_back(previous);
    double step = 0, unused = 0;
    minimize_polynomial(find_interpolating_polynomial(samples), min_step_size, max_step_size, &step, &unused);
    return step;
}

} // namespace

void Solve(const SolverOptions &opt, Problem *problem, SolverSummary *summary)
{
    *summary = SolverSummary();
    Program prog;
    prog.problem = problem;
    const int nblocks = (int)problem->blocks.size();
    prog.tangent_offset.assign(nblocks, -1);
    prog.ambient_offset.assign(nblocks, -1);
    // reduced program: drop constant blocks and blocks no residual touches (ceres/reduced_program)
    std::vector<char> used(nblocks, 0);
    for (const auto &rb : problem->residuals)
        for (int b : rb.blocks)
            used[b] = 1;
    for (int b = 0; b < nblocks; b++)
        if (used[b] && !problem->blocks[b].constant)
        {
            prog.var_blocks.push_back(b);
            prog.tangent_offset[b] = prog.num_tangent;
            prog.ambient_offset[b] = prog.num_ambient;
            prog.num_tangent += problem->blocks[b].tangent_size();
            prog.num_ambient += problem->blocks[b].size;
        }
    std::vector<double> x(prog.num_ambient);
    for (int b : prog.var_blocks)
        for (int k = 0; k < problem->blocks[b].size; k++)
            x[prog.ambient_offset[b] + k] = problem->blocks[b].data[k];
    for (size_t r = 0; r < problem->residuals.size(); r++)
    {
        bool any = false;
        for (int b : problem->residuals[r].blocks)
            any |= prog.ambient_offset[b] >= 0;
        if (any)
        {
            prog.active_residuals.push_back((int)r);
            prog.num_res += problem->residuals[r].cost->num_residuals;
        }
        else
        {
            double c = 0;
            std::vector<double> res;
            if (eval_block(prog, problem->residuals[r], x, false, &c, res, nullptr))
                prog.fixed_cost += c;
        }
    }
    summary->fixed_cost = prog.fixed_cost;
    summary->num_parameters_reduced = prog.num_tangent;
    summary->num_residuals_reduced = prog.num_res;
    if (prog.var_blocks.empty() || prog.active_residuals.empty())
    {
        summary->message = "no non-constant parameter blocks";
        summary->initial_cost = summary->final_cost = prog.fixed_cost;
        summary->usable = true;
        return;
    }

    const int n = prog.num_tangent;
    auto write_back = [&](const std::vector<double> &state) {
        for (int b : prog.var_blocks)
            for (int k = 0; k < problem->blocks[b].size; k++)
                problem->blocks[b].data[k] = state[prog.ambient_offset[b] + k];
    };
    auto plus = [&](const std::vector<double> &state, const std::vector<double> &delta, std::vector<double> &out) {
        out.resize(state.size());
        for (int b : prog.var_blocks)
        {
            const ParameterBlock &pb = problem->blocks[b];
            const double *s = &state[prog.ambient_offset[b]];
            const double *d = &delta[prog.tangent_offset[b]];
            double *o = &out[prog.ambient_offset[b]];
            if (pb.manifold == Manifold::EIGEN_QUATERNION)
                quat_plus(s, d, o);
            else if (pb.manifold == Manifold::SUBSET)
            {
                int t = 0;
                for (int k = 0; k < pb.size; k++)
                    o[k] = pb.is_subset_constant(k) ? s[k] : s[k] + d[t++];
            }
            else
                for (int k = 0; k < pb.size; k++)
                    o[k] = s[k] + d[k];
            // ParameterBlock::Plus projects onto the box constraints (ceres/parameter_block.h)
            if (!pb.lower.empty())
                for (int k = 0; k < pb.size; k++)
                    o[k] = std::max(o[k], pb.lower[k]);
            if (!pb.upper.empty())
                for (int k = 0; k < pb.size; k++)
                    o[k] = std::min(o[k], pb.upper[k]);
        }
    };
    bool is_constrained = false;
    for (int b : prog.var_blocks)
        is_constrained |= !problem->blocks[b].lower.empty() || !problem->blocks[b].upper.empty();
    // scatter block Jacobians into (optionally scaled) J'J and J'r
    auto normal_equations = [&](const Evaluation &ev, const std::vector<double> &scale, std::vector<double> &JtJ,
                                std::vector<double> &Jtr, std::vector<double> &colnorm2) {
        JtJ.assign((size_t)n * n, 0.0);
        Jtr.assign(n, 0.0);
        colnorm2.assign(n, 0.0);
        size_t roff = 0;
        for (size_t k = 0; k < prog.active_residuals.size(); k++)
        {
            const ResidualBlock &rb = problem->residuals[prog.active_residuals[k]];
            const int nr = rb.cost->num_residuals;
            std::vector<int> cols;
            for (int b : rb.blocks)
                if (prog.tangent_offset[b] >= 0)
                    for (int c = 0; c < problem->blocks[b].tangent_size(); c++)
                        cols.push_back(prog.tangent_offset[b] + c);
            const int tc = (int)cols.size();
            const std::vector<double> &J = ev.jac[k];
            for (int a = 0; a < tc; a++)
            {
                double g = 0;
                for (int r = 0; r < nr; r++)
                    g += J[(size_t)r * tc + a] * scale[cols[a]] * ev.residuals[roff + r];
                Jtr[cols[a]] += g;
                for (int b2 = 0; b2 < tc; b2++)
                {
                    double s = 0;
                    for (int r = 0; r < nr; r++)
                        s += J[(size_t)r * tc + a] * J[(size_t)r * tc + b2];
                    JtJ[(size_t)cols[a] * n + cols[b2]] += s * scale[cols[a]] * scale[cols[b2]];
                }
            }
            roff += nr;
        }
        for (int i = 0; i < n; i++)
            colnorm2[i] = JtJ[(size_t)i * n + i];
    };
    auto max_abs = [](const std::vector<double> &v) {
        double m = 0;
        for (double e : v)
            m = std::max(m, std::abs(e));
        return m;
    };
    auto norm = [](const std::vector<double> &v) {
        double s = 0;
        for (double e : v)
            s += e * e;
        return std::sqrt(s);
    };

    // ---- Init (trust_region_minimizer.cc: Init + IterationZero)
    Evaluation ev;
    if (!evaluate(prog, x, true, &ev))
    {
        summary->message = "initial evaluation failed";
        return;
    }
    std::vector<double> ones(n, 1.0), JtJ, g, cn2, scale(n, 1.0);
    normal_equations(ev, ones, JtJ, g, cn2);
    if (opt.jacobi_scaling)
        for (int i = 0; i < n; i++)
            scale[i] = 1.0 / (1.0 + std::sqrt(cn2[i]));
    double x_cost = ev.cost, x_norm = norm(x);
    summary->initial_cost = x_cost + prog.fixed_cost;
    double radius = opt.initial_trust_region_radius, decrease_factor = 2.0;
    bool reuse_diagonal = false;
    std::vector<double> diagonal(n, 0.0);
    int num_consecutive_invalid = 0;

    IterationSummary it0;
    it0.iteration = 0;
    it0.step_is_valid = it0.step_is_successful = true;
    it0.cost = x_cost + prog.fixed_cost;
    auto gradient_max_norm = [&](const std::vector<double> &state, const std::vector<double> &grad) {
        if (!is_constrained)
            return max_abs(grad);
        // |x - Plus(x, -gradient)| (ambient, projected onto the bounds): TrustRegionMinimizer::EvaluateGradientAndJacobian
        std::vector<double> neg(grad.size()), moved;
        for (size_t i = 0; i < grad.size(); i++)
            neg[i] = -grad[i];
        plus(state, neg, moved);
        double m = 0;
        for (size_t i = 0; i < state.size(); i++)
            m = std::max(m, std::abs(state[i] - moved[i]));
        return m;
    };
    it0.gradient_max_norm = gradient_max_norm(x, g);
    it0.trust_region_radius = radius;
    summary->iterations.push_back(it0);
    summary->usable = true;
    auto finish = [&](const char *msg) {
        summary->message = msg;
        summary->final_cost = x_cost + prog.fixed_cost;
        write_back(x);
    };
    if (it0.gradient_max_norm <= opt.gradient_tolerance)
        return finish("Gradient tolerance reached");

    std::vector<double> sJtJ, sg, scn2, A, step, delta, cand;
    normal_equations(ev, scale, sJtJ, sg, scn2);
    while (true)
    {
        IterationSummary is;
        is.iteration = summary->iterations.back().iteration + 1;
        // FinalizeIterationAndCheckIfMinimizerCanContinue checks of the previous iteration
        if (summary->iterations.back().iteration >= opt.max_num_iterations)
            return finish("Maximum number of iterations reached");
        if (radius <= opt.min_trust_region_radius)
            return finish("Minimum trust region radius reached");

        // ---- ComputeTrustRegionStep (LevenbergMarquardtStrategy::ComputeStep)
        if (!reuse_diagonal)
            for (int i = 0; i < n; i++)
                diagonal[i] = std::min(std::max(scn2[i], opt.min_lm_diagonal), opt.max_lm_diagonal);
        A = sJtJ;
        for (int i = 0; i < n; i++)
        {
            const double d = std::sqrt(diagonal[i] / radius);
            A[(size_t)i * n + i] += d * d;
        }
        step = sg;
        bool solved = cholesky_solve(A, step, n);
        for (double &s : step)
        {
            if (!std::isfinite(s))
                solved = false;
            s = -s;
        }
        reuse_diagonal = true;
        double model_cost_change = 0;
        if (solved)
        {
            // model_cost_change = -(J step)'(r + J step / 2) = -(step'g + step'J'J step / 2)
            double sg_dot = 0, quad = 0;
            for (int i = 0; i < n; i++)
            {
                sg_dot += step[i] * sg[i];
                double row = 0;
                for (int j = 0; j < n; j++)
                    row += sJtJ[(size_t)i * n + j] * step[j];
                quad += step[i] * row;
            }
            model_cost_change = -(sg_dot + quad / 2.0);
        }
        is.step_is_valid = solved && model_cost_change > 0.0;
        static const bool verbose = getenv("OC_RELAX_VERBOSE") != nullptr;
        if (verbose)
            fprintf(stderr, "[oracle relax] n=%d iter=%d cost=%.17g radius=%.6g model=%.17g solved=%d gmax=%.6g\n", n,
                    is.iteration, x_cost, radius, model_cost_change, (int)solved, summary->iterations.back().gradient_max_norm);
        if (!is.step_is_valid)
        {
            // HandleInvalidStep
            if (++num_consecutive_invalid >= opt.max_num_consecutive_invalid_steps)
            {
                summary->iterations.push_back(is);
                return finish("Too many consecutive invalid steps");
            }
            radius *= 0.5; // LevenbergMarquardtStrategy::StepIsInvalid
            reuse_diagonal = true;
            is.cost = x_cost + prog.fixed_cost;
            is.trust_region_radius = radius;
            is.gradient_max_norm = summary->iterations.back().gradient_max_norm;
            summary->iterations.push_back(is);
            continue;
        }
        num_consecutive_invalid = 0;
        delta.resize(n);
        for (int i = 0; i < n; i++)
            delta[i] = step[i] * scale[i];
        if (is_constrained && opt.max_num_line_search_step_size_iterations > 0)
        {
            // TrustRegionMinimizer::DoLineSearch: a projected Armijo search (cubic interpolation, sufficient decrease 1e-4,
            // contraction within [1e-3, 0.6], at most 20 steps) along delta; on success delta *= the step found
            double gdd = 0;
            for (int i = 0; i < n; i++)
                gdd += g[i] * delta[i];
            double dmax = 0;
            for (int i = 0; i < n; i++)
                dmax = std::max(dmax, std::abs(delta[i]));
            auto phi = [&](double a, FunctionSample *out) {
                std::vector<double> sd(n), xa;
                for (int i = 0; i < n; i++)
                    sd[i] = a * delta[i];
                plus(x, sd, xa);
                Evaluation eva;
                out->x = a;
                out->value_is_valid = out->gradient_is_valid = false;
                if (!evaluate(prog, xa, true, &eva))
                    return;
                std::vector<double> JtJa, ga, cna;
                normal_equations(eva, ones, JtJa, ga, cna);
                out->value = eva.cost;
                out->value_is_valid = std::isfinite(eva.cost);
                double gd = 0;
                for (int i = 0; i < n; i++)
                    gd += ga[i] * delta[i];
                out->gradient = gd;
                out->gradient_is_valid = out->value_is_valid && std::isfinite(gd);
            };
            FunctionSample initial, previous, current;
            initial.x = 0;
            initial.value = x_cost;
            initial.gradient = gdd;
            initial.value_is_valid = initial.gradient_is_valid = true;
            phi(1.0, &current);
            bool success = true;
            int ls_iterations = 0;
            while (!current.value_is_valid || current.value > (x_cost + 1e-4 * gdd * current.x))
            {
                if (++ls_iterations >= opt.max_num_line_search_step_size_iterations)
                {
                    success = false;
                    break;
                }
                const double next = interpolating_step(initial, previous, current, 1e-3 * current.x, 0.6 * current.x);
                if (next * dmax < 1e-9)
                {
                    success = false;
                    break;
                }
                previous = current;
                phi(next, &current);
            }
            if (success)
                for (int i = 0; i < n; i++)
                    delta[i] *= current.x;
        }
        plus(x, delta, cand);
        Evaluation cev;
        double candidate_cost = std::numeric_limits<double>::max();
        if (evaluate(prog, cand, false, &cev))
            candidate_cost = cev.cost;

        // ParameterToleranceReached
        {
            double s = 0;
            for (size_t i = 0; i < x.size(); i++)
                s += (x[i] - cand[i]) * (x[i] - cand[i]);
            is.step_norm = std::sqrt(s);
        }
        if (is.step_norm <= opt.parameter_tolerance * (x_norm + opt.parameter_tolerance))
        {
            summary->iterations.push_back(is);
            return finish("Parameter tolerance reached");
        }
        // FunctionToleranceReached
        is.cost_change = x_cost - candidate_cost;
        if (std::abs(is.cost_change) <= opt.function_tolerance * x_cost)
        {
            summary->iterations.push_back(is);
            return finish("Function tolerance reached");
        }
        is.relative_decrease = is.cost_change / model_cost_change;
        if (is.relative_decrease > opt.min_relative_decrease)
        {
            // HandleSuccessfulStep
            x = cand;
            x_norm = norm(x);
            if (!evaluate(prog, x, true, &ev))
            {
                summary->iterations.push_back(is);
                return finish("Jacobian evaluation failed");
            }
            x_cost = ev.cost;
            normal_equations(ev, ones, JtJ, g, cn2);
            normal_equations(ev, scale, sJtJ, sg, scn2);
            is.step_is_successful = true;
            is.gradient_max_norm = gradient_max_norm(x, g);
            const double t = 2.0 * is.relative_decrease - 1.0;
            radius = radius / std::max(1.0 / 3.0, 1.0 - t * t * t);
            radius = std::min(opt.max_trust_region_radius, radius);
            decrease_factor = 2.0;
            reuse_diagonal = false;
            summary->num_successful_steps++;
        }
        else
        {
            radius = radius / decrease_factor;
            decrease_factor *= 2.0;
            reuse_diagonal = true;
            is.gradient_max_norm = summary->iterations.back().gradient_max_norm;
            summary->num_unsuccessful_steps++;
        }
        is.cost = x_cost + prog.fixed_cost;
        is.trust_region_radius = radius;
        summary->iterations.push_back(is);
        if (is.step_is_successful && is.gradient_max_norm <= opt.gradient_tolerance)
            return finish("Gradient tolerance reached");
    }
}

} // namespace mc
} // namespace oracle
