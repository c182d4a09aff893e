// ORACLE — test infrastructure only (see oracle.hpp).
//
// Restatement of the parts of Ceres Solver 2.x the reference's relax stage uses [3P — Ceres is not
// under /root/reference; restated from its published documentation and algorithm descriptions]:
//   * Problem with parameter blocks, EigenQuaternionManifold, constant blocks (ceres/problem.h)
//   * AutoDiffCostFunction on Jets (ceres/autodiff_cost_function.h, jet.h -> oracle/jet.hpp)
//   * HuberLoss + the Triggs corrector (ceres/loss_function.h, corrector.cc)
//   * TrustRegionMinimizer with LevenbergMarquardtStrategy, jacobi scaling, monotonic steps
//     (trust_region_minimizer.cc, levenberg_marquardt_strategy.cc)
//   * SPARSE_NORMAL_CHOLESKY is an exact solve of (J'J + D'D) y = J'r: restated as a dense Cholesky.
// Pinned (tolerance level) by restated test/test_relax.cpp cases in tests/test_oracle_relax.py.
#pragma once

#include "jet.hpp"

#include <array>
#include <cmath>
#include <functional>
#include <memory>
#include <string>
#include <unordered_map>
#include <vector>

namespace oracle
{
namespace mc
{

struct CostFunction
{
    virtual ~CostFunction() = default;
    int num_residuals = 0;
    std::vector<int> block_sizes;
    // jacobians[i] may be null; otherwise row-major num_residuals x block_sizes[i]
    virtual bool Evaluate(double const *const *params, double *residuals, double **jacobians) const = 0;
};

// AutoDiffCostFunction<Functor, kNumResiduals, N0, N1, ...>: Functor::operator()(const T* p0, ..., T* residuals)
template <typename Functor, int kNumResiduals, int... Ns> struct AutoDiffCostFunction : CostFunction
{
    static constexpr int kNumBlocks = sizeof...(Ns);
    static constexpr int kTotal = (Ns + ...);
    std::unique_ptr<Functor> functor;
    explicit AutoDiffCostFunction(Functor *f) : functor(f)
    {
        num_residuals = kNumResiduals;
        block_sizes = {Ns...};
    }
    template <typename T, size_t... I> bool call(const T *const *p, T *r, std::index_sequence<I...>) const
    {
        return (*functor)(p[I]..., r);
    }
    bool Evaluate(double const *const *params, double *residuals, double **jacobians) const override
    {
        if (!jacobians)
            return call<double>(params, residuals, std::make_index_sequence<kNumBlocks>{});
        using J = Jet<kTotal>;
        std::vector<J> x(kTotal);
        const J *ptrs[kNumBlocks];
        int off = 0;
        for (int b = 0; b < kNumBlocks; b++)
        {
            ptrs[b] = x.data() + off;
            for (int k = 0; k < block_sizes[b]; k++)
                x[off + k] = J(params[b][k], off + k);
            off += block_sizes[b];
        }
        J out[kNumResiduals];
        if (!call<J>(ptrs, out, std::make_index_sequence<kNumBlocks>{}))
            return false;
        off = 0;
        for (int b = 0; b < kNumBlocks; b++)
        {
            if (jacobians[b])
                for (int r = 0; r < kNumResiduals; r++)
                    for (int k = 0; k < block_sizes[b]; k++)
                        jacobians[b][r * block_sizes[b] + k] = out[r].v[off + k];
            off += block_sizes[b];
        }
        for (int r = 0; r < kNumResiduals; r++)
            residuals[r] = out[r].a;
        return true;
    }
};

struct LossFunction
{
    virtual ~LossFunction() = default;
    virtual void Evaluate(double s, double rho[3]) const = 0;
};
struct HuberLoss : LossFunction // ceres/loss_function.cc
{
    double a, b;
    explicit HuberLoss(double a_) : a(a_), b(a_ * a_)
    {
    }
    void Evaluate(double s, double rho[3]) const override
    {
        if (s > b)
        {
            const double r = std::sqrt(s);
            rho[0] = 2.0 * a * r - b;
            rho[1] = std::max(std::numeric_limits<double>::min(), a / r);
            rho[2] = -rho[1] / (2.0 * s);
        }
        else
        {
            rho[0] = s;
            rho[1] = 1.0;
            rho[2] = 0.0;
        }
    }
};

enum class Manifold
{
    EUCLIDEAN,
    EIGEN_QUATERNION,
    SUBSET // ceres::SubsetManifold(size, constant_parameters): the listed coordinates are held constant
};

struct ParameterBlock
{
    double *data = nullptr;
    int size = 0;
    Manifold manifold = Manifold::EUCLIDEAN;
    bool constant = false;
    std::vector<int> subset_constant;       // Manifold::SUBSET: constant coordinates
    std::vector<double> lower, upper;       // empty = unbounded (SetParameterLowerBound / UpperBound)
    int tangent_size() const
    {
        if (manifold == Manifold::EIGEN_QUATERNION)
            return 3;
        if (manifold == Manifold::SUBSET)
            return size - (int)subset_constant.size();
        return size;
    }
    bool is_subset_constant(int k) const
    {
        for (int c : subset_constant)
            if (c == k)
                return true;
        return false;
    }
};

struct ResidualBlock
{
    std::unique_ptr<CostFunction> cost;
    const LossFunction *loss = nullptr; // not owned
    std::vector<int> blocks;            // indices into Problem::blocks
};

class Problem
{
  public:
    int AddParameterBlock(double *p, int size);
    void AddResidualBlock(CostFunction *cost, const LossFunction *loss, const std::vector<double *> &params);
    void SetManifold(double *p, Manifold m);
    void SetSubsetManifold(double *p, const std::vector<int> &constant_parameters);
    void SetParameterLowerBound(double *p, int index, double v);
    void SetParameterUpperBound(double *p, int index, double v);
    void SetParameterBlockConstant(double *p);
    void SetParameterBlockVariable(double *p);
    bool IsParameterBlockConstant(double *p) const;
    bool HasParameterBlock(double *p) const
    {
        return index.count(p) != 0;
    }
    int NumParameterBlocks() const
    {
        return (int)blocks.size();
    }
    int NumResidualBlocks() const
    {
        return (int)residuals.size();
    }
    std::vector<double *> GetParameterBlocks() const;

    std::vector<ParameterBlock> blocks; // insertion order (Ceres orders the program the same way)
    std::vector<ResidualBlock> residuals;
    std::unordered_map<double *, int> index;
};

struct SolverOptions
{
    int max_num_iterations = 100;
    double initial_trust_region_radius = 1.0;
    double max_trust_region_radius = 1e16;
    double min_trust_region_radius = 1e-32;
    double min_relative_decrease = 1e-3;
    double min_lm_diagonal = 1e-6;
    double max_lm_diagonal = 1e32;
    int max_num_consecutive_invalid_steps = 5;
    double function_tolerance = 1e-6;
    double gradient_tolerance = 1e-10;
    double parameter_tolerance = 1e-8;
    bool jacobi_scaling = true;
    int max_num_line_search_step_size_iterations = 20; // projected line search of bounds-constrained problems
};

struct IterationSummary
{
    int iteration = 0;
    bool step_is_valid = false, step_is_successful = false;
    double cost = 0, cost_change = 0, gradient_max_norm = 0, step_norm = 0, relative_decrease = 0,
           trust_region_radius = 0;
};

struct SolverSummary
{
    std::vector<IterationSummary> iterations;
    double initial_cost = 0, final_cost = 0, fixed_cost = 0;
    int num_successful_steps = 0, num_unsuccessful_steps = 0;
    int num_parameters_reduced = 0, num_residuals_reduced = 0;
    std::string message;
    bool usable = false;
};

void Solve(const SolverOptions &options, Problem *problem, SolverSummary *summary);

} // namespace mc
} // namespace oracle
