// levenberg_marquardt_sparse_hip.cpp -- see the header.  Compiled against the reference's headers.
#include "levenberg_marquardt_sparse_hip.h"

#include <corbo-core/console.h>
#include <corbo-optimization/hyper_graph/hyper_graph_optimization_problem_base.h>
#include <corbo-optimization/hyper_graph/vertex_interface.h>

#include <cmath>
#include <cstring>

namespace corbo {

LevenbergMarquardtSparseHip::LevenbergMarquardtSparseHip()
{
    corbo_hip_default_lm_opts(&_opts);
    std::memset(&_desc, 0, sizeof(_desc));
    std::memset(&_dims, 0, sizeof(_dims));
    std::memset(&_stats, 0, sizeof(_stats));
}

LevenbergMarquardtSparseHip::~LevenbergMarquardtSparseHip() { releaseHandle(); }

void LevenbergMarquardtSparseHip::releaseHandle()
{
    if (_handle) corbo_hip_destroy(_handle);
    _handle = nullptr;
}

void LevenbergMarquardtSparseHip::setPenaltyWeights(double weight_eq, double weight_ineq, double weight_bounds)
{
    _opts.weight_eq     = weight_eq;
    _opts.weight_ineq   = weight_ineq;
    _opts.weight_bounds = weight_bounds;
}

void LevenbergMarquardtSparseHip::setWeightAdapation(double factor_eq, double factor_ineq, double factor_bounds, double max_eq, double max_ineq,
                                                     double max_bounds)
{
    _opts.adapt_factor_eq     = factor_eq;
    _opts.adapt_factor_ineq   = factor_ineq;
    _opts.adapt_factor_bounds = factor_bounds;
    _opts.adapt_max_eq        = max_eq;
    _opts.adapt_max_ineq      = max_ineq;
    _opts.adapt_max_bounds    = max_bounds;
}

bool LevenbergMarquardtSparseHip::initialize(OptimizationProblemInterface* problem)
{
    // same check as LevenbergMarquardtSparse::initialize (levenberg_marquardt_sparse.cpp:33-42)
    if (problem && !problem->isLeastSquaresProblem())
    {
        PRINT_ERROR("LevenbergMarquardtSparseHip(): cannot handle non-least-squares objectives or LS objectives in non-LS form.");
        return false;
    }
    if (!_have_desc)
    {
        PRINT_ERROR("LevenbergMarquardtSparseHip(): setDeviceModel() must be called before initialize().");
        return false;
    }
    return true;
}

void LevenbergMarquardtSparseHip::clear()
{
    releaseHandle();
    std::memset(&_stats, 0, sizeof(_stats));
}

// Stacked residual [lsq | w_eq eq | w_ineq max(0, c) | w_b bound distance] of the graph's own edges (the reference's
// LevenbergMarquardtSparse::computeValues, levenberg_marquardt_sparse.cpp:222-246) against the device's at the uploaded vertex values.
bool LevenbergMarquardtSparseHip::modelMatchesGraph(OptimizationProblemInterface& problem)
{
    const double w_eq = _opts.weight_eq, w_ineq = _opts.weight_ineq, w_b = _opts.weight_bounds;
    Eigen::VectorXd host(_dims.m), dev(_dims.m);
    int idx = 0;
    if (_dims.lsq > 0) problem.computeValuesLsqObjective(host.segment(idx, _dims.lsq));
    idx += _dims.lsq;
    if (_dims.eq > 0)
    {
        problem.computeValuesEquality(host.segment(idx, _dims.eq));
        host.segment(idx, _dims.eq) *= w_eq;
    }
    idx += _dims.eq;
    if (_dims.ineq > 0) problem.computeValuesActiveInequality(host.segment(idx, _dims.ineq), w_ineq);
    idx += _dims.ineq;
    if (_dims.bounds > 0)
    {
        problem.computeDistanceFiniteCombinedBounds(host.segment(idx, _dims.bounds));
        host.segment(idx, _dims.bounds) *= w_b;
    }
    if (corbo_hip_eval(_handle, w_eq, w_ineq, w_b, dev.data(), nullptr) != CORBO_HIP_OK)
    {
        PRINT_ERROR("LevenbergMarquardtSparseHip(): " << corbo_hip_last_error());
        return false;
    }
    for (int i = 0; i < _dims.m; ++i)
    {
        const double tol = 1e-9 * (1.0 + std::abs(host[i]));
        if (!(std::abs(host[i] - dev[i]) <= tol))
        {
            const char* part = i < _dims.lsq ? "lsq objective" : i < _dims.lsq + _dims.eq ? "equality" : i < _dims.lsq + _dims.eq + _dims.ineq ? "inequality" : "bounds";
            PRINT_ERROR("LevenbergMarquardtSparseHip(): the device model does not describe this hypergraph: residual row "
                        << i << " (" << part << ") is " << host[i] << " on the graph's edges and " << dev[i]
                        << " on the device; refusing to solve (no CPU fallback).");
            return false;
        }
    }
    return true;
}

SolverStatus LevenbergMarquardtSparseHip::solve(OptimizationProblemInterface& problem, bool new_structure, bool new_run, double* obj_value)
{
    if (obj_value) *obj_value = -1;
    auto* hg = dynamic_cast<BaseHyperGraphOptimizationProblem*>(&problem);
    if (!hg || !hg->getGraph().hasVertexSet())
    {
        PRINT_ERROR("LevenbergMarquardtSparseHip(): the problem is not a hypergraph optimization problem.");
        return SolverStatus::Error;
    }
    // vertices in the grid's order: x_0..x_{N-2}, u_0..u_{N-2}, x_f, dt, (u_prev, u_ref, u_prev_dt)
    // (FullDiscretizationGridBase::getVertices, full_discretization_grid_base.cpp:499-512)
    std::vector<VertexInterface*> vtx;
    hg->getGraph().getVertexSetRaw()->getVertices(vtx);
    const int nx = _desc.nx, nu = _desc.nu, s = nx + nu;
    if ((int)vtx.size() < 4 || ((int)vtx.size() - 5) % 2 != 0)
    {
        PRINT_ERROR("LevenbergMarquardtSparseHip(): unexpected vertex set (not a full-discretization / shooting grid with 1 control per interval).");
        return SolverStatus::Error;
    }
    const int N = ((int)vtx.size() - 5) / 2 + 1;
    VertexInterface* xf_v = vtx[2 * (N - 1)];
    VertexInterface* dt_v = vtx[2 * (N - 1) + 1];
    // FullDiscretizationGridBase lists all states, then all controls; ShootingGridBase (shooting_grid_base.cpp:567-581) interleaves
    // them interval by interval: s_0, u_0, s_1, u_1, ...
    const bool interleaved = (_desc.grid == CORBO_HIP_GRID_MS);
    auto xv = [&](int k) { return interleaved ? vtx[2 * k] : vtx[k]; };
    auto uv = [&](int k) { return interleaved ? vtx[2 * k + 1] : vtx[N - 1 + k]; };

    bool verify_now = false;
    if (new_structure || !_handle || _desc.N != N)
    {
        verify_now = _verify;
        // describe the structure, then verify it against what the graph reports
        for (int k = 0; k < N - 1; ++k)
            if (xv(k)->getDimension() != nx || uv(k)->getDimension() != nu)
            {
                PRINT_ERROR("LevenbergMarquardtSparseHip(): vertex dimensions do not match the device model (nx, nu).");
                return SolverStatus::Error;
            }
        _desc.N             = N;
        _desc.xf_fixed_mask = 0;
        for (int i = 0; i < nx; ++i)
            if (xf_v->isFixedComponent(i)) _desc.xf_fixed_mask |= (1u << i);
        const bool dt_free = !dt_v->isFixedComponent(0);
        if (dt_free != (_desc.grid == CORBO_HIP_GRID_FD_VARIABLE))
        {
            PRINT_ERROR("LevenbergMarquardtSparseHip(): dt fixed/free does not match the device model's grid kind.");
            return SolverStatus::Error;
        }
        if (dt_free) { _desc.dt_lb = dt_v->getLowerBounds()[0]; _desc.dt_ub = dt_v->getUpperBounds()[0]; }
        // bound pattern: shared along the horizon in the reference (NlpFunctions::x_lb ...), read from x_1 / u_0 / x_f
        VertexInterface* xb = (N > 2) ? xv(1) : xf_v;
        for (int i = 0; i < nx; ++i) { _desc.x_lb[i] = xb->getLowerBounds()[i]; _desc.x_ub[i] = xb->getUpperBounds()[i]; }
        for (int i = 0; i < nu; ++i) { _desc.u_lb[i] = uv(0)->getLowerBounds()[i]; _desc.u_ub[i] = uv(0)->getUpperBounds()[i]; }
        _desc.dt_ref = dt_v->getData()[0];
        if (corbo_hip_get_dims(&_desc, &_dims) != CORBO_HIP_OK)
        {
            PRINT_ERROR("LevenbergMarquardtSparseHip(): " << corbo_hip_last_error());
            return SolverStatus::Error;
        }
        if (_dims.n != problem.getParameterDimension() || _dims.lsq != problem.getLsqObjectiveDimension() ||
            _dims.eq != problem.getEqualityDimension() || _dims.ineq != problem.getInequalityDimension() ||
            _dims.bounds != problem.finiteCombinedBoundsDimension())
        {
            PRINT_ERROR("LevenbergMarquardtSparseHip(): hypergraph dimensions (n=" << problem.getParameterDimension() << ", lsq="
                                                                                  << problem.getLsqObjectiveDimension() << ", eq=" << problem.getEqualityDimension()
                                                                                  << ") do not match the device model; refusing to solve (no CPU fallback).");
            return SolverStatus::Error;
        }
        releaseHandle();
        if (corbo_hip_create(&_desc, 1, _device, &_handle) != CORBO_HIP_OK)
        {
            PRINT_ERROR("LevenbergMarquardtSparseHip(): " << corbo_hip_last_error());
            _handle = nullptr;
            return SolverStatus::Error;
        }
        _x.assign(_dims.nv, 0.0);
        _lb.assign(_dims.nv, 0.0);
        _ub.assign(_dims.nv, 0.0);
    }
    const bool dt_free = (_desc.grid == CORBO_HIP_GRID_FD_VARIABLE);

    // ---- gather vertex values and bounds into the C-ABI's vertex layout
    auto pack = [&](VertexInterface* v, int off, int dim) {
        std::memcpy(&_x[off], v->getData(), dim * sizeof(double));
        std::memcpy(&_lb[off], v->getLowerBounds(), dim * sizeof(double));
        std::memcpy(&_ub[off], v->getUpperBounds(), dim * sizeof(double));
    };
    for (int k = 0; k < N - 1; ++k) { pack(xv(k), k * s, nx); pack(uv(k), k * s + nx, nu); }
    pack(xf_v, (N - 1) * s, nx);
    if (dt_free) pack(dt_v, (N - 1) * s + nx, 1);

    std::vector<double> xref(nx, 0.0);
    if (_xref.size() == nx)
        for (int i = 0; i < nx; ++i) xref[i] = _xref[i];

    if (corbo_hip_set_instance_data(_handle, _x.data(), _lb.data(), _ub.data(), xref.data()) != CORBO_HIP_OK)
    {
        PRINT_ERROR("LevenbergMarquardtSparseHip(): " << corbo_hip_last_error());
        return SolverStatus::Error;
    }
    if (verify_now && !modelMatchesGraph(problem))
    {
        releaseHandle();   // the next call re-checks
        return SolverStatus::Error;
    }
    if (corbo_hip_solve(_handle, &_opts, new_run ? 1 : 0) != CORBO_HIP_OK)
    {
        PRINT_ERROR("LevenbergMarquardtSparseHip(): " << corbo_hip_last_error());
        return SolverStatus::Error;
    }
    double chi2    = -1;
    int32_t status = CORBO_HIP_SOLVER_ERROR;
    if (corbo_hip_get_solution(_handle, _x.data(), &chi2, &status) != CORBO_HIP_OK)
    {
        PRINT_ERROR("LevenbergMarquardtSparseHip(): " << corbo_hip_last_error());
        return SolverStatus::Error;
    }
    corbo_hip_get_stats(_handle, &_stats);

    // ---- scatter the last accepted iterate back into the vertices (what callers read after solve(),
    //      full_discretization_grid_base.cpp:324-331,529-565); fixed components are left untouched; backup stacks stay empty
    auto unpack = [&](VertexInterface* v, int off, int dim) {
        for (int i = 0; i < dim; ++i)
            if (!v->isFixedComponent(i)) v->setData(i, _x[off + i]);
    };
    for (int k = 0; k < N - 1; ++k) { unpack(xv(k), k * s, nx); unpack(uv(k), k * s + nx, nu); }
    unpack(xf_v, (N - 1) * s, nx);
    if (dt_free) unpack(dt_v, (N - 1) * s + nx, 1);

    if (obj_value) *obj_value = chi2;
    switch (status)
    {
        case CORBO_HIP_SOLVER_CONVERGED: return SolverStatus::Converged;
        case CORBO_HIP_SOLVER_EARLY_TERMINATED: return SolverStatus::EarlyTerminated;
        case CORBO_HIP_SOLVER_INFEASIBLE: return SolverStatus::Infeasible;
        default: return SolverStatus::Error;
    }
}

}  // namespace corbo
