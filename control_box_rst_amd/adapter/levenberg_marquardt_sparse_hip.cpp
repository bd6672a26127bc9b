// levenberg_marquardt_sparse_hip.cpp -- see the header.  Compiled against the reference's headers.
#include "levenberg_marquardt_sparse_hip.h"

#include "graph_recogniser.h"

#include <corbo-core/console.h>
#include <corbo-optimization/hyper_graph/hyper_graph_optimization_problem_base.h>
#include <corbo-optimization/hyper_graph/vertex_interface.h>

#include <Eigen/Sparse>
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
    return true;
}

void LevenbergMarquardtSparseHip::clear()
{
    _hess_run_tracked = false;
    releaseHandle();
    std::memset(&_stats, 0, sizeof(_stats));
}

// Stacked residual [lsq | w_eq eq | w_ineq max(0, c) | w_b bound distance] and combined sparse Jacobian of the graph's own edges (the
// reference's LevenbergMarquardtSparse::computeValues, levenberg_marquardt_sparse.cpp:222-246, and
// computeCombinedSparseJacobian, hyper_graph_optimization_problem_edge_based.cpp:1480-1753) against the device's at the uploaded
// vertex values.
bool LevenbergMarquardtSparseHip::modelMatchesGraph(OptimizationProblemInterface& problem, bool perturbed)
{
    const char* where0 = perturbed ? "at the perturbed probe point" : "at the current vertex values";
    if (_desc.cost_nonlsq)
    {   // a stated model for the exact-Hessian path (plain / integral objective edges: no least-squares residual to compare, and
        // corbo_hip_eval refuses the handle like LevenbergMarquardtSparse refuses the problem): objective value and gradient, the
        // constraint values through the linear form's bounds (lbA = -c_eq, ubA = -c_ineq)
        const int n = _dims.n, rows = _dims.eq + _dims.ineq + _dims.bounds;
        Eigen::VectorXd gd(n), gh(n), ceq(_dims.eq), cineq(_dims.ineq), lbA(rows), ubA(rows);
        double od = 0;
        int32_t lnnz = 0, lrows = 0;
        if (corbo_hip_eval_objective_gradient(_handle, gd.data(), &od) != CORBO_HIP_OK || corbo_hip_linear_form_structure(&_desc, &lnnz, &lrows, nullptr, nullptr) != CORBO_HIP_OK)
        {
            PRINT_ERROR("LevenbergMarquardtSparseHip(): " << corbo_hip_last_error());
            return false;
        }
        Eigen::VectorXd lvals(lnnz);
        if (lrows != rows || corbo_hip_eval_linear_form(_handle, lvals.data(), lbA.data(), ubA.data()) != CORBO_HIP_OK)
        {
            PRINT_ERROR("LevenbergMarquardtSparseHip(): linear form of the device model: " << corbo_hip_last_error());
            return false;
        }
        const double oh = problem.computeValueObjective();
        problem.computeGradientObjective(gh);
        if (_dims.eq > 0) problem.computeValuesEquality(ceq);
        if (_dims.ineq > 0) problem.computeValuesInequality(cineq);
        bool ok = std::abs(oh - od) <= 1e-9 * (1.0 + std::abs(oh));
        const double gmax = std::max(1.0, gh.cwiseAbs().maxCoeff());
        for (int i = 0; i < n && ok; ++i) ok = std::abs(gh[i] - gd[i]) <= 2e-6 * gmax;
        for (int i = 0; i < _dims.eq && ok; ++i) ok = std::abs(ceq[i] + lbA[i]) <= 1e-9 * (1.0 + std::abs(ceq[i]));
        for (int i = 0; i < _dims.ineq && ok; ++i) ok = std::abs(cineq[i] + ubA[_dims.eq + i]) <= 1e-9 * (1.0 + std::abs(cineq[i]));
        if (!ok)
            PRINT_ERROR("LevenbergMarquardtSparseHip(): the stated device model does not describe this hypergraph (objective value " << oh << " on the graph's edges, " << od
                                                                                                                                    << " on the device, or its gradient / constraint values differ) "
                                                                                                                                    << where0 << "; refusing (no CPU fallback).");
        return ok;
    }
    const double w_eq = _w_eq, w_ineq = _w_ineq, w_b = _w_b;
    Eigen::VectorXd host(_dims.m), dev(_dims.m), devj(_dims.nnz);
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
    if (corbo_hip_eval(_handle, w_eq, w_ineq, w_b, dev.data(), devj.data()) != CORBO_HIP_OK)
    {
        PRINT_ERROR("LevenbergMarquardtSparseHip(): " << corbo_hip_last_error());
        return false;
    }
    const char* where = perturbed ? "at the perturbed probe point" : "at the current vertex values";
    for (int i = 0; i < _dims.m; ++i)
    {
        const double tol = 1e-9 * (1.0 + std::abs(host[i]));
        if (!(std::abs(host[i] - dev[i]) <= tol))
        {
            const char* part = i < _dims.lsq ? "lsq objective" : i < _dims.lsq + _dims.eq ? "equality" : i < _dims.lsq + _dims.eq + _dims.ineq ? "inequality" : "bounds";
            PRINT_ERROR("LevenbergMarquardtSparseHip(): the device model does not describe this hypergraph: residual row "
                        << i << " (" << part << ") is " << host[i] << " on the graph's edges and " << dev[i] << " on the device " << where
                        << "; refusing to solve (no CPU fallback).");
            return false;
        }
    }
    // combined sparse Jacobian: same (row, column) pattern, values to the finite-difference noise level (delta = 1e-9: ~1e-7 relative)
    std::vector<int32_t> rows(_dims.nnz), cols(_dims.nnz);
    if (corbo_hip_get_structure(&_desc, rows.data(), cols.data()) != CORBO_HIP_OK)
    {
        PRINT_ERROR("LevenbergMarquardtSparseHip(): " << corbo_hip_last_error());
        return false;
    }
    Eigen::SparseMatrix<double> Jh(_dims.m, _dims.n);
    Jh.reserve(_dims.nnz);
    problem.computeCombinedSparseJacobian(Jh, true, true, true, true, true, w_eq, w_ineq, w_b, &host);
    double jmax = 1.0;
    for (int k = 0; k < Jh.outerSize(); ++k)
        for (Eigen::SparseMatrix<double>::InnerIterator it(Jh, k); it; ++it) jmax = std::max(jmax, std::abs(it.value()));
    if ((int)Jh.nonZeros() != _dims.nnz)
    {
        PRINT_ERROR("LevenbergMarquardtSparseHip(): the graph's combined Jacobian has " << Jh.nonZeros() << " structural non-zeros, the device model " << _dims.nnz
                                                                                       << "; refusing to solve.");
        return false;
    }
    for (int k = 0; k < _dims.nnz; ++k)
    {
        const double jh = Jh.coeff(rows[k], cols[k]);
        if (!(std::abs(jh - devj[k]) <= 2e-6 * jmax))
        {
            PRINT_ERROR("LevenbergMarquardtSparseHip(): the device model does not describe this hypergraph: Jacobian entry (" << rows[k] << ", " << cols[k] << ") is "
                        << jh << " on the graph's edges and " << devj[k] << " on the device " << where << "; refusing to solve (no CPU fallback).");
            return false;
        }
    }
    return true;
}

bool LevenbergMarquardtSparseHip::uploadVertices(const std::vector<VertexInterface*>& xs, const std::vector<VertexInterface*>& us, VertexInterface* xf,
                                                 VertexInterface* dt)
{
    const int nx = _desc.nx, nu = _desc.nu, s = nx + nu, N = _desc.N;
    auto pack = [&](VertexInterface* v, int off, int dim) {
        std::memcpy(&_x[off], v->getData(), dim * sizeof(double));
        std::memcpy(&_lb[off], v->getLowerBounds(), dim * sizeof(double));
        std::memcpy(&_ub[off], v->getUpperBounds(), dim * sizeof(double));
    };
    for (int k = 0; k < N - 1; ++k) { pack(xs[k], k * s, nx); pack(us[k], k * s + nx, nu); }
    pack(xf, (N - 1) * s, nx);
    if ((_desc.grid == CORBO_HIP_GRID_FD_VARIABLE || _desc.grid == CORBO_HIP_GRID_MS_VARIABLE)) pack(dt, (N - 1) * s + nx, 1);
    std::vector<double> xref(nx, 0.0);
    if (_xref.size() == nx)
        for (int i = 0; i < nx; ++i) xref[i] = _xref[i];
    if (corbo_hip_set_instance_data(_handle, _x.data(), _lb.data(), _ub.data(), xref.data()) != CORBO_HIP_OK)
    {
        PRINT_ERROR("LevenbergMarquardtSparseHip(): " << corbo_hip_last_error());
        return false;
    }
    // time-varying state reference: one reference per vertex component (corbo_hip_set_references), else the static one above
    const double* refs = nullptr;
    if (_xref_traj.rows() == N && _xref_traj.cols() == nx)
    {
        _ref.assign(_x.size(), 0.0);
        for (int k = 0; k < N; ++k)
            for (int i = 0; i < nx; ++i) _ref[k * s + i] = _xref_traj(k, i);
        refs = _ref.data();
    }
    if (corbo_hip_set_references(_handle, refs) != CORBO_HIP_OK)
    {
        PRINT_ERROR("LevenbergMarquardtSparseHip(): " << corbo_hip_last_error());
        return false;
    }
    if (_desc.ctrl_dev && _uprev_v && _uprev_dt_v)
    {   // what setPreviousControlInput left in the grid's fixed vertices (structured_optimal_control_problem.cpp:119, full_discretization_grid_base.cpp:66-70)
        const double dtp = _uprev_dt_v->getData()[0];
        if (_uprev_v->getDimension() != nu || corbo_hip_set_previous_control(_handle, _uprev_v->getData(), &dtp) != CORBO_HIP_OK)
        {
            PRINT_ERROR("LevenbergMarquardtSparseHip(): previous control: " << corbo_hip_last_error());
            return false;
        }
    }
    return true;
}

// Everything between "here is a hypergraph" and "its vertex values are on the device": recognise (new structure), describe, verify, upload.
bool LevenbergMarquardtSparseHip::attach(OptimizationProblemInterface& problem, bool new_structure, bool new_run)
{
    auto* hg = dynamic_cast<BaseHyperGraphOptimizationProblem*>(&problem);
    if (!hg || !hg->getGraph().hasVertexSet())
    {
        PRINT_ERROR("LevenbergMarquardtSparseHip(): the problem is not a hypergraph optimization problem.");
        return false;
    }
    // vertices in the grid's order: x_0..x_{N-2}, u_0..u_{N-2}, x_f, dt, (u_prev, u_ref, u_prev_dt)
    // (FullDiscretizationGridBase::getVertices, full_discretization_grid_base.cpp:499-512)
    std::vector<VertexInterface*> vtx;
    hg->getGraph().getVertexSetRaw()->getVertices(vtx);
    if ((int)vtx.size() < 7 || ((int)vtx.size() - 5) % 2 != 0)
    {
        PRINT_ERROR("LevenbergMarquardtSparseHip(): unexpected vertex set (not a full-discretization / shooting grid with 1 control per interval).");
        return false;
    }
    const int N = ((int)vtx.size() - 5) / 2 + 1;
    VertexInterface* xf_v = vtx[2 * (N - 1)];
    VertexInterface* dt_v = vtx[2 * (N - 1) + 1];
    _uprev_v = vtx[2 * (N - 1) + 2]; _uprev_dt_v = vtx[2 * (N - 1) + 4];   // the grid's fixed vertices _u_prev / _u_prev_dt (control-deviation edge of interval 0)

    // a fixed dt that changed without a structure change (setDtRef between runs) is a new problem for the device
    const bool dt_changed = _handle && !(_desc.grid == CORBO_HIP_GRID_FD_VARIABLE || _desc.grid == CORBO_HIP_GRID_MS_VARIABLE) && dt_v->getData()[0] != _desc.dt_ref;
    bool verify_now = false;
    // Same structure, new run: the reference's cost / constraint objects may have been given another reference, other weights or other
    // parameters WITHOUT a structure change (QuadraticFormCost::update returns false, setpoints move between runs).  The model
    // is derived again from the graph and compared with the resident one: only-the-references-moved keeps the handle, anything else
    // rebuilds it like a new structure.  (_tracking = false: the caller vouches that nothing but the vertex values changes, like setDeviceModel.)
    bool model_changed = false;
    if (!(new_structure || !_handle || _desc.N != N || dt_changed) && _recognised && new_run && _tracking)
    {
        HipRecognisedModel m, resident;
        resident.desc = _desc; resident.xref = _xref; resident.xref_traj = _xref_traj;   // checked term by term first (graph_recogniser.h)
        std::string why;
        if (!recogniseHyperGraphForHip(*hg, &m, &why, &resident))
        {
            PRINT_ERROR("LevenbergMarquardtSparseHip(): this hypergraph has no device description any more: " << why << "; refusing to solve (no CPU fallback).");
            return false;
        }
        corbo_hip_problem_desc cmp = m.desc;   // the fields the adapter reads from the vertices are not the recogniser's
        cmp.N = _desc.N; cmp.xf_fixed_mask = _desc.xf_fixed_mask; cmp.dt_ref = _desc.dt_ref; cmp.dt_lb = _desc.dt_lb; cmp.dt_ub = _desc.dt_ub;
        std::memcpy(cmp.x_lb, _desc.x_lb, sizeof(cmp.x_lb)); std::memcpy(cmp.x_ub, _desc.x_ub, sizeof(cmp.x_ub));
        std::memcpy(cmp.u_lb, _desc.u_lb, sizeof(cmp.u_lb)); std::memcpy(cmp.u_ub, _desc.u_ub, sizeof(cmp.u_ub));
        model_changed = std::memcmp(&cmp, &_desc, sizeof(cmp)) != 0;
        if (model_changed) _desc = m.desc;
        _xref      = m.xref;
        _xref_traj = m.xref_traj;
    }
    if (new_structure || !_handle || _desc.N != N || dt_changed)
    {
        verify_now = _verify;
        if (!_have_desc || _recognised)
        {   // derive the device model from the graph
            HipRecognisedModel m;
            std::string why;
            if (!recogniseHyperGraphForHip(*hg, &m, &why))
            {
                PRINT_ERROR("LevenbergMarquardtSparseHip(): this hypergraph has no device description: " << why << "; refusing to solve (no CPU fallback).");
                return false;
            }
            _desc       = m.desc;
            _xref       = m.xref;
            _xref_traj  = m.xref_traj;
            _have_desc  = true;
            _recognised = true;
        }
    }
    const int nx = _desc.nx, nu = _desc.nu, s = nx + nu;
    // FullDiscretizationGridBase lists all states, then all controls; ShootingGridBase (shooting_grid_base.cpp:567-581) interleaves
    // them interval by interval: s_0, u_0, s_1, u_1, ...
    const bool interleaved = (_desc.grid == CORBO_HIP_GRID_MS || _desc.grid == CORBO_HIP_GRID_MS_VARIABLE);
    std::vector<VertexInterface*>& xs = _xs; std::vector<VertexInterface*>& us = _us;
    xs.assign(N - 1, nullptr); us.assign(N - 1, nullptr);
    for (int k = 0; k < N - 1; ++k) { xs[k] = interleaved ? vtx[2 * k] : vtx[k]; us[k] = interleaved ? vtx[2 * k + 1] : vtx[N - 1 + k]; }

    if (model_changed) verify_now = _verify;
    if (new_structure || !_handle || _desc.N != N || dt_changed || model_changed)
    {
        // describe the structure, then verify it against what the graph reports
        for (int k = 0; k < N - 1; ++k)
            if (xs[k]->getDimension() != nx || us[k]->getDimension() != nu)
            {
                PRINT_ERROR("LevenbergMarquardtSparseHip(): vertex dimensions do not match the device model (nx, nu).");
                return false;
            }
        _desc.N             = N;
        _desc.xf_fixed_mask = 0;
        for (int i = 0; i < nx; ++i)
            if (xf_v->isFixedComponent(i)) _desc.xf_fixed_mask |= (1u << i);
        const bool dt_free = !dt_v->isFixedComponent(0);
        if (dt_free != ((_desc.grid == CORBO_HIP_GRID_FD_VARIABLE || _desc.grid == CORBO_HIP_GRID_MS_VARIABLE)))
        {
            PRINT_ERROR("LevenbergMarquardtSparseHip(): dt fixed/free does not match the device model's grid kind.");
            return false;
        }
        if (dt_free) { _desc.dt_lb = dt_v->getLowerBounds()[0]; _desc.dt_ub = dt_v->getUpperBounds()[0]; }
        // bound pattern: shared along the horizon in the reference (NlpFunctions::x_lb ...), read from x_1 / u_0 / x_f
        VertexInterface* xb = (N > 2) ? xs[1] : xf_v;
        for (int i = 0; i < nx; ++i) { _desc.x_lb[i] = xb->getLowerBounds()[i]; _desc.x_ub[i] = xb->getUpperBounds()[i]; }
        for (int i = 0; i < nu; ++i) { _desc.u_lb[i] = us[0]->getLowerBounds()[i]; _desc.u_ub[i] = us[0]->getUpperBounds()[i]; }
        _desc.dt_ref = dt_v->getData()[0];
        if (corbo_hip_get_dims(&_desc, &_dims) != CORBO_HIP_OK)
        {
            PRINT_ERROR("LevenbergMarquardtSparseHip(): " << corbo_hip_last_error());
            return false;
        }
        if (_dims.n != problem.getParameterDimension() || _dims.lsq != problem.getLsqObjectiveDimension() ||
            _dims.eq != problem.getEqualityDimension() || _dims.ineq != problem.getInequalityDimension() ||
            _dims.bounds != problem.finiteCombinedBoundsDimension())
        {
            PRINT_ERROR("LevenbergMarquardtSparseHip(): hypergraph dimensions (n=" << problem.getParameterDimension() << ", lsq="
                                                                                  << problem.getLsqObjectiveDimension() << ", eq=" << problem.getEqualityDimension()
                                                                                  << ") do not match the device model; refusing to solve (no CPU fallback).");
            return false;
        }
        releaseHandle();
        if (corbo_hip_create(&_desc, 1, _device, &_handle) != CORBO_HIP_OK)
        {
            PRINT_ERROR("LevenbergMarquardtSparseHip(): " << corbo_hip_last_error());
            _handle = nullptr;
            return false;
        }
        corbo_hip_set_result_sink(_handle, 1);
        corbo_hip_set_option(_handle, "solve_timing", 0);   // no HIP timing events around the one launch of a solve: 10 - 13 us of a batch-1 solve   // one OCP per solve() whose result goes back into the vertices: let the solve kernel deliver it
        _x.assign(_dims.nv, 0.0);
        _lb.assign(_dims.nv, 0.0);
        _ub.assign(_dims.nv, 0.0);
    }
    const bool dt_free = ((_desc.grid == CORBO_HIP_GRID_FD_VARIABLE || _desc.grid == CORBO_HIP_GRID_MS_VARIABLE));

    if (verify_now)
    {
        // (1) at the current vertex values, (2) at a deterministically perturbed point: every unfixed component moved by a fixed
        // non-zero pattern through the problem's own increment operator, then restored
        if (!uploadVertices(xs, us, xf_v, dt_v) || !modelMatchesGraph(problem, false))
        {
            releaseHandle();   // the next call re-checks
            return false;
        }
        Eigen::VectorXd inc(_dims.n);
        for (int i = 0; i < _dims.n; ++i) inc[i] = ((i & 1) ? -1.0 : 1.0) * 0.01 * (1.0 + (i % 7) / 7.0);
        problem.backupParameters();
        problem.applyIncrement(inc);
        const bool ok = uploadVertices(xs, us, xf_v, dt_v) && modelMatchesGraph(problem, true);
        problem.restoreBackupParameters(false);
        if (!ok)
        {
            releaseHandle();
            return false;
        }
    }
    // ---- gather vertex values and bounds into the C-ABI's vertex layout
    if (!uploadVertices(xs, us, xf_v, dt_v)) return false;
    _xf_v = xf_v; _dt_v = dt_v;
    return true;

}

SolverStatus LevenbergMarquardtSparseHip::solve(OptimizationProblemInterface& problem, bool new_structure, bool new_run, double* obj_value)
{
    if (obj_value) *obj_value = -1;
    _hess_run_tracked = false;   // (a Hessian-path call after this solve tracks the model again)
    // penalty weights: resetWeights / adaptWeights (levenberg_marquardt_sparse.cpp:83-86, 270-287).  Kept here, not in the device
    // handle: the reference's _weight_* survive a structure change, the handle does not.
    if (new_run) { _w_eq = _opts.weight_eq; _w_ineq = _opts.weight_ineq; _w_b = _opts.weight_bounds; }
    else
    {
        _w_eq *= _opts.adapt_factor_eq;       if (_w_eq > _opts.adapt_max_eq) _w_eq = _opts.adapt_max_eq;
        _w_ineq *= _opts.adapt_factor_ineq;   if (_w_ineq > _opts.adapt_max_ineq) _w_ineq = _opts.adapt_max_ineq;
        _w_b *= _opts.adapt_factor_bounds;    if (_w_b > _opts.adapt_max_bounds) _w_b = _opts.adapt_max_bounds;
    }
    if (!attach(problem, new_structure, new_run)) return SolverStatus::Error;

    corbo_hip_lm_opts o = _opts;   // the weights of THIS solve, stated explicitly (new_run = 1 makes the library take them as they are)
    o.weight_eq = _w_eq; o.weight_ineq = _w_ineq; o.weight_bounds = _w_b;
    if (corbo_hip_solve(_handle, &o, 1) != CORBO_HIP_OK)
    {
        PRINT_ERROR("LevenbergMarquardtSparseHip(): " << corbo_hip_last_error());
        return SolverStatus::Error;
    }
    double chi2    = -1;
    int32_t status = CORBO_HIP_SOLVER_ERROR;
    // the solve kernel has written the accepted iterate and the LM state into the handle's pinned host memory itself (result sink, enabled at
    // create): views, no copy kernel and no second synchronisation on the per-solve path (families without a run-to-completion kernel: the
    // library copies behind the solve)
    const double* xp    = nullptr;
    const double* chi2p = nullptr;
    const int32_t* stp  = nullptr;
    int32_t stride      = 0;
    if (corbo_hip_fetch_solution(_handle, &xp, &stride, &chi2p, &stp) != CORBO_HIP_OK || !xp || !chi2p || !stp)
    {
        PRINT_ERROR("LevenbergMarquardtSparseHip(): " << corbo_hip_last_error());
        return SolverStatus::Error;
    }
    std::memcpy(_x.data(), xp, sizeof(double) * (size_t)_dims.nv);
    chi2   = chi2p[0];
    status = stp[0];
    corbo_hip_get_stats(_handle, &_stats);

    // ---- scatter the last accepted iterate back into the vertices (what callers read after solve(),
    //      full_discretization_grid_base.cpp:324-331,529-565); fixed components are left untouched; backup stacks stay empty
    auto unpack = [&](VertexInterface* v, int off, int dim) {
        for (int i = 0; i < dim; ++i)
            if (!v->isFixedComponent(i)) v->setData(i, _x[off + i]);
    };
    const int nx = _desc.nx, nu = _desc.nu, s = nx + nu, N = _desc.N;
    for (int k = 0; k < N - 1; ++k) { unpack(_xs[k], k * s, nx); unpack(_us[k], k * s + nx, nu); }
    unpack(_xf_v, (N - 1) * s, nx);
    if ((_desc.grid == CORBO_HIP_GRID_FD_VARIABLE || _desc.grid == CORBO_HIP_GRID_MS_VARIABLE)) unpack(_dt_v, (N - 1) * s + nx, 1);

    if (obj_value) *obj_value = chi2;
    switch (status)
    {
        case CORBO_HIP_SOLVER_CONVERGED: return SolverStatus::Converged;
        case CORBO_HIP_SOLVER_EARLY_TERMINATED: return SolverStatus::EarlyTerminated;
        case CORBO_HIP_SOLVER_INFEASIBLE: return SolverStatus::Infeasible;
        default: return SolverStatus::Error;
    }
}


// ---- operators of the exact-Hessian path (what an interior-point / SQP solver asks the problem for), evaluated on the device for the
//      hypergraph's current vertex values; signatures of OptimizationProblemInterface::computeSparseHessians{NNZ,Structure,Values}
//      (optimization_problem_interface.h) with the problem as the first argument
// attach for the Hessian-path entry points: the vertex values are uploaded on every call; the model tracking (one recogniser pass over all cost edges)
// runs on every call too unless setHessianTrackOncePerRun(true) limits it to the first call of an outer run (newHessianRun(), solve(), clear() start the next one)
bool LevenbergMarquardtSparseHip::attachHessianPath(OptimizationProblemInterface& problem)
{
    const bool first = !_hess_track_once || !_hess_run_tracked || _handle == nullptr;
    if (!attach(problem, _handle == nullptr, first)) return false;
    _hess_run_tracked = true;
    return true;
}

bool LevenbergMarquardtSparseHip::computeSparseHessiansNNZ(OptimizationProblemInterface& problem, int& nnz_obj, int& nnz_eq, int& nnz_ineq, bool lower_part_only)
{
    if (!attachHessianPath(problem)) return false;
    int32_t nnz[3];
    if (corbo_hip_hessian_nnz(&_desc, lower_part_only ? 1 : 0, nnz) != CORBO_HIP_OK) { PRINT_ERROR("LevenbergMarquardtSparseHip(): " << corbo_hip_last_error()); return false; }
    nnz_obj = nnz[0]; nnz_eq = nnz[1]; nnz_ineq = nnz[2];
    return true;
}

bool LevenbergMarquardtSparseHip::computeSparseHessiansStructure(OptimizationProblemInterface& problem, Eigen::Ref<Eigen::VectorXi> i_row_obj,
                                                                 Eigen::Ref<Eigen::VectorXi> j_col_obj, Eigen::Ref<Eigen::VectorXi> i_row_eq,
                                                                 Eigen::Ref<Eigen::VectorXi> j_col_eq, Eigen::Ref<Eigen::VectorXi> i_row_ineq,
                                                                 Eigen::Ref<Eigen::VectorXi> j_col_ineq, bool lower_part_only)
{
    if (!attachHessianPath(problem)) return false;
    static_assert(sizeof(int) == sizeof(int32_t), "Eigen::VectorXi is handed to the C-ABI as int32_t");
    if (corbo_hip_hessian_structure(&_desc, lower_part_only ? 1 : 0, i_row_obj.data(), j_col_obj.data(), i_row_eq.data(), j_col_eq.data(), i_row_ineq.data(),
                                    j_col_ineq.data()) != CORBO_HIP_OK)
    {
        PRINT_ERROR("LevenbergMarquardtSparseHip(): " << corbo_hip_last_error());
        return false;
    }
    return true;
}

bool LevenbergMarquardtSparseHip::computeGradientObjective(OptimizationProblemInterface& problem, Eigen::Ref<Eigen::VectorXd> gradient, double* obj_value)
{
    if (!attachHessianPath(problem)) return false;
    if (gradient.size() != _dims.n) { PRINT_ERROR("LevenbergMarquardtSparseHip(): gradient vector of the wrong size."); return false; }
    if (corbo_hip_eval_objective_gradient(_handle, gradient.data(), obj_value) != CORBO_HIP_OK)
    {
        PRINT_ERROR("LevenbergMarquardtSparseHip(): " << corbo_hip_last_error());
        return false;
    }
    return true;
}

bool LevenbergMarquardtSparseHip::computeSparseHessiansValues(OptimizationProblemInterface& problem, Eigen::Ref<Eigen::VectorXd> values_obj,
                                                              Eigen::Ref<Eigen::VectorXd> values_eq, Eigen::Ref<Eigen::VectorXd> values_ineq, double multiplier_obj,
                                                              const double* multipliers_eq, const double* multipliers_ineq, bool lower_part_only)
{
    if (!attachHessianPath(problem)) return false;   // uploads the current vertex values
    if (corbo_hip_eval_hessians(_handle, lower_part_only ? 1 : 0, multiplier_obj, multipliers_eq, multipliers_ineq, values_obj.data(), values_eq.data(),
                                values_ineq.data()) != CORBO_HIP_OK)
    {
        PRINT_ERROR("LevenbergMarquardtSparseHip(): " << corbo_hip_last_error());
        return false;
    }
    return true;
}

}  // namespace corbo
