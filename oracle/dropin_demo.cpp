// oracle/dropin_demo.cpp -- TEST INFRASTRUCTURE ONLY.
//
// Drop-in proof: the GENUINE reference's StructuredOptimalControlProblem (compiled from /root/reference into
// oracle/_ref/libcorbo_ref.a) is run twice on the same OCP, once with its own corbo::LevenbergMarquardtSparse and once with
// corbo::LevenbergMarquardtSparseHip (control_box_rst_amd/adapter/, over the C-ABI in libcorbo_hip.so) injected through the
// unchanged NlpSolverInterface plug-in point.  Prints one JSON line per scenario with the max trajectory difference.
// Built here (needs the reference's headers), runs on the GPU box (tests/test_gpu_dropin.py).
#include <corbo-core/reference_trajectory.h>
#include <corbo-core/time.h>
#include <corbo-optimal-control/functions/final_state_cost.h>
#include <corbo-optimal-control/functions/minimum_time.h>
#include <corbo-optimal-control/functions/quadratic_cost.h>
#include <corbo-optimal-control/structured_ocp/discretization_grids/finite_differences_grid.h>
#include <corbo-optimal-control/structured_ocp/discretization_grids/finite_differences_variable_grid.h>
#include <corbo-optimal-control/structured_ocp/structured_optimal_control_problem.h>
#include <corbo-optimization/hyper_graph/hyper_graph_optimization_problem_edge_based.h>
#include <corbo-optimization/solver/levenberg_marquardt_sparse.h>
#include <corbo-systems/benchmark/linear_benchmark_systems.h>
#include <corbo-systems/benchmark/nonlinear_benchmark_systems.h>

#include <cmath>
#include <cstdio>
#include <cstring>
#include <memory>
#include <string>

#include "../control_box_rst_amd/adapter/levenberg_marquardt_sparse_hip.h"

using namespace corbo;

class UnicycleRef : public SystemDynamicsInterface
{
 public:
    Ptr getInstance() const override { return std::make_shared<UnicycleRef>(); }
    bool isContinuousTime() const override { return true; }
    bool isLinear() const override { return false; }
    int getInputDimension() const override { return 2; }
    int getStateDimension() const override { return 3; }
    void dynamics(const Eigen::Ref<const StateVector>& x, const Eigen::Ref<const ControlVector>& u, Eigen::Ref<StateVector> f) const override
    {
        f[0] = u[0] * std::cos(x[2]);
        f[1] = u[0] * std::sin(x[2]);
        f[2] = u[1];
    }
};

struct Run
{
    Eigen::VectorXd traj;
    double chi2 = 0;
    bool ok     = false;
};

static Eigen::VectorXd trajectory(StructuredOptimalControlProblem& ocp, FiniteDifferencesGrid& grid)
{
    auto xs = std::make_shared<TimeSeries>();
    auto us = std::make_shared<TimeSeries>();
    ocp.getTimeSeries(xs, us);
    Eigen::MatrixXd X = xs->getValuesMatrixView(), U = us->getValuesMatrixView();
    Eigen::VectorXd out(X.size() + U.size() + 1);
    out << Eigen::Map<Eigen::VectorXd>(X.data(), X.size()), Eigen::Map<Eigen::VectorXd>(U.data(), U.size()), grid.getDt();
    return out;
}

// scenario "unicycle": cfg 3 single instance; "dint": cfg 2 (free dt, 5 consecutive solves, new_run only first)
static Run run(const std::string& scenario, bool hip, int N)
{
    Run r;
    SystemDynamicsInterface::Ptr dyn;
    std::shared_ptr<FiniteDifferencesGrid> grid;
    auto hg = std::make_shared<HyperGraphOptimizationProblemEdgeBased>();
    NlpSolverInterface::Ptr solver;
    corbo_hip_problem_desc d;
    std::memset(&d, 0, sizeof(d));
    for (int i = 0; i < CORBO_HIP_MAX_NX; ++i) { d.x_lb[i] = -CORBO_HIP_INF; d.x_ub[i] = CORBO_HIP_INF; }
    for (int i = 0; i < CORBO_HIP_MAX_NU; ++i) { d.u_lb[i] = -CORBO_HIP_INF; d.u_ub[i] = CORBO_HIP_INF; }
    double w;
    Eigen::VectorXd x0, xf;
    int solves = 1;
    if (scenario == "unicycle")
    {
        dyn  = std::make_shared<UnicycleRef>();
        grid = std::make_shared<FiniteDifferencesGrid>();
        w    = 10;
        x0   = Eigen::Vector3d(0, 0, 0);
        xf   = Eigen::Vector3d(2, 1, 0.5);
        d.grid = CORBO_HIP_GRID_FD; d.defect = CORBO_HIP_DEFECT_CRANK_NICOLSON; d.dynamics = CORBO_HIP_DYN_UNICYCLE;
        d.stage_cost = CORBO_HIP_COST_QUADRATIC_LSQ; d.final_cost = 1; d.nx = 3; d.nu = 2;
        const double q[3] = {1, 1, 0.1}, rr[2] = {0.1, 0.05};
        for (int i = 0; i < 3; ++i) { d.q_diag[i] = q[i]; d.qf_diag[i] = 10.0 * q[i]; }
        for (int i = 0; i < 2; ++i) d.r_diag[i] = rr[i];
    }
    else
    {
        dyn       = std::make_shared<SerialIntegratorSystem>(2);
        auto vg   = std::make_shared<FiniteDifferencesVariableGrid>();
        vg->setDtBounds(0.01, 10.0);
        Eigen::Matrix<bool, -1, 1> fixed(2);
        fixed.setConstant(true);
        vg->setXfFixed(fixed);
        grid   = vg;
        w      = 100;
        x0     = Eigen::Vector2d(0, 0);
        xf     = Eigen::Vector2d(1, 0);
        solves = 5;
        d.grid = CORBO_HIP_GRID_FD_VARIABLE; d.defect = CORBO_HIP_DEFECT_CRANK_NICOLSON; d.dynamics = CORBO_HIP_DYN_SERIAL_INTEGRATOR;
        d.stage_cost = CORBO_HIP_COST_MIN_TIME_LSQ; d.final_cost = 0; d.nx = 2; d.nu = 1; d.dyn_params[0] = 1.0;
    }
    d.N = N; d.dt_ref = 0.1;
    if (hip)
    {
        auto s = std::make_shared<LevenbergMarquardtSparseHip>();
        s->setIterations(10);
        s->setPenaltyWeights(w, w, w);
        s->setDeviceModel(d);
        s->setStateReference(xf);
        solver = s;
    }
    else
    {
        auto s = std::make_shared<LevenbergMarquardtSparse>();
        s->setIterations(10);
        s->setPenaltyWeights(w, w, w);
        solver = s;
    }
    grid->setNRef(N);
    grid->setDtRef(0.1);
    grid->setCostIntegrationRule(FullDiscretizationGridBase::CostIntegrationRule::LeftSum);
    StructuredOptimalControlProblem ocp(grid, dyn, hg, solver);
    if (scenario == "unicycle")
    {
        Eigen::MatrixXd Q  = Eigen::Vector3d(1, 1, 0.1).asDiagonal();
        Eigen::MatrixXd R  = Eigen::Vector2d(0.1, 0.05).asDiagonal();
        Eigen::MatrixXd Qf = 10.0 * Q;
        ocp.setStageCost(std::make_shared<QuadraticFormCost>(Q, R, false, true));
        ocp.setFinalStageCost(std::make_shared<QuadraticFinalStateCost>(Qf, true));
        ocp.setBounds(Eigen::Vector3d::Constant(-10), Eigen::Vector3d::Constant(10), Eigen::Vector2d::Constant(-1), Eigen::Vector2d::Constant(1));
    }
    else
    {
        ocp.setStageCost(std::make_shared<MinimumTime>(true));
        ocp.setControlBounds(Eigen::VectorXd::Constant(1, -1), Eigen::VectorXd::Constant(1, 1));
    }
    if (!ocp.initialize()) return r;
    StaticReference xref(xf);
    ZeroReference uref(d.nu);
    r.ok = true;
    for (int i = 0; i < solves; ++i) r.ok = ocp.compute(x0, xref, uref, nullptr, Time(0), i == 0) && r.ok;
    r.traj = trajectory(ocp, *grid);
    r.chi2 = ocp.getCurrentObjectiveValue();
    return r;
}

int main(int argc, char** argv)
{
    int rc = 0;
    for (const char* sc : {"unicycle", "dint"})
    {
        const int N = std::string(sc) == "unicycle" ? 100 : 50;
        Run a = run(sc, false, N);
        Run b = run(sc, true, N);
        double diff = (a.ok && b.ok && a.traj.size() == b.traj.size()) ? (a.traj - b.traj).cwiseAbs().maxCoeff() : 1e300;
        printf("{\"scenario\": \"%s\", \"ok_reference\": %d, \"ok_hip\": %d, \"chi2_reference\": %.17g, \"chi2_hip\": %.17g, \"max_abs_diff\": %.6e}\n", sc,
               a.ok ? 1 : 0, b.ok ? 1 : 0, a.chi2, b.chi2, diff);
        if (!(diff < 1e-5)) rc = 1;
    }
    return rc;
}
