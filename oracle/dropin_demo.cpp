// oracle/dropin_demo.cpp -- TEST INFRASTRUCTURE ONLY.
//
// Drop-in proof: the GENUINE reference's StructuredOptimalControlProblem (compiled from /root/reference into
// oracle/_ref/libcorbo_ref.a) is run twice on the same OCP, once with its own corbo::LevenbergMarquardtSparse and once with
// corbo::LevenbergMarquardtSparseHip (control_box_rst_amd/adapter/, over the C-ABI in libcorbo_hip.so) injected through the
// unchanged NlpSolverInterface plug-in point.  Prints one JSON line per scenario with the max trajectory difference.
// Built here (needs the reference's headers), runs on the GPU box (tests/test_gpu_dropin.py).
#include <corbo-core/reference_trajectory.h>
#include <corbo-core/time.h>
#include <corbo-optimal-control/functions/final_state_constraints.h>
#include <corbo-optimal-control/functions/final_state_cost.h>
#include <corbo-optimal-control/functions/minimum_time.h>
#include <corbo-optimal-control/functions/quadratic_cost.h>
#include <corbo-optimal-control/structured_ocp/discretization_grids/finite_differences_grid.h>
#include <corbo-numerics/explicit_integrators.h>
#include <corbo-optimal-control/functions/stage_functions.h>
#include <corbo-optimal-control/structured_ocp/discretization_grids/finite_differences_variable_grid.h>
#include <corbo-optimal-control/structured_ocp/discretization_grids/multiple_shooting_grid.h>
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

class QuadrotorRef : public SystemDynamicsInterface  // same expressions as oracle/ref_driver.cpp
{
 public:
    Ptr getInstance() const override { return std::make_shared<QuadrotorRef>(); }
    bool isContinuousTime() const override { return true; }
    bool isLinear() const override { return false; }
    int getInputDimension() const override { return 4; }
    int getStateDimension() const override { return 12; }
    void dynamics(const Eigen::Ref<const StateVector>& x, const Eigen::Ref<const ControlVector>& u, Eigen::Ref<StateVector> f) const override
    {
        const double g = 9.81, m = 1.0, Ixx = 0.01, Iyy = 0.01, Izz = 0.02;
        double sphi = std::sin(x[6]), cphi = std::cos(x[6]), sth = std::sin(x[7]), cth = std::cos(x[7]), spsi = std::sin(x[8]), cpsi = std::cos(x[8]);
        double tm = u[0] / m;
        f[0]  = x[3];
        f[1]  = x[4];
        f[2]  = x[5];
        f[3]  = (cphi * sth * cpsi + sphi * spsi) * tm;
        f[4]  = (cphi * sth * spsi - sphi * cpsi) * tm;
        f[5]  = cphi * cth * tm - g;
        f[6]  = x[9] + (x[10] * sphi + x[11] * cphi) * (sth / cth);
        f[7]  = x[10] * cphi - x[11] * sphi;
        f[8]  = (x[10] * sphi + x[11] * cphi) / cth;
        f[9]  = ((Iyy - Izz) * x[10] * x[11] + u[1]) / Ixx;
        f[10] = ((Izz - Ixx) * x[9] * x[11] + u[2]) / Iyy;
        f[11] = ((Ixx - Iyy) * x[9] * x[10] + u[3]) / Izz;
    }
};

class BallKeepOut : public StageInequalityConstraint
{
 public:
    BallKeepOut(double cx, double cy, double cz, double r) : _cx(cx), _cy(cy), _cz(cz), _r(r) {}
    StageInequalityConstraint::Ptr getInstance() const override { return std::make_shared<BallKeepOut>(_cx, _cy, _cz, _r); }
    int getNonIntegralStateTermDimension(int k) const override { return 1; }
    void computeNonIntegralStateTerm(int k, const Eigen::Ref<const Eigen::VectorXd>& x, Eigen::Ref<Eigen::VectorXd> cost) const override
    {
        double dx = x[0] - _cx, dy = x[1] - _cy, dz = x[2] - _cz;
        cost[0]   = _r * _r - (dx * dx + dy * dy + dz * dz);
    }

 private:
    double _cx, _cy, _cz, _r;
};

struct Run
{
    Eigen::VectorXd traj;
    double chi2 = 0;
    bool ok     = false;
};

static Eigen::VectorXd trajectory(StructuredOptimalControlProblem& ocp, DiscretizationGridInterface& grid)
{
    auto xs = std::make_shared<TimeSeries>();
    auto us = std::make_shared<TimeSeries>();
    ocp.getTimeSeries(xs, us);
    Eigen::MatrixXd X = xs->getValuesMatrixView(), U = us->getValuesMatrixView();
    Eigen::VectorXd out(X.size() + U.size() + 1);
    out << Eigen::Map<Eigen::VectorXd>(X.data(), X.size()), Eigen::Map<Eigen::VectorXd>(U.data(), U.size()), grid.getFirstDt();
    return out;
}

// scenario "unicycle": cfg 3 single instance; "dint": cfg 2 (free dt, 5 consecutive solves, new_run only first)
static Run run(const std::string& scenario_in, bool hip, int N)
{
    // "<scenario>_mismatch": the OCP of <scenario>, but the device model handed to the HIP solver states another state weight
    const bool mismatch        = scenario_in.size() > 9 && scenario_in.compare(scenario_in.size() - 9, 9, "_mismatch") == 0;
    const std::string scenario = mismatch ? scenario_in.substr(0, scenario_in.size() - 9) : scenario_in;
    Run r;
    SystemDynamicsInterface::Ptr dyn;
    std::shared_ptr<FiniteDifferencesGrid> grid;
    std::shared_ptr<MultipleShootingGrid> ms_grid;
    auto hg = std::make_shared<HyperGraphOptimizationProblemEdgeBased>();
    NlpSolverInterface::Ptr solver;
    corbo_hip_problem_desc d;
    std::memset(&d, 0, sizeof(d));
    for (int i = 0; i < CORBO_HIP_MAX_NX; ++i) { d.x_lb[i] = -CORBO_HIP_INF; d.x_ub[i] = CORBO_HIP_INF; }
    for (int i = 0; i < CORBO_HIP_MAX_NU; ++i) { d.u_lb[i] = -CORBO_HIP_INF; d.u_ub[i] = CORBO_HIP_INF; }
    double w;
    Eigen::VectorXd x0, xf;
    int solves = 1;
    const bool tball = (scenario == "unicycle_tball");   // cfg 3 structure, short horizon, TerminalBall final-stage constraint
    if (scenario == "unicycle" || tball)
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
        if (tball)
        {
            d.final_ineq = CORBO_HIP_FINAL_INEQ_TERMINAL_BALL;
            d.final_ineq_params[0] = 1.0; d.final_ineq_params[1] = 1.0; d.final_ineq_params[2] = 0.1; d.final_ineq_params[3] = 0.05;  // S, gamma
        }
    }
    else if (scenario == "quad")
    {
        dyn     = std::make_shared<QuadrotorRef>();
        ms_grid = std::make_shared<MultipleShootingGrid>();
        ms_grid->setNumericalIntegrator(std::make_shared<IntegratorExplicitRungeKutta4>());
        w  = 10;
        x0 = Eigen::VectorXd::Zero(12);
        xf = Eigen::VectorXd::Zero(12);
        xf[0] = 2; xf[1] = 1; xf[2] = 1;
        d.grid = CORBO_HIP_GRID_MS; d.defect = CORBO_HIP_DEFECT_RK4_SHOOTING; d.dynamics = CORBO_HIP_DYN_QUADROTOR;
        d.stage_cost = CORBO_HIP_COST_QUADRATIC_LSQ; d.final_cost = 1; d.nx = 12; d.nu = 4; d.stage_ineq = CORBO_HIP_INEQ_BALL;
        const double q[12] = {1, 1, 1, 0.1, 0.1, 0.1, 0.5, 0.5, 0.5, 0.05, 0.05, 0.05}, rr[4] = {0.01, 0.1, 0.1, 0.1};
        for (int i = 0; i < 12; ++i) { d.q_diag[i] = q[i]; d.qf_diag[i] = 10.0 * q[i]; }
        for (int i = 0; i < 4; ++i) d.r_diag[i] = rr[i];
        d.ineq_params[0] = 1.0; d.ineq_params[1] = 0.5; d.ineq_params[2] = 0.6; d.ineq_params[3] = 0.4;
        d.dyn_params[0] = 9.81; d.dyn_params[1] = 1.0; d.dyn_params[2] = 0.01; d.dyn_params[3] = 0.01; d.dyn_params[4] = 0.02;
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
    const double dt = (scenario == "quad") ? 0.05 : 0.1;
    d.N = N; d.dt_ref = dt;
    if (mismatch) d.q_diag[1] = 2.0 * d.q_diag[1];
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
    DiscretizationGridInterface::Ptr any_grid;
    if (grid)
    {
        grid->setNRef(N);
        grid->setDtRef(dt);
        grid->setCostIntegrationRule(FullDiscretizationGridBase::CostIntegrationRule::LeftSum);
        any_grid = grid;
    }
    else
    {
        ms_grid->setNRef(N);
        ms_grid->setDtRef(dt);
        any_grid = ms_grid;
    }
    StructuredOptimalControlProblem ocp(any_grid, dyn, hg, solver);
    if (scenario == "unicycle" || tball)
    {
        Eigen::MatrixXd Q  = Eigen::Vector3d(1, 1, 0.1).asDiagonal();
        Eigen::MatrixXd R  = Eigen::Vector2d(0.1, 0.05).asDiagonal();
        Eigen::MatrixXd Qf = 10.0 * Q;
        ocp.setStageCost(std::make_shared<QuadraticFormCost>(Q, R, false, true));
        ocp.setFinalStageCost(std::make_shared<QuadraticFinalStateCost>(Qf, true));
        ocp.setBounds(Eigen::Vector3d::Constant(-10), Eigen::Vector3d::Constant(10), Eigen::Vector2d::Constant(-1), Eigen::Vector2d::Constant(1));
        if (tball)
        {
            Eigen::MatrixXd Sm = Eigen::Vector3d(1, 1, 0.1).asDiagonal();
            ocp.setFinalStageConstraint(std::make_shared<TerminalBall>(Sm, 0.05));
        }
    }
    else if (scenario == "quad")
    {
        Eigen::VectorXd q(12), rr(4), ulb(4), uub(4);
        q << 1, 1, 1, 0.1, 0.1, 0.1, 0.5, 0.5, 0.5, 0.05, 0.05, 0.05;
        rr << 0.01, 0.1, 0.1, 0.1;
        ulb << 0, -1, -1, -1;
        uub << 20, 1, 1, 1;
        Eigen::MatrixXd Q = q.asDiagonal(), R = rr.asDiagonal(), Qf = 10.0 * Q;
        ocp.setStageCost(std::make_shared<QuadraticFormCost>(Q, R, false, true));
        ocp.setFinalStageCost(std::make_shared<QuadraticFinalStateCost>(Qf, true));
        ocp.setControlBounds(ulb, uub);
        ocp.setStageInequalityConstraint(std::make_shared<BallKeepOut>(1.0, 0.5, 0.6, 0.4));
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
    r.traj = trajectory(ocp, *any_grid);
    r.chi2 = ocp.getCurrentObjectiveValue();
    return r;
}

int main(int argc, char** argv)
{
    int rc = 0;
    for (const char* sc : {"unicycle", "dint", "quad", "unicycle_tball"})
    {
        const int N = std::string(sc) == "unicycle" ? 100 : std::string(sc) == "dint" ? 50 : 30;
        Run a = run(sc, false, N);
        Run b = run(sc, true, N);
        double diff = (a.ok && b.ok && a.traj.size() == b.traj.size()) ? (a.traj - b.traj).cwiseAbs().maxCoeff() : 1e300;
        printf("{\"scenario\": \"%s\", \"ok_reference\": %d, \"ok_hip\": %d, \"chi2_reference\": %.17g, \"chi2_hip\": %.17g, \"max_abs_diff\": %.6e}\n", sc,
               a.ok ? 1 : 0, b.ok ? 1 : 0, a.chi2, b.chi2, diff);
        if (!(diff < (std::string(sc) == "quad" ? 5e-3 : 1e-5))) rc = 1;
    }
    {   // a device model that does not describe the graph must be refused (SolverStatus::Error -> compute() fails), not solved
        Run c = run("unicycle_mismatch", true, 30);
        printf("{\"scenario\": \"unicycle_mismatch\", \"ok_hip\": %d}\n", c.ok ? 1 : 0);
        if (c.ok) rc = 1;
    }
    return rc;
}
