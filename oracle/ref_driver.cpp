// oracle/ref_driver.cpp -- TEST INFRASTRUCTURE ONLY.
//
// Drives the GENUINE reference (control_box_rst, compiled from /root/reference by oracle/Makefile into
// oracle/_ref/libcorbo_ref.a) through its public C++ setters, exactly the way SURVEY.md 8(c) describes:
//   grid + dynamics + HyperGraphOptimizationProblemEdgeBased + LevenbergMarquardtSparse
//   -> StructuredOptimalControlProblem::compute()  (structured_optimal_control_problem.cpp:77-154)
// It is this repo's own code (no reference source is copied); it only *calls* the reference.
//
// Modes
//   ref_driver dump  scenario=<name> [N=..] [iters=..] [x0=a,b,..] [xf=a,b,..] [solves=..]
//        -> JSON on stdout: dimensions, initial parameter vector, values and Jacobian triplets at the initial
//           point (hot-path rows a2-a6 of SURVEY 8a), and the vertex trajectory + chi2 after k = 1..iters LM
//           iterations (each obtained by a fresh OCP run with setIterations(k); pins row a1).
//   ref_driver bench scenario=<name> batch=<B> [seed=..] [iters=..]
//        -> solves B seeded instances sequentially on one core, prints JSON with solver wall time
//           (OptimalControlProblemStatistics::solving_time, i.e. only levenberg_marquardt_sparse.cpp:44-220).
//
//   ref_driver loop  scenario=<name> [steps=..] [iters=..] [shift=0|1] [integrator=euler|rk4] [disturbance=amp]
//        -> closed loop with the reference's SimulatedPlant (see loop() below).
//   ref_driver kat
//        -> the known-answer cases of the reference's own LM solver test re-run on non-OCP problems (see kat() below).
//
// Scenarios: unicycle (cfg 3), vdp (cfg 1), dint (cfg 2: time-optimal double integrator).
#include <corbo-core/reference_trajectory.h>
#include <corbo-core/time.h>
#include <corbo-numerics/finite_differences_collocation.h>
#include <corbo-optimal-control/functions/final_state_constraints.h>
#include <corbo-optimal-control/functions/final_state_cost.h>
#include <corbo-optimal-control/functions/hybrid_cost.h>
#include <corbo-optimal-control/functions/minimum_time.h>
#include <corbo-optimal-control/functions/quadratic_control_cost.h>
#include <corbo-optimal-control/functions/quadratic_state_cost.h>
#include <corbo-optimal-control/functions/quadratic_cost.h>
#include <corbo-optimal-control/statistics.h>
#include <corbo-optimal-control/structured_ocp/discretization_grids/finite_differences_grid.h>
#include <corbo-numerics/explicit_integrators.h>
#include <corbo-optimal-control/functions/stage_functions.h>
#include <corbo-optimal-control/structured_ocp/discretization_grids/finite_differences_variable_grid.h>
#include <corbo-optimal-control/structured_ocp/discretization_grids/multiple_shooting_grid.h>
#include <corbo-optimal-control/structured_ocp/discretization_grids/multiple_shooting_variable_grid.h>
#include <corbo-optimal-control/structured_ocp/structured_optimal_control_problem.h>
#include <corbo-optimization/hyper_graph/hyper_graph_optimization_problem_edge_based.h>
#include <corbo-optimization/simple_optimization_problem.h>
#include <corbo-plants/disturbance_interface.h>
#include <corbo-plants/simulated_plant.h>
#include <corbo-systems/output_function_interface.h>
#include <corbo-optimization/solver/levenberg_marquardt_sparse.h>
#include <corbo-systems/benchmark/linear_benchmark_systems.h>
#include <corbo-systems/benchmark/nonlinear_benchmark_systems.h>

#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <map>
#include <memory>
#include <random>
#include <sstream>
#include <string>
#include <vector>

using namespace corbo;

// Unicycle (not part of the reference; a user plug-in through SystemDynamicsInterface):
//   xdot = u1 cos(theta), ydot = u1 sin(theta), thetadot = u2
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

// Kinematic car (the user-model example of control_box_rst_amd/csrc/models/kinematic_car.hpp): state (x, y, theta), controls (v, delta),
// wheelbase L.  The expressions are character-for-character those of the device model and of oracle/corbo_oracle.c.
class KinematicCarRef : public SystemDynamicsInterface
{
 public:
    explicit KinematicCarRef(double wheelbase = 2.5) : L(wheelbase) {}
    Ptr getInstance() const override { return std::make_shared<KinematicCarRef>(); }
    bool isContinuousTime() const override { return true; }
    bool isLinear() const override { return false; }
    int getInputDimension() const override { return 2; }
    int getStateDimension() const override { return 3; }
    void dynamics(const Eigen::Ref<const StateVector>& x, const Eigen::Ref<const ControlVector>& u, Eigen::Ref<StateVector> f) const override
    {
        f[0] = u[0] * std::cos(x[2]);
        f[1] = u[0] * std::sin(x[2]);
        f[2] = u[0] / L * std::tan(u[1]);
    }
    double L;
};

// Planar quadrotor (the big-block user-model example of control_box_rst_amd/csrc/models/planar_quadrotor.hpp): state (x, z, theta, x', z',
// theta'), controls = the two rotor thrusts; m, I, l, g.  The expressions are those of the device model and of oracle/corbo_oracle.c.
class PlanarQuadrotorRef : public SystemDynamicsInterface
{
 public:
    Ptr getInstance() const override { return std::make_shared<PlanarQuadrotorRef>(); }
    bool isContinuousTime() const override { return true; }
    bool isLinear() const override { return false; }
    int getInputDimension() const override { return 2; }
    int getStateDimension() const override { return 6; }
    void dynamics(const Eigen::Ref<const StateVector>& x, const Eigen::Ref<const ControlVector>& u, Eigen::Ref<StateVector> f) const override
    {
        const double m = 1.0, I = 0.05, l = 0.25, g = 9.81;
        const double T = u[0] + u[1];
        f[0] = x[3];
        f[1] = x[4];
        f[2] = x[5];
        f[3] = -(T * std::sin(x[2])) / m;
        f[4] = (T * std::cos(x[2])) / m - g;
        f[5] = (u[0] - u[1]) * l / I;
    }
};

// Quadrotor (user plug-in, DESIGN.md "quadrotor"): x = [p(3) v(3) roll pitch yaw  body rates(3)], u = [thrust, torques(3)],
// params g, m, Ixx, Iyy, Izz.  The expressions are character-for-character those of oracle/corbo_oracle.c and the device model.
class QuadrotorRef : public SystemDynamicsInterface
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

// keep-out ball on the position: c(x) = r^2 - |x[0:3] - center|^2 <= 0   (one nonlinear stage inequality per grid point, cfg 5)
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

// A user's stage inequalities as ONE object (the OCP holds a single StageInequalityConstraint): the keep-out ball either as the non-integral
// state term (= BallKeepOut) or as the INTEGRAL state-control term (the grid then creates TrapezoidalIntegralInequalityEdge / LeftSumInequalityEdge,
// finite_differences_grid.cpp:107-122), and an input-rate limit ((u_k - u_prev) / dt_prev)^2 - r_max^2 <= 0 per control as the
// control-deviation term (TernaryVectorScalarVertexEdge on (u_k, u_{k-1}, dt), nlp_functions.cpp:117-131, 152-186).
class UserStageInequalities : public StageInequalityConstraint
{
 public:
    Eigen::VectorXd ball;        // cx, cy, cz, r or empty
    bool ball_integral = false;
    Eigen::VectorXd rate;        // r_max per control, or empty
    double tilt  = 0;            // > 0: a tilt cone x[6]^2 + x[7]^2 - tilt^2 <= 0 as the non-integral state term (instead of the ball; csrc/stage_functions/tilt_cone.hpp)
    double unorm = 0;            // > 0: an input-magnitude bound |u|^2 - unorm^2 <= 0 as the non-integral CONTROL term (csrc/stage_functions/control_norm.hpp)
    StageInequalityConstraint::Ptr getInstance() const override { return std::make_shared<UserStageInequalities>(*this); }
    int getNonIntegralStateTermDimension(int k) const override { return ((ball.size() == 4 && !ball_integral) || tilt > 0) ? 1 : 0; }
    void computeNonIntegralStateTerm(int k, const Eigen::Ref<const Eigen::VectorXd>& x, Eigen::Ref<Eigen::VectorXd> cost) const override
    {
        if (tilt > 0) cost[0] = (x[6] * x[6] + x[7] * x[7]) - tilt * tilt;
        else cost[0] = ballValue(x);
    }
    int getNonIntegralControlTermDimension(int k) const override { return unorm > 0 ? 1 : 0; }
    void computeNonIntegralControlTerm(int k, const Eigen::Ref<const Eigen::VectorXd>& u, Eigen::Ref<Eigen::VectorXd> cost) const override
    {
        double acc = 0.0;
        for (int i = 0; i < u.size(); ++i) acc += u[i] * u[i];
        cost[0] = acc - unorm * unorm;
    }
    int getIntegralStateControlTermDimension(int k) const override { return (ball.size() == 4 && ball_integral) ? 1 : 0; }
    void computeIntegralStateControlTerm(int k, const Eigen::Ref<const Eigen::VectorXd>& x, const Eigen::Ref<const Eigen::VectorXd>& u,
                                         Eigen::Ref<Eigen::VectorXd> cost) const override
    {
        cost[0] = ballValue(x);
    }
    int getNonIntegralControlDeviationTermDimension(int k) const override { return (int)rate.size(); }
    void computeNonIntegralControlDeviationTerm(int k, const Eigen::Ref<const Eigen::VectorXd>& u_k, const Eigen::Ref<const Eigen::VectorXd>& u_prev,
                                                double dt_prev, Eigen::Ref<Eigen::VectorXd> cost) const override
    {
        for (int i = 0; i < rate.size(); ++i)
        {
            const double d = (u_k[i] - u_prev[i]) / dt_prev;
            cost[i]        = d * d - rate[i] * rate[i];
        }
    }

 private:
    double ballValue(const Eigen::Ref<const Eigen::VectorXd>& x) const
    {
        double dx = x[0] - ball[0], dy = x[1] - ball[1], dz = x[2] - ball[2];
        return ball[3] * ball[3] - (dx * dx + dy * dy + dz * dz);
    }
};

// A user's stage equality in INTEGRAL form: a^T x + b^T u - c (one row; e.g. a path constraint that has to hold on average over every interval).  The
// grid integrates it with its cost integration rule: appended to the dynamics edge (TrapezoidalIntegralEqualityDynamicsEdge) or as a
// LeftSumEqualityEdge in front of it (finite_differences_grid.cpp:80-106).
class LinearIntegralEquality : public StageEqualityConstraint
{
 public:
    Eigen::VectorXd a, b;
    double c = 0;
    StageEqualityConstraint::Ptr getInstance() const override { return std::make_shared<LinearIntegralEquality>(*this); }
    int getIntegralStateControlTermDimension(int k) const override { return 1; }
    void computeIntegralStateControlTerm(int k, const Eigen::Ref<const Eigen::VectorXd>& x, const Eigen::Ref<const Eigen::VectorXd>& u,
                                         Eigen::Ref<Eigen::VectorXd> cost) const override
    {
        double acc = 0.0;
        for (int i = 0; i < a.size(); ++i) acc += a[i] * x[i];
        for (int i = 0; i < b.size(); ++i) acc += b[i] * u[i];
        cost[0] = acc - c;
    }
};

// MinTimeQuadratic::_only_last_n has no setter outside fromMessage (hybrid_cost.h:300): reach the protected member through a subclass
struct MinTimeQuadraticLastN : public MinTimeQuadratic
{
    MinTimeQuadraticLastN(const Eigen::MatrixXd& Q, const Eigen::MatrixXd& R, int last_n, bool integral_form = false, bool lsq_form = true) : MinTimeQuadratic(Q, R, integral_form, lsq_form) { _only_last_n = last_n; }
};

struct Scenario
{
    std::string name;
    int nx = 0, nu = 0, N = 0;
    double dt = 0.1;
    int iters  = 10;
    int solves = 1;  // consecutive compute() calls (new_run only for the first one)
    double w_eq = 2, w_ineq = 2, w_b = 2;
    Eigen::VectorXd x0, xf;
    std::string collocation = "crank_nicolson";
    Eigen::VectorXd ball;       // ball=cx,cy,cz,r: BallKeepOut stage inequality on the first three state components (unicycle)
    // integral-form constraints and the control-deviation term (user stage functions above; FiniteDifferencesGrid / ...VariableGrid):
    std::string crule;          // crule=trap|left: the grid's integration rule for the integral constraint edges (setCostIntegrationRule)
    bool noball = false;        // noball=1 (pquad): without the keep-out ball the scenario carries by default
    bool ball_integral = false; // ball_int=1 (with ball=): the ball as the INTEGRAL state-control term of the stage inequalities
    double tilt = 0;            // tilt=<alpha>: the tilt cone as the stage inequalities' state term (user stage function, slot 0)
    double unorm = 0;           // unorm=<r>: the input-magnitude bound as their control term (user stage function, slot 1)
    Eigen::VectorXd eq_lin;     // eq_lin=a_1..a_nx,b_1..b_nu,c: LinearIntegralEquality
    Eigen::VectorXd rate;       // rate=r_1..r_nu: input-rate limit as the control-deviation term of the stage inequalities
    Eigen::VectorXd u_prev;     // u_prev=... (with rate=): the previously applied control (setPreviousControlInput), default zero
    double u_prev_dt = 0;       // u_prev_dt=<dt>: its age (default: the grid's dt)
    Eigen::VectorXd xlb, xub, ulb, uub;   // xlb=/xub=/ulb=/uub= comma lists ("inf" = unbounded): replace the scenario's box bounds
    // cost=mtq|qstate|qctrl|mtqs|mtqc: replace the scenario's stage cost by MinTimeQuadratic / QuadraticStateCost / QuadraticControlCost /
    // MinTimeQuadraticStates / MinTimeQuadraticControls (lsq form), Q = diag(1, 0.5, 0.2, 0.1)[:nx], R = diag(0.1, 0.2, 0.05)[:nu]
    std::string cost;
    int last_n = 0;             // last_n=<n> with cost=mtq: MinTimeQuadratic's only_last_n
    std::string integral;       // integral=trap|left (unicycle, vdp; with lsq=0): QuadraticFormCost in integral form, the grid's cost integration rule
    mutable Eigen::MatrixXd Qfull, Rfull, Qffull;   // what fullq = 1 configured (filled by build(); dump prints their factors)
    std::string ms_integrator;  // ms_integrator=euler|rk2|rk3 (shooting grids): IntegratorExplicitEuler / RungeKutta2 / RungeKutta3 instead of RungeKutta4
    bool fullq = false;         // fullq=1 (unicycle, vdp, par2 / par3 / lin, the zoo): NON-DIAGONAL Q, R, Qf = 10 Q (off-diagonal entries 0.25 sqrt(w_i w_j)): the dense
                                // branch of QuadraticFormCost::setWeightQ / setWeightR and QuadraticFinalStateCost::setWeightQf (upper Cholesky factors)
    bool nonlsq = false;        // lsq=0 (unicycle, vdp, dint, int3 vargrid, cost=mtq; hess mode): QuadraticFormCost / QuadraticFinalStateCost with lsq_form = false -- scalar terms
    int xf_fixed = -1;          // xf_fixed=<bit mask>: partially fixed goal state (setXfFixed), unicycle / vdp
    int final_cost = -1;        // final_cost=0: no final-state cost
    bool vargrid = false;       // vargrid=1 (int3): FiniteDifferencesVariableGrid, x_f fixed, MinimumTime(lsq)
    bool teq = false;           // teq=1: TerminalEqualityConstraint(xf) final-stage constraint
    int teq_mask = 0;           // teq_mask=m (with teq=1): TerminalPartialEqualityConstraint, component i active iff bit i of m
    bool fd = false;            // grid=fd (scenarios quad / pquad, whose default is the shooting grid): FiniteDifferencesGrid + collocation
    bool ms = false;            // grid=ms: MultipleShootingGrid + RK4 instead of the finite-differences grid (vdp, unicycle)
    double tball_gamma = 0;     // tball=<gamma>: TerminalBall(S, gamma) final-stage constraint, S = tball_s (diagonal)
    Eigen::VectorXd tball_s;    // empty = no terminal ball
    Eigen::VectorXd lin_a, lin_b;   // scenario lin: LinearStateSpaceModel matrices, row-major (lin_a= / lin_b= with nx= / nu=)
    // adapt=single|aggressive|shrink (variable grids): FiniteDifferencesVariableGrid::setGridAdaptTimeBased* / SimpleShrinkingHorizon
    // (finite_differences_variable_grid.cpp:44-66) with nmax= / nmin= / hyst= / adapt_first=
    std::string adapt;
    int n_max = 1000, n_min = 2;
    double hyst = 0.1;
    bool adapt_first = false;
    // xref_traj=1: a time-varying state reference (DiscreteTimeReferenceTrajectory, one sample per grid point, zero-order hold) that ends in
    // xf; uref=<csv>: a static non-zero control reference (StaticReference) instead of ZeroReference
    bool xref_traj = false;
    Eigen::VectorXd uref;
    // qdiag= / rdiag= / qfdiag= (unicycle): diagonal weights instead of the scenario's; start=<file>: the parameter vector (the hypergraph's
    // active parameters, in its order) the solve starts from instead of the grid's straight-line initial guess -- see startFrom() below
    Eigen::VectorXd qdiag, rdiag, qfdiag;
    std::string start;
};

// the reference's other benchmark systems (nonlinear_benchmark_systems.h), default parameters: nx = 2 except the rocket (3) and the cart-pole (4)
static bool isZoo(const std::string& n) { return n == "duffing" || n == "rocket" || n == "pendulum" || n == "mpendulum" || n == "toy" || n == "artstein" || n == "cartpole"; }

struct Built
{
    std::shared_ptr<FullDiscretizationGridBase> grid;  // FD grids
    std::shared_ptr<MultipleShootingGrid> ms_grid;     // multiple-shooting grid (scenario quad)
    DiscretizationGridInterface::Ptr any_grid;
    std::shared_ptr<HyperGraphOptimizationProblemEdgeBased> hg;
    std::shared_ptr<LevenbergMarquardtSparse> solver;
    std::shared_ptr<StructuredOptimalControlProblem> ocp;
    std::shared_ptr<OptimalControlProblemStatistics> stats;
    ReferenceTrajectoryInterface::Ptr xref;
    ReferenceTrajectoryInterface::Ptr uref;
    SystemDynamicsInterface::Ptr dyn;
};

static FiniteDifferencesCollocationInterface::Ptr makeCollocation(const std::string& n)
{
    if (n == "forward") return std::make_shared<ForwardDiffCollocation>();
    if (n == "backward") return std::make_shared<BackwardDiffCollocation>();
    if (n == "midpoint") return std::make_shared<MidpointDiffCollocation>();
    return std::make_shared<CrankNicolsonDiffCollocation>();
}

// fullq = 1: the weight with off-diagonal entries 0.25 sqrt(w_i w_j) (symmetric positive definite for these sizes)
static Eigen::MatrixXd fullWeight(const Eigen::MatrixXd& D)
{
    Eigen::MatrixXd W = D;
    for (int i = 0; i < W.rows(); ++i)
        for (int j = 0; j < W.cols(); ++j)
            if (i != j) W(i, j) = 0.25 * std::sqrt(D(i, i) * D(j, j));
    return W;
}

static NumericalIntegratorExplicitInterface::Ptr shootingIntegrator(const Scenario& s)
{
    if (s.ms_integrator == "euler") return std::make_shared<IntegratorExplicitEuler>();
    if (s.ms_integrator == "rk2") return std::make_shared<IntegratorExplicitRungeKutta2>();
    if (s.ms_integrator == "rk3") return std::make_shared<IntegratorExplicitRungeKutta3>();
    if (s.ms_integrator == "rk5") return std::make_shared<IntegratorExplicitRungeKutta5>();
    if (s.ms_integrator == "rk6") return std::make_shared<IntegratorExplicitRungeKutta6>();
    if (s.ms_integrator == "rk7") return std::make_shared<IntegratorExplicitRungeKutta7>();
    return std::make_shared<IntegratorExplicitRungeKutta4>();
}

static Built build(const Scenario& s, int iterations)
{
    Built b;
    SystemDynamicsInterface::Ptr dyn;
    b.hg     = std::make_shared<HyperGraphOptimizationProblemEdgeBased>();
    b.solver = std::make_shared<LevenbergMarquardtSparse>();
    b.solver->setIterations(iterations);
    b.solver->setPenaltyWeights(s.w_eq, s.w_ineq, s.w_b);
    b.stats = std::make_shared<OptimalControlProblemStatistics>();

    auto make_ms = [&]() {
        b.ms_grid = std::make_shared<MultipleShootingGrid>();
        b.ms_grid->setNumericalIntegrator(shootingIntegrator(s));
        b.ms_grid->setNRef(s.N);
        b.ms_grid->setDtRef(s.dt);
        if (s.xf_fixed >= 0)
        {
            Eigen::Matrix<bool, -1, 1> fixed(s.nx);
            for (int i = 0; i < s.nx; ++i) fixed[i] = (s.xf_fixed >> i) & 1;
            b.ms_grid->setXfFixed(fixed);
        }
        b.any_grid = b.ms_grid;
    };
    if (s.name == "unicycle" || s.name == "kcar")   // kcar: the same OCP around the kinematic-car user model (wheelbase 2.5)
    {
        if (s.name == "kcar") dyn = std::make_shared<KinematicCarRef>(2.5);
        else dyn = std::make_shared<UnicycleRef>();
        if (s.ms) make_ms(); else b.grid = std::make_shared<FiniteDifferencesGrid>();
    }
    else if (s.name == "vdp")
    {
        dyn = std::make_shared<VanDerPolOscillator>();
        if (s.ms) make_ms(); else b.grid = std::make_shared<FiniteDifferencesGrid>();
    }
    else if (s.name == "lin")   // LinearStateSpaceModel(A, B)
    {
        Eigen::MatrixXd A(s.nx, s.nx), B(s.nx, s.nu);
        for (int i = 0; i < s.nx; ++i)
        {
            for (int j = 0; j < s.nx; ++j) A(i, j) = s.lin_a[i * s.nx + j];
            for (int j = 0; j < s.nu; ++j) B(i, j) = s.lin_b[i * s.nu + j];
        }
        auto sys = std::make_shared<LinearStateSpaceModel>();
        sys->setParameters(A, B);
        dyn = sys;
        if (s.ms) make_ms(); else b.grid = std::make_shared<FiniteDifferencesGrid>();
    }
    else if (s.name == "par2" || s.name == "par3")   // ParallelIntegratorSystem(p), time constant 1
    {
        auto sys = std::make_shared<ParallelIntegratorSystem>();
        sys->setDimension(s.nx);
        dyn = sys;
        if (s.ms) make_ms(); else b.grid = std::make_shared<FiniteDifferencesGrid>();
    }
    else if (isZoo(s.name))
    {
        if (s.name == "duffing") dyn = std::make_shared<DuffingOscillator>();
        else if (s.name == "rocket") dyn = std::make_shared<FreeSpaceRocket>();
        else if (s.name == "pendulum") dyn = std::make_shared<SimplePendulum>();
        else if (s.name == "mpendulum") dyn = std::make_shared<MasslessPendulum>();
        else if (s.name == "toy") dyn = std::make_shared<ToyExample>();
        else if (s.name == "cartpole") dyn = std::make_shared<CartPole>();
        else dyn = std::make_shared<ArtsteinsCircle>();
        if (s.ms) make_ms(); else b.grid = std::make_shared<FiniteDifferencesGrid>();
    }
    else if (s.name == "int3")
    {
        dyn = std::make_shared<SerialIntegratorSystem>(3);
        if (s.vargrid && s.ms)
        {   // time-optimal on the shooting grid: MultipleShootingVariableGrid (free dt), RK4, x_f fixed
            auto grid = std::make_shared<MultipleShootingVariableGrid>();
            grid->setNumericalIntegrator(shootingIntegrator(s));
            grid->setNRef(s.N);
            grid->setDtRef(s.dt);
            grid->setDtBounds(0.01, 10.0);
            Eigen::Matrix<bool, -1, 1> fixed(3);
            fixed.setConstant(true);
            grid->setXfFixed(fixed);
            b.ms_grid  = grid;
            b.any_grid = grid;
        }
        else if (s.vargrid)
        {
            auto grid = std::make_shared<FiniteDifferencesVariableGrid>();
            grid->setDtBounds(0.01, 10.0);
            Eigen::Matrix<bool, -1, 1> fixed(3);
            fixed.setConstant(true);
            grid->setXfFixed(fixed);
            b.grid = grid;
        }
        else if (s.ms) make_ms();
        else b.grid = std::make_shared<FiniteDifferencesGrid>();
    }
    else if (s.name == "dint")
    {
        dyn = std::make_shared<SerialIntegratorSystem>(2);
        Eigen::Matrix<bool, -1, 1> fixed(2);
        fixed.setConstant(true);
        if (s.ms)
        {   // cfg 2 on the shooting grid: MultipleShootingVariableGrid (free dt), RK4
            auto grid = std::make_shared<MultipleShootingVariableGrid>();
            grid->setNumericalIntegrator(shootingIntegrator(s));
            grid->setNRef(s.N);
            grid->setDtRef(s.dt);
            grid->setDtBounds(0.01, 10.0);
            grid->setXfFixed(fixed);
            b.ms_grid  = grid;
            b.any_grid = grid;
        }
        else
        {
            auto grid = std::make_shared<FiniteDifferencesVariableGrid>();
            grid->setDtBounds(0.01, 10.0);
            grid->setXfFixed(fixed);
            b.grid = grid;
        }
    }
    else if (s.name == "quad" || s.name == "pquad")   // pquad: the same kind of OCP around the 6-state planar quadrotor (a user model of the big-block family)
    {
        dyn = std::make_shared<QuadrotorRef>();
        if (s.name == "pquad") dyn = std::make_shared<PlanarQuadrotorRef>();
        if (s.vargrid)
        {   // vargrid=1: time-optimal transfer to a fixed x_f with a free dt (MultipleShootingVariableGrid, or FiniteDifferencesVariableGrid with grid=fd)
            Eigen::Matrix<bool, -1, 1> fixed(s.name == "pquad" ? 6 : 12);
            fixed.setConstant(true);
            if (s.fd)
            {
                auto grid = std::make_shared<FiniteDifferencesVariableGrid>();
                grid->setDtBounds(0.01, 10.0);
                grid->setXfFixed(fixed);
                b.grid = grid;
            }
            else
            {
                auto grid = std::make_shared<MultipleShootingVariableGrid>();
                grid->setNumericalIntegrator(shootingIntegrator(s));
                grid->setNRef(s.N);
                grid->setDtRef(s.dt);
                grid->setDtBounds(0.01, 10.0);
                grid->setXfFixed(fixed);
                b.ms_grid  = grid;
                b.any_grid = grid;
            }
        }
        else if (s.fd) b.grid = std::make_shared<FiniteDifferencesGrid>();   // grid=fd: the same OCP on the collocation grid
        else
        {
            b.ms_grid = std::make_shared<MultipleShootingGrid>();
            b.ms_grid->setNumericalIntegrator(shootingIntegrator(s));
            b.ms_grid->setNRef(s.N);
            b.ms_grid->setDtRef(s.dt);
            b.any_grid = b.ms_grid;
        }
    }
    else
    {
        fprintf(stderr, "unknown scenario %s\n", s.name.c_str());
        exit(2);
    }
    if (!s.adapt.empty())
    {
        auto vg = std::dynamic_pointer_cast<FiniteDifferencesVariableGrid>(b.grid);
        auto mg = std::dynamic_pointer_cast<MultipleShootingVariableGrid>(b.ms_grid);
        if (!vg && !mg)
        {
            fprintf(stderr, "adapt= needs a FiniteDifferencesVariableGrid / MultipleShootingVariableGrid scenario\n");
            exit(2);
        }
        if (mg)
        {   // (multiple_shooting_variable_grid.h:52-56: no adapt_first_iter argument, the member stays false)
            if (s.adapt_first) { fprintf(stderr, "adapt_first=1: not settable on a MultipleShootingVariableGrid\n"); exit(2); }
            mg->setNmin(s.n_min);
            if (s.adapt == "single") mg->setGridAdaptTimeBasedSingleStep(s.n_max, s.hyst);
            else if (s.adapt == "aggressive") mg->setGridAdaptTimeBasedAggressiveEstimate(s.n_max, s.hyst);
            else if (s.adapt == "shrink") mg->setGridAdaptSimpleShrinkingHorizon();
            else { fprintf(stderr, "unknown adapt=%s\n", s.adapt.c_str()); exit(2); }
        }
        else
        {
            vg->setNmin(s.n_min);
            if (s.adapt == "single") vg->setGridAdaptTimeBasedSingleStep(s.n_max, s.hyst, s.adapt_first);
            else if (s.adapt == "aggressive") vg->setGridAdaptTimeBasedAggressiveEstimate(s.n_max, s.hyst, s.adapt_first);
            else if (s.adapt == "shrink") vg->setGridAdaptSimpleShrinkingHorizon(s.adapt_first);
            else { fprintf(stderr, "unknown adapt=%s\n", s.adapt.c_str()); exit(2); }
        }
    }
    if (b.grid)
    {
        b.grid->setNRef(s.N);
        b.grid->setDtRef(s.dt);
        b.grid->setCostIntegrationRule((s.integral == "trap" || s.crule == "trap") ? FullDiscretizationGridBase::CostIntegrationRule::TrapezoidalRule : FullDiscretizationGridBase::CostIntegrationRule::LeftSum);
        b.grid->setFiniteDifferencesCollocationMethod(makeCollocation(s.collocation));
        if (s.xf_fixed >= 0)
        {
            Eigen::Matrix<bool, -1, 1> fixed(s.nx);
            for (int i = 0; i < s.nx; ++i) fixed[i] = (s.xf_fixed >> i) & 1;
            b.grid->setXfFixed(fixed);
        }
        b.any_grid = b.grid;
    }

    b.dyn = dyn;
    b.ocp = std::make_shared<StructuredOptimalControlProblem>(b.any_grid, dyn, b.hg, b.solver);
    b.ocp->setStatisticsObject(b.stats);

    if (s.name == "unicycle" || s.name == "kcar")
    {
        Eigen::MatrixXd Q = Eigen::Vector3d(1, 1, 0.1).asDiagonal();
        Eigen::MatrixXd R = Eigen::Vector2d(0.1, 0.05).asDiagonal();
        if (s.qdiag.size() == 3) Q = s.qdiag.asDiagonal();
        if (s.rdiag.size() == 2) R = s.rdiag.asDiagonal();
        if (s.fullq) { Q = fullWeight(Q); R = fullWeight(R); }
        Eigen::MatrixXd Qf = 10.0 * Q;
        if (s.qfdiag.size() == 3) Qf = s.qfdiag.asDiagonal();
        s.Qfull = Q; s.Rfull = R; s.Qffull = Qf;
        b.ocp->setStageCost(std::make_shared<QuadraticFormCost>(Q, R, !s.integral.empty(), !s.nonlsq));
        b.ocp->setFinalStageCost(std::make_shared<QuadraticFinalStateCost>(Qf, !s.nonlsq));
        b.ocp->setBounds(Eigen::Vector3d::Constant(-10), Eigen::Vector3d::Constant(10), Eigen::Vector2d::Constant(-1), Eigen::Vector2d::Constant(1));
    }
    else if (s.name == "vdp")
    {
        Eigen::MatrixXd Q = Eigen::Vector2d(1, 1).asDiagonal();
        Eigen::MatrixXd R = Eigen::MatrixXd::Constant(1, 1, 0.1);
        if (s.fullq) Q = fullWeight(Q);
        Eigen::MatrixXd Qf = 10.0 * Q;
        s.Qfull = Q; s.Rfull = R; s.Qffull = Qf;
        b.ocp->setStageCost(std::make_shared<QuadraticFormCost>(Q, R, !s.integral.empty(), !s.nonlsq));
        b.ocp->setFinalStageCost(std::make_shared<QuadraticFinalStateCost>(Qf, !s.nonlsq));
        b.ocp->setControlBounds(Eigen::VectorXd::Constant(1, -1), Eigen::VectorXd::Constant(1, 1));
    }
    else if (s.name == "par2" || s.name == "par3" || s.name == "lin")   // Q = diag(1, 0.5, 0.2, 0.1), R = diag(0.1, 0.2, 0.05), Qf = 10 Q, |u_i| <= 1.5
    {
        Eigen::VectorXd q(s.nx), r(s.nu);
        const double qv[4] = {1.0, 0.5, 0.2, 0.1}, rv[3] = {0.1, 0.2, 0.05};
        for (int i = 0; i < s.nx; ++i) q[i] = qv[i];
        for (int i = 0; i < s.nu; ++i) r[i] = rv[i];
        Eigen::MatrixXd Q = q.asDiagonal(), R = r.asDiagonal();
        if (s.fullq) { Q = fullWeight(Q); R = fullWeight(R); }
        Eigen::MatrixXd Qf = 10.0 * Q;
        s.Qfull = Q; s.Rfull = R; s.Qffull = Qf;
        b.ocp->setStageCost(std::make_shared<QuadraticFormCost>(Q, R, false, true));
        b.ocp->setFinalStageCost(std::make_shared<QuadraticFinalStateCost>(Qf, true));
        b.ocp->setControlBounds(Eigen::VectorXd::Constant(s.nu, -1.5), Eigen::VectorXd::Constant(s.nu, 1.5));
    }
    else if (isZoo(s.name))   // Q = diag(1, 0.5[, 0.2]), R = 0.1, Qf = 10 Q, |u| <= 1.5
    {
        Eigen::VectorXd q(s.nx);
        const double qv[4] = {1.0, 0.5, 0.2, 0.1};
        for (int i = 0; i < s.nx; ++i) q[i] = qv[i];
        Eigen::MatrixXd Q = q.asDiagonal();
        Eigen::MatrixXd R = Eigen::MatrixXd::Constant(1, 1, 0.1);
        if (s.fullq) Q = fullWeight(Q);
        Eigen::MatrixXd Qf = 10.0 * Q;
        s.Qfull = Q; s.Rfull = R; s.Qffull = Qf;
        b.ocp->setStageCost(std::make_shared<QuadraticFormCost>(Q, R, false, true));
        b.ocp->setFinalStageCost(std::make_shared<QuadraticFinalStateCost>(Qf, true));
        b.ocp->setControlBounds(Eigen::VectorXd::Constant(1, -1.5), Eigen::VectorXd::Constant(1, 1.5));
    }
    else if (s.name == "int3")
    {
        if (s.vargrid) b.ocp->setStageCost(std::make_shared<MinimumTime>(!s.nonlsq));
        else
        {
            Eigen::MatrixXd Q = Eigen::Vector3d(1, 0.5, 0.1).asDiagonal();
            Eigen::MatrixXd R = Eigen::MatrixXd::Constant(1, 1, 0.1);
            Eigen::MatrixXd Qf = 10.0 * Q;
            b.ocp->setStageCost(std::make_shared<QuadraticFormCost>(Q, R, false, true));
            b.ocp->setFinalStageCost(std::make_shared<QuadraticFinalStateCost>(Qf, true));
        }
        b.ocp->setControlBounds(Eigen::VectorXd::Constant(1, -1), Eigen::VectorXd::Constant(1, 1));
    }
    else if (s.name == "dint")
    {
        b.ocp->setStageCost(std::make_shared<MinimumTime>(!s.nonlsq));
        b.ocp->setControlBounds(Eigen::VectorXd::Constant(1, -1), Eigen::VectorXd::Constant(1, 1));
    }
    else if (s.name == "pquad")
    {
        Eigen::VectorXd q(6), r(2);
        q << 1, 1, 0.5, 0.1, 0.1, 0.05;
        r << 0.02, 0.02;
        Eigen::MatrixXd Q = q.asDiagonal(), R = r.asDiagonal();
        Eigen::MatrixXd Qf = 10.0 * Q;
        if (s.vargrid) b.ocp->setStageCost(std::make_shared<MinimumTime>(true));
        else
        {   // (lsq=0 integral=..: the cost forms of the Hessian path -- on the shooting grid the intervals' edges become MultipleShootingEdgeSingleControl)
            b.ocp->setStageCost(std::make_shared<QuadraticFormCost>(Q, R, !s.integral.empty(), !s.nonlsq));
            b.ocp->setFinalStageCost(std::make_shared<QuadraticFinalStateCost>(Qf, !s.nonlsq));
        }
        b.ocp->setControlBounds(Eigen::Vector2d(0, 0), Eigen::Vector2d(12, 12));
        if (!s.noball) b.ocp->setStageInequalityConstraint(std::make_shared<BallKeepOut>(1.0, 0.5, 0.0, 0.3));
    }
    else if (s.name == "quad")
    {
        Eigen::VectorXd q(12), r(4);
        q << 1, 1, 1, 0.1, 0.1, 0.1, 0.5, 0.5, 0.5, 0.05, 0.05, 0.05;
        r << 0.01, 0.1, 0.1, 0.1;
        Eigen::MatrixXd Q = q.asDiagonal(), R = r.asDiagonal();
        Eigen::MatrixXd Qf = 10.0 * Q;
        if (s.vargrid) b.ocp->setStageCost(std::make_shared<MinimumTime>(true));
        else
        {
            b.ocp->setStageCost(std::make_shared<QuadraticFormCost>(Q, R, !s.integral.empty(), !s.nonlsq));
            b.ocp->setFinalStageCost(std::make_shared<QuadraticFinalStateCost>(Qf, !s.nonlsq));
        }
        Eigen::VectorXd ulb(4), uub(4);
        ulb << 0, -1, -1, -1;
        uub << 20, 1, 1, 1;
        b.ocp->setControlBounds(ulb, uub);
        if (!s.noball) b.ocp->setStageInequalityConstraint(std::make_shared<BallKeepOut>(1.0, 0.5, 0.6, 0.4));
    }
    if (s.xlb.size() > 0 || s.ulb.size() > 0)
    {
        Eigen::VectorXd xl = s.xlb.size() ? s.xlb : Eigen::VectorXd::Constant(s.nx, -CORBO_INF_DBL);
        Eigen::VectorXd xu = s.xub.size() ? s.xub : Eigen::VectorXd::Constant(s.nx, CORBO_INF_DBL);
        Eigen::VectorXd ul = s.ulb.size() ? s.ulb : Eigen::VectorXd::Constant(s.nu, -CORBO_INF_DBL);
        Eigen::VectorXd uu = s.uub.size() ? s.uub : Eigen::VectorXd::Constant(s.nu, CORBO_INF_DBL);
        b.ocp->setBounds(xl, xu, ul, uu);
    }
    if (!s.cost.empty())
    {
        Eigen::VectorXd q(s.nx), r(s.nu);
        const double qv[4] = {1.0, 0.5, 0.2, 0.1}, rv[3] = {0.1, 0.2, 0.05};
        for (int i = 0; i < s.nx; ++i) q[i] = qv[i % 4];
        for (int i = 0; i < s.nu; ++i) r[i] = rv[i % 3];
        Eigen::MatrixXd Q = q.asDiagonal(), R = r.asDiagonal();
        if (s.cost == "mtq" && s.last_n > 0) b.ocp->setStageCost(std::make_shared<MinTimeQuadraticLastN>(Q, R, s.last_n, !s.integral.empty(), !s.nonlsq));
        else if (s.cost == "mtq") b.ocp->setStageCost(std::make_shared<MinTimeQuadratic>(Q, R, !s.integral.empty(), !s.nonlsq));
        else if (s.cost == "qstate") b.ocp->setStageCost(std::make_shared<QuadraticStateCost>(Q, false, true));
        else if (s.cost == "qctrl") b.ocp->setStageCost(std::make_shared<QuadraticControlCost>(R, false, true));
        else if (s.cost == "mtqs") b.ocp->setStageCost(std::make_shared<MinTimeQuadraticStates>(Q, false, true));
        else if (s.cost == "mtqc") b.ocp->setStageCost(std::make_shared<MinTimeQuadraticControls>(R, false, true));
        else { fprintf(stderr, "unknown cost=%s\n", s.cost.c_str()); exit(2); }
    }
    if (s.final_cost == 0) b.ocp->setFinalStageCost({});
    if (s.ball.size() == 4 && s.name != "quad" && s.name != "pquad" && !s.ball_integral && s.rate.size() == 0)
        b.ocp->setStageInequalityConstraint(std::make_shared<BallKeepOut>(s.ball[0], s.ball[1], s.ball[2], s.ball[3]));
    else if (s.ball_integral || s.rate.size() > 0 || s.tilt > 0 || s.unorm > 0)
    {
        auto c = std::make_shared<UserStageInequalities>();
        c->ball = s.ball; c->ball_integral = s.ball_integral; c->rate = s.rate; c->tilt = s.tilt; c->unorm = s.unorm;
        b.ocp->setStageInequalityConstraint(c);
    }
    if (s.eq_lin.size() > 0)
    {
        if (s.eq_lin.size() != s.nx + s.nu + 1) { fprintf(stderr, "eq_lin needs nx + nu + 1 numbers\n"); exit(2); }
        auto c = std::make_shared<LinearIntegralEquality>();
        c->a = s.eq_lin.head(s.nx); c->b = s.eq_lin.segment(s.nx, s.nu); c->c = s.eq_lin[s.nx + s.nu];
        b.ocp->setStageEqualityConstraint(c);
    }
    if (s.u_prev.size() > 0 || s.u_prev_dt > 0)
        b.ocp->setPreviousControlInput(s.u_prev.size() > 0 ? s.u_prev : Eigen::VectorXd(Eigen::VectorXd::Zero(s.nu)), s.u_prev_dt > 0 ? s.u_prev_dt : s.dt);
    if (s.teq && s.teq_mask)
    {
        Eigen::Matrix<bool, -1, 1> active(s.nx);
        for (int i = 0; i < s.nx; ++i) active[i] = (s.teq_mask >> i) & 1;
        auto c = std::make_shared<TerminalPartialEqualityConstraint>();
        c->setXRef(s.xf, active);
        b.ocp->setFinalStageConstraint(c);
    }
    else if (s.teq) b.ocp->setFinalStageConstraint(std::make_shared<TerminalEqualityConstraint>(s.xf));
    if (s.tball_s.size() > 0)
    {
        Eigen::MatrixXd Sm = s.tball_s.asDiagonal();
        b.ocp->setFinalStageConstraint(std::make_shared<TerminalBall>(Sm, s.tball_gamma));
    }
    if (!b.ocp->initialize())
    {
        fprintf(stderr, "ocp initialize failed\n");
        exit(3);
    }
    if (s.xref_traj)
    {
        auto ts = std::make_shared<TimeSeries>();
        ts->setValueDimension(s.nx);
        for (int k = 0; k < s.N; ++k)
        {
            Eigen::VectorXd r = s.xf;
            const double fade = double(s.N - 1 - k) / double(s.N - 1);   // the last sample is xf itself
            for (int i = 0; i < s.nx; ++i) r[i] += 0.3 * fade * std::sin(0.37 * k + 1.1 * i);
            ts->add(k * s.dt, r);
        }
        b.xref = std::make_shared<DiscreteTimeReferenceTrajectory>(ts, TimeSeries::Interpolation::ZeroOrderHold);
    }
    else
        b.xref = std::make_shared<StaticReference>(s.xf);
    if (s.uref.size() == s.nu) b.uref = std::make_shared<StaticReference>(s.uref);
    else b.uref = std::make_shared<ZeroReference>(s.nu);
    return b;
}

static bool run(Built& b, const Scenario& s, int solves, double* solve_seconds = nullptr)
{
    bool ok  = true;
    double t = 0;
    for (int i = 0; i < solves; ++i)
    {
        ok = b.ocp->compute(s.x0, *b.xref, *b.uref, nullptr, Time(0), i == 0) && ok;
        t += b.stats->solving_time.toSec();
    }
    if (solve_seconds) *solve_seconds = t;
    return ok;
}

// start=<file>: the OCP as it stands after a 0-iteration compute() (graph built, straight-line guess in the vertices), its active parameters
// overwritten with the file's values (OptimizationProblemInterface::setParameterVector); the LM iterations then run in a second compute()
// with new_run = false, which keeps the vertices (full_discretization_grid_base.cpp:83-108; the shooting grids need their warm-start flag for that, shooting_grid_base.cpp:85).
// Pins randomized-start cases (tests/test_gpu_fuzz.py seeds) to the reference itself.
static Built buildStarted(const Scenario& s, int iterations, bool* ok = nullptr)
{
    Built b = build(s, s.start.empty() ? iterations : 0);
    bool r  = b.ocp->compute(s.x0, *b.xref, *b.uref, nullptr, Time(0), true);
    if (!s.start.empty())
    {
        const int n = b.hg->getParameterDimension();
        Eigen::VectorXd p(n);
        FILE* f = fopen(s.start.c_str(), "r");
        if (!f) { fprintf(stderr, "cannot open %s\n", s.start.c_str()); exit(4); }
        for (int i = 0; i < n; ++i)
            if (fscanf(f, "%lf", &p[i]) != 1) { fprintf(stderr, "start: %d values expected\n", n); exit(4); }
        fclose(f);
        b.hg->setParameterVector(p);
        b.any_grid->setModified(true);   // the solver analyses the sparsity pattern inside its first iteration only (levenberg_marquardt_sparse.cpp:121): the 0-iteration
                                         // solve left that undone, so the second compute() has to see a new structure (edges re-created, vertices kept)
        if (b.ms_grid) b.ms_grid->setWarmStart(true);   // (shooting_grid_base.cpp:85-96: without it every update() re-initialises the sequences)
        if (iterations > 0)
        {
            b.solver->setIterations(iterations);
            r = b.ocp->compute(s.x0, *b.xref, *b.uref, nullptr, Time(0), false);
        }
    }
    if (ok) *ok = r;
    return b;
}

static void printVec(const char* key, const Eigen::VectorXd& v, bool comma = true)
{
    printf("\"%s\": [", key);
    for (int i = 0; i < v.size(); ++i) printf("%s%.17g", i ? ", " : "", v[i]);
    printf("]%s\n", comma ? "," : "");
}

// all vertex values in grid order  x_0 u_0 | x_1 u_1 | ... | x_f | (dt)
static Eigen::VectorXd vertexValues(Built& b, const Scenario& s)
{
    auto xs = std::make_shared<TimeSeries>();
    auto us = std::make_shared<TimeSeries>();
    b.ocp->getTimeSeries(xs, us);
    int n = b.any_grid->getN();
    std::vector<double> out;
    for (int k = 0; k < n - 1; ++k)
    {
        for (int i = 0; i < s.nx; ++i) out.push_back(xs->getValuesMatrixView()(i, k));
        for (int i = 0; i < s.nu; ++i) out.push_back(us->getValuesMatrixView()(i, k));
    }
    for (int i = 0; i < s.nx; ++i) out.push_back(xs->getValuesMatrixView()(i, n - 1));
    out.push_back(b.any_grid->getFirstDt());
    return Eigen::Map<Eigen::VectorXd>(out.data(), out.size());
}


// the references the cost terms actually used (ReferenceTrajectoryInterface::getReferenceCached(k) after the grid's precompute):
// "ref_vertex" in the vertex layout x_0 u_0 | x_1 u_1 | ... | x_f | (dt -> 0)
static void printReferences(Built& b, const Scenario& s)
{
    if (!s.xref_traj && s.uref.size() == 0) return;
    const int n = b.any_grid->getN();
    std::vector<double> out;
    for (int k = 0; k < n - 1; ++k)
    {
        const auto& xr = b.xref->getReferenceCached(k);
        for (int i = 0; i < s.nx; ++i) out.push_back(xr[i]);
        const auto& ur = b.uref->getReferenceCached(k);
        for (int i = 0; i < s.nu; ++i) out.push_back(ur[i]);
    }
    const auto& xr = b.xref->getReferenceCached(n - 1);
    for (int i = 0; i < s.nx; ++i) out.push_back(xr[i]);
    out.push_back(0.0);
    printVec("ref_vertex", Eigen::Map<Eigen::VectorXd>(out.data(), out.size()));
}

static Scenario parse(int argc, char** argv, std::map<std::string, std::string>& kv)
{
    for (int i = 2; i < argc; ++i)
    {
        std::string a(argv[i]);
        size_t p = a.find('=');
        if (p == std::string::npos) continue;
        kv[a.substr(0, p)] = a.substr(p + 1);
    }
    Scenario s;
    s.name = kv.count("scenario") ? kv["scenario"] : "unicycle";
    auto vec = [](const std::string& str) {
        std::vector<double> v;
        std::stringstream ss(str);
        std::string item;
        while (std::getline(ss, item, ',')) v.push_back(strtod(item.c_str(), nullptr));
        return Eigen::VectorXd(Eigen::Map<Eigen::VectorXd>(v.data(), v.size()));
    };
    if (s.name == "unicycle" || s.name == "kcar")
    {
        s.nx = 3; s.nu = 2; s.N = 100; s.dt = 0.1;
        s.w_eq = s.w_ineq = s.w_b = 10;
        s.x0 = Eigen::Vector3d(0, 0, 0);
        s.xf = Eigen::Vector3d(2, 1, 0.5);
    }
    else if (s.name == "vdp")
    {
        s.nx = 2; s.nu = 1; s.N = 20; s.dt = 0.1;
        s.w_eq = s.w_ineq = s.w_b = 2;
        s.x0 = Eigen::Vector2d(1, 0);
        s.xf = Eigen::Vector2d(0, 0);
    }
    else if (s.name == "lin")
    {
        s.nx = kv.count("nx") ? atoi(kv["nx"].c_str()) : 2;
        s.nu = kv.count("nu") ? atoi(kv["nu"].c_str()) : 1;
        s.N = 24; s.dt = 0.1;
        s.w_eq = s.w_ineq = s.w_b = 5;
        s.x0 = Eigen::VectorXd::Zero(s.nx);
        s.xf = Eigen::VectorXd::Constant(s.nx, 0.5);
        s.lin_a = vec(kv["lin_a"]);
        s.lin_b = vec(kv["lin_b"]);
        if (s.lin_a.size() != s.nx * s.nx || s.lin_b.size() != s.nx * s.nu) { fprintf(stderr, "lin: lin_a / lin_b sizes\n"); exit(1); }
    }
    else if (s.name == "par2" || s.name == "par3")
    {
        s.nx = s.nu = (s.name == "par2") ? 2 : 3; s.N = 24; s.dt = 0.1;
        s.w_eq = s.w_ineq = s.w_b = 5;
        s.x0 = Eigen::VectorXd::Zero(s.nx);
        s.xf = (s.nx == 2) ? Eigen::VectorXd(Eigen::Vector2d(1.0, -0.5)) : Eigen::VectorXd(Eigen::Vector3d(1.0, -0.5, 0.3));
    }
    else if (isZoo(s.name))
    {
        s.nx = (s.name == "rocket") ? 3 : (s.name == "cartpole") ? 4 : 2; s.nu = 1; s.N = 24; s.dt = 0.1;
        s.w_eq = s.w_ineq = s.w_b = 5;
        if (s.name == "rocket") { s.x0 = Eigen::Vector3d(0, 0, 1); s.xf = Eigen::Vector3d(0.6, 0, 0.98); }   // position, speed, mass
        else if (s.name == "cartpole") { s.x0 = Eigen::Vector4d(0, 0.4, 0, 0); s.xf = Eigen::Vector4d(0.5, 0, 0, 0); }   // [x phi xdot phidot]
        else if (s.name == "pendulum" || s.name == "mpendulum") { s.x0 = Eigen::Vector2d(0.8, 0); s.xf = Eigen::Vector2d(0, 0); }
        else if (s.name == "artstein") { s.x0 = Eigen::Vector2d(0.6, 0.4); s.xf = Eigen::Vector2d(0.1, 0); }
        else { s.x0 = Eigen::Vector2d(0.8, -0.2); s.xf = Eigen::Vector2d(0, 0); }
    }
    else if (s.name == "dint")
    {
        s.nx = 2; s.nu = 1; s.N = 50; s.dt = 0.1;
        s.w_eq = s.w_ineq = s.w_b = 100;
        s.x0 = Eigen::Vector2d(0, 0);
        s.xf = Eigen::Vector2d(1, 0);
        s.solves = 5;
    }
    else if (s.name == "int3")   // SerialIntegratorSystem(3): fixed grid + quadratic cost, or (vargrid=1) time-optimal like cfg 2
    {
        s.nx = 3; s.nu = 1; s.N = 30; s.dt = 0.1;
        s.w_eq = s.w_ineq = s.w_b = 10;
        s.x0 = Eigen::Vector3d(0, 0, 0);
        s.xf = Eigen::Vector3d(1, 0, 0);
    }
    else if (s.name == "pquad")
    {
        s.nx = 6; s.nu = 2; s.N = 20; s.dt = 0.05;
        s.w_eq = s.w_ineq = s.w_b = 10;
        s.x0 = Eigen::VectorXd::Zero(6);
        s.xf = Eigen::VectorXd::Zero(6);
        s.xf[0] = 2; s.xf[1] = 1;
    }
    else if (s.name == "quad")
    {
        s.nx = 12; s.nu = 4; s.N = 20; s.dt = 0.05;
        s.w_eq = s.w_ineq = s.w_b = 10;
        s.x0 = Eigen::VectorXd::Zero(12);
        s.xf = Eigen::VectorXd::Zero(12);
        s.xf[0] = 2; s.xf[1] = 1; s.xf[2] = 1;
    }
    if (kv.count("N")) s.N = atoi(kv["N"].c_str());
    if (kv.count("dt")) s.dt = atof(kv["dt"].c_str());
    if (kv.count("iters")) s.iters = atoi(kv["iters"].c_str());
    if (kv.count("solves")) s.solves = atoi(kv["solves"].c_str());
    if (kv.count("w"))
    {
        Eigen::VectorXd w = vec(kv["w"]);
        s.w_eq = w[0]; s.w_ineq = w[1]; s.w_b = w[2];
    }
    if (kv.count("x0")) s.x0 = vec(kv["x0"]);
    if (kv.count("xf")) s.xf = vec(kv["xf"]);
    if (kv.count("collocation")) s.collocation = kv["collocation"];
    if (kv.count("grid")) { s.ms = (kv["grid"] == "ms"); s.fd = (kv["grid"] == "fd"); }
    auto bvec = [&](const std::string& str) {   // like vec(), "inf" / "-inf" = +-CORBO_INF_DBL
        std::vector<double> v;
        std::stringstream ss(str);
        std::string item;
        while (std::getline(ss, item, ',')) v.push_back(item == "inf" ? CORBO_INF_DBL : item == "-inf" ? -CORBO_INF_DBL : strtod(item.c_str(), nullptr));
        return Eigen::VectorXd(Eigen::Map<Eigen::VectorXd>(v.data(), v.size()));
    };
    if (kv.count("xlb")) s.xlb = bvec(kv["xlb"]);
    if (kv.count("xub")) s.xub = bvec(kv["xub"]);
    if (kv.count("ulb")) s.ulb = bvec(kv["ulb"]);
    if (kv.count("uub")) s.uub = bvec(kv["uub"]);
    if (kv.count("xf_fixed")) s.xf_fixed = atoi(kv["xf_fixed"].c_str());
    if (kv.count("final_cost")) s.final_cost = atoi(kv["final_cost"].c_str());
    if (kv.count("ball")) s.ball = vec(kv["ball"]);
    if (kv.count("crule")) s.crule = kv["crule"];
    if (kv.count("ball_int")) s.ball_integral = atoi(kv["ball_int"].c_str()) != 0;
    if (kv.count("noball")) s.noball = atoi(kv["noball"].c_str()) != 0;
    if (kv.count("eq_lin")) s.eq_lin = vec(kv["eq_lin"]);
    if (kv.count("rate")) s.rate = vec(kv["rate"]);
    if (kv.count("tilt")) s.tilt = atof(kv["tilt"].c_str());
    if (kv.count("unorm")) s.unorm = atof(kv["unorm"].c_str());
    if (kv.count("u_prev")) s.u_prev = vec(kv["u_prev"]);
    if (kv.count("u_prev_dt")) s.u_prev_dt = atof(kv["u_prev_dt"].c_str());
    if (kv.count("teq")) s.teq = atoi(kv["teq"].c_str()) != 0;
    if (kv.count("teq_mask")) s.teq_mask = atoi(kv["teq_mask"].c_str());
    if (kv.count("xref_traj")) s.xref_traj = atoi(kv["xref_traj"].c_str()) != 0;
    if (kv.count("uref")) s.uref = vec(kv["uref"]);
    if (kv.count("qdiag")) s.qdiag = vec(kv["qdiag"]);
    if (kv.count("rdiag")) s.rdiag = vec(kv["rdiag"]);
    if (kv.count("qfdiag")) s.qfdiag = vec(kv["qfdiag"]);
    if (kv.count("start")) s.start = kv["start"];
    if (kv.count("vargrid")) s.vargrid = atoi(kv["vargrid"].c_str()) != 0;
    if (kv.count("cost")) s.cost = kv["cost"];
    if (kv.count("last_n")) s.last_n = atoi(kv["last_n"].c_str());
    if (kv.count("lsq")) s.nonlsq = atoi(kv["lsq"].c_str()) == 0;
    if (kv.count("fullq")) s.fullq = atoi(kv["fullq"].c_str()) != 0;
    if (kv.count("ms_integrator")) s.ms_integrator = kv["ms_integrator"];
    if (kv.count("integral")) s.integral = kv["integral"];
    if (kv.count("adapt")) s.adapt = kv["adapt"];
    if (kv.count("nmax")) s.n_max = atoi(kv["nmax"].c_str());
    if (kv.count("nmin")) s.n_min = atoi(kv["nmin"].c_str());
    if (kv.count("hyst")) s.hyst = strtod(kv["hyst"].c_str(), nullptr);
    if (kv.count("adapt_first")) s.adapt_first = atoi(kv["adapt_first"].c_str()) != 0;
    if (kv.count("tball"))
    {
        s.tball_gamma = atof(kv["tball"].c_str());
        s.tball_s     = kv.count("tball_s") ? vec(kv["tball_s"]) : Eigen::VectorXd::Ones(s.nx);
    }
    return s;
}

// the upper Cholesky factor exactly as QuadraticFormCost::setWeightQ keeps it (quadratic_cost.cpp:52-54), row-major
static void printFactor(const char* name, const Eigen::MatrixXd& W)
{
    Eigen::LLT<Eigen::MatrixXd, Eigen::Upper> chol(W);
    const Eigen::MatrixXd U = chol.matrixU();
    printf("\"%s\": [", name);
    for (int i = 0; i < U.rows(); ++i)
        for (int j = 0; j < U.cols(); ++j) printf("%s%.17g", (i + j) ? ", " : "", U(i, j));
    printf("],\n");
}

static int dump(const Scenario& s)
{
    printf("{\n\"scenario\": \"%s\", \"nx\": %d, \"nu\": %d, \"N\": %d, \"dt\": %.17g, \"iters\": %d, \"solves\": %d,\n", s.name.c_str(), s.nx, s.nu,
           s.N, s.dt, s.iters, s.solves);
    printf("\"collocation\": \"%s\", \"weights\": [%.17g, %.17g, %.17g],\n", s.collocation.c_str(), s.w_eq, s.w_ineq, s.w_b);
    if (s.ms) printf("\"grid\": \"ms\",\n");
    if (s.fd) printf("\"grid\": \"fd\",\n");
    if (!s.ms_integrator.empty()) printf("\"ms_integrator\": \"%s\",\n", s.ms_integrator.c_str());
    if (s.lin_a.size()) { printVec("lin_a", s.lin_a); printVec("lin_b", s.lin_b); }
    if (s.ball.size() == 4) printVec("ball", s.ball);
    if (!s.crule.empty()) printf("\"crule\": \"%s\",\n", s.crule.c_str());
    if (s.ball_integral) printf("\"ball_int\": 1,\n");
    if (s.eq_lin.size()) printVec("eq_lin", s.eq_lin);
    if (s.rate.size()) printVec("rate", s.rate);
    if (s.tilt > 0) printf("\"tilt\": %.17g,\n", s.tilt);
    if (s.unorm > 0) printf("\"unorm\": %.17g,\n", s.unorm);
    if (s.u_prev.size()) printVec("u_prev", s.u_prev);
    if (s.u_prev_dt > 0) printf("\"u_prev_dt\": %.17g,\n", s.u_prev_dt);
    if (s.noball) printf("\"noball\": 1,\n");
    if (s.teq) printf("\"teq\": 1,\n");
    if (s.teq && s.teq_mask) printf("\"teq_mask\": %d,\n", s.teq_mask);
    if (s.vargrid) printf("\"vargrid\": 1,\n");
    if (!s.cost.empty()) printf("\"cost\": \"%s\",\n", s.cost.c_str());
    if (s.last_n > 0) printf("\"last_n\": %d,\n", s.last_n);
    if (s.nonlsq) printf("\"lsq\": 0,\n");
    if (!s.integral.empty()) printf("\"integral\": \"%s\",\n", s.integral.c_str());
    if (s.xlb.size()) printVec("xlb", s.xlb);
    if (s.xub.size()) printVec("xub", s.xub);
    if (s.ulb.size()) printVec("ulb", s.ulb);
    if (s.uub.size()) printVec("uub", s.uub);
    if (s.qdiag.size()) printVec("qdiag", s.qdiag);
    if (s.rdiag.size()) printVec("rdiag", s.rdiag);
    if (s.qfdiag.size()) printVec("qfdiag", s.qfdiag);
    if (!s.start.empty()) printf("\"start\": 1,\n");
    if (s.xf_fixed >= 0) printf("\"xf_fixed\": %d,\n", s.xf_fixed);
    if (s.final_cost >= 0) printf("\"final_cost\": %d,\n", s.final_cost);
    printVec("x0", s.x0);
    printVec("xf", s.xf);
    if (s.tball_s.size() > 0)
    {
        printf("\"tball_gamma\": %.17g, ", s.tball_gamma);
        printVec("tball_s", s.tball_s);
    }

    // ---- hot-path pieces at the initial point: LM with 0 iterations builds the graph, evaluates once and returns
    {
        Built b = buildStarted(s, 0);
        auto& hg = *b.hg;
        int n = hg.getParameterDimension(), lsq = hg.getLsqObjectiveDimension(), eq = hg.getEqualityDimension(), ineq = hg.getInequalityDimension(),
            nb = hg.finiteCombinedBoundsDimension();
        int m = lsq + eq + ineq + nb;
        printf("\"n\": %d, \"lsq\": %d, \"eq\": %d, \"ineq\": %d, \"bounds\": %d, \"m\": %d,\n", n, lsq, eq, ineq, nb, m);
        Eigen::VectorXd p(n), lb(n), ub(n);
        hg.getParameterVector(p);
        hg.getBounds(lb, ub);
        printVec("param_init", p);
        printVec("param_lb", lb);
        printVec("param_ub", ub);
        printVec("vertex_init", vertexValues(b, s));
        printReferences(b, s);
        if (s.fullq)
        {
            printf("\"fullq\": 1,\n");
            printFactor("q_sqrt", s.Qfull);
            if (s.nu > 1) printFactor("r_sqrt", s.Rfull);
            printFactor("qf_sqrt", s.Qffull);
        }
        // stacked residual exactly as LevenbergMarquardtSparse::computeValues (levenberg_marquardt_sparse.cpp:222-246)
        Eigen::VectorXd values(m);
        if (lsq) hg.computeValuesLsqObjective(values.segment(0, lsq));
        if (eq)
        {
            hg.computeValuesEquality(values.segment(lsq, eq));
            values.segment(lsq, eq) *= s.w_eq;
        }
        if (ineq) hg.computeValuesActiveInequality(values.segment(lsq + eq, ineq), s.w_ineq);
        if (nb)
        {
            hg.computeDistanceFiniteCombinedBounds(values.segment(lsq + eq + ineq, nb));
            values.segment(lsq + eq + ineq, nb) *= s.w_b;
        }
        printVec("values_init", values);
        Eigen::SparseMatrix<double> J(m, n);
        hg.computeCombinedSparseJacobian(J, true, true, true, true, true, s.w_eq, s.w_ineq, s.w_b, &values);
        std::vector<int> rows, cols;
        std::vector<double> vals;
        for (int k = 0; k < J.outerSize(); ++k)
            for (Eigen::SparseMatrix<double>::InnerIterator it(J, k); it; ++it)
            {
                rows.push_back(it.row());
                cols.push_back(it.col());
                vals.push_back(it.value());
            }
        printf("\"nnz\": %d,\n\"jac_rows\": [", (int)rows.size());
        for (size_t i = 0; i < rows.size(); ++i) printf("%s%d", i ? ", " : "", rows[i]);
        printf("],\n\"jac_cols\": [");
        for (size_t i = 0; i < cols.size(); ++i) printf("%s%d", i ? ", " : "", cols[i]);
        printf("],\n");
        printVec("jac_vals", Eigen::Map<Eigen::VectorXd>(vals.data(), vals.size()));
        // H = J^T J diag and rhs (levenberg_marquardt_sparse.cpp:97-100)
        Eigen::SparseMatrix<double> H = J.transpose() * J;
        Eigen::VectorXd rhs           = J.transpose() * -values;
        printVec("H_diag_init", H.diagonal());
        printVec("rhs_init", rhs);
        printf("\"nnz_H\": %d,\n", (int)H.nonZeros());
    }

    // ---- trajectory after k LM iterations (fresh OCP each time), last solve of `solves`
    printf("\"after_iter\": [\n");
    for (int k = 1; k <= s.iters; ++k)
    {
        bool ok = true;
        Built b = s.start.empty() ? build(s, k) : buildStarted(s, k, &ok);
        if (s.start.empty()) ok = run(b, s, s.solves);
        printf("{\"k\": %d, \"ok\": %d, \"chi2\": %.17g, ", k, ok ? 1 : 0, b.ocp->getCurrentObjectiveValue());
        printVec("vertex", vertexValues(b, s), false);
        printf("}%s\n", k < s.iters ? "," : "");
    }
    printf("]\n}\n");
    return 0;
}

static int bench(const Scenario& s0, std::map<std::string, std::string>& kv)
{
    int batch         = kv.count("batch") ? atoi(kv["batch"].c_str()) : 16;
    unsigned long seed = kv.count("seed") ? strtoul(kv["seed"].c_str(), nullptr, 10) : 20260928UL;
    double total = 0, chi2_sum = 0;
    // instances=<file>: one instance per line, "x0[0..nx) xf[0..nx)" (the same seeded sample the GPU run uses)
    std::vector<std::vector<double>> inst;
    if (kv.count("instances"))
    {
        FILE* f = fopen(kv["instances"].c_str(), "r");
        if (!f)
        {
            fprintf(stderr, "cannot open %s\n", kv["instances"].c_str());
            return 4;
        }
        std::vector<double> row(2 * s0.nx);
        for (;;)
        {
            bool ok = true;
            for (int j = 0; j < 2 * s0.nx; ++j)
                if (fscanf(f, "%lf", &row[j]) != 1) ok = false;
            if (!ok) break;
            inst.push_back(row);
        }
        fclose(f);
        batch = (int)inst.size();
    }
    auto w0 = std::chrono::steady_clock::now();
    for (int i = 0; i < batch; ++i)
    {
        Scenario s = s0;
        std::mt19937_64 rng(seed + i);
        std::uniform_real_distribution<double> U(-1.0, 1.0);
        if (!inst.empty())
        {
            s.x0 = Eigen::Map<Eigen::VectorXd>(inst[i].data(), s.nx);
            s.xf = Eigen::Map<Eigen::VectorXd>(inst[i].data() + s.nx, s.nx);
        }
        else if (s.name == "unicycle")
        {
            s.x0 = Eigen::Vector3d(U(rng), U(rng), U(rng) * M_PI / 4);
            s.xf = Eigen::Vector3d(2 + 0.5 * U(rng), 1 + 0.5 * U(rng), 0.5 + 0.5 * U(rng));
        }
        Built b = build(s, s.iters);
        double t = 0;
        run(b, s, s.solves, &t);
        total += t;
        chi2_sum += b.ocp->getCurrentObjectiveValue();
    }
    double wall = std::chrono::duration<double>(std::chrono::steady_clock::now() - w0).count();
    printf("{\"scenario\": \"%s\", \"batch\": %d, \"iters\": %d, \"solves\": %d, \"solve_seconds\": %.9g, \"wall_seconds\": %.9g, "
           "\"iter_per_s\": %.9g, \"chi2_sum\": %.17g}\n",
           s0.name.c_str(), batch, s0.iters, s0.solves, total, wall, batch * (double)s0.iters * s0.solves / total, chi2_sum);
    return 0;
}

// Moving-horizon sequence: `steps` calls of StructuredOptimalControlProblem::compute with new_run = true on ONE OCP object; the
// measured state handed to step s+1 is x_1 of the solution of step s plus a deterministic disturbance.  shift=1 turns on the grid's
// moving-horizon warm start (FullDiscretizationGridBase::setWarmStart -> warmStartShifting, full_discretization_grid_base.cpp:96,
// 230-283); otherwise the previous solution stays in place and only x_0 is overwritten (:98-108).  Dumps x0 and the vertex values
// after every step (iters=0: the pure warm-started initial guess).
static int mpc(const Scenario& s, std::map<std::string, std::string>& kv)
{
    const int steps  = kv.count("steps") ? atoi(kv["steps"].c_str()) : 4;
    const bool shift = kv.count("shift") ? atoi(kv["shift"].c_str()) != 0 : true;
    const int iters0 = kv.count("iters0") ? atoi(kv["iters0"].c_str()) : 10;   // LM iterations of step 0 (builds the first trajectory)
    // ocp_iters=K: K compute() calls per step like PredictiveController::step (predictive_controller.cpp:46-80), new_run only for the
    // first -- the calls in which a variable grid adapts its resolution (adaptGrid skips new runs unless adapt_first)
    const int ocp_iters = kv.count("ocp_iters") ? atoi(kv["ocp_iters"].c_str()) : 1;
    Built b = build(s, iters0);
    if (shift && b.grid) b.grid->setWarmStart(true);
    // ShootingGridBase: the same shifting (shooting_grid_base.cpp:99-113,292-352).  Without warm start a shooting grid re-initialises its
    // sequences in every update (:85-96) -- an adapting grid keeps them (the variable grid never shifts: isMovingHorizonWarmStartActive() false)
    if ((shift || !s.adapt.empty()) && b.ms_grid) b.ms_grid->setWarmStart(true);
    printf("{\n\"scenario\": \"%s\", \"nx\": %d, \"nu\": %d, \"N\": %d, \"dt\": %.17g, \"iters0\": %d, \"iters\": %d, \"shift\": %d,\n", s.name.c_str(),
           s.nx, s.nu, s.N, s.dt, iters0, s.iters, shift ? 1 : 0);
    printf("\"ocp_iters\": %d, \"adapt\": \"%s\", \"nmax\": %d, \"nmin\": %d, \"hyst\": %.17g, \"adapt_first\": %d,\n", ocp_iters, s.adapt.c_str(), s.n_max, s.n_min,
           s.hyst, s.adapt_first ? 1 : 0);
    if (s.ms) printf("\"grid\": \"ms\",\n");
    if (s.fd) printf("\"grid\": \"fd\",\n");
    if (!s.ms_integrator.empty()) printf("\"ms_integrator\": \"%s\",\n", s.ms_integrator.c_str());
    printf("\"collocation\": \"%s\", \"weights\": [%.17g, %.17g, %.17g],\n", s.collocation.c_str(), s.w_eq, s.w_ineq, s.w_b);
    if (s.lin_a.size()) { printVec("lin_a", s.lin_a); printVec("lin_b", s.lin_b); }
    printVec("xf", s.xf);
    if (kv.count("plant_a")) printf("\"plant_a\": %.17g,\n", atof(kv["plant_a"].c_str()));
    printf("\"steps\": [\n");
    Eigen::VectorXd x0 = s.x0;
    for (int st = 0; st < steps; ++st)
    {
        if (st == 1) b.solver->setIterations(s.iters);
        bool ok = true;
        std::vector<int> n_seq;
        for (int it = 0; it < ocp_iters; ++it)
        {
            ok = b.ocp->compute(x0, *b.xref, *b.uref, nullptr, Time(st * s.dt), it == 0) && ok;
            n_seq.push_back(b.any_grid->getN());
        }
        Eigen::VectorXd v = vertexValues(b, s);
        printf("{\"ok\": %d, \"chi2\": %.17g, \"n\": %d, \"n_seq\": [", ok ? 1 : 0, b.ocp->getCurrentObjectiveValue(), b.any_grid->getN());
        for (size_t i = 0; i < n_seq.size(); ++i) printf("%s%d", i ? ", " : "", n_seq[i]);
        printf("], ");
        printVec("x0", x0);
        printVec("vertex", v, false);
        printf("}%s\n", st + 1 < steps ? "," : "");
        for (int i = 0; i < s.nx; ++i) x0[i] = v[s.nx + s.nu + i] + 0.01 * std::sin(1.0 + st + 0.5 * i);   // x_1 + disturbance
    }
    printf("]\n}\n");
    return 0;
}

// ---------------------------------------------------------------------------------------------------------------------
// loop: closed loop of the OCP with the reference's own plant, the step either side of the solve:
//   y = SimulatedPlant::output (full state)  ->  StructuredOptimalControlProblem::compute(y, ..., new_run = true)
//   ->  SimulatedPlant::control(u_sequence, x_sequence, dt, t)   (plants/src/simulated_plant.cpp:97-160: the first control of the
//       sequence is held over dt and integrated with the plant's integrator; then the state disturbance is applied)
// i.e. what task_closed_loop_control.cpp:153-235 does per step with a PredictiveController.  plant = the OCP's own dynamics,
// integrator=euler|rk4 (SimulatedPlant's default is explicit Euler), a deterministic state disturbance (an object of this file
// behind the reference's DisturbanceInterface).  Dumps per step the measured state, chi2, the vertex values after the solve and
// the plant state after control().
class DeterministicStateDisturbance : public DisturbanceInterface
{
 public:
    explicit DeterministicStateDisturbance(double dt = 0.1, double amplitude = 0.0) : _dt(dt), _amp(amplitude) {}
    Ptr getInstance() const override { return std::make_shared<DeterministicStateDisturbance>(); }
    void disturb(const Time& t, const Eigen::Ref<const Eigen::VectorXd>& values, Eigen::Ref<Eigen::VectorXd> disturbed_values) override
    {
        const int step = (int)std::lround(t.toSec() / _dt);
        last.resize(values.size());
        for (int i = 0; i < values.size(); ++i)
        {
            last[i]             = _amp * std::sin(1.0 + step + 0.5 * i);
            disturbed_values[i] = values[i] + last[i];
        }
    }
    Eigen::VectorXd last;   // what the last call added (recorded in the fixture: a checker need not share this host's libm)
    void reset() override {}

 private:
    double _dt, _amp;
};

static int loop(const Scenario& s, std::map<std::string, std::string>& kv)
{
    const int steps         = kv.count("steps") ? atoi(kv["steps"].c_str()) : 5;
    const bool shift        = kv.count("shift") ? atoi(kv["shift"].c_str()) != 0 : true;
    const std::string integ = kv.count("integrator") ? kv["integrator"] : "rk4";
    const double amp        = kv.count("disturbance") ? atof(kv["disturbance"].c_str()) : 0.01;
    Built b = build(s, s.iters);
    if (shift && b.grid) b.grid->setWarmStart(true);
    if (shift && b.ms_grid) b.ms_grid->setWarmStart(true);
    // plant_a=<a>: the plant is a Van der Pol oscillator with ANOTHER damping coefficient than the controller's model
    SystemDynamicsInterface::Ptr plant_dyn = b.dyn;
    if (kv.count("plant_a"))
    {
        auto pd = std::make_shared<VanDerPolOscillator>();
        pd->setDampingCoefficient(atof(kv["plant_a"].c_str()));
        plant_dyn = pd;
    }
    SimulatedPlant plant(plant_dyn, std::make_shared<FullStateSystemOutput>());
    if (integ == "rk4") plant.setIntegrator(std::make_shared<IntegratorExplicitRungeKutta4>());   // else: the default explicit Euler
    auto disturbance = std::make_shared<DeterministicStateDisturbance>(s.dt, amp);
    plant.setStateDisturbance(disturbance);
    plant.setInitialState(s.x0);
    plant.reset();
    printf("{\n\"scenario\": \"%s\", \"nx\": %d, \"nu\": %d, \"N\": %d, \"dt\": %.17g, \"iters\": %d, \"shift\": %d, \"integrator\": \"%s\", \"disturbance\": %.17g,\n",
           s.name.c_str(), s.nx, s.nu, s.N, s.dt, s.iters, shift ? 1 : 0, integ.c_str(), amp);
    printf("\"collocation\": \"%s\", \"weights\": [%.17g, %.17g, %.17g],\n", s.collocation.c_str(), s.w_eq, s.w_ineq, s.w_b);
    if (s.lin_a.size()) { printVec("lin_a", s.lin_a); printVec("lin_b", s.lin_b); }
    printVec("xf", s.xf);
    if (kv.count("plant_a")) printf("\"plant_a\": %.17g,\n", atof(kv["plant_a"].c_str()));
    printf("\"steps\": [\n");
    for (int st = 0; st < steps; ++st)
    {
        const Time t(st * s.dt);
        Eigen::VectorXd y(s.nx);
        plant.output(y, t);
        bool ok           = b.ocp->compute(y, *b.xref, *b.uref, nullptr, t, true);
        Eigen::VectorXd v = vertexValues(b, s);
        auto xs = std::make_shared<TimeSeries>();
        auto us = std::make_shared<TimeSeries>();
        b.ocp->getTimeSeries(xs, us);
        ok = plant.control(us, xs, Duration(s.dt), t) && ok;
        Eigen::VectorXd after(s.nx);
        plant.output(after, t);
        printf("{\"ok\": %d, \"chi2\": %.17g, ", ok ? 1 : 0, b.ocp->getCurrentObjectiveValue());
        printVec("x0", y);
        printReferences(b, s);   // (time-varying reference: what getReferenceCached handed out at this control step)
        printVec("vertex", v);
        {   // the interval the plant really integrates over: its time-stamped control buffer hands out (t + dt) - t, rounded
            // (systems/src/time_value_buffer.cpp:68-73), e.g. 0.10000000000000003 at t = 0.2
            const double ts = t.toSec();
            printf("\"plant_dt\": %.17g, ", (ts + Duration(s.dt).toSec()) - ts);
        }
        printVec("disturbance", disturbance->last);
        printVec("plant_after", after, false);
        printf("}%s\n", st + 1 < steps ? "," : "");
    }
    printf("]\n}\n");
    return 0;
}

// ---------------------------------------------------------------------------------------------------------------------
// kat: the known-answer cases of the reference's own solver test (optimization/test/test_levenberg_marquardt_sparse.cpp:71-371;
// written against an older class name, `StandardOptimizationProblemWithCallbacks` = today's SimpleOptimizationProblemWithCallbacks)
// re-run against the compiled reference.  The problem definitions below are this file's own restatement of what those tests set
// up (same functions, starts, bounds, weights, iteration counts, call order -- including the tests' "setParameterValue(0, ..)
// twice" slip); the output pins LevenbergMarquardtSparse::solve on problems that are not OCPs.

static void printIVec(const char* key, const Eigen::VectorXi& v, bool comma = true)
{
    printf("\"%s\": [", key);
    for (int i = 0; i < v.size(); ++i) printf("%s%d", i ? ", " : "", v[i]);
    printf("]%s\n", comma ? "," : "");
}

// hess: the operators of the exact-Hessian path at a generic point of the scenario's hypergraph -- what
// IpoptWrapper::eval_h hands to the interior-point solver (nlp_solver_ipopt_wrapper.cpp:249-271, lower_part_only = true) and the
// two-side-bounded linear form lbA <= A x <= ubA of the QP interface (hyper_graph_optimization_problem_edge_based.cpp:4762-4968).
static int hess(const Scenario& s)
{
    Built b = build(s, 0);
    run(b, s, 1);   // 0 iterations: graph built, evaluated once
    auto& hg = *b.hg;
    const int n = hg.getParameterDimension(), eq = hg.getEqualityDimension(), ineq = hg.getInequalityDimension();
    Eigen::VectorXd inc(n);
    for (int i = 0; i < n; ++i) inc[i] = 0.05 * std::sin(1.3 * i + 0.4);
    hg.applyIncrement(inc);   // away from the structured initial guess (u = 0, straight line)
    printf("{\n\"scenario\": \"%s\", \"nx\": %d, \"nu\": %d, \"N\": %d, \"dt\": %.17g,\n", s.name.c_str(), s.nx, s.nu, s.N, s.dt);
    printf("\"collocation\": \"%s\",\n", s.collocation.c_str());
    if (s.ms) printf("\"grid\": \"ms\",\n");
    if (s.fd) printf("\"grid\": \"fd\",\n");
    if (!s.ms_integrator.empty()) printf("\"ms_integrator\": \"%s\",\n", s.ms_integrator.c_str());
    if (s.ball.size() == 4) printVec("ball", s.ball);
    if (s.lin_a.size()) { printVec("lin_a", s.lin_a); printVec("lin_b", s.lin_b); }
    if (s.noball) printf("\"noball\": 1,\n");
    if (s.teq) printf("\"teq\": 1,\n");
    if (s.teq && s.teq_mask) printf("\"teq_mask\": %d,\n", s.teq_mask);
    if (s.vargrid) printf("\"vargrid\": 1,\n");
    if (!s.cost.empty()) printf("\"cost\": \"%s\",\n", s.cost.c_str());
    if (s.last_n > 0) printf("\"last_n\": %d,\n", s.last_n);
    if (s.nonlsq) printf("\"lsq\": 0,\n");
    if (!s.integral.empty()) printf("\"integral\": \"%s\",\n", s.integral.c_str());
    if (s.xf_fixed >= 0) printf("\"xf_fixed\": %d,\n", s.xf_fixed);
    if (s.final_cost >= 0) printf("\"final_cost\": %d,\n", s.final_cost);
    if (s.xlb.size()) printVec("xlb", s.xlb);
    if (s.xub.size()) printVec("xub", s.xub);
    if (s.ulb.size()) printVec("ulb", s.ulb);
    if (s.uub.size()) printVec("uub", s.uub);
    printVec("x0", s.x0);
    printVec("xf", s.xf);
    if (s.tball_s.size() > 0)
    {
        printf("\"tball_gamma\": %.17g, ", s.tball_gamma);
        printVec("tball_s", s.tball_s);
    }
    printf("\"n\": %d, \"eq\": %d, \"ineq\": %d, \"bounds\": %d,\n", n, eq, ineq, hg.finiteCombinedBoundsDimension());
    printVec("vertex_point", vertexValues(b, s));
    printReferences(b, s);
    if (s.fullq)
    {
        printf("\"fullq\": 1,\n");
        printFactor("q_sqrt", s.Qfull);
        if (s.nu > 1) printFactor("r_sqrt", s.Rfull);
        printFactor("qf_sqrt", s.Qffull);
    }
    Eigen::VectorXd meq(eq), mineq(ineq);
    for (int i = 0; i < eq; ++i) meq[i] = 0.5 + 0.25 * std::cos(0.7 * i);
    for (int i = 0; i < ineq; ++i) mineq[i] = 0.3 + 0.125 * (i % 5);
    const double mobj = 1.5;
    printf("\"mult_obj\": %.17g,\n", mobj);
    printVec("mult_eq", meq);
    printVec("mult_ineq", mineq);
    for (int lower = 0; lower < 2; ++lower)
    {
        int no = 0, ne = 0, ni = 0;
        hg.computeSparseHessiansNNZ(no, ne, ni, lower != 0);
        Eigen::VectorXi io(no), jo(no), ie(ne), je(ne), ii(ni), ji(ni);
        hg.computeSparseHessiansStructure(io, jo, ie, je, ii, ji, lower != 0);
        Eigen::VectorXd vo(no), ve(ne), vi(ni);
        hg.computeSparseHessiansValues(vo, ve, vi, mobj, meq.data(), ineq ? mineq.data() : nullptr, lower != 0);
        const char* tag = lower ? "lower" : "full";
        char key[64];
        auto K = [&](const char* base) { snprintf(key, sizeof(key), "%s_%s", base, tag); return key; };
        printIVec(K("hobj_rows"), io); printIVec(K("hobj_cols"), jo); printVec(K("hobj_vals"), vo);
        printIVec(K("heq_rows"), ie);  printIVec(K("heq_cols"), je);  printVec(K("heq_vals"), ve);
        printIVec(K("hineq_rows"), ii); printIVec(K("hineq_cols"), ji); printVec(K("hineq_vals"), vi);
    }
    {   // linear form
        const int nnz = hg.computeSparseJacobianTwoSideBoundedLinearFormNNZ(true);
        Eigen::VectorXi ir(nnz), jc(nnz);
        Eigen::VectorXd va(nnz);
        hg.computeSparseJacobianTwoSideBoundedLinearFormStructure(ir, jc, true);
        hg.computeSparseJacobianTwoSideBoundedLinearFormValues(va, true);
        const int ma = eq + ineq + hg.finiteCombinedBoundsDimension();
        Eigen::VectorXd lbA(ma), ubA(ma);
        hg.computeBoundsForTwoSideBoundedLinearForm(lbA, ubA, true);
        printIVec("lin_rows", ir); printIVec("lin_cols", jc); printVec("lin_vals", va);
        printVec("lin_lbA", lbA); printVec("lin_ubA", ubA);
    }
    {   // the other callbacks of IpoptWrapper (nlp_solver_ipopt_wrapper.cpp:128-230): eval_f, eval_grad_f, eval_g, eval_jac_g
        Eigen::VectorXd grad(n);
        hg.computeGradientObjective(grad);
        printVec("grad_obj", grad);
        printf("\"obj_value\": %.17g,\n", hg.computeValueObjective());
        Eigen::VectorXd g(eq + ineq);
        if (eq) hg.computeValuesEquality(g.head(eq));
        if (ineq) hg.computeValuesInequality(g.tail(ineq));
        printVec("g_values", g);
        const int nj = hg.computeCombinedSparseJacobiansNNZ(false, true, true);
        Eigen::VectorXi ir(nj), jc(nj);
        Eigen::VectorXd vj(nj);
        hg.computeCombinedSparseJacobiansStructure(ir, jc, false, true, true);
        hg.computeCombinedSparseJacobiansValues(vj, false, true, true);
        printIVec("jacg_rows", ir); printIVec("jacg_cols", jc); printVec("jacg_vals", vj);
    }
    printVec("vertex_after", vertexValues(b, s), false);   // (the in-place perturbations of the finite differences leave the point a few ulps off)
    printf("}\n");
    return 0;
}

static int g_kat_iter_cap = 0;   // kat cap=K: every phase runs at most K LM iterations (to compare iterate by iterate)

struct KatPhase
{
    std::vector<std::pair<int, double>> set_values;  // setParameterValue calls before the phase
    double w[3]     = {2, 2, 2};                     // levenberg_marquardt_sparse.h:126-128 defaults
    double adapt[6] = {1, 1, 1, 500, 500, 500};
    int iterations  = 100;                           // the fixture's SetUp
    int solves      = 1;                             // new_run only for the first
    bool set_weights = false, set_adapt = false, initialize = true;
};

static void katRun(const char* name, const char* fun, int n, int lsq, int eq, int ineq, const Eigen::VectorXd& x_start, const Eigen::VectorXd& lb,
                   const Eigen::VectorXd& ub, std::function<void(const Eigen::VectorXd&, Eigen::Ref<Eigen::VectorXd>)> f_obj,
                   std::function<void(const Eigen::VectorXd&, Eigen::Ref<Eigen::VectorXd>)> f_eq,
                   std::function<void(const Eigen::VectorXd&, Eigen::Ref<Eigen::VectorXd>)> f_ineq, const std::vector<KatPhase>& phases, bool last)
{
    SimpleOptimizationProblemWithCallbacks optim;
    LevenbergMarquardtSparse solver;
    solver.setIterations(100);
    optim.resizeParameterVector(n);
    if (x_start.size() == n) optim.setX(x_start);
    for (int i = 0; i < n; ++i)
    {
        if (lb.size() == n) optim.setLowerBound(i, lb[i]);
        if (ub.size() == n) optim.setUpperBound(i, ub[i]);
    }
    optim.setObjectiveFunction(f_obj, lsq, true);
    if (eq) optim.setEqualityConstraint(f_eq, eq);
    if (ineq) optim.setInequalityConstraint(f_ineq, ineq);
    printf("{\"name\": \"%s\", \"fun\": \"%s\", \"n\": %d, \"lsq\": %d, \"eq\": %d, \"ineq\": %d,\n", name, fun, n, lsq, eq, ineq);
    Eigen::VectorXd lbv(n), ubv(n);
    for (int i = 0; i < n; ++i) { lbv[i] = optim.getLowerBound(i); ubv[i] = optim.getUpperBound(i); }
    printVec("lb", lbv);
    printVec("ub", ubv);
    printf("\"phases\": [\n");
    for (size_t ph = 0; ph < phases.size(); ++ph)
    {
        const KatPhase& P = phases[ph];
        for (auto& sv : P.set_values) optim.setParameterValue(sv.first, sv.second);
        if (P.set_weights) solver.setPenaltyWeights(P.w[0], P.w[1], P.w[2]);
        if (P.set_adapt) solver.setWeightAdapation(P.adapt[0], P.adapt[1], P.adapt[2], P.adapt[3], P.adapt[4], P.adapt[5]);
        const int iters = (g_kat_iter_cap > 0 && g_kat_iter_cap < P.iterations) ? g_kat_iter_cap : P.iterations;
        solver.setIterations(iters);
        if (P.initialize) solver.initialize(&optim);
        printf("{");
        printVec("x_init", optim.getX());
        SolverStatus st = SolverStatus::Error;
        double obj      = -1;
        for (int i = 0; i < P.solves; ++i) st = solver.solve(optim, true, i == 0, &obj);
        printf("\"weights\": [%.17g, %.17g, %.17g], \"adapt\": [%.17g, %.17g, %.17g, %.17g, %.17g, %.17g], \"iterations\": %d, \"solves\": %d,\n", P.w[0],
               P.w[1], P.w[2], P.adapt[0], P.adapt[1], P.adapt[2], P.adapt[3], P.adapt[4], P.adapt[5], iters, P.solves);
        printf("\"status\": %d, \"chi2\": %.17g, ", (int)st, obj);
        printVec("x_final", optim.getX(), false);
        printf("}%s\n", ph + 1 < phases.size() ? "," : "");
    }
    printf("]}%s\n", last ? "" : ",");
}

static int kat()
{
    using V  = Eigen::VectorXd;
    using R  = Eigen::Ref<Eigen::VectorXd>;
    V none;
    auto shift1  = [](const V& x, R v) { v[0] = x[0] - 2; };
    auto affine3 = [](const V& x, R v) { v[0] = x[0] - 5; v[1] = x[1] + 3; v[2] = x[2]; };
    auto rosen   = [](const V& x, R v) { v[0] = std::sqrt(100) * (x[1] - x[0] * x[0]); v[1] = 1 - x[0]; };
    auto eq3     = [](const V& x, R v) { v[0] = x[0] - 3; };
    auto ineq3   = [](const V& x, R v) { v[0] = -x[0] + 3; };
    auto betts   = [](const V& x, R v) { v[0] = std::sqrt(0.01) * x[0]; v[1] = x[1]; };
    auto bettsc  = [](const V& x, R v) { v[0] = x[1] - 10.0 * x[0] + 10.0; };
    KatPhase def;
    KatPhase w100;
    w100.set_weights = true;
    w100.w[0] = w100.w[1] = w100.w[2] = 100;
    printf("{\"source\": \"optimization/test/test_levenberg_marquardt_sparse.cpp:71-371 re-run against the compiled reference\",\n\"cases\": [\n");
    katRun("solve_unconstr_1", "shift1", 1, 1, 0, 0, V::Ones(1), none, none, shift1, nullptr, nullptr, {def}, false);
    katRun("solve_unconstr_2", "affine3", 3, 3, 0, 0, V::Ones(3), none, none, affine3, nullptr, nullptr, {def}, false);
    katRun("solve_rosenbrock_unconstr", "rosenbrock", 2, 2, 0, 0, V::Ones(2), none, none, rosen, nullptr, nullptr, {def}, false);
    {   // the classic start (SURVEY 8c quotes its result)
        V x(2);
        x << -1.2, 1;
        katRun("rosenbrock_classic_start", "rosenbrock", 2, 2, 0, 0, x, none, none, rosen, nullptr, nullptr, {def}, false);
    }
    katRun("solve_eqconstr_1", "shift1_eq3", 1, 1, 1, 0, V::Ones(1), none, none, shift1, eq3, nullptr, {w100}, false);
    katRun("solve_ineqconstr_1", "shift1_ineq3", 1, 1, 0, 1, V::Ones(1), none, none, shift1, nullptr, ineq3, {w100}, false);
    {
        V lb(1);
        lb[0] = 5;
        katRun("solve_lower_bounds", "shift1", 1, 1, 0, 0, V::Ones(1), lb, none, shift1, nullptr, nullptr, {w100}, false);
        V ub(1);
        ub[0] = -1;
        katRun("solve_upper_bounds", "shift1", 1, 1, 0, 0, V::Ones(1), none, ub, shift1, nullptr, nullptr, {w100}, false);
    }
    {
        V lb(2), ub(2);
        lb << 2, -50;
        ub << 50, 50;
        KatPhase a;   // "feasible start": no initialize() call in the test, default weights
        a.set_values = {{0, 5.0}, {0, -5.0}};
        a.initialize = false;
        KatPhase b;   // "infeasible start"
        b.set_values  = {{0, -1.0}, {0, -1.0}};
        b.set_weights = true;
        b.w[0] = 1; b.w[1] = 10; b.w[2] = 10;
        b.iterations = 5000;
        katRun("solve_betts_fun_constr", "betts", 2, 2, 0, 1, none, lb, ub, betts, nullptr, bettsc, {a, b}, false);
        KatPhase c;
        c.set_values  = {{0, 5.0}, {0, -5.0}};
        c.set_weights = true;
        c.set_adapt   = true;
        c.adapt[0] = c.adapt[1] = c.adapt[2] = 5;
        c.iterations = 5;
        c.solves     = 5;
        KatPhase d = c;
        d.set_values = {{0, -1.0}, {0, -1.0}};
        d.initialize = false;
        katRun("solve_betts_fun_constr_weight_adapt", "betts", 2, 2, 0, 1, none, lb, ub, betts, nullptr, bettsc, {c, d}, true);
    }
    printf("]}\n");
    return 0;
}

int main(int argc, char** argv)
{
    if (argc < 2)
    {
        fprintf(stderr, "usage: ref_driver dump|bench|mpc|loop|hess|kat key=value ...\n");
        return 1;
    }
    std::string mode(argv[1]);
    if (mode == "kat")
    {
        if (argc > 2 && std::string(argv[2]).rfind("cap=", 0) == 0) g_kat_iter_cap = std::atoi(argv[2] + 4);
        return kat();
    }
    std::map<std::string, std::string> kv;
    Scenario s = parse(argc, argv, kv);
    if (mode == "dump") return dump(s);
    if (mode == "bench") return bench(s, kv);
    if (mode == "mpc") return mpc(s, kv);
    if (mode == "loop") return loop(s, kv);
    if (mode == "hess") return hess(s);
    fprintf(stderr, "unknown mode\n");
    return 1;
}
