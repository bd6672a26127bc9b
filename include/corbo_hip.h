/*
 * corbo_hip.h -- C-ABI of the MI355X-native NLP inner loop for control_box_rst hypergraph OCPs.
 *
 * This is the drop-in boundary.  Nothing like it exists in the reference: the reference's
 * plug-in point is the C++ class corbo::NlpSolverInterface
 *     /root/reference/src/optimization/include/corbo-optimization/solver/nlp_solver_interface.h:67-115
 * whose solve() is called from
 *     /root/reference/src/optimal_control/src/structured_ocp/structured_optimal_control_problem.cpp:134 .
 * The C++ adapter in control_box_rst_amd/adapter/ implements that class on top of the functions below
 * (binding shown in INTEGRATION.md).  Each entry point cites the reference code it replaces.
 *
 * Conventions: plain C, POD structs, caller owns every host buffer, the library owns device buffers and one
 * HIP stream per handle.  Return value 0 = OK, negative = corbo_hip_status.  A handle is not thread-safe;
 * distinct handles are independent.  All floating point is IEEE fp64 (the reference's only arithmetic type).
 *
 * Data layout ("vertex layout", one row per OCP instance, row stride = corbo_hip_dims.nv doubles):
 *     [ x_0 u_0 | x_1 u_1 | ... | x_{N-2} u_{N-2} | x_f | dt (only when dt is a free parameter) ]
 * which is the order of FullDiscretizationGridBase's vertices
 *     /root/reference/src/optimal_control/src/structured_ocp/discretization_grids/full_discretization_grid_base.cpp:514-527 .
 * "Parameter layout" (length n) is the same sequence with every fixed component removed (x_0, fixed x_f
 * components, fixed dt) -- the reference's column order (vertex_set.cpp:405-418).
 */
#ifndef CORBO_HIP_H_
#define CORBO_HIP_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define CORBO_HIP_MAX_NX 16
#define CORBO_HIP_MAX_NU 8
#define CORBO_HIP_INF 2e30 /* CORBO_INF_DBL, /root/reference/src/core/include/corbo-core/types.h:52 */

typedef enum corbo_hip_status {
    CORBO_HIP_OK              = 0,
    CORBO_HIP_ERR_INVALID     = -1, /* bad argument / unsupported descriptor */
    CORBO_HIP_ERR_DEVICE      = -2, /* HIP runtime error (see corbo_hip_last_error) */
    CORBO_HIP_ERR_UNSUPPORTED = -3, /* descriptor is valid for the reference but has no device kernel yet */
    CORBO_HIP_ERR_STATE       = -4  /* call order violated (e.g. solve before set_instance_data) */
} corbo_hip_status;

/* SolverStatus of the reference, /root/reference/src/optimization/include/corbo-optimization/types.h:30 */
typedef enum corbo_hip_solver_status {
    CORBO_HIP_SOLVER_CONVERGED        = 0,
    CORBO_HIP_SOLVER_EARLY_TERMINATED = 1,
    CORBO_HIP_SOLVER_INFEASIBLE       = 2,
    CORBO_HIP_SOLVER_ERROR            = 3
} corbo_hip_solver_status;

/* Discretization grid that produced the hypergraph (decides vertex set and edge creation order). */
typedef enum corbo_hip_grid {
    CORBO_HIP_GRID_FD          = 0, /* FiniteDifferencesGrid          (finite_differences_grid.cpp:38-154), fixed dt   */
    CORBO_HIP_GRID_FD_VARIABLE = 1, /* FiniteDifferencesVariableGrid  (finite_differences_variable_grid.h:34-89), free dt */
    CORBO_HIP_GRID_MS          = 2, /* MultipleShootingGrid, 1 control per interval (multiple_shooting_grid.cpp:38-197)  */
    CORBO_HIP_GRID_MS_VARIABLE = 3  /* MultipleShootingVariableGrid (multiple_shooting_variable_grid.h:34-90): the same with a free dt */
} corbo_hip_grid;

/* Dynamics-defect formula of the equality edge between (x_k, u_k, x_{k+1}, dt). */
typedef enum corbo_hip_defect {
    CORBO_HIP_DEFECT_FORWARD        = 0, /* finite_differences_collocation.h:126-134 */
    CORBO_HIP_DEFECT_BACKWARD       = 1, /* :160-168 */
    CORBO_HIP_DEFECT_MIDPOINT       = 2, /* :194-202 */
    CORBO_HIP_DEFECT_CRANK_NICOLSON = 3, /* :228-238 (reference default, full_discretization_grid_base.h:138) */
    CORBO_HIP_DEFECT_RK4_SHOOTING   = 4  /* multiple_shooting_edges.h:125-134 + explicit_integrators.h:280-295 */
} corbo_hip_defect;

/* System dynamics f(x,u) (SystemDynamicsInterface::dynamics, system_dynamics_interface.h:121). */
typedef enum corbo_hip_dynamics {
    CORBO_HIP_DYN_VAN_DER_POL       = 0, /* nonlinear_benchmark_systems.h:52-60, params[0] = a              nx=2 nu=1 */
    CORBO_HIP_DYN_SERIAL_INTEGRATOR = 1, /* linear_benchmark_systems.h:72-83,    params[0] = time constant  nx=p nu=1 */
    CORBO_HIP_DYN_UNICYCLE          = 2, /* user plug-in: xdot=u1 cos th, ydot=u1 sin th, thdot=u2          nx=3 nu=2 */
    CORBO_HIP_DYN_QUADROTOR         = 3, /* user plug-in: 12-state rigid body, see DESIGN.md                nx=12 nu=4 */
    /* the reference's other benchmark systems (nonlinear_benchmark_systems.h), parameters in the order of their setters */
    CORBO_HIP_DYN_DUFFING           = 4, /* DuffingOscillator :88-148,  params = damping, spring_alpha, spring_beta  nx=2 nu=1 */
    CORBO_HIP_DYN_FREE_SPACE_ROCKET = 5, /* FreeSpaceRocket :154-184    (no parameters)                              nx=3 nu=1 */
    CORBO_HIP_DYN_SIMPLE_PENDULUM   = 6, /* SimplePendulum :187-258,    params = mass, length, gravitation, friction  nx=2 nu=1 */
    CORBO_HIP_DYN_MASSLESS_PENDULUM = 7, /* MasslessPendulum :261-314,  params[0] = omega0                            nx=2 nu=1 */
    CORBO_HIP_DYN_TOY_EXAMPLE       = 8, /* ToyExample :406-460,        params[0] = mu                                nx=2 nu=1 */
    CORBO_HIP_DYN_ARTSTEINS_CIRCLE  = 9, /* ArtsteinsCircle :463-509    (no parameters)                              nx=2 nu=1 */
    CORBO_HIP_DYN_CART_POLE         = 10, /* CartPole :317-390          (fixed parameters), state [x phi xdot phidot] nx=4 nu=1 */
    CORBO_HIP_DYN_PARALLEL_INTEGRATOR = 11, /* linear_benchmark_systems.h:120-183, f = T u, params[0] = T      nx=nu=2|3 */
    CORBO_HIP_DYN_LINEAR_STATE_SPACE  = 12  /* LinearStateSpaceModel :186-262, f = A x + B u, matrices in lin_a / lin_b;
                                             * (nx, nu) in {(2,1), (2,2), (3,1), (3,2), (3,3), (4,1)} */
} corbo_hip_dynamics;
/* User dynamics models: a `Dynamics<>` specialisation dropped into control_box_rst_amd/csrc/models/ (README.md there) is registered by
 * the build under the id CORBO_HIP_DYN_USER + slot, slot = 0 .. 15 -- the device-side counterpart of a user's own
 * corbo::SystemDynamicsInterface subclass (system_dynamics_interface.h:66-121).  Shipped example: slot 0 = kinematic car (nx = 3, nu = 2). */
#define CORBO_HIP_DYN_USER 1000

typedef enum corbo_hip_stage_cost {
    CORBO_HIP_COST_NONE          = 0,
    CORBO_HIP_COST_QUADRATIC_LSQ = 1, /* QuadraticFormCost(Q,R, integral=false, lsq=true), diagonal Q,R, zero uref
                                         (quadratic_cost.cpp:100-184) */
    CORBO_HIP_COST_MIN_TIME_LSQ  = 2, /* MinimumTime(lsq=true) (minimum_time.h:49-78); dt edge created twice
                                         (nlp_functions.cpp:91-107) */
    CORBO_HIP_COST_MIN_TIME_QUADRATIC_LSQ = 3 /* MinTimeQuadratic(Q,R, integral=false, lsq=true), only_last_n = 0 (hybrid_cost.h:189-303): the
                                         terms of both, per interval in the edge order of nlp_functions.cpp:70-107 (state term, control
                                         term, at k = 0 the dt term twice); free-dt grids.  (QuadraticStateCost / QuadraticControlCost and
                                         the hybrids built on them, MinTimeQuadraticStates / ...Controls, create NO least-squares term for a
                                         diagonal weight: their diagonal setWeight overload leaves _Q / _R empty and the term dimension is
                                         _Q.rows() -- quadratic_state_cost.cpp:33-62, quadratic_state_cost.h:50; such graphs are what is
                                         left of them, COST_NONE resp. COST_MIN_TIME_LSQ, and the adapter's recogniser maps them so.) */
} corbo_hip_stage_cost;
/* which terms a stage cost has: bit 0 state term, bit 1 control term, bit 2 dt term (free-dt grids only) */
#define CORBO_HIP_COST_TERMS(c) ((c) == 1 ? 3 : (c) == 2 ? 4 : (c) == 3 ? 7 : 0)

typedef enum corbo_hip_stage_ineq {
    CORBO_HIP_INEQ_NONE = 0,
    CORBO_HIP_INEQ_BALL = 1 /* c(x) = r^2 - |x[0:3]-c|^2 <= 0 : spherical keep-out, params = cx,cy,cz,r (cfg 5) */
} corbo_hip_stage_ineq;
/* User stage functions: a `StageFunction<slot>` specialisation dropped into control_box_rst_amd/csrc/stage_functions/ (README.md there) is registered by
 * the build under the id CORBO_HIP_STAGE_FN_USER + slot, slot = 0 .. 15 -- the device-side counterpart of a user's own corbo::StageInequalityConstraint
 * subclass (stage_functions.h:276-310).  kind = state_ineq: its non-integral STATE term c(x_k) (dimension 1), named in `stage_ineq`, parameters
 * `ineq_params`; kind = control_ineq: its non-integral CONTROL term c(u_k) (dimension 1), named in `stage_ineq_control`, parameters `ineq_control_params`.
 * Shipped examples: slot 0 = tilt cone x[6]^2 + x[7]^2 - alpha^2 (state), slot 1 = input magnitude |u|^2 - r^2 (control). */
#define CORBO_HIP_STAGE_FN_USER 1000

typedef enum corbo_hip_final_ineq {
    CORBO_HIP_FINAL_INEQ_NONE = 0,
    /* TerminalBall, diagonal S (optimal_control/include/corbo-optimal-control/functions/final_state_constraints.h:38-96,
     * src/functions/final_state_constraints.cpp:60-80): c(x_f) = (x_f - xref)^T S (x_f - xref) - gamma <= 0, one inequality row
     * on x_f, created after the stage inequalities (finite_differences_grid.cpp:135-143).  params = S_11 .. S_nn, gamma. */
    CORBO_HIP_FINAL_INEQ_TERMINAL_BALL = 1
} corbo_hip_final_ineq;

/* Problem descriptor shared by every instance of a batch (POD). */
typedef struct corbo_hip_problem_desc {
    int32_t grid;          /* corbo_hip_grid */
    int32_t defect;        /* corbo_hip_defect */
    int32_t dynamics;      /* corbo_hip_dynamics */
    int32_t stage_cost;    /* corbo_hip_stage_cost */
    int32_t final_cost;    /* 0 = none, 1 = QuadraticFinalStateCost(Qf, lsq=true), diagonal (final_state_cost.cpp:72-112) */
    int32_t stage_ineq;    /* corbo_hip_stage_ineq, or CORBO_HIP_STAGE_FN_USER + slot: a user state function (csrc/stage_functions/) */
    int32_t nx, nu, N;     /* state dim, control dim, grid points (N-1 intervals) */
    uint32_t xf_fixed_mask; /* bit i set = component i of x_f is fixed (setXfFixed, full_discretization_grid_base.h:89-93) */
    double dt_ref;         /* dt (fixed grids) or initial dt (free-dt grid) */
    double dt_lb, dt_ub;   /* free-dt grid only (FiniteDifferencesVariableGrid::setDtBounds) */
    /* Box bounds shared along the horizon (NlpFunctions::x_lb/x_ub/u_lb/u_ub, set by
     * StructuredOptimalControlProblem::setBounds, structured_optimal_control_problem.cpp:167-176).  +-CORBO_HIP_INF =
     * unbounded.  Their finiteness pattern fixes which bound rows exist (vector_vertex.h:174-184); per-instance
     * values may be overridden with corbo_hip_set_instance_data but must keep that pattern. */
    double x_lb[CORBO_HIP_MAX_NX], x_ub[CORBO_HIP_MAX_NX];
    double u_lb[CORBO_HIP_MAX_NU], u_ub[CORBO_HIP_MAX_NU];
    double q_diag[CORBO_HIP_MAX_NX];
    double r_diag[CORBO_HIP_MAX_NU];
    double qf_diag[CORBO_HIP_MAX_NX];
    double dyn_params[8];
    double ineq_params[8];
    int32_t final_ineq;    /* enum corbo_hip_final_ineq; families with nx <= 4 and the 12-state big-block family */
    int32_t final_eq;      /* 1 = TerminalEqualityConstraint(xref) (final_state_constraints.h:130-160): nx equality rows x_f - xref after the
                            * defect rows (finite_differences_grid.cpp:135-141), xref = the instance's state reference; nx <= 4 or 12 */
    double final_ineq_params[CORBO_HIP_MAX_NX + 1];
    /* CORBO_HIP_DYN_LINEAR_STATE_SPACE (LinearStateSpaceModel::setParameters(A, B)): row-major A[i * nx + j], B[i * nu + j] */
    double lin_a[16];
    double lin_b[12];
    /* CORBO_HIP_COST_MIN_TIME_QUADRATIC_LSQ only: first interval that carries the quadratic form's terms -- MinTimeQuadratic's
     * only_last_n option, _quad_k_min = max(N - only_last_n, 0) (hybrid_cost.h:224-237); 0 = every interval */
    int32_t quad_first_interval;
    /* 1: the quadratic stage cost and the final cost are NOT in least-squares form -- QuadraticFormCost(Q, R, false, lsq_form = false),
     * QuadraticFinalStateCost(Qf, false): one scalar term xd^T Q xd / u^T R u / xd^T Qf xd per edge (quadratic_cost.cpp:133-138,165-170,
     * final_state_cost.cpp:102-108), filed as plain objective edges (edge_set.h:118-125).  What the reference's IPOPT / QP callers use.
     * Such a problem is not a least-squares problem: corbo_hip_solve / corbo_hip_eval refuse it like LevenbergMarquardtSparse::solve does
     * (levenberg_marquardt_sparse.cpp:48-55); the operators of the exact-Hessian path work on it (objective edges: finite-difference Hessians
     * weighted with the objective multiplier, gradient = Jacobian rows, value = sum).  Every stage cost kind: MinimumTime(false) is
     * (N - 1) dt (minimum_time.h:60), not flagged linear (stage_functions.h:73) -- its Hessian entries are finite differences too. */
    int32_t cost_nonlsq;
    /* QuadraticFormCost(Q, R, integral_form = true) on a FiniteDifferencesGrid: instead of the per-vertex terms ONE objective edge per
     * interval that integrates c(x, u) = (x - xref)^T Q (x - xref) + u^T R u over it (finite_differences_grid.cpp:62-77) --
     * 1: TrapezoidalIntegralCostEdge on (x_k, u_k, x_{k+1}, dt), 0.5 dt (c(x_k, u_k) + c(x_{k+1}, u_k));  2: LeftSumCostEdge on (x_k, u_k, dt),
     * dt c(x_k, u_k) (finite_differences_collocation_edges.h:98-152, 323-368; FullDiscretizationGridBase::CostIntegrationRule).  Plain objective
     * edges: needs cost_nonlsq = 1 (the final cost is then QuadraticFinalStateCost(Qf, false)); Hessian-path operators only.
     * With CORBO_HIP_COST_MIN_TIME_QUADRATIC_LSQ on the FiniteDifferencesVariableGrid: MinTimeQuadratic(Q, R, integral_form = true, lsq_form = false)
     * (hybrid_cost.h:189-303) -- its quadratic part in integral form (integral edges on the intervals k >= quad_first_interval), its dt term (plain,
     * created twice at k = 0) filed BEFORE interval 0's integral edge (finite_differences_grid.cpp:58-77).
     * On a MultipleShootingGrid (CORBO_HIP_GRID_MS; either value): the grid creates ONE MultipleShootingEdgeSingleControl per interval INSTEAD of
     * the dynamics-only edge (multiple_shooting_grid.cpp:70-77; multiple_shooting_edges.h:151-303) -- a mixed edge on (x_k, u_k, dt, x_{k+1}) whose
     * objective part is the cost integrated along the shooting step by the grid's integrator (augmented state [cost; x], :214-229, :251-281) and
     * whose equality part is the defect.  Its blocks are the last ones of the Hessian lists, its equality rows follow the terminal equality's.
     * Diagonal Q / R, no stage inequality, nx <= 4. */
    int32_t cost_integral;
    /* Non-diagonal weights (QuadraticFormCost::setWeightQ / setWeightR with a full matrix, quadratic_cost.cpp:36-55, 77-95;
     * QuadraticFinalStateCost::setWeightQf, final_state_cost.cpp:36-58; QuadraticFinalStateCostRiccati, final_state_cost.h:103, whose Qf is
     * always dense): the reference keeps the UPPER Cholesky factor U (Q = U^T U, Eigen::LLT<.., Upper>::matrixU()) and the least-squares
     * term is U (x - xref) / U u (quadratic_cost.cpp:116-118, 148-150; final_state_cost.cpp:88-90) -- a dense product in Eigen's gemv
     * order, its Jacobian block a dense (upper-triangular) nx x nx block.  Bit 0: q_sqrt is used instead of q_diag, bit 1: r_sqrt instead of
     * r_diag, bit 2: qf_sqrt instead of qf_diag.  Row-major [i * nx + j] (r_sqrt: [i * nu + j]), entries below the diagonal zero.  Families
     * with nx <= 4 on the Levenberg-Marquardt path and the Hessian-path operators in least-squares form (cost_nonlsq = 0). */
    int32_t weights_dense;
    /* Integrator of the shooting grids' defect edges (MultipleShootingGrid::setNumericalIntegrator; explicit_integrators.h):
     * 0 = IntegratorExplicitRungeKutta4 (:244-295, what the reference's examples use), 1 = IntegratorExplicitEuler (:47-72),
     * 2 = IntegratorExplicitRungeKutta2 (:97-138), 3 = IntegratorExplicitRungeKutta3 (:167-213), 5 / 6 / 7 = IntegratorExplicitRungeKutta5 / 6 / 7
     * (:327-394, :429-503, :541-628; every family on the Levenberg-Marquardt path -- around a big-block model not together with extra edges or with a free
     * dt on odd block sizes; the Hessian-path operators: families with nx <= 4).
     * Travels to the kernels in slot 7 of the dynamics parameters (no model uses more than 5). */
    int32_t shooting_integrator;
    /* TerminalPartialEqualityConstraint (final_state_constraints.h:198-300): with final_eq = 1 and a non-zero mask the equality rows exist for the
     * components whose bit is set only -- row idx = the number of active components before it, x_f[i] - xref[i]; 0 = every component
     * (TerminalEqualityConstraint).  Every family, both the Levenberg-Marquardt path and the Hessian-path operators. */
    uint32_t final_eq_mask;
    double q_sqrt[16];
    double r_sqrt[16];
    double qf_sqrt[16];
    /* ---- Integral-form constraints and the control-deviation term (FiniteDifferencesGrid / FiniteDifferencesVariableGrid,
     * families with nx <= 4, horizons up to 256 grid points).  The reference creates these edges from USER stage functions only (no stock
     * class has such terms); the device knows the plug-in functions named here.
     * constraint_integration: the grid's integration rule for them (FullDiscretizationGridBase::CostIntegrationRule; the same member that
     *   integrates an integral cost): 1 = TrapezoidalRule, 2 = LeftSum (finite_differences_grid.cpp:80-125); required with stage_ineq_integral / stage_eq.
     * stage_ineq_integral = 1: stage_ineq is the stage inequalities' INTEGRAL state-control term c(x, u) (getIntegralStateControlTermDimension = 1)
     *   instead of their non-integral state term: one TrapezoidalIntegralInequalityEdge on (x_k, u_k, x_{k+1}, dt), 0.5 dt (c(x_k, u_k) + c(x_{k+1}, u_k)),
     *   or one LeftSumInequalityEdge on (x_k, u_k, dt), c(x_k, u_k) dt, per interval (finite_differences_collocation_edges.h:271-321, 412-459).
     * stage_eq: corbo_hip_stage_eq, the stage equalities' integral state-control term e(x, u) (one row).  TrapezoidalRule: the row is APPENDED to the
     *   interval's dynamics edge (TrapezoidalIntegralEqualityDynamicsEdge, :149-216: dimension nx + 1); LeftSum: a LeftSumEqualityEdge on (x_k, u_k, dt)
     *   in front of the dynamics edge (:368-410).
     * ctrl_dev: corbo_hip_ctrl_dev, the stage inequalities' control-deviation term (getNonIntegralControlDeviationTermDimension = nu): one
     *   TernaryVectorScalarVertexEdge per interval on (u_k, u_{k-1}, dt) -- k = 0: on (u_0, the previously applied control, its age), both fixed,
     *   corbo_hip_set_previous_control -- and one behind the final-stage edges on (u_ref = 0, u_{N-2}, dt) (nlp_functions.cpp:117-131, 152-186,
     *   finite_differences_grid.cpp:145-153).  It couples the controls of neighbouring intervals.
     * Handles with any of these run the LM pass as separate launches with the band factorisation (band_factor_kernel), not the fused kernels. */
    int32_t constraint_integration;
    int32_t stage_ineq_integral;
    int32_t stage_eq;
    int32_t ctrl_dev;
    double stage_eq_params[CORBO_HIP_MAX_NX + CORBO_HIP_MAX_NU + 1];   /* CORBO_HIP_STAGE_EQ_LINEAR: a_1 .. a_nx, b_1 .. b_nu, c */
    double ctrl_dev_params[CORBO_HIP_MAX_NU];                          /* CORBO_HIP_CTRL_DEV_RATE: r_max per control */
    /* The stage inequalities' non-integral CONTROL term (getNonIntegralControlTermDimension = 1): 0 = none, else CORBO_HIP_STAGE_FN_USER + slot of a
     * registered `control_ineq` function -- one UnaryVectorVertexEdge on u_k per interval, created behind the state term's edge and in front of the
     * control-deviation edge (nlp_functions.cpp:82-89).  An edge on u_k alone; every family and grid on the Levenberg-Marquardt path (handles with it take
     * the extra-edge routes like the control-deviation term does). */
    int32_t stage_ineq_control;
    int32_t reserved0;
    double ineq_control_params[8];
} corbo_hip_problem_desc;

typedef enum corbo_hip_stage_eq {
    CORBO_HIP_STAGE_EQ_NONE   = 0,
    CORBO_HIP_STAGE_EQ_LINEAR = 1   /* e(x, u) = a^T x + b^T u - c (summed left to right: the x terms, then the u terms, then - c) */
} corbo_hip_stage_eq;
typedef enum corbo_hip_ctrl_dev {
    CORBO_HIP_CTRL_DEV_NONE = 0,
    CORBO_HIP_CTRL_DEV_RATE = 1     /* row i: ((u_k[i] - u_prev[i]) / dt_prev)^2 - r_max[i]^2 <= 0 (an input-rate limit) */
} corbo_hip_ctrl_dev;

/* Sizes derived from a descriptor (corbo_hip_get_dims). */
typedef struct corbo_hip_dims {
    int32_t nv;      /* doubles per instance in vertex layout */
    int32_t n;       /* parameters (columns of J)  = getParameterDimension()            */
    int32_t lsq;     /* getLsqObjectiveDimension()                                       */
    int32_t eq;      /* getEqualityDimension()                                           */
    int32_t ineq;    /* getInequalityDimension()                                         */
    int32_t bounds;  /* finiteCombinedBoundsDimension(): one row per unfixed component with a finite lb or ub */
    int32_t m;       /* rows of J = lsq + eq + ineq + bounds                             */
    int32_t nnz;     /* structural non-zeros of J (hyper_graph_optimization_problem_edge_based.cpp:193-222) */
} corbo_hip_dims;

/* Levenberg-Marquardt options = LevenbergMarquardtSparse's setters
 * (/root/reference/src/optimization/include/corbo-optimization/solver/levenberg_marquardt_sparse.h:86-124). */
typedef struct corbo_hip_lm_opts {
    int32_t iterations;                                       /* setIterations, default 10 */
    double weight_eq, weight_ineq, weight_bounds;             /* setPenaltyWeights, default 2/2/2 */
    double adapt_factor_eq, adapt_factor_ineq, adapt_factor_bounds; /* setWeightAdapation, default 1/1/1 */
    double adapt_max_eq, adapt_max_ineq, adapt_max_bounds;    /* default 500/500/500 */
} corbo_hip_lm_opts;

typedef struct corbo_hip_stats {
    int64_t lm_iterations;      /* sum over instances of outer iterations run (= batch * iterations) */
    int64_t accepted_steps;
    int64_t rejected_steps;
    int64_t jacobian_sweeps;    /* per-instance Jacobian evaluations, summed */
    int64_t residual_sweeps;    /* per-instance residual evaluations, summed */
    int64_t factorizations;
    int32_t passes;             /* device passes (kernel rounds) the batch needed */
    float   solve_ms;           /* HIP-event time of the last corbo_hip_solve (device side) */
    float   sweep_ms;           /* accumulated time of the edge/Jacobian sweep kernel inside it (0 if not profiled) */
    float   factor_ms;          /* accumulated time of the assemble/factor/solve kernel (0 if not profiled) */
    int32_t inner_loop_cuts;    /* instances whose inner loop was cut after 64 consecutive rejections (status ERROR); 0 in every test */
    int64_t counted_iterations; /* of lm_iterations (and of factorizations, one each): outer iterations that followed a converged step
                                 * (|delta| <= eps2 / 2) and were COUNTED, not executed (option "ff_converged", default on; the reference
                                 * executes them and changes nothing a caller sees, levenberg_marquardt_sparse.cpp:129-154).  Rates of
                                 * executed work: lm_iterations - counted_iterations, factorizations - counted_iterations. */
    int64_t speculative_takeovers; /* big-block family, reject-streak speculation (option "reject_speculation"): how often an instance whose step was
                                 * rejected took over the state of one of its damping candidates that had run alongside in spare rows.  The candidates ARE the
                                 * instance's own next passes (same arithmetic, same numbers): every other field and every result is the same with or without. */
} corbo_hip_stats;

typedef struct corbo_hip_solver* corbo_hip_handle;

/* Fill `opts` with the reference defaults (levenberg_marquardt_sparse.h:112-124). */
void corbo_hip_default_lm_opts(corbo_hip_lm_opts* opts);

/* Validate a descriptor and compute the static sizes.  Replaces the dimension bookkeeping of
 * LevenbergMarquardtSparse::solve (levenberg_marquardt_sparse.cpp:48-80) and
 * OptimizationEdgeSet::getDimensions / VertexSetInterface::computeVertexIndices (edge_set.cpp:198-236,
 * vertex_set.cpp:405-418).  Needs no GPU. */
int corbo_hip_get_dims(const corbo_hip_problem_desc* desc, corbo_hip_dims* dims);

/* Static sparsity pattern of the combined Jacobian in the library's value order (length dims.nnz each).
 * Replaces computeCombinedSparseJacobiansStructure (hyper_graph_optimization_problem_edge_based.cpp:1181-1478).
 * Needs no GPU. */
int corbo_hip_get_structure(const corbo_hip_problem_desc* desc, int32_t* rows, int32_t* cols);

/* Initial trajectory exactly as the grid writes it into the vertices before the first solve
 * (FullDiscretizationGridBase::initializeSequences, full_discretization_grid_base.cpp:134-179): linear
 * interpolation x0 -> xf, u = 0, dt = dt_ref.  x0, xf: [batch][nx]; x_out: [batch][nv].  Host-only helper. */
int corbo_hip_init_trajectory(const corbo_hip_problem_desc* desc, int batch, const double* x0, const double* xf, double* x_out);

/* Create a solver for `batch` independent instances of `desc` on HIP device `device`.  Every entry point below runs on that device
 * and restores the caller's current device before it returns.  No C++ exception leaves the library.  A handle is not thread-safe;
 * different handles (also on the same device) are independent. */
int corbo_hip_create(const corbo_hip_problem_desc* desc, int batch, int device, corbo_hip_handle* out);
/* The same with an explicit choice of the factorisation ROUTE where a descriptor has two (A/B measurements and the cross-route parity tests;
 * the library reads no environment variable).  `route` = OR of:
 *   CORBO_HIP_ROUTE_FREE_DT_BAND  a free dt around a big-block model through the general band factorisation instead of the stage / chain kernels' border column
 *   CORBO_HIP_ROUTE_XE_BAND       extra edges (control-deviation term, integral-form constraint edges, a user control inequality) of the small-block families
 *                                 up to 256 grid points through the general band factorisation instead of the block-tridiagonal route (one launch per solve)
 * 0 = what corbo_hip_create chooses.  A flag that does not apply to the descriptor is ignored. */
#define CORBO_HIP_ROUTE_FREE_DT_BAND 1u
#define CORBO_HIP_ROUTE_XE_BAND      2u
int corbo_hip_create_routed(const corbo_hip_problem_desc* desc, int batch, int device, uint32_t route, corbo_hip_handle* out);
void corbo_hip_destroy(corbo_hip_handle h);
/* Which factorisation a handle's solves run through (decided in corbo_hip_create from the descriptor's structure; for logs, A/B scripts and the route tests):
 *   CORBO_HIP_FACTOR_STAGE_CR     small-block families (nx <= 4): controls first, block cyclic reduction on the state blocks; one launch per solve up to 256 grid
 *                                 points, host-launched passes with the workspace in HBM beyond
 *   CORBO_HIP_FACTOR_STAGE_CHAIN  big-block family (5 <= nx <= 12): stage kernel + partitioned chain (a free dt as a second right-hand side)
 *   CORBO_HIP_FACTOR_BAND         general band factorisation of J^T J (host-launched passes)
 *   CORBO_HIP_FACTOR_BLOCK_TRI    small-block families with extra edges: block cyclic reduction on the (x_k, u_k) blocks, one launch per solve
 * Returns the value, or a negative CORBO_HIP_ERR_* for a null handle. */
#define CORBO_HIP_FACTOR_STAGE_CR    0
#define CORBO_HIP_FACTOR_STAGE_CHAIN 1
#define CORBO_HIP_FACTOR_BAND        2
#define CORBO_HIP_FACTOR_BLOCK_TRI   3
int corbo_hip_factor_route(corbo_hip_handle h);

/* Upload per-instance data (what the grid holds in its vertices when solve() is entered):
 *   x  [batch][nv] vertex values, lb/ub [batch][nv] bounds (+-CORBO_HIP_INF = unbounded; entries of fixed
 *   components are ignored), xref [batch][nx] static state reference (StaticReference).
 * lb, ub may be NULL = the descriptor's box bounds replicated along the horizon (filled in on the device); xref may be NULL = zeros.
 * The caller's arrays are plain host memory and are free again when the call returns: they are repacked into a pinned staging
 * buffer owned by the handle and copied on the handle's stream. */
int corbo_hip_set_instance_data(corbo_hip_handle h, const double* x, const double* lb, const double* ub, const double* xref);

/* The previously applied control and its age, per instance (StructuredOptimalControlProblem::setPreviousControlInput,
 * structured_optimal_control_problem.h:73-79; the fixed vertices _u_prev / _u_prev_dt of the grid, full_discretization_grid_base.cpp:66-70): what the
 * control-deviation edge of interval 0 sees (corbo_hip_problem_desc::ctrl_dev).  u_prev [batch][nu], dt_prev [batch]; NULL = zeros resp. the
 * descriptor's dt_ref (the reference's defaults, structured_optimal_control_problem.cpp:67-71). */
int corbo_hip_set_previous_control(corbo_hip_handle h, const double* u_prev, const double* dt_prev);


/* Time-varying state reference (what ReferenceTrajectoryInterface::getReferenceCached(k) hands the cost and final-stage terms,
 * optimal_control/src/functions/quadratic_cost.cpp:100-119, final_state_cost.cpp:72-92, final_state_constraints.cpp:60-80): one
 * reference per vertex component, in the vertex layout of corbo_hip_set_instance_data -- ref [batch][nv]: the entries of x_k are the
 * state reference at grid point k, those of x_f the reference of the final-stage terms (final cost, TerminalBall,
 * TerminalEqualityConstraint); the dt entry is ignored.  The cost row of a state component is w * (x - ref).  The control entries
 * must be zero (CORBO_HIP_ERR_INVALID otherwise): the reference's least-squares control term with a non-zero control reference
 * assigns the scalar ud^T R^(1/2) ud to the nu-vector (quadratic_cost.cpp:160-163) -- there is nothing well-defined to reproduce.
 * Stays in force until the next call; NULL returns to the static reference of corbo_hip_set_instance_data.  The moving-horizon shift
 * does not move references: a caller that advances time uploads the references of the new grid. */
int corbo_hip_set_references(corbo_hip_handle h, const double* ref);
/* The same for a whole tracking run, resident on the device: traj [batch][T][nx] = the state reference sampled at the grid's dt
 * (DiscreteTimeReferenceTrajectory with one sample per dt, zero-order hold; the last sample is held beyond the end,
 * core/include/corbo-core/reference_trajectory.h:383-400).  Control step `step` sees samples step .. step + N - 1; the call sets the
 * window of `step`, and every control step of corbo_hip_closed_loop moves it one sample on before its solve (what sampling at
 * t + k dt does in PredictiveController::step).  NULL ends it (static reference again). */
int corbo_hip_set_reference_trajectory(corbo_hip_handle h, const double* traj, int T, int step);

/* Re-arm the resident batch: copy the x uploaded by the last corbo_hip_set_instance_data back into the iterate,
 * device to device, asynchronously on the handle's stream (what the grid does when it re-initialises its vertices for a
 * new problem, full_discretization_grid_base.cpp:134-179).  Lets a caller re-solve the same batch without a PCIe trip. */
int corbo_hip_restore_instance_data(corbo_hip_handle h);

/* Start of a new moving-horizon (MPC) run on the trajectories resident in HBM = the new_run branch of
 * FullDiscretizationGridBase::update (full_discretization_grid_base.cpp:91-108), per instance, on the device:
 *   shift != 0  (the caller's grid->setWarmStart(true); honoured by fixed-dt grids only, like
 *               isMovingHorizonWarmStartActive, full_discretization_grid_base.h:133 / finite_differences_variable_grid.h:77):
 *               warmStartShifting(x0) (:230-283) -- the nearest stored state within 20 samples decides the shift
 *               (findNearestState, :285-317), states and controls move forward, the tail is extrapolated linearly;
 *   always:     x_0 = x0_new (:101), fixed components of x_f = the state reference (:103-106).
 * ShootingGridBase (MultipleShootingGrid) does the same on its shooting intervals (shooting_grid_base.cpp:99-113, 292-388).
 * x0_new [batch][nx] (host; copied into pinned memory that the kernel reads directly, free again on return).  Asynchronous on the
 * handle's stream.  Follow with corbo_hip_solve(h, opts, new_run = 1). */
int corbo_hip_warm_start(corbo_hip_handle h, const double* x0_new, int shift);

/* ---- Time-optimal grids whose resolution adapts per instance (FiniteDifferencesVariableGrid::adaptGrid*, finite_differences_variable_grid.cpp:
 * 66-163): N differs from instance to instance, a handle has ONE N -- a caller keeps one handle per N (a "bucket") and moves an
 * instance whose N changes into the bucket of its new N with a resampled trajectory, all on the device
 * (control_box_rst_amd/adaptive_grid.py is that caller).
 *   corbo_hip_prepare_slots  a handle that is only ever filled by corbo_hip_resample_into: the descriptor's bound pattern in every
 *                            slot, zero iterates; and / or sets how many of the handle's slots are in use: instances [0, active) take
 *                            part in corbo_hip_solve / corbo_hip_warm_start / corbo_hip_get_stats (default after create: all).
 *   corbo_hip_get_dt         dt of the active instances (what adaptGridTimeBased* decides on), dt_out [active].  Synchronises.
 *   corbo_hip_resample_into  FullDiscretizationGridBase::resampleTrajectory(N_dst) (full_discretization_grid_base.cpp:397-474) of
 *                            instance src_index[q] of `src` into slot dst_index[q] of `dst`, q < count: start sample and x_f kept, interior
 *                            states interpolated linearly in time, controls held, dt_new = dt_old (N_src - 1) / (N_dst - 1); the state
 *                            reference travels along.  N_src == N_dst is a plain move (src and dst may be the same handle: compaction).
 *                            Bit-identical to the oracle's restatement, which is pinned bit for bit to sequences of the reference.
 *                            CORBO_HIP_GRID_MS_VARIABLE handles: ShootingGridBase::resampleTrajectory (shooting_grid_base.cpp:473-547),
 *                            the same operation on the same vertex layout. */
int corbo_hip_prepare_slots(corbo_hip_handle h, int active);
int corbo_hip_get_dt(corbo_hip_handle h, double* dt_out);
int corbo_hip_resample_into(corbo_hip_handle src, corbo_hip_handle dst, int count, const int32_t* src_index, const int32_t* dst_index);

/* u_0 of every instance's current trajectory = FullDiscretizationGridBase::getFirstControlInput
 * (full_discretization_grid_base.cpp:324-331), what a predictive controller applies to its plant.  u0_out [batch][nu] (host);
 * batch * nu doubles, packed by a small kernel into pinned host memory, instead of the whole trajectories.  Synchronises. */
int corbo_hip_get_first_control(corbo_hip_handle h, double* u0_out);

/* ---- The plant side of a closed loop, on the device -----------------------------------------------------------
 * One plant per instance whose model is the descriptor's dynamics: the reference's SimulatedPlant(dynamics) with a full-state output
 * (plants/src/simulated_plant.cpp).  With these four entries a batch of predictive controllers runs
 *     corbo_hip_plant_step -> corbo_hip_warm_start_from_plant -> corbo_hip_solve
 * (= per step of task_closed_loop_control.cpp:153-235: plant.output, controller.step, plant.control) without any state or control
 * crossing PCIe. */
typedef enum corbo_hip_integrator {
    CORBO_HIP_INTEGRATOR_EULER = 0, /* IntegratorExplicitEuler (SimulatedPlant's default), explicit_integrators.h:66-72 */
    CORBO_HIP_INTEGRATOR_RK4   = 1  /* IntegratorExplicitRungeKutta4, explicit_integrators.h:280-295 */
} corbo_hip_integrator;
/* SimulatedPlant::setInitialState + reset: x [batch][nx] (host). */
int corbo_hip_plant_set_state(corbo_hip_handle h, const double* x);
/* SimulatedPlant::control(u_sequence, ., dt, t) without dead time (simulated_plant.cpp:97-160): the first control of each instance's
 * resident trajectory is held over dt, x_plant <- integrator.solveIVP(x_plant, u_0, dt), then the state disturbance: `disturbance`
 * [batch][nx] (host, may be NULL) is ADDED to the new state (what a DisturbanceInterface object does to it, :141).  Asynchronous. */
int corbo_hip_plant_step(corbo_hip_handle h, int integrator, double dt, const double* disturbance);
/* A plant that differs from the controller's model: SimulatedPlant takes its OWN SystemDynamicsInterface object
 * (plants/include/corbo-plants/simulated_plant.h), e.g. the same system class with other parameters.  params [batch][8]: the model
 * parameters (order of the descriptor's dyn_params) of every instance's plant; NULL = the controller's again.  The controller keeps
 * the descriptor's parameters.  Not for CORBO_HIP_DYN_LINEAR_STATE_SPACE. */
int corbo_hip_plant_set_params(corbo_hip_handle h, const double* params);
/* Per-instance parameters of the CONTROLLER's dynamics (the `params` argument of SURVEY 8b's set_instance_data sketch): a batch of
 * controllers for plants of one class that differ in their parameters -- in the reference one SystemDynamicsInterface object each
 * (e.g. VanDerPolOscillator::setParameters(a), nonlinear_benchmark_systems.h) handed to its own StructuredOptimalControlProblem.
 * params [batch][8] in the order of the descriptor's dyn_params; NULL = the descriptor's for every instance again.  Used by every
 * kernel that evaluates the dynamics (LM solve, corbo_hip_eval, the Hessian-path operators, and the simulated plants unless
 * corbo_hip_plant_set_params gave them their own).  Not for CORBO_HIP_DYN_LINEAR_STATE_SPACE.  Synchronises. */
int corbo_hip_set_instance_params(corbo_hip_handle h, const double* params);
/* SimulatedPlant::output with FullStateSystemOutput: x_out [batch][nx] (host).  Synchronises. */
int corbo_hip_plant_get_state(corbo_hip_handle h, double* x_out);
/* corbo_hip_warm_start with x0_new = the device-resident plant states (the measured state a controller is handed). */
int corbo_hip_warm_start_from_plant(corbo_hip_handle h, int shift);

/* A batch of predictive controllers with simulated plants: `steps` control steps without returning to the host in between, i.e.
 * steps x [plant.control; controller.step] of task_closed_loop_control.cpp:153-235 with PredictiveController::step
 * (controllers/src/predictive_controller.cpp:46-80: `ocp_iterations` solves per step, new_run only for the first) per instance:
 *     for s in 0 .. steps-1:   corbo_hip_plant_step(integrator, dt, disturbance[s]);   corbo_hip_warm_start_from_plant(shift);
 *                              corbo_hip_solve(new_run = 1);   (ocp_iterations - 1) x corbo_hip_solve(new_run = 0)
 * For the families with a run-to-completion solve kernel everything is enqueued on the handle's stream and synchronised once at
 * the end (3 launches per step); the others run the same sequence step by step.  Needs resident, solved trajectories and a plant
 * state.  disturbance [steps][batch][nx] (host) or NULL.  states_out [steps][batch][nx], controls_out [steps][batch][nu] (host, may
 * be NULL): the plant state after each step and the control that was applied during it. */
int corbo_hip_closed_loop(corbo_hip_handle h, const corbo_hip_lm_opts* opts, int steps, int ocp_iterations, int integrator, double dt, int shift,
                          const double* disturbance, double* states_out, double* controls_out);

/* The NLP inner loop for the whole batch = LevenbergMarquardtSparse::solve
 * (levenberg_marquardt_sparse.cpp:44-220) per instance.  new_run: reset (1) or adapt (0) the penalty weights
 * (:83-86); 2 = 1 with the iterates of the last corbo_hip_set_instance_data as the start (corbo_hip_restore_instance_data and new_run = 1
 * in one call: the run-to-completion kernel reads its start from the shadow copy itself, other handles copy first).  One run-to-completion launch on the handle's stream; the call returns once every instance has finished its outer
 * iterations; results stay resident in HBM.  CORBO_HIP_ERR_DEVICE "pass limit reached" if an instance is still unfinished after
 * 4096 LM passes (never seen; the reference would loop).  One deviation from the reference is guarded, not silent: an inner loop
 * that rejects 64 trial steps in a row is cut (the reference would keep multiplying the damping); such an instance finishes with
 * status CORBO_HIP_SOLVER_ERROR and is counted in corbo_hip_stats.inner_loop_cuts. */
int corbo_hip_solve(corbo_hip_handle h, const corbo_hip_lm_opts* opts, int new_run);
/* The same solve WITHOUT the wait (round 4): everything is enqueued on the handle's stream and the call returns, so the host side of the next solve --
 * re-arm / warm start, the launch -- overlaps this one's kernel (the reference has no counterpart: LevenbergMarquardtSparse::solve is a blocking call;
 * a caller that streams batch after batch through one handle uses this).  Handles that solve in one launch (run to completion: the small-block families
 * up to 256 grid points); every other handle solves synchronously like corbo_hip_solve.  Results, statistics, the HIP-event times of corbo_hip_get_timing
 * and the pass-limit check of the enqueued solves become available with the next corbo_hip_synchronize / corbo_hip_solve / corbo_hip_get_* /
 * corbo_hip_fetch_solution call (an error of an enqueued solve is reported there).  Entry points that CHANGE the handle's data (set_instance_data,
 * warm_start, set_references ...) wait for the enqueued solves and then do their work; they never return an enqueued solve's error -- it stays pending
 * for the next of the calls named above. */
int corbo_hip_solve_async(corbo_hip_handle h, const corbo_hip_lm_opts* opts, int new_run);

/* Block until the handle's stream is idle. */
int corbo_hip_synchronize(corbo_hip_handle h);

/* Download results (synchronises): x_out [batch][nv] last accepted iterate (what callers read from the vertices,
 * full_discretization_grid_base.cpp:324-331,529-565), chi2_out [batch] (*obj_value), status_out [batch]
 * (corbo_hip_solver_status).  Any pointer may be NULL. */
int corbo_hip_get_solution(corbo_hip_handle h, double* x_out, double* chi2_out, int32_t* status_out);
int corbo_hip_get_stats(corbo_hip_handle h, corbo_hip_stats* stats);

/* Results into HOST-VISIBLE (pinned) memory owned by the handle, without the repacking copy of corbo_hip_get_solution: the
 * accepted iterates [batch][*x_row_stride] (the first dims.nv doubles of a row are the vertex layout; the rest is the library's
 * padding), chi2 [batch], status [batch].  Asynchronous copies behind the solve on the handle's stream, one synchronisation.  The
 * views stay valid until the next call that stages data through the handle (set_instance_data, get_solution, fetch_solution,
 * plant_get_state).  This is "last result resident on host-visible memory" of SURVEY.md 8d.  Any pointer may be NULL. */
int corbo_hip_fetch_solution(corbo_hip_handle h, const double** x_pinned, int32_t* x_row_stride, const double** chi2_pinned,
                             const int32_t** status_pinned);

/* Result sink: with enable != 0 the run-to-completion solve kernel writes every instance's accepted iterate and LM state into the
 * handle's pinned host memory itself, as soon as that instance has finished (posted PCIe writes, overlapped with the instances still
 * iterating); corbo_hip_fetch_solution after such a solve returns the views without any copy.  Off by default (a moving-horizon
 * caller keeps its results on the device).  Handles without a run-to-completion kernel (big-block family, band route: passes launched
 * from the host) deliver the same with a copy instead when the batch's results are 1 MB or more: a device-side snapshot behind the last pass, then the copy engine moves it into
 * pinned memory on a second stream, overlapping whatever the caller enqueues next; corbo_hip_fetch_solution / corbo_hip_synchronize
 * wait for it (smaller batches: corbo_hip_fetch_solution copies, as without the sink). */
int corbo_hip_set_result_sink(corbo_hip_handle h, int enable);

/* Accumulated HIP-event time [ms] (handle's stream, first launch to last kernel end) and number of corbo_hip_solve calls since the
 * last reset: the average launch duration of the run-to-completion solve kernel, measured live (bench.py roofline object). */
int corbo_hip_get_timing(corbo_hip_handle h, double* solve_ms_sum, int64_t* solves, int reset);

/* Measurement hook (bench.py `roofline_sweep_phase`, profiles/rNN_pass_phases.json): with corbo_hip_set_option(h, "phase_cycles", 1) the
 * run-to-completion solve kernel adds up, per instance, the shader-clock cycles its workgroup spends in the phases of an LM pass -- the hot path of
 * LevenbergMarquardtSparse::solve (levenberg_marquardt_sparse.cpp:129-215) split where the reference splits it: residual + Jacobian sweeps
 * (computeValues + computeCombinedSparseJacobian, :97-100, :178-199), residual-only sweeps of rejected trial steps (:160-169), factor phases
 * (H + mu I, SimplicialLLT, solve: :135-158).  out [batch][8] of the LAST solve: cycles [0] Jacobian sweep phases, [1] residual-only sweep phases,
 * [2] factor phases; counts [3], [4], [5]; [6], [7] unused.  Two clock reads per phase: the solve is ~ 1 % slower with the option on. */
int corbo_hip_get_phase_cycles(corbo_hip_handle h, int64_t* out);

/* Parity hook for SURVEY rows a2-a6: evaluate the stacked residual
 * (LevenbergMarquardtSparse::computeValues, levenberg_marquardt_sparse.cpp:222-246) and, if jac_out != NULL, the
 * combined Jacobian values (computeCombinedSparseJacobian, hyper_graph_optimization_problem_edge_based.cpp:1480-1753)
 * at the resident x with the given weights.  values_out [batch][m], jac_out [batch][nnz] (order of
 * corbo_hip_get_structure).  Synchronises. */
int corbo_hip_eval(corbo_hip_handle h, double w_eq, double w_ineq, double w_bounds, double* values_out, double* jac_out);

/* ---- Several GPUs of one node: the batch is the sharding unit -- independent OCP instances, no collective on the data
 * path.  A caller (one process per GPU, or one thread per handle) creates one handle per device on its slice:
 *     corbo_hip_device_count(&n);  corbo_hip_shard_bounds(global_batch, world, rank, &first, &count);  corbo_hip_create(desc, count, rank % n, &h);
 * Results of all ranks are collected straight from device memory: corbo_hip_device_views gives the [batch][row_stride] iterate array and the
 * handle's stream for an RCCL all-gather (control_box_rst_amd/sharding.py gather_trajectories_device does it through torch.distributed). */
int corbo_hip_device_count(int* count);
/* Contiguous slice of `rank` out of `world` (remainders go to the low ranks) -- the same rule as control_box_rst_amd.sharding.shard_bounds. */
int corbo_hip_shard_bounds(int global_batch, int world, int rank, int* first, int* count);


/* ---- Operators of the exact-Hessian path ------------------------------------------------------------------
 * What an interior-point / SQP solver behind corbo::NlpSolverInterface asks the hypergraph for instead of the Gauss-Newton
 * system: replaces HyperGraphOptimizationProblemEdgeBased::computeSparseHessians{NNZ,Structure,Values}
 * (optimization/src/hyper_graph/hyper_graph_optimization_problem_edge_based.cpp:2087-3760; called by IpoptWrapper::eval_h,
 * optimization/src/solver/nlp_solver_ipopt_wrapper.cpp:249-271, with lower_part_only = true) and
 * computeSparseJacobianTwoSideBoundedLinearForm{NNZ,Structure,Values} + computeBoundsForTwoSideBoundedLinearForm (:4762-4968,
 * optimization/src/optimization_problem_interface.cpp:1141-1183), evaluated for every instance at its resident iterate.
 * Three triplet lists -- objective (least-squares edges: 2 m J^T J blocks), equalities, inequalities (forward differences,
 * HESSIAN_DELTA = 1e-2, of the central-difference edge Jacobians, weighted by the multipliers) -- in the reference's entry order.
 * Entry order quirk kept from the reference: a rectangular off-diagonal vertex pair is LISTED row-major but FILLED column-major
 * (:2993-3003 against :3550-3552).  The structure entries are host-only functions of the descriptor (like corbo_hip_get_structure); the
 * eval entries take and return host arrays. */
int corbo_hip_hessian_nnz(const corbo_hip_problem_desc* desc, int lower_part_only, int32_t* nnz_out /* [3]: objective, equalities, inequalities */);
int corbo_hip_hessian_structure(const corbo_hip_problem_desc* desc, int lower_part_only, int32_t* rows_obj, int32_t* cols_obj, int32_t* rows_eq,
                                int32_t* cols_eq, int32_t* rows_ineq, int32_t* cols_ineq);
/* mult_eq [batch][dims.eq], mult_ineq [batch][dims.ineq] (NULL = all ones, as the reference treats a null pointer);
 * vals_* [batch][nnz of the list] */
int corbo_hip_eval_hessians(corbo_hip_handle h, int lower_part_only, double mult_obj, const double* mult_eq, const double* mult_ineq, double* vals_obj,
                            double* vals_eq, double* vals_ineq);
/* The first-order callbacks of the same interface (IpoptWrapper::eval_grad_f / eval_f, nlp_solver_ipopt_wrapper.cpp:128-169):
 * computeGradientObjective (hyper_graph_optimization_problem_edge_based.cpp:31-102) -> grad [batch][n], and computeValueObjective
 * (hyper_graph_optimization_problem_base.cpp:127-161) -> obj [batch] (may be NULL).  eval_g / eval_jac_g are covered by
 * corbo_hip_eval_linear_form: the constraint values are -lbA (equalities) and -ubA (inequalities), the Jacobian list of
 * computeCombinedSparseJacobiansValues(false, true, true) is the linear form's value list without the bound rows, same order. */
int corbo_hip_eval_objective_gradient(corbo_hip_handle h, double* grad, double* obj);
/* lbA <= A dx <= ubA with the finite bounds as identity rows: structure (rows / cols may be NULL for the size query), then
 * vals [batch][nnz], lbA / ubA [batch][n_rows].  (ubA of a bound row is x - ub, as the reference computes it.) */
/* The same lists WITHOUT the copy into caller arrays (a batch of 1024 cfg-3 instances returns 36 MB: the copy, not the kernel, is what a
 * call costs): device_views = 0 -> views into pinned host memory owned by the handle (one PCIe transfer by a copy kernel), device_views = 1 ->
 * pointers into the handle's HBM buffers (no transfer at all: a solver that lives on the GPU).  [batch][nnz] each, valid until the next
 * Hessian-path call on the handle; a NULL view pointer skips that list, an empty list yields NULL. */
int corbo_hip_eval_hessians_views(corbo_hip_handle h, int lower_part_only, double mult_obj, const double* mult_eq, const double* mult_ineq, int device_views,
                                  const double** vals_obj, const double** vals_eq, const double** vals_ineq);
int corbo_hip_linear_form_structure(const corbo_hip_problem_desc* desc, int32_t* nnz_out, int32_t* n_rows_out, int32_t* rows, int32_t* cols);
int corbo_hip_eval_linear_form(corbo_hip_handle h, double* vals, double* lbA, double* ubA);

/* Device-resident views for callers that already live on the GPU (torch tensors, RCCL gathers): pointers into
 * the library's HBM buffers, valid until corbo_hip_destroy.  x: [batch][nv], chi2: [batch]. */
int corbo_hip_device_views(corbo_hip_handle h, double** x_dev, double** chi2_dev, void** hip_stream);
/* Row stride (doubles) of the x view: dims.nv values + the library's padding (a fixed dt lives there too). */
int corbo_hip_device_row_stride(corbo_hip_handle h, int32_t* row_stride);

/* Launch only the edge/Jacobian sweep kernel `repeat` times on the resident data and return the average
 * per-launch time in ms measured with HIP events on the handle's stream (bench.py roofline leg). */
int corbo_hip_time_sweep(corbo_hip_handle h, double w_eq, double w_ineq, double w_bounds, int with_jacobian, int repeat,
                         float* ms_per_launch);

/* The same launches, each bracketed by its own pair of HIP events on the handle's stream (the launches do not overlap): ms_each
 * [repeat] = the duration of every single launch, the quantity a rocprofv3 kernel trace reports per dispatch. */
int corbo_hip_time_sweep_each(corbo_hip_handle h, double w_eq, double w_ineq, double w_bounds, int with_jacobian, int repeat,
                              float* ms_each);

/* Same for the assemble/factor/solve kernel: runs the LM prologue on the resident data, then launches the kernel `repeat`
 * times.  timeline8 (may be NULL) receives 8 shader-clock stamps of workgroup 0 taken at the kernel's phase boundaries
 * (load | assemble+control elimination | state blocks | cyclic reduction | root | back-substitution | step) -- diagnostics. */
int corbo_hip_time_factor(corbo_hip_handle h, int repeat, float* ms_per_launch, long long* timeline8);

/* Per-kernel HIP-event timing inside corbo_hip_solve (fills corbo_hip_stats.sweep_ms / factor_ms); off by default. */
int corbo_hip_set_profiling(corbo_hip_handle h, int enable);

/* f(x, u) of the descriptor's dynamics (SystemDynamicsInterface::dynamics, system_dynamics_interface.h:121) as the DEVICE evaluates
 * it, for n points: x [n][nx], u [n][nu] -> f [n][nx] (host arrays).  Only dynamics, nx, nu, dyn_params (and lin_a / lin_b) of
 * the descriptor are read.  Runs on the current HIP device.  Lets a caller check that a dynamics object it holds is the model a
 * descriptor names (the adapter's recogniser matches user systems against the library's plug-in models with it). */
int corbo_hip_eval_dynamics(const corbo_hip_problem_desc* desc, int n, const double* x, const double* u, double* f);

/* User stage functions (csrc/stage_functions/): kind of a registered id -- 0 state inequality, 1 control inequality, -1 = not registered -- and the value
 * c(v, prm) of such a function (or of CORBO_HIP_INEQ_BALL) at n points, evaluated on the HOST by the very template the kernels compile: v [n][dim], out [n].
 * Lets a caller check that a stage-function object it holds is the function an id names (the adapter's recogniser matches a graph's inequality edges with it,
 * like corbo_hip_eval_dynamics for a dynamics object).  Needs no GPU. */
int corbo_hip_stage_function_kind(int id);
int corbo_hip_eval_stage_function(int id, int dim, int n, const double* v, const double* prm, double* out);

/* Diagnostics and test hooks of a handle (nothing here is read from the environment):
 *   "pass_limit"        value > 0: the run-to-completion kernel gives up after `value` LM passes per instance instead of 4096
 *   "run_to_completion" 0: one launch per LM pass (the host counts unfinished instances) instead of one launch per solve
 *   "pass_timeline"     value >= 0: corbo_hip_solve prints the per-pass shader-clock stamps of instance `value` on stderr; -1: off
 *   "sweep_timeline"    1: corbo_hip_time_sweep prints the phase stamps of instance 0 on stderr
 *   "solve_timing"      0: corbo_hip_solve records no HIP timing events around its launches (corbo_hip_stats::solve_ms and corbo_hip_get_timing stay 0):
 *                       7 - 13 us less per call, what a caller that solves ONE problem per call wants (the drop-in adapter sets it)
 *   "ff_converged"      0: compute the outer iterations that follow a converged step instead of counting them (A/B and tests)
 *   "lag_priority"      0: no lag-based issue priority in the run-to-completion kernel
 *   "phase_cycles"      1: per-instance phase totals of the run-to-completion kernel (corbo_hip_get_phase_cycles)
 *   "raw_stamps"        1: "pass_timeline" prints raw stamp offsets (development builds that re-purpose the stamp slots)
 *   "band_wide"         1: the band route (integral-form constraint edges / control-deviation term) keeps the eight-wave factor kernel for every
 *                       half-bandwidth; 0 (default): half-bandwidths up to 7 take the one-wave-per-instance kernel
 *   "bt_waves"          2 / 3: the block-tridiagonal route's kernel for two / three workgroups per CU whatever the batch size; 0 (default): chosen by the
 *                       number of rounds the batch needs (A/B) */
int corbo_hip_set_option(corbo_hip_handle h, const char* name, int value);

/* Text of the last error on this thread. */
const char* corbo_hip_last_error(void);

/* Layout check for bindings that mirror the PODs above by hand (ctypes, cgo, JNI ...): sizeof of what THIS library was compiled with --
 * which = 0: corbo_hip_problem_desc, 1: corbo_hip_dims, 2: corbo_hip_lm_opts, 3: corbo_hip_stats; anything else: 0.  A binding compares
 * it with its own mirror when it loads the library and refuses on a mismatch (control_box_rst_amd/capi.py does). */
size_t corbo_hip_sizeof(int which);

#ifdef __cplusplus
}
#endif
#endif /* CORBO_HIP_H_ */
