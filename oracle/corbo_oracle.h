/*
 * corbo_oracle.h -- TEST INFRASTRUCTURE ONLY.  CPU restatement (plain C99) of the reference's hypergraph
 * NLP inner loop.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this.
 * The product (control_box_rst_amd/) never links, imports or calls it.
 *
 * Parity status: PINNED -- checked against golden vectors produced by the genuine reference compiled from
 * /root/reference (oracle/Makefile target `ref`, generator oracle/gen_golden.py, fixtures tests/golden/*.json).
 *
 * The descriptor PODs are shared with the product's public header (types only, no code).
 */
#ifndef CORBO_ORACLE_H_
#define CORBO_ORACLE_H_

#include "../include/corbo_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct oracle_problem oracle_problem;

/* per-outer-iteration trace of LevenbergMarquardtSparse::solve (for tests) */
typedef struct oracle_trace_entry {
    int32_t k;            /* outer iteration */
    int32_t inner_passes; /* factorisations in this outer iteration */
    int32_t accepted;     /* 1 if the last inner pass was accepted */
    double mu;            /* damping after the iteration */
    double rho;           /* last gain ratio */
    double chi2;          /* accepted chi2 after the iteration */
    double delta_norm;    /* |delta| of the last inner pass */
} oracle_trace_entry;

/* Build the hypergraph of one OCP instance (vertices + edges in the reference's creation order). NULL if invalid. */
oracle_problem* oracle_create(const corbo_hip_problem_desc* desc);
void oracle_destroy(oracle_problem* p);

int oracle_get_dims(const oracle_problem* p, corbo_hip_dims* dims);
/* (row, col) of every structural non-zero of the combined Jacobian, in the oracle's value order */
int oracle_get_structure(const oracle_problem* p, int32_t* rows, int32_t* cols);

/* FullDiscretizationGridBase::initializeSequences (full_discretization_grid_base.cpp:134-179) */
int oracle_init_trajectory(const corbo_hip_problem_desc* desc, const double* x0, const double* xf, double* x_out);

/* vertex values / bounds in vertex layout (nv doubles); lb/ub NULL = descriptor box bounds; xref NULL = zeros */
int oracle_set_data(oracle_problem* p, const double* x, const double* lb, const double* ub, const double* xref);
/* the previously applied control and its age (setPreviousControlInput): the fixed vertices the control-deviation edge of interval 0 sees */
int oracle_set_previous_control(oracle_problem* p, const double* u_prev, double dt_prev);
/* time-varying state reference: one per vertex component (state reference at grid point k in the x_k entries, the final-stage terms
 * use the x_f entries; control / dt entries are not read); NULL = the static reference of oracle_set_data */
int oracle_set_references(oracle_problem* p, const double* ref);
int oracle_warm_start(oracle_problem* p, const double* x0, int shift);
int oracle_get_x(const oracle_problem* p, double* x_out);
/* SimulatedPlant::control without dead time (plants/src/simulated_plant.cpp:97-160): x_plant <- integrator.solveIVP(x_plant, u_0 of
 * the stored trajectory, dt), then x_plant += disturbance (may be NULL).  integrator: corbo_hip_integrator.  x_plant: nx doubles, in/out. */
/* resampleTrajectory (full_discretization_grid_base.cpp:397-474) on the vertex layout of a free-dt grid; the grid-adaptation rules of
 * FiniteDifferencesVariableGrid (finite_differences_variable_grid.cpp:101-163) */
int oracle_resample_trajectory(int nx, int nu, int n, const double* x_old, int n_new, double* x_new);
int oracle_adapt_grid_n(int strategy, int n, double dt, double dt_ref, double hyst, int n_min, int n_max);
/* sizeof(corbo_hip_problem_desc) this checker was compiled with (oracle.py rebuilds a stale liboracle.so on a mismatch) */
int oracle_sizeof_problem_desc(void);
int oracle_plant_step(const oracle_problem* p, int integrator, double dt, const double* disturbance, double* x_plant);

/* Callback problem (SimpleOptimizationProblemWithCallbacks): n parameters, f fills the lsq / equality / inequality value vectors at x
 * (any of them may have dimension 0).  lb / ub NULL = unbounded.  Runs through the same oracle_solve() as the OCPs; used to pin the LM
 * loop against the known-answer cases of the reference's own solver test.  set_data / get_x / eval / solve work as for an OCP (nv = n). */
typedef void (*oracle_generic_fun)(const double* x, double* lsq, double* eq, double* ineq);
oracle_problem* oracle_create_generic(int n, int dim_lsq, int dim_eq, int dim_ineq, const double* lb, const double* ub, oracle_generic_fun f);

/* LevenbergMarquardtSparse::computeValues + computeCombinedSparseJacobian at the current x (jac may be NULL) */
int oracle_eval(oracle_problem* p, double w_eq, double w_ineq, double w_bounds, double* values, double* jac);

/* Operators of the exact-Hessian path at the current x.  Three triplet lists -- objective, equalities, inequalities -- exactly as
 * computeSparseHessians{NNZ,Structure,Values} (hyper_graph_optimization_problem_edge_based.cpp:2087-3760) produce them, entry order
 * included; lower_part_only as IpoptWrapper::eval_h passes it (true).  Multipliers: one per equality / inequality row (NULL = 1). */
/* offset in the vertex layout of every optimisation parameter (n entries) */
int oracle_get_param_offsets(const oracle_problem* p, int32_t* off_out);
int oracle_hessian_nnz(oracle_problem* p, int lower_part_only, int32_t nnz_out[3]);
int oracle_hessian_structure(oracle_problem* p, int lower_part_only, int32_t* rows_obj, int32_t* cols_obj, int32_t* rows_eq, int32_t* cols_eq,
                             int32_t* rows_ineq, int32_t* cols_ineq);
int oracle_hessian_values(oracle_problem* p, int lower_part_only, double mult_obj, const double* mult_eq, const double* mult_ineq,
                          double* vals_obj, double* vals_eq, double* vals_ineq);
/* lbA <= A dx <= ubA: computeSparseJacobianTwoSideBoundedLinearForm* with the finite bounds (:4762-4968) and
 * computeBoundsForTwoSideBoundedLinearForm (optimization_problem_interface.cpp:1141-1183).  Any output may be NULL. */
/* eval_grad_f / eval_f of the interior-point interface: computeGradientObjective (n entries) and computeValueObjective */
int oracle_objective_gradient(oracle_problem* p, double* grad, double* obj_out);
int oracle_linear_form(oracle_problem* p, int32_t* nnz_out, int32_t* rows, int32_t* cols, double* vals, double* lbA, double* ubA);

/* LevenbergMarquardtSparse::solve.  Returns corbo_hip_solver_status; *chi2_out = *obj_value.
 * trace (may be NULL) must hold opts->iterations entries. */
int oracle_solve(oracle_problem* p, const corbo_hip_lm_opts* opts, int new_run, double* chi2_out, oracle_trace_entry* trace);

/* Convenience for the CPU baseline: solve `batch` instances one after another on the calling thread.
 * x: [batch][nv] in/out, xref: [batch][nx], chi2_out: [batch], status_out: [batch]. */
int oracle_solve_batch(const corbo_hip_problem_desc* desc, int batch, double* x, const double* xref, const corbo_hip_lm_opts* opts,
                       double* chi2_out, int32_t* status_out);

#ifdef __cplusplus
}
#endif
#endif
