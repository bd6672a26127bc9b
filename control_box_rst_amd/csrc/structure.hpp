// structure.hpp -- static structure of the hypergraph NLP described by a corbo_hip_problem_desc.
//
// Host-side, computed once per handle ("new_structure" work of the reference:
// LevenbergMarquardtSparse::solve, levenberg_marquardt_sparse.cpp:48-80; OptimizationEdgeSet::computeEdgeIndices,
// edge_set.cpp:31-42; VertexSetInterface::computeVertexIndices, vertex_set.cpp:405-418;
// computeSparseJacobian*NNZ / *Structure, hyper_graph_optimization_problem_edge_based.cpp:193-322,1181-1478).
// The device kernels are table driven: they never see edges as objects, only the flat task tables built here.
#pragma once

#include <cstdint>
#include <string>
#include <vector>

#include "../../include/corbo_hip.h"

namespace corbo_hip {

// Big-block family (kernels.hip, BigLds: stage kernels + chain kernel, factor workspace in HBM; multiple shooting with RK4, fixed dt):
// the dimensions it is built for -- one lane per (column, side) of an interval's finite differences (nx + nu <= 16), the diag pass parks
// 4 nx + 2 numbers in the nx x nx factor slot (nx >= 5), the controls are eliminated by one lane (nu <= 4).  The quadrotor (12, 4) and
// every user model (csrc/models/) of that size.
constexpr bool big_family_dims(int nx, int nu) { return nx >= 5 && nx <= 12 && nu >= 1 && nu <= 4 && nu <= nx && nx + nu <= 16; }

// edge kinds the device can evaluate (closed set, DESIGN.md "device-describable edges")
enum EdgeKind : int32_t {
    EK_STATE_COST   = 0,  // sqrt(Q) .* (x_k - xref)            quadratic_cost.cpp:100-119
    EK_CONTROL_COST = 1,  // sqrt(R) .* u_k                     quadratic_cost.cpp:140-154
    EK_FINAL_COST   = 2,  // sqrt(Qf) .* (x_f - xref)           final_state_cost.cpp:72-92
    EK_DT_COST      = 3,  // sqrt(N-1) * dt                     minimum_time.h:49-78
    EK_DEFECT       = 4,  // dynamics defect (x_k,u_k,x_{k+1},dt)
    EK_STAGE_INEQ   = 5,  // stage inequality on x_k
    EK_FINAL_INEQ   = 6,  // final-stage inequality on x_f (TerminalBall)
    EK_FINAL_EQ     = 7,  // final-stage equality x_f - xref (TerminalEqualityConstraint)
    // plain objective edges (cost_nonlsq: QuadraticFormCost / QuadraticFinalStateCost with lsq_form = false), one scalar term each;
    // Hessian-path operators only
    EK_STATE_QCOST   = 8,
    EK_CONTROL_QCOST = 9,
    EK_FINAL_QCOST   = 10,
    EK_DT_QCOST      = 11,  // MinimumTime(lsq_form = false): (N - 1) dt (minimum_time.h:60), not flagged linear
    // QuadraticFormCost in integral form: one objective edge per interval (finite_differences_collocation_edges.h:98-152, 323-368)
    EK_INTEGRAL_TRAP = 12,  // TrapezoidalIntegralCostEdge on (x_k, u_k, x_{k+1}, dt)
    EK_INTEGRAL_LEFT = 13,  // LeftSumCostEdge on (x_k, u_k, dt)
    // MultipleShootingEdgeSingleControl on (x_k, u_k, dt, x_{k+1}) (multiple_shooting_edges.h:151-303): a mixed edge -- what a MultipleShootingGrid
    // creates instead of the dynamics-only edge when the stage cost has integral terms (multiple_shooting_grid.cpp:70-77)
    EK_MIXED_OBJ = 14,      // objective part: the cost integrated along the shooting step (1 value)
    EK_MIXED_EQ  = 15,      // equality part: the defect (nx values)
    EK_MIXED_JOINT = 16,    // both parts as one value vector [objective; equalities] (the Jacobian of the Hessian walk: one perturbation cycle for both)
    // integral-form constraints and the control-deviation term (user stage functions; corbo_hip_problem_desc::constraint_integration ...)
    EK_XI_INEQ    = 17,     // TrapezoidalIntegralInequalityEdge (x_k, u_k, x_{k+1}, dt) / LeftSumInequalityEdge (x_k, u_k, dt)
    EK_XI_EQ_LEFT = 18,     // LeftSumEqualityEdge (x_k, u_k, dt)
    EK_XI_EQ_ROW  = 19,     // the integral row the trapezoidal rule appends to the dynamics edge (TrapezoidalIntegralEqualityDynamicsEdge: dimension nx + 1)
    EK_CTRL_DEV   = 20,     // TernaryVectorScalarVertexEdge<computeNonIntegralControlDeviationTerm> (u_k, u_prev, dt_prev)
    EK_U_INEQ     = 21      // UnaryVectorVertexEdge<computeNonIntegralControlTerm> on u_k: a user stage inequality's control term (csrc/stage_functions/)
};

// One "extra" edge (the kinds above), evaluated by a lane of the sweep kernel's generic loop (sweep_body, XE): attached vertices with their storage
// offsets (voff < 0: the grid's always-fixed vertices -- -1 _u_prev, -2 _u_ref, -3 _u_prev_dt, full_discretization_grid_base.cpp:509-511), the
// residual row of its first value and, per vertex, the Jacobian value offset of its block (columns of the unfixed components only, column stride
// = the EDGE's dimension `edim`, this entry's first row inside the edge = `rie`).
struct XEdge {
    int32_t kind, k, row, dim, edim, rie, nverts, scale;
    int32_t voff[4], vdim[4], joff[4];
    uint32_t fixed[4];
};

// per-stage view of the Jacobian for the assembly of H = J^T J (levenberg_marquardt_sparse.cpp:97-100):
// defect edge k has the dense local Jacobian [A | B | C | d] w.r.t. (x_k, u_k, x_{k+1}, dt); entry = offset of the
// column's first value in the Jacobian value array, -1 when that component is fixed.
struct StageCols {
    int32_t col[CORBO_HIP_MAX_NX + CORBO_HIP_MAX_NU + CORBO_HIP_MAX_NX + 1];
};

// per component of the vertex storage: diagonal contributions to H (cost row, bound row, inequality row)
struct CompInfo {
    int32_t fixed;      // 1 = not a parameter
    int32_t param;      // parameter (column) index or -1
    int32_t cost_joff;  // Jacobian value index of d(cost row)/d(component) (diagonal of the cost block) or -1
    int32_t cost_row;   // residual row of that cost value or -1
    int32_t bnd_joff;   // bound row entry or -1
    int32_t bnd_row;
    int32_t cost2_joff; // second cost row on the same component (duplicated dt edge, nlp_functions.cpp:91-107) or -1
    int32_t cost2_row;
};

struct Structure {
    corbo_hip_problem_desc desc{};
    corbo_hip_dims dims{};
    int nx = 0, nu = 0, N = 0, s = 0;
    bool dt_free = false;
    int off_xf = 0, off_dt = 0;  // vertex-storage offsets
    int nvs = 0;                 // device vertex storage per instance (always holds dt), padded to even
    int eq_row0 = 0, ineq_row0 = 0, bnd_row0 = 0;
    int defect_joff0 = 0;        // first Jacobian value of the equality blocks
    double sq[CORBO_HIP_MAX_NX]{}, sr[CORBO_HIP_MAX_NU]{}, sqf[CORBO_HIP_MAX_NX]{};
    double dt_weight = 0;

    std::vector<StageCols> stage_cols;   // N-1 defect edges
    std::vector<int32_t> ineq_cols;      // (N-1)*nx: Jacobian value index of d(ineq_k)/d(x_k[i]) or -1
    std::vector<int32_t> ineq_rows;      // N-1 residual rows (or empty)
    int fin_row = -1;                    // residual row of the final-stage inequality or -1
    int fin_eq_row0 = -1, fin_eq_dim = 0; // first row / number of rows of the final-stage equality (TerminalEqualityConstraint: nx, partial: active components)
    int fin_joff[CORBO_HIP_MAX_NX];      // Jacobian value index of d(final ineq)/d(x_f[i]) or -1
    std::vector<CompInfo> comp;          // nvs entries
    std::vector<int32_t> jac_rows, jac_cols;  // structure in value order
    std::vector<int32_t> param_voff;     // parameter -> vertex storage offset
    std::vector<XEdge> xedges;           // integral-form constraint edges / control-deviation edges (empty for every other descriptor)
    int eq_stride = 0;                   // residual rows per interval in the equality section (nx; nx + 1 with an integral equality row)
    int eq_defect_off = 0;               // row of the dynamics defect inside the interval's equality rows (1 behind a LeftSumEqualityEdge)
    bool has_extra() const { return !xedges.empty(); }

    int x_off(int k) const { return k * s; }             // k in [0, N-1]; k == N-1 is x_f
    int u_off(int k) const { return k * s + nx; }
};

// Exact-Hessian path: the three triplet lists of computeSparseHessians{NNZ,Structure,Values}
// (hyper_graph_optimization_problem_edge_based.cpp:2087-3760) -- objective, equalities, inequalities -- in the reference's entry order,
// and the two-side-bounded linear form with the finite bounds (:4762-4968).
struct HessianStructure {
    int32_t nnz[3] = {0, 0, 0};
    std::vector<int32_t> rows[3], cols[3];
    // per stage k (N entries, k = N-1 = final stage), six numbers: first value of the stage's edges in the lists, -1 = no such edge:
    //   [0] objective: state cost (final stage: final cost)   [1] objective: control cost
    //   [2] equalities: defect edge (final stage: terminal equality)   [3] inequalities: stage inequality (final stage: terminal inequality)
    //   [4] first equality row of [2]   [5] first inequality row of [3]   (multiplier / linear-form row indices)
    // shooting grid with an integral-form cost (mixed edges): [0] / [2] of an interval = first value of the mixed edge's objective / equality blocks
    std::vector<int32_t> stage_off;
    int32_t dt_cost_off = -1;       // objective: first value of the two dt cost edges (they follow stage 0's state / control terms), -1 = none
    int32_t lin_nnz = 0, lin_bounds0 = 0;
    std::vector<int32_t> lin_rows, lin_cols;
    std::vector<int32_t> lin_off;   // [N][2]: first linear-form value of the stage's equality edge / inequality edge, -1 = none
};
void build_hessian_structure(const Structure& S, bool lower_part_only, HessianStructure& out);

// Block-tridiagonal route of the small-block families with extra edges (kernels.hip, bt_factor.hpp): H = J^T J in the stage
// blocks z_k = (x_k, u_k) (S = nx + nu rows; the last block: x_f, padded), a free dt as a border.  The dynamics-defect edges are assembled by the kernel
// from their dense local Jacobians; what every OTHER row of J adds to an entry -- lower part of the diagonal blocks D_k, the couplings
// F_k = H(block k + 1, block k), the right-hand side, the border -- is a sum of products of two operands of
// the array [J (nnz_pad) | values (m_pad) | 0]; the entries are sorted by list length and dealt out in SUPER-ROUNDS of 4 * `threads` entries (four per
// lane) whose lists are padded to the super-round's longest, a multiple of four (ELL layout: the two operands' BYTE OFFSETS of product i of the lane's entry c at
// pairs[((off[sr] + i) * threads + lane) * 8 + 2 c + {0, 1}], padding = the zero operand; four more steps of padding behind the last one).
struct BtTables {
    int S = 0, NB = 0, szp = 0, threads = 0, rounds = 0;
    bool arrow = false;
    std::vector<uint32_t> pairs;     // operand byte offsets
    std::vector<int32_t> off;        // [rounds + 1] steps (one step = one 16-byte group of four pairs per lane); rounds = super-rounds
    std::vector<uint32_t> target;    // [rounds * threads * 4] (lane-major groups of four): LDS slot (bits 0..27) | flags -- bit 28 diagonal of a parameter (+ damping), 29 identity
                                     // row (fixed component / pad: the value is 1), 30 right-hand side (negated sum), 31 corner H(dt, dt); no entry: the trash slot
};
// false (with a reason) when H is not block tridiagonal in the stage blocks
bool build_bt_tables(const Structure& S, const std::vector<int32_t>& jmap, int nnz_pad, int m_pad, int threads, BtTables& out, std::string* why);

// returns "" on success, otherwise an error text
std::string validate_desc(const corbo_hip_problem_desc& d);
std::string build_structure(const corbo_hip_problem_desc& d, Structure& out);
void init_trajectory(const corbo_hip_problem_desc& d, int batch, const double* x0, const double* xf, double* x_out);

}  // namespace corbo_hip
