/*
 * corbo_oracle.c -- TEST INFRASTRUCTURE ONLY (see corbo_oracle.h).
 *
 * Plain-C restatement of the reference's hypergraph NLP inner loop, operation by operation.  Every function cites
 * the reference code it follows (paths relative to /root/reference/src).  It keeps the reference's data model:
 * vertices (value storage + fixed mask + bounds), edges (attached vertices, computeValues), central-difference
 * block Jacobians that perturb the shared vertex storage in place, the stacked residual
 * [lsq | w_eq*eq | w_ineq*max(0,ineq) | w_b*bounds], J^T J and the Levenberg-Marquardt loop with all of its quirks.
 * Differences to the reference that are rounding-level only (documented in DESIGN.md):
 *   - squared norms / dot products are summed left to right (Eigen uses packet-wise partial sums);
 *   - the linear system is solved by an envelope (skyline) Cholesky in natural order (Eigen: SimplicialLLT + AMD).
 * Parity status: PINNED against tests/golden/ (genuine reference outputs), see tests/test_oracle_golden.py.
 */
#include "corbo_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------------------------------------------------ */
/* hyper-graph data model (optimization/include/corbo-optimization/hyper_graph/{vector_vertex,scalar_vertex}.h)  */

typedef struct {
    int off;        /* offset of the values in the vertex storage */
    int dim;
    unsigned fixed; /* bit i = component i fixed (PartiallyFixedVectorVertex, vector_vertex.h:276-446) */
    int n_unfixed;
    int col;        /* getVertexIdx(): first parameter index, -1 when not active (vertex_set.cpp:405-418) */
} o_vertex;

enum { E_STATE_COST, E_CONTROL_COST, E_FINAL_COST, E_DT_COST, E_DEFECT, E_STAGE_INEQ, E_FINAL_INEQ, E_FINAL_EQ, E_INTEGRAL_COST,
       E_MS_MIXED_OBJ, E_MS_MIXED_EQ, /* the objective part and the equality part of one MultipleShootingEdgeSingleControl (a BaseMixedEdge) */
       E_INT_INEQ,     /* TrapezoidalIntegralInequalityEdge (x1, u1, x2, dt) / LeftSumInequalityEdge (x1, u1, dt) */
       E_INT_EQ_LEFT,  /* LeftSumEqualityEdge (x1, u1, dt); the trapezoidal rule appends its row to the dynamics edge instead (E_DEFECT with dim = nx + 1) */
       E_CTRL_DEV,     /* TernaryVectorScalarVertexEdge<computeNonIntegralControlDeviationTerm> (u_k, u_prev, dt_prev) */
       E_U_INEQ };     /* UnaryVectorVertexEdge<computeNonIntegralControlTerm> (u_k): a user stage inequality's control term */

typedef struct {
    int type;
    int k;        /* stage index */
    int nverts;
    int vert[4];  /* attached vertices in the reference's order, e.g. defect: (x1,u1,x2,dt) */
    int dim;      /* getDimension() */
    int row;      /* global row of the first value in the stacked residual */
    int scale;    /* 0 lsq (unscaled), 1 equality (w_eq), 2 inequality (w_ineq, active set) */
    int nonlsq;   /* objective edge that is NOT in least-squares form (scalar term; Hessian-path operators only: row = -1, no LM rows) */
    int partner;  /* E_MS_MIXED_OBJ: index of the E_MS_MIXED_EQ part of the same mixed edge (and vice versa); else -1 */
} o_edge;

typedef struct { /* one (edge, vertex) Jacobian block, values stored column-major like Eigen::MatrixXd */
    int edge, vtx_idx;
    int row0, col0, rows, cols;
    int off; /* offset into the Jacobian value array */
} o_block;

struct oracle_problem {
    corbo_hip_problem_desc d;
    corbo_hip_dims dims;
    int n_vertices, n_edges, n_blocks;
    int has_nonlsq; /* objective edges that are not in least-squares form: LevenbergMarquardtSparse refuses the problem (:48-55), so do the LM entries here */
    o_vertex* v;
    o_edge* e;
    o_block* b;
    int first_bound_nnz;  /* offset of the bound entries in the Jacobian value array */
    int* bound_vert_off;  /* per bound row: offset of the component in the vertex storage */
    int* bound_col;       /* per bound row: parameter index */
    int* param_off;       /* per parameter: offset in the vertex storage (applyIncrementNonFixed, vertex_set.cpp:357-367) */
    double *x, *lb, *ub, *xref, *backup;
    double* refvec; /* NULL, or one reference per vertex component (time-varying references, oracle_set_references) */
    double sq[CORBO_HIP_MAX_NX], sr[CORBO_HIP_MAX_NU], sqf[CORBO_HIP_MAX_NX]; /* cwiseSqrt of the weights */
    double dt_weight;
    /* static row-wise view of J (for J^T J in Eigen's summation order) */
    int* csr_ptr;  /* m+1 */
    int* csr_col;  /* nnz */
    int* csr_val;  /* nnz: index into the value array */
    /* envelope storage of H */
    int* env_first;  /* n: first structurally non-zero column of row i */
    int* env_ptr;    /* n+1 */
    int env_size;
    /* LM work space */
    double *values, *jac, *H, *L, *rhs, *delta, *tmp;
    double w_eq, w_ineq, w_b; /* current penalty weights (levenberg_marquardt_sparse.h:126-128) */
    oracle_generic_fun gen;   /* != NULL: a callback problem (oracle_create_generic) instead of a hypergraph */
};

/* ------------------------------------------------------------------------------------------------------------ */
/* dynamics (SystemDynamicsInterface::dynamics)                                                                  */

static void dynamics(const corbo_hip_problem_desc* d, const double* x, const double* u, double* f)
{
    switch (d->dynamics) {
        case CORBO_HIP_DYN_VAN_DER_POL: { /* systems/include/corbo-systems/benchmark/nonlinear_benchmark_systems.h:52-60 */
            double a = d->dyn_params[0];
            f[0]     = x[1];
            f[1]     = -a * (x[0] * x[0] - 1) * x[1] - x[0] + u[0];
            break;
        }
        case CORBO_HIP_DYN_SERIAL_INTEGRATOR: { /* linear_benchmark_systems.h:72-83 */
            int n = d->nx;
            for (int i = 0; i < n - 1; ++i) f[i] = x[i + 1];
            f[n - 1] = u[0] / d->dyn_params[0];
            break;
        }
        case CORBO_HIP_DYN_USER + 0: { /* user model csrc/models/kinematic_car.hpp = class KinematicCarRef of oracle/ref_driver.cpp (scenario kcar) */
            f[0] = u[0] * cos(x[2]);
            f[1] = u[0] * sin(x[2]);
            f[2] = u[0] / d->dyn_params[0] * tan(u[1]);
            break;
        }
        case CORBO_HIP_DYN_USER + 1: { /* user model csrc/models/planar_quadrotor.hpp = class PlanarQuadrotorRef of oracle/ref_driver.cpp (scenario pquad) */
            double m = d->dyn_params[0], I = d->dyn_params[1], l = d->dyn_params[2], g = d->dyn_params[3];
            double T = u[0] + u[1];
            f[0] = x[3];
            f[1] = x[4];
            f[2] = x[5];
            f[3] = -(T * sin(x[2])) / m;
            f[4] = (T * cos(x[2])) / m - g;
            f[5] = (u[0] - u[1]) * l / I;
            break;
        }
        case CORBO_HIP_DYN_UNICYCLE: { /* user plug-in, same formula as oracle/ref_driver.cpp */
            f[0] = u[0] * cos(x[2]);
            f[1] = u[0] * sin(x[2]);
            f[2] = u[1];
            break;
        }
        case CORBO_HIP_DYN_QUADROTOR: { /* user plug-in, DESIGN.md "quadrotor"; same formula as oracle/ref_driver.cpp */
            double g = d->dyn_params[0], m = d->dyn_params[1], Ixx = d->dyn_params[2], Iyy = d->dyn_params[3], Izz = d->dyn_params[4];
            double sphi = sin(x[6]), cphi = cos(x[6]), sth = sin(x[7]), cth = cos(x[7]), spsi = sin(x[8]), cpsi = cos(x[8]);
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
            break;
        }
        /* the reference's other benchmark systems (nonlinear_benchmark_systems.h) */
        case CORBO_HIP_DYN_DUFFING: { /* :108-115 */
            double damping = d->dyn_params[0], alpha = d->dyn_params[1], beta = d->dyn_params[2];
            f[0] = x[1];
            f[1] = -damping * x[1] - alpha * x[0] - beta * x[0] * x[0] * x[0] + u[0];
            break;
        }
        case CORBO_HIP_DYN_FREE_SPACE_ROCKET: /* :174-183 */
            f[0] = x[1];
            f[1] = (u[0] - 0.02 * x[1] * x[1]) / x[2];
            f[2] = -0.01 * u[0] * u[0];
            break;
        case CORBO_HIP_DYN_SIMPLE_PENDULUM: { /* :207-215 */
            double m = d->dyn_params[0], l = d->dyn_params[1], g = d->dyn_params[2], rho = d->dyn_params[3];
            f[0] = x[1];
            f[1] = u[0] - rho / (m * l * l) * x[1] - g / l * sin(x[0]);
            break;
        }
        case CORBO_HIP_DYN_MASSLESS_PENDULUM: /* :281-289 */
            f[0] = x[1];
            f[1] = u[0] - d->dyn_params[0] * sin(x[0]);
            break;
        case CORBO_HIP_DYN_TOY_EXAMPLE: { /* :426-436 */
            double mu = d->dyn_params[0];
            f[0] = x[1] + u[0] * (mu + (1.0 - mu) * x[0]);
            f[1] = x[0] + u[0] * (mu - 4.0 * (1.0 - mu) * x[1]);
            break;
        }
        case CORBO_HIP_DYN_ARTSTEINS_CIRCLE: /* :483-491 */
            f[0] = (x[0] * x[0] - x[1] * x[1]) * u[0];
            f[1] = 2 * x[0] * x[1] * u[0];
            break;
        case CORBO_HIP_DYN_LINEAR_STATE_SPACE: /* linear_benchmark_systems.h:206-213: f = A x + B u, evaluated by Eigen as one running sum
                                                * per row (dst = A x; dst += B u; column-major gemv, column by column) */
            for (int i = 0; i < d->nx; ++i) {
                double acc = 0.0;
                if (d->nx == 4) { /* a full block of four columns: Eigen's gemv kernel adds it pairwise (GeneralMatrixVector.h) */
                    const double* a = d->lin_a + i * 4;
                    acc = (a[0] * x[0] + a[1] * x[1]) + (a[2] * x[2] + a[3] * x[3]);
                }
                else
                    for (int j = 0; j < d->nx; ++j) acc += d->lin_a[i * d->nx + j] * x[j];
                for (int j = 0; j < d->nu; ++j) acc += d->lin_b[i * d->nu + j] * u[j];
                f[i] = acc;
            }
            break;
        case CORBO_HIP_DYN_PARALLEL_INTEGRATOR: /* linear_benchmark_systems.h:142-148 */
            for (int i = 0; i < d->nx; ++i) f[i] = d->dyn_params[0] * u[i];
            break;
        case CORBO_HIP_DYN_CART_POLE: { /* :337-355, state [x phi xdot phidot], the reference's fixed parameters */
            const double mc = 1.0, mp = 0.3, l = 0.5, g = 9.81;
            double sin_phi_phidot_sq = sin(x[1]) * x[3] * x[3];
            double denum             = mc + mp * (1 - cos(x[1]) * cos(x[1])); /* std::pow(cos, 2): exactly cos * cos */
            f[0] = x[2];
            f[1] = x[3];
            f[2] = (l * mp * sin_phi_phidot_sq + u[0] + mp * g * cos(x[1]) * sin(x[1])) / denum;
            f[3] = -(l * mp * cos(x[1]) * sin_phi_phidot_sq + u[0] * cos(x[1]) + (mp + mc) * g * sin(x[1])) / (l * denum);
            break;
        }
        default: break;
    }
}

/* ------------------------------------------------------------------------------------------------------------ */
/* edge values                                                                                                    */

/* sum of the n terms of a quadratic form  x^T W_diag x  in the order the reference's arithmetic takes: the inner product is Eigen's
 * (lhs.transpose().cwiseProduct(rhs)).sum() (GeneralProduct, InnerProduct), a vectorised reduction with packets of two doubles (the
 * reference is compiled for baseline x86-64: SSE2) and two packet accumulators -- src/extern/eigen3/Eigen/src/Core/Redux.h,
 * redux_impl<Func, Derived, LinearVectorizedTraversal, NoUnrolling>: ((t0 + t4 + ..) + (t2 + t6 + ..)) lane by lane, a leftover packet, the two
 * lanes added, then the odd element.  Up to three terms this is the plain left-to-right sum; from four terms on the rounding differs. */
static double eigen_sum(const double* t, int n)
{
    const int as2 = (n / 4) * 4, as1 = (n / 2) * 2;
    if (n <= 0) return 0.0;
    if (!as1) return t[0];
    double a0 = t[0], a1 = t[1];
    if (as1 > 2) {
        double b0 = t[2], b1 = t[3];
        for (int i = 4; i < as2; i += 4) { a0 += t[i]; a1 += t[i + 1]; b0 += t[i + 2]; b1 += t[i + 3]; }
        a0 += b0; a1 += b1;
        if (as1 > as2) { a0 += t[as2]; a1 += t[as2 + 1]; }
    }
    double res = a0 + a1;
    for (int i = as1; i < n; ++i) res += t[i];
    return res;
}

/* QuadraticFormCost::computeIntegralStateControlTerm (quadratic_cost.cpp:186-230), diagonal weights: cost = 0; cost += xd^T Q xd; cost += u^T R u */
static void integral_cost_term(const oracle_problem* p, const double* rk, const double* x, const double* u, double* cost_out)
{
    const corbo_hip_problem_desc* d = &p->d;
    double cost = 0.0, acc = 0.0;
    { double t_[CORBO_HIP_MAX_NX]; for (int i = 0; i < d->nx; ++i) { double xd = x[i] - rk[i]; t_[i] = (xd * d->q_diag[i]) * xd; } acc += eigen_sum(t_, d->nx); }
    cost += acc;
    acc = 0.0;
    { double t_[CORBO_HIP_MAX_NX]; for (int i = 0; i < d->nu; ++i) t_[i] = (u[i] * d->r_diag[i]) * u[i]; acc += eigen_sum(t_, d->nu); }
    cost += acc;
    cost_out[0] = cost;
}

/* One step of the shooting grids' explicit integrator (MultipleShootingGrid::setNumericalIntegrator): the end state x(t + dt).
 * aug = 0: solveIVP(x1, u1, dt, system, x2) on the system dynamics (MSVariableDynamicsOnlyEdge, multiple_shooting_edges.h:125-134);
 * aug = 1: solveIVP(current, dt, integrand, values) of MultipleShootingEdgeSingleControl (multiple_shooting_edges.h:214-229, 251-281) on
 *          current = [cost; x] (n = nx + 1): integrand = [c(x, u) against reference k; f(x, u)] -- both overloads of every integrator are the
 *          same expressions (explicit_integrators.h). */
static void shooting_end_state(const oracle_problem* p, int aug, const double* rk, const double* x1, const double* u1, double dt, double* err)
{
    const corbo_hip_problem_desc* d = &p->d;
    const int n = aug ? d->nx + 1 : d->nx;
    double t[CORBO_HIP_MAX_NX + 1];
#define SHOOT_RHS(X, U, F)                                                                                   \
    do {                                                                                                     \
        if (aug) { dynamics(d, (X) + 1, (U), (F) + 1); integral_cost_term(p, rk, (X) + 1, (U), (F)); }       \
        else dynamics(d, (X), (U), (F));                                                                     \
    } while (0)
        {
            double k1[CORBO_HIP_MAX_NX + 1], k2[CORBO_HIP_MAX_NX + 1], k3[CORBO_HIP_MAX_NX + 1], k4[CORBO_HIP_MAX_NX + 1];
            if (d->shooting_integrator >= 5) { /* IntegratorExplicitRungeKutta5 / 6 / 7 (:371-394, :479-503, :600-628), coefficient-wise like Eigen, left to right */
                double K[11][CORBO_HIP_MAX_NX + 1];
                double *q1 = K[0], *q2 = K[1], *q3 = K[2], *q4 = K[3], *q5 = K[4], *q6 = K[5], *q7 = K[6], *q8 = K[7], *q9 = K[8], *q10 = K[9], *q11 = K[10];
#define RK_STAGE(Q, EXPR)                                   \
    {                                                       \
        for (int i = 0; i < n; ++i) t[i] = EXPR;           \
        SHOOT_RHS(t, u1, Q);                              \
        for (int i = 0; i < n; ++i) Q[i] *= dt;            \
    }
                RK_STAGE(q1, x1[i])
                if (d->shooting_integrator == 5) {
                    const double s6 = sqrt(6.0);
                    RK_STAGE(q2, x1[i] + 4.0 * q1[i] / 11.0)
                    RK_STAGE(q3, x1[i] + (9.0 * q1[i] + 11.0 * q2[i]) / 50.0)
                    RK_STAGE(q4, x1[i] + (-11.0 * q2[i] + 15.0 * q3[i]) / 4.0)
                    RK_STAGE(q5, x1[i] + ((81.0 + 9.0 * s6) * q1[i] + (255.0 - 55.0 * s6) * q3[i] + (24.0 - 14.0 * s6) * q4[i]) / 600.0)
                    RK_STAGE(q6, x1[i] + ((81.0 - 9.0 * s6) * q1[i] + (255.0 + 55.0 * s6) * q3[i] + (24.0 + 14.0 * s6) * q4[i]) / 600.0)
                    for (int i = 0; i < n; ++i) err[i] = x1[i] + (4.0 * q1[i] + (16.0 + s6) * q5[i] + (16.0 - s6) * q6[i]) / 36.0;
                }
                else if (d->shooting_integrator == 6) {
                    RK_STAGE(q2, x1[i] + 2.0 * q1[i] / 33.0)
                    RK_STAGE(q3, x1[i] + 4.0 * q2[i] / 33.0)
                    RK_STAGE(q4, x1[i] + (q1[i] + 3.0 * q3[i]) / 22.0)
                    RK_STAGE(q5, x1[i] + (43.0 * q1[i] - 165.0 * q3[i] + 144.0 * q4[i]) / 64.0)
                    RK_STAGE(q6, x1[i] + (-4053483.0 * q1[i] + 16334703.0 * q3[i] - 12787632.0 * q4[i] + 1057536.0 * q5[i]) / 826686.0)
                    RK_STAGE(q7, x1[i] + (169364139.0 * q1[i] - 663893307.0 * q3[i] + 558275718.0 * q4[i] - 29964480.0 * q5[i] + 35395542.0 * q6[i]) / 80707214.0)
                    RK_STAGE(q8, x1[i] + (-733.0 * q1[i] + 3102.0 * q3[i]) / 176.0 - (335763.0 * q4[i] / 23296.0) + (216.0 * q5[i] / 77.0) - (4617.0 * q6[i] / 2816.0) + (7203.0 * q7[i] / 9152.0))
                    for (int i = 0; i < n; ++i)
                        err[i] = x1[i] + (336336.0 * q1[i] + 1771561.0 * q4[i] + 1916928.0 * q5[i] + 597051.0 * q6[i] + 1411788.0 * q7[i] + 256256.0 * q8[i]) / 6289920.0;
                }
                else {
                    RK_STAGE(q2, x1[i] + 2.0 * q1[i] / 27.0)
                    RK_STAGE(q3, x1[i] + (q1[i] + 3.0 * q2[i]) / 36.0)
                    RK_STAGE(q4, x1[i] + (q1[i] + 3.0 * q3[i]) / 24.0)
                    RK_STAGE(q5, x1[i] + (80.0 * q1[i] - 300.0 * q3[i] + 300.0 * q4[i]) / 192.0)
                    RK_STAGE(q6, x1[i] + (q1[i] + 5.0 * q4[i] + 4.0 * q5[i]) / 20.0)
                    RK_STAGE(q7, x1[i] + (-25.0 * q1[i] + 125.0 * q4[i] - 260.0 * q5[i] + 250.0 * q6[i]) / 108.0)
                    RK_STAGE(q8, x1[i] + (93.0 * q1[i] + 244.0 * q5[i] - 200.0 * q6[i] + 13.0 * q7[i]) / 900.0)
                    RK_STAGE(q9, x1[i] + (12.0 * q1[i] - 53.0 * q4[i]) / 6.0 + (1408.0 * q5[i] - 1070.0 * q6[i] + 67.0 * q7[i] + 270.0 * q8[i]) / 90.0)
                    RK_STAGE(q10, x1[i] + (-12285.0 * q1[i] + 3105.0 * q4[i] - 105408.0 * q5[i] + 83970.0 * q6[i] - 4617.0 * q7[i] + 41310.0 * q8[i] - 1215.0 * q9[i]) / 14580.0)
                    RK_STAGE(q11, x1[i] + (2383.0 * q1[i] - 8525.0 * q4[i] + 17984.0 * q5[i] - 15050.0 * q6[i] + 2133.0 * q7[i] + 2250.0 * q8[i] + 1125.0 * q9[i] + 1800.0 * q10[i]) / 4100.0)
                    for (int i = 0; i < n; ++i)
                        err[i] = x1[i] + (41.0 * q1[i] + 272.0 * q6[i] + 216.0 * q7[i] + 216.0 * q8[i] + 27.0 * q9[i] + 27.0 * q10[i] + 41.0 * q11[i]) / 840.0;
                }
#undef RK_STAGE
                return;
            }
            if (d->shooting_integrator == 1) { /* IntegratorExplicitEuler (:66-72): x2 = f; x2 *= dt; x2 += x1 */
                SHOOT_RHS(x1, u1, err);
                for (int i = 0; i < n; ++i) err[i] *= dt;
                for (int i = 0; i < n; ++i) err[i] += x1[i];
                return;
            }
            if (d->shooting_integrator == 2) { /* IntegratorExplicitRungeKutta2 (:127-138) */
                SHOOT_RHS(x1, u1, k1);
                for (int i = 0; i < n; ++i) k1[i] *= dt;
                for (int i = 0; i < n; ++i) t[i] = x1[i] + k1[i];
                SHOOT_RHS(t, u1, k2);
                for (int i = 0; i < n; ++i) k2[i] *= dt;
                for (int i = 0; i < n; ++i) err[i] = x1[i] + (k1[i] + k2[i]) / 2.0;
                return;
            }
            if (d->shooting_integrator == 3) { /* IntegratorExplicitRungeKutta3 (:200-213) */
                SHOOT_RHS(x1, u1, k1);
                for (int i = 0; i < n; ++i) k1[i] *= dt;
                for (int i = 0; i < n; ++i) t[i] = x1[i] + (k1[i] / 2.0);
                SHOOT_RHS(t, u1, k2);
                for (int i = 0; i < n; ++i) k2[i] *= dt;
                for (int i = 0; i < n; ++i) t[i] = x1[i] - k1[i] + 2.0 * k2[i];
                SHOOT_RHS(t, u1, k3);
                for (int i = 0; i < n; ++i) k3[i] *= dt;
                for (int i = 0; i < n; ++i) err[i] = x1[i] + (k1[i] + 4.0 * k2[i] + k3[i]) / 6.0;
                return;
            }
            SHOOT_RHS(x1, u1, k1);
            for (int i = 0; i < n; ++i) k1[i] *= dt;
            for (int i = 0; i < n; ++i) t[i] = x1[i] + k1[i] / 2.0;
            SHOOT_RHS(t, u1, k2);
            for (int i = 0; i < n; ++i) k2[i] *= dt;
            for (int i = 0; i < n; ++i) t[i] = x1[i] + k2[i] / 2.0;
            SHOOT_RHS(t, u1, k3);
            for (int i = 0; i < n; ++i) k3[i] *= dt;
            for (int i = 0; i < n; ++i) t[i] = x1[i] + k3[i];
            SHOOT_RHS(t, u1, k4);
            for (int i = 0; i < n; ++i) k4[i] *= dt;
            for (int i = 0; i < n; ++i) err[i] = x1[i] + (k1[i] + 2.0 * k2[i] + 2.0 * k3[i] + k4[i]) / 6.0;
        }
#undef SHOOT_RHS
}

static void defect_values(const oracle_problem* p, const double* x1, const double* u1, const double* x2, double dt, double* err)
{
    const corbo_hip_problem_desc* d = &p->d;
    int nx = d->nx;
    double f1[CORBO_HIP_MAX_NX], t[CORBO_HIP_MAX_NX];
    switch (d->defect) {
        case CORBO_HIP_DEFECT_FORWARD: /* numerics/include/corbo-numerics/finite_differences_collocation.h:126-134 */
            dynamics(d, x1, u1, err);
            for (int i = 0; i < nx; ++i) err[i] -= (x2[i] - x1[i]) / dt;
            break;
        case CORBO_HIP_DEFECT_BACKWARD: /* :160-168 */
            dynamics(d, x2, u1, err);
            for (int i = 0; i < nx; ++i) err[i] -= (x2[i] - x1[i]) / dt;
            break;
        case CORBO_HIP_DEFECT_MIDPOINT: /* :194-202 */
            for (int i = 0; i < nx; ++i) t[i] = 0.5 * (x1[i] + x2[i]);
            dynamics(d, t, u1, err);
            for (int i = 0; i < nx; ++i) err[i] -= (x2[i] - x1[i]) / dt;
            break;
        case CORBO_HIP_DEFECT_CRANK_NICOLSON: /* :228-238 */
            dynamics(d, x1, u1, f1);
            dynamics(d, x2, u1, err);
            for (int i = 0; i < nx; ++i) err[i] = (x2[i] - x1[i]) / dt - 0.5 * (f1[i] + err[i]);
            break;
        case CORBO_HIP_DEFECT_RK4_SHOOTING: /* explicit_integrators.h:280-295 + integrator_interface.h:217-222 */
            shooting_end_state(p, 0, NULL, x1, u1, dt, err);
            for (int i = 0; i < nx; ++i) err[i] -= x2[i];
            break;
        default: break;
    }
}

/* BaseEdge::computeValues of the edge classes on the path */
/* U * xd for a NON-DIAGONAL weight: `cost.noalias() = _Q_sqrt * xd` (quadratic_cost.cpp:116-118, 148-150; final_state_cost.cpp:88-90) with U the
 * upper Cholesky factor kept by setWeightQ / setWeightR / setWeightQf (quadratic_cost.cpp:36-55, final_state_cost.cpp:38-58).  Eigen evaluates the
 * dynamic-size product as a column-major gemv into a zeroed destination: one running sum per row over the columns, a full block of FOUR columns
 * added pairwise (the order the linear state-space model's A x + B u is restated in; pinned bit for bit by the *_fullq fixtures). */
/* ... and with THREE columns the kernel's column order depends on where the destination lies: for a destination that starts 8 bytes past a
 * 16-byte boundary (an odd row of the solver's residual vector) it skips the first column to line the packets up, runs columns 1, 2 and adds
 * column 0 last (Eigen/src/Core/products/GeneralMatrixVector.h: skipColumns, with alignmentStep = 1 for an odd leading dimension); pinned bit
 * for bit by unicycle_n300_fullq (150 odd stages).  `odd` = that case; temporaries (finite differences, Hessian path) start aligned. */
static void dense_weight_times(const double* U, int n, const double* xd, double* out, int odd)
{
    for (int i = 0; i < n; ++i) {
        const double* u = U + i * n;
        if (n == 3 && odd) { out[i] = ((0.0 + u[1] * xd[1]) + u[2] * xd[2]) + u[0] * xd[0]; continue; }
        if (n == 4) { out[i] = 0.0 + ((u[0] * xd[0] + u[1] * xd[1]) + (u[2] * xd[2] + u[3] * xd[3])); continue; }
        double acc = 0.0;
        for (int j = 0; j < n; ++j) acc += u[j] * xd[j];
        out[i] = acc;
    }
}

/* the plug-in stage functions (user classes in the reference; oracle/ref_driver.cpp UserStageInequalities / LinearIntegralEquality) */
/* the stage inequalities' non-integral STATE term (a user StageInequalityConstraint of the reference, stage_functions.h:276-310; oracle/ref_driver.cpp
 * UserStageInequalities): the keep-out ball r^2 - |pos - c|^2, or a user function by id -- CORBO_HIP_STAGE_FN_USER + 0: the tilt cone
 * x[6]^2 + x[7]^2 - alpha^2 (restated from the host class like csrc/stage_functions/tilt_cone.hpp restates it) */
static double stage_ineq_value(const corbo_hip_problem_desc* d, const double* xk)
{
    if (d->stage_ineq == CORBO_HIP_STAGE_FN_USER + 0) return (xk[6] * xk[6] + xk[7] * xk[7]) - d->ineq_params[0] * d->ineq_params[0];
    double dx = xk[0] - d->ineq_params[0], dy = xk[1] - d->ineq_params[1], dz = xk[2] - d->ineq_params[2];
    return d->ineq_params[3] * d->ineq_params[3] - (dx * dx + dy * dy + dz * dz);
}
/* ... and their non-integral CONTROL term: CORBO_HIP_STAGE_FN_USER + 1 = the input-magnitude bound u[0]^2 + ... - r^2 (summed left to right) */
static double stage_ineq_control_value(const corbo_hip_problem_desc* d, const double* uk)
{
    double acc = 0.0;
    for (int i = 0; i < d->nu; ++i) acc += uk[i] * uk[i];
    return acc - d->ineq_control_params[0] * d->ineq_control_params[0];
}
static int stage_ineq_known(const corbo_hip_problem_desc* d)
{
    if (d->stage_ineq == CORBO_HIP_INEQ_NONE) return 1;
    if (d->stage_ineq == CORBO_HIP_INEQ_BALL) return d->nx >= 3;
    if (d->stage_ineq == CORBO_HIP_STAGE_FN_USER + 0) return d->nx >= 8;
    return 0;
}
static double stage_eq_value(const corbo_hip_problem_desc* d, const double* xk, const double* uk) /* a^T x + b^T u - c, summed left to right */
{
    double acc = 0.0;
    for (int i = 0; i < d->nx; ++i) acc += d->stage_eq_params[i] * xk[i];
    for (int i = 0; i < d->nu; ++i) acc += d->stage_eq_params[d->nx + i] * uk[i];
    return acc - d->stage_eq_params[d->nx + d->nu];
}

static void edge_values_at(const oracle_problem* p, const o_edge* e, double* out, int odd);
static void edge_values(const oracle_problem* p, const o_edge* e, double* out) { edge_values_at(p, e, out, 0); }
/* odd: `out` is an odd row of the solver's residual vector (see dense_weight_times) */
static void edge_values_at(const oracle_problem* p, const o_edge* e, double* out, int odd)
{
    const corbo_hip_problem_desc* d = &p->d;
    const double* x = p->x;
    switch (e->type) {
        case E_STATE_COST: { /* optimal_control/src/functions/quadratic_cost.cpp:100-119 (lsq form, diagonal Q) */
            const double* xk = x + p->v[e->vert[0]].off;
            const double* rk = p->refvec ? p->refvec + p->v[e->vert[0]].off : p->xref; /* getReferenceCached(k) */
            if (e->nonlsq) { /* lsq_form = false, diagonal mode: xd^T * Q_diag * xd (quadratic_cost.cpp:133-138; the expression of TerminalBall) */
                double acc = 0.0;
                { double t_[CORBO_HIP_MAX_NX]; for (int i = 0; i < d->nx; ++i) { double xd = xk[i] - rk[i]; t_[i] = (xd * d->q_diag[i]) * xd; } acc += eigen_sum(t_, d->nx); }
                out[0] = acc;
                break;
            }
            if (d->weights_dense & 1) { /* non-diagonal Q: xd = x_k - xref(k); cost = Q_sqrt * xd */
                double xd[CORBO_HIP_MAX_NX];
                for (int i = 0; i < d->nx; ++i) xd[i] = xk[i] - rk[i];
                dense_weight_times(d->q_sqrt, d->nx, xd, out, odd);
                break;
            }
            for (int i = 0; i < d->nx; ++i) out[i] = p->sq[i] * (xk[i] - rk[i]);
            break;
        }
        case E_CONTROL_COST: { /* quadratic_cost.cpp:140-154 (lsq form, zero uref, diagonal R).  A non-zero uref is not restated: the
                                * reference assigns the scalar ud^T R^(1/2) ud to the nu-vector there (:160-163) */
            const double* uk = x + p->v[e->vert[0]].off;
            if (e->nonlsq) { /* u_k^T * R_diag * u_k (quadratic_cost.cpp:165-170) */
                double acc = 0.0;
                { double t_[CORBO_HIP_MAX_NX]; for (int i = 0; i < d->nu; ++i) t_[i] = (uk[i] * d->r_diag[i]) * uk[i]; acc += eigen_sum(t_, d->nu); }
                out[0] = acc;
                break;
            }
            if (d->weights_dense & 2) { dense_weight_times(d->r_sqrt, d->nu, uk, out, odd); break; } /* R_sqrt * u_k (quadratic_cost.cpp:148-150) */
            for (int i = 0; i < d->nu; ++i) out[i] = p->sr[i] * uk[i];
            break;
        }
        case E_FINAL_COST: { /* optimal_control/src/functions/final_state_cost.cpp:72-92 */
            const double* xk = x + p->v[e->vert[0]].off;
            const double* rk = p->refvec ? p->refvec + p->v[e->vert[0]].off : p->xref;
            if (e->nonlsq) { /* xd^T * Qf_diag * xd (final_state_cost.cpp:102-108) */
                double acc = 0.0;
                { double t_[CORBO_HIP_MAX_NX]; for (int i = 0; i < d->nx; ++i) { double xd = xk[i] - rk[i]; t_[i] = (xd * d->qf_diag[i]) * xd; } acc += eigen_sum(t_, d->nx); }
                out[0] = acc;
                break;
            }
            if (d->weights_dense & 4) { /* Qf_sqrt * xd (final_state_cost.cpp:88-90) */
                double xd[CORBO_HIP_MAX_NX];
                for (int i = 0; i < d->nx; ++i) xd[i] = xk[i] - rk[i];
                dense_weight_times(d->qf_sqrt, d->nx, xd, out, odd);
                break;
            }
            for (int i = 0; i < d->nx; ++i) out[i] = p->sqf[i] * (xk[i] - rk[i]);
            break;
        }
        case E_INTEGRAL_COST: { /* TrapezoidalIntegralCostEdge (x1, u1, x2, dt) / LeftSumCostEdge (x1, u1, dt)
                                 * (finite_differences_collocation_edges.h:98-152, 323-368) over QuadraticFormCost::computeIntegralStateControlTerm
                                 * (quadratic_cost.cpp:186-230): cost = 0; cost += xd^T Q_diag xd; cost += u^T R_diag u; both ends use reference k */
            const double* x1 = x + p->v[e->vert[0]].off;
            const double* u1 = x + p->v[e->vert[1]].off;
            const double* rk = p->refvec ? p->refvec + p->v[e->vert[0]].off : p->xref;
            const int trap   = (e->nverts == 4);
            const double dt  = x[p->v[e->vert[trap ? 3 : 2]].off];
            double c[2];
            for (int end = 0; end < (trap ? 2 : 1); ++end) {
                const double* xe = end ? x + p->v[e->vert[2]].off : x1;
                double cost = 0.0, acc = 0.0;
                { double t_[CORBO_HIP_MAX_NX]; for (int i = 0; i < d->nx; ++i) { double xd = xe[i] - rk[i]; t_[i] = (xd * d->q_diag[i]) * xd; } acc += eigen_sum(t_, d->nx); }
                cost += acc;
                acc = 0.0;
                { double t_[CORBO_HIP_MAX_NX]; for (int i = 0; i < d->nu; ++i) t_[i] = (u1[i] * d->r_diag[i]) * u1[i]; acc += eigen_sum(t_, d->nu); }
                cost += acc;
                c[end] = cost;
            }
            if (trap) out[0] = 0.5 * dt * (c[0] + c[1]);
            else { out[0] = c[0]; out[0] *= dt; }
            break;
        }
        case E_MS_MIXED_OBJ: case E_MS_MIXED_EQ: { /* MultipleShootingEdgeSingleControl (x_k, u_k, dt, x_{k+1}), multiple_shooting_edges.h:151-303:
                                                   * precompute() integrates current = [0; x_k] with the grid's integrator over dt, integrand
                                                   * [c(x, u_k) against reference k; f(x, u_k)] (:214-229, :251-281); objective value = values[0]
                                                   * (:230-233), equality values = values[1 .. nx] - x_{k+1} (:234-242) */
            const double* x1 = x + p->v[e->vert[0]].off;
            const double* u1 = x + p->v[e->vert[1]].off;
            const double dt  = x[p->v[e->vert[2]].off];
            const double* x2 = x + p->v[e->vert[3]].off;
            const double* rk = p->refvec ? p->refvec + p->v[e->vert[0]].off : p->xref;
            double cur[CORBO_HIP_MAX_NX + 1], val[CORBO_HIP_MAX_NX + 1];
            cur[0] = 0;
            for (int i = 0; i < d->nx; ++i) cur[1 + i] = x1[i];
            shooting_end_state(p, 1, rk, cur, u1, dt, val);
            if (e->type == E_MS_MIXED_OBJ) out[0] = val[0];
            else
                for (int i = 0; i < d->nx; ++i) out[i] = val[1 + i] - x2[i];
            break;
        }
        case E_DT_COST: /* optimal_control/include/corbo-optimal-control/functions/minimum_time.h:70-78 */
            out[0] = p->dt_weight * x[p->v[e->vert[0]].off];
            break;
        case E_DEFECT: { /* FDCollocationEdge / MSVariableDynamicsOnlyEdge ::computeValues */
            const double* x1 = x + p->v[e->vert[0]].off;
            const double* u1 = x + p->v[e->vert[1]].off;
            const double* x2 = x + p->v[e->vert[2]].off;
            double dt        = x[p->v[e->vert[3]].off];
            defect_values(p, x1, u1, x2, dt, out);
            if (e->dim > d->nx) { /* TrapezoidalIntegralEqualityDynamicsEdge (:149-216): values.tail = 0.5 * dt * (e(x1, u1) + e(x2, u1)) */
                const double e1 = stage_eq_value(d, x1, u1), e2 = stage_eq_value(d, x2, u1);
                out[d->nx] = 0.5 * dt * (e1 + e2);
            }
            break;
        }
        case E_STAGE_INEQ: { /* user StageInequalityConstraint (state term): keep-out ball c = r^2 - |pos - c|^2 <= 0, or a user function by id */
            out[0] = stage_ineq_value(d, x + p->v[e->vert[0]].off);
            break;
        }
        case E_U_INEQ: { /* user StageInequalityConstraint (control term): UnaryVectorVertexEdge on u_k (nlp_functions.cpp:82-89) */
            out[0] = stage_ineq_control_value(d, x + p->v[e->vert[0]].off);
            break;
        }
        case E_INT_INEQ: { /* finite_differences_collocation_edges.h:271-321 (0.5 * dt * (c1 + c2), both ends with u1) / :412-459 (c1, then *= dt) */
            const double* x1 = x + p->v[e->vert[0]].off;
            const int trap   = (e->nverts == 4);
            const double dt  = x[p->v[e->vert[trap ? 3 : 2]].off];
            const double c1  = stage_ineq_value(d, x1);
            if (trap) { const double c2 = stage_ineq_value(d, x + p->v[e->vert[2]].off); out[0] = 0.5 * dt * (c1 + c2); }
            else { out[0] = c1; out[0] *= dt; }
            break;
        }
        case E_INT_EQ_LEFT: { /* LeftSumEqualityEdge::computeValues (:368-410): e(x1, u1), then *= dt */
            out[0] = stage_eq_value(d, x + p->v[e->vert[0]].off, x + p->v[e->vert[1]].off);
            out[0] *= x[p->v[e->vert[2]].off];
            break;
        }
        case E_CTRL_DEV: { /* user computeNonIntegralControlDeviationTerm(k, u_k, u_prev, dt_prev): ((u_k - u_prev) / dt_prev)^2 - r_max^2 per control */
            const double* uk = x + p->v[e->vert[0]].off;
            const double* up = x + p->v[e->vert[1]].off;
            const double dtp = x[p->v[e->vert[2]].off];
            for (int i = 0; i < d->nu; ++i) {
                const double dd = (uk[i] - up[i]) / dtp;
                out[i] = dd * dd - d->ctrl_dev_params[i] * d->ctrl_dev_params[i];
            }
            break;
        }
        case E_FINAL_EQ: { /* TerminalEqualityConstraint::computeNonIntegralStateTerm (final_state_constraints.h:149-154): x_k - xref */
            const double* xk = x + p->v[e->vert[0]].off;
            const double* rk = p->refvec ? p->refvec + p->v[e->vert[0]].off : p->xref;
            if (d->final_eq_mask) { /* TerminalPartialEqualityConstraint (final_state_constraints.h:236-252): the active components only, in order */
                int idx = 0;
                for (int i = 0; i < d->nx; ++i)
                    if ((d->final_eq_mask >> i) & 1u) out[idx++] = xk[i] - rk[i];
                break;
            }
            for (int i = 0; i < d->nx; ++i) out[i] = xk[i] - rk[i];
            break;
        }
        case E_FINAL_INEQ: { /* TerminalBall, diagonal mode, non-zero reference (final_state_constraints.cpp:72-76):
                              * xd = x_k - xref; cost = xd^T * S_diag * xd - gamma  (row vector times diagonal, then the inner product) */
            const double* xk = x + p->v[e->vert[0]].off;
            const double* rk = p->refvec ? p->refvec + p->v[e->vert[0]].off : p->xref;
            double acc = 0.0;
            { double t_[CORBO_HIP_MAX_NX]; for (int i = 0; i < d->nx; ++i) { double xd = xk[i] - rk[i]; t_[i] = (xd * d->final_ineq_params[i]) * xd; } acc += eigen_sum(t_, d->nx); }
            out[0] = acc - d->final_ineq_params[d->nx];
            break;
        }
        default: break;
    }
}

/* BaseEdge::computeJacobian (optimization/src/hyper_graph/edge_interface.cpp:55-96): central differences, delta=1e-9,
 * the vertex is perturbed IN PLACE and reverted by a third addition (not exact in floating point). */
static void edge_jacobian(oracle_problem* p, const o_edge* e, int vtx_idx, double* block /* dim x n_unfixed, col-major */)
{
    const double delta     = 1e-9;
    const double neg2delta = -2 * delta;
    const double scalar    = 1.0 / (2 * delta);
    const o_vertex* v      = &p->v[e->vert[vtx_idx]];
    double values1[CORBO_HIP_MAX_NX], values2[CORBO_HIP_MAX_NX];
    int col_idx = 0;
    for (int i = 0; i < v->dim; ++i) {
        if (v->fixed & (1u << i)) continue;
        p->x[v->off + i] += delta;
        edge_values(p, e, values2);
        p->x[v->off + i] += neg2delta;
        edge_values(p, e, values1);
        for (int j = 0; j < e->dim; ++j) block[col_idx * e->dim + j] = scalar * (values2[j] - values1[j]);
        p->x[v->off + i] += delta;
        ++col_idx;
    }
}

/* ------------------------------------------------------------------------------------------------------------ */
/* graph construction                                                                                             */

static int is_finite_bound(double lb, double ub) { return lb > -CORBO_HIP_INF || ub < CORBO_HIP_INF; } /* vector_vertex.h:174-184 */

static int validate(const corbo_hip_problem_desc* d)
{
    if (!d) return 0;
    if (d->nx < 1 || d->nx > CORBO_HIP_MAX_NX || d->nu < 1 || d->nu > CORBO_HIP_MAX_NU || d->N < 2) return 0;
    if (d->grid < 0 || d->grid > CORBO_HIP_GRID_MS_VARIABLE) return 0;
    if (d->defect < 0 || d->defect > CORBO_HIP_DEFECT_RK4_SHOOTING) return 0;
    if ((d->grid == CORBO_HIP_GRID_MS || d->grid == CORBO_HIP_GRID_MS_VARIABLE) != (d->defect == CORBO_HIP_DEFECT_RK4_SHOOTING)) return 0;
    switch (d->dynamics) {
        case CORBO_HIP_DYN_VAN_DER_POL: if (d->nx != 2 || d->nu != 1) return 0; break;
        case CORBO_HIP_DYN_SERIAL_INTEGRATOR: if (d->nu != 1) return 0; break;
        case CORBO_HIP_DYN_UNICYCLE: if (d->nx != 3 || d->nu != 2) return 0; break;
        case CORBO_HIP_DYN_USER + 0: if (d->nx != 3 || d->nu != 2) return 0; break; /* kinematic car (csrc/models/kinematic_car.hpp) */
        case CORBO_HIP_DYN_USER + 1: if (d->nx != 6 || d->nu != 2) return 0; break; /* planar quadrotor (csrc/models/planar_quadrotor.hpp) */
        case CORBO_HIP_DYN_QUADROTOR: if (d->nx != 12 || d->nu != 4) return 0; break;
        case CORBO_HIP_DYN_DUFFING:
        case CORBO_HIP_DYN_SIMPLE_PENDULUM:
        case CORBO_HIP_DYN_MASSLESS_PENDULUM:
        case CORBO_HIP_DYN_TOY_EXAMPLE:
        case CORBO_HIP_DYN_ARTSTEINS_CIRCLE: if (d->nx != 2 || d->nu != 1) return 0; break;
        case CORBO_HIP_DYN_FREE_SPACE_ROCKET: if (d->nx != 3 || d->nu != 1) return 0; break;
        case CORBO_HIP_DYN_CART_POLE: if (d->nx != 4 || d->nu != 1) return 0; break;
        case CORBO_HIP_DYN_PARALLEL_INTEGRATOR: if (d->nx != d->nu || d->nx < 2 || d->nx > 3) return 0; break;
        case CORBO_HIP_DYN_LINEAR_STATE_SPACE:
            if (!((d->nx == 2 && (d->nu == 1 || d->nu == 2)) || (d->nx == 3 && d->nu >= 1 && d->nu <= 3) || (d->nx == 4 && d->nu == 1))) return 0;
            break;
        default: return 0;
    }
    if (d->stage_cost < 0 || d->stage_cost > CORBO_HIP_COST_MIN_TIME_QUADRATIC_LSQ) return 0;
    if (d->stage_cost > CORBO_HIP_COST_MIN_TIME_LSQ && (CORBO_HIP_COST_TERMS(d->stage_cost) & 4) && d->grid != CORBO_HIP_GRID_FD_VARIABLE &&
        d->grid != CORBO_HIP_GRID_MS_VARIABLE)
        return 0;
    if (d->cost_nonlsq != 0 && d->cost_nonlsq != 1) return 0;
    if (d->cost_integral < 0 || d->cost_integral > 2) return 0;
    /* integral-form quadratic cost: QuadraticFormCost on a FiniteDifferencesGrid / FiniteDifferencesVariableGrid / MultipleShootingGrid, MinTimeQuadratic
     * (quadratic part in integral form next to its dt terms, hybrid_cost.h:189-303) on the FiniteDifferencesVariableGrid */
    if (d->cost_integral && (!d->cost_nonlsq || (d->stage_cost != CORBO_HIP_COST_QUADRATIC_LSQ && d->stage_cost != CORBO_HIP_COST_MIN_TIME_QUADRATIC_LSQ) ||
                             d->grid == CORBO_HIP_GRID_MS_VARIABLE)) return 0;
    if (d->cost_integral && d->stage_cost == CORBO_HIP_COST_MIN_TIME_QUADRATIC_LSQ && d->grid != CORBO_HIP_GRID_FD_VARIABLE) return 0;
    if (d->cost_integral && d->stage_cost == CORBO_HIP_COST_QUADRATIC_LSQ && d->grid == CORBO_HIP_GRID_FD_VARIABLE) return 0; /* (no fixture of the reference pins it) */
    if (d->cost_integral && d->grid == CORBO_HIP_GRID_MS && (d->stage_ineq || (d->weights_dense & 3))) return 0; /* MultipleShootingEdgeSingleControl: diagonal Q / R, no stage inequality */
    if (d->quad_first_interval < 0 || d->quad_first_interval > d->N - 1) return 0;
    if (d->quad_first_interval != 0 && d->stage_cost != CORBO_HIP_COST_MIN_TIME_QUADRATIC_LSQ) return 0;
    if (!stage_ineq_known(d)) return 0;
    if (d->stage_ineq_control != 0 && d->stage_ineq_control != CORBO_HIP_STAGE_FN_USER + 1) return 0;
    if (!(d->dt_ref > 0)) return 0;
    return 1;
}

/* integral-form constraints (FiniteDifferencesGrid and FiniteDifferencesVariableGrid) / control-deviation term (every grid), least-squares problems */
static int validate_extra(const corbo_hip_problem_desc* d)
{
    const int any = d->stage_ineq_integral || d->stage_eq || d->ctrl_dev;
    if (d->constraint_integration < 0 || d->constraint_integration > 2) return 0;
    if (d->stage_ineq_integral && (d->stage_ineq == CORBO_HIP_INEQ_NONE || !d->constraint_integration)) return 0;
    if (d->stage_eq && (d->stage_eq != CORBO_HIP_STAGE_EQ_LINEAR || !d->constraint_integration)) return 0;
    if (d->ctrl_dev && d->ctrl_dev != CORBO_HIP_CTRL_DEV_RATE) return 0;
    /* the integral-form edges are classes of the finite-differences grids; the control-deviation term is a non-integral term that every grid creates
     * (multiple_shooting_grid.cpp:62, 193-197) */
    if ((d->stage_ineq_integral || d->stage_eq) && (d->grid == CORBO_HIP_GRID_MS || d->grid == CORBO_HIP_GRID_MS_VARIABLE)) return 0;
    if (any && (d->cost_nonlsq || d->cost_integral || d->N < 3)) return 0;
    return 1;
}

static int dt_is_free(const corbo_hip_problem_desc* d) { return d->grid == CORBO_HIP_GRID_FD_VARIABLE || d->grid == CORBO_HIP_GRID_MS_VARIABLE; }

/* static row-wise view of J, envelope of H = J^T J and the LM work space (needs dims and the structure) */
static void finish_linear_algebra_setup(oracle_problem* p)
{
    int n = p->dims.n, m = p->dims.m, nnz = p->dims.nnz;
    /* ---- static row-wise view of J */
    int32_t* rows = (int32_t*)calloc(nnz, sizeof(int32_t));
    int32_t* cols = (int32_t*)calloc(nnz, sizeof(int32_t));
    oracle_get_structure(p, rows, cols);
    p->csr_ptr = (int*)calloc(m + 1, sizeof(int));
    p->csr_col = (int*)calloc(nnz, sizeof(int));
    p->csr_val = (int*)calloc(nnz, sizeof(int));
    for (int i = 0; i < nnz; ++i) p->csr_ptr[rows[i] + 1]++;
    for (int i = 0; i < m; ++i) p->csr_ptr[i + 1] += p->csr_ptr[i];
    int* fill = (int*)calloc((size_t)(m > 0 ? m : 1), sizeof(int));
    for (int i = 0; i < nnz; ++i) {
        int r = rows[i], pos = p->csr_ptr[r] + fill[r]++;
        p->csr_col[pos] = cols[i];
        p->csr_val[pos] = i;
    }
    for (int r = 0; r < m; ++r) /* sort every row by column (insertion sort, rows are short) */
        for (int a = p->csr_ptr[r] + 1; a < p->csr_ptr[r + 1]; ++a) {
            int c = p->csr_col[a], vv = p->csr_val[a], b2 = a - 1;
            while (b2 >= p->csr_ptr[r] && p->csr_col[b2] > c) { p->csr_col[b2 + 1] = p->csr_col[b2]; p->csr_val[b2 + 1] = p->csr_val[b2]; --b2; }
            p->csr_col[b2 + 1] = c; p->csr_val[b2 + 1] = vv;
        }
    free(fill); free(rows); free(cols);
    /* ---- envelope of H = J^T J */
    p->env_first = (int*)calloc(n, sizeof(int));
    p->env_ptr   = (int*)calloc(n + 1, sizeof(int));
    for (int i = 0; i < n; ++i) p->env_first[i] = i;
    for (int r = 0; r < m; ++r) {
        if (p->csr_ptr[r] == p->csr_ptr[r + 1]) continue;
        int cmin = p->csr_col[p->csr_ptr[r]];
        for (int a = p->csr_ptr[r]; a < p->csr_ptr[r + 1]; ++a)
            if (cmin < p->env_first[p->csr_col[a]]) p->env_first[p->csr_col[a]] = cmin;
    }
    for (int i = 0; i < n; ++i) p->env_ptr[i + 1] = p->env_ptr[i] + (i - p->env_first[i] + 1);
    p->env_size = p->env_ptr[n];

    p->values = (double*)calloc(m, sizeof(double));
    p->jac    = (double*)calloc(nnz, sizeof(double));
    p->H      = (double*)calloc(p->env_size, sizeof(double));
    p->L      = (double*)calloc(p->env_size, sizeof(double));
    p->rhs    = (double*)calloc(n, sizeof(double));
    p->delta  = (double*)calloc(n, sizeof(double));
    p->tmp    = (double*)calloc(n, sizeof(double));
    p->w_eq = p->w_ineq = p->w_b = 2; /* levenberg_marquardt_sparse.h:126-128 */
}


oracle_problem* oracle_create(const corbo_hip_problem_desc* desc)
{
    if (!validate(desc) || !validate_extra(desc)) return NULL;
    oracle_problem* p = (oracle_problem*)calloc(1, sizeof(oracle_problem));
    p->d              = *desc;
    const corbo_hip_problem_desc* d = &p->d;
    int nx = d->nx, nu = d->nu, N = d->N, s = nx + nu;
    int free_dt = dt_is_free(d);

    /* ---- vertices: x_0 u_0 | ... | x_{N-2} u_{N-2} | x_f | dt   (full_discretization_grid_base.cpp:134-179) */
    /* + the grid's always-fixed vertices _u_prev, _u_ref, _u_prev_dt (full_discretization_grid_base.cpp:509-511), stored behind dt */
    p->n_vertices = 2 * (N - 1) + 2 + 3;
    p->v          = (o_vertex*)calloc(p->n_vertices, sizeof(o_vertex));
    int nv = (N - 1) * s + nx + 1; /* storage always holds dt (fixed dt: not part of the public vertex layout) */
    const int v_uprev = 2 * (N - 1) + 2, v_uref = v_uprev + 1, v_uprev_dt = v_uprev + 2;
    p->v[v_uprev].off = nv;             p->v[v_uprev].dim = nu;  p->v[v_uprev].fixed = (1u << nu) - 1;
    p->v[v_uref].off = nv + nu;         p->v[v_uref].dim = nu;   p->v[v_uref].fixed = (1u << nu) - 1;
    p->v[v_uprev_dt].off = nv + 2 * nu; p->v[v_uprev_dt].dim = 1; p->v[v_uprev_dt].fixed = 1;
    const int nv_store = nv + 2 * nu + 1;
    for (int k = 0; k < N - 1; ++k) {
        o_vertex* xv = &p->v[2 * k];
        o_vertex* uv = &p->v[2 * k + 1];
        xv->off = k * s;       xv->dim = nx; xv->fixed = (k == 0) ? ((1u << nx) - 1) : 0; /* x_seq.front().setFixed(true) :170 */
        uv->off = k * s + nx;  uv->dim = nu; uv->fixed = 0;
    }
    o_vertex* xf = &p->v[2 * (N - 1)];
    xf->off = (N - 1) * s; xf->dim = nx; xf->fixed = d->xf_fixed_mask & ((1u << nx) - 1);
    o_vertex* dtv = &p->v[2 * (N - 1) + 1];
    dtv->off = (N - 1) * s + nx; dtv->dim = 1; dtv->fixed = free_dt ? 0 : 1; /* _dt.set(.., isDtFixedIntended()) :173 */
    /* active vertices + column indices (full_discretization_grid_base.cpp:514-527, vertex_set.cpp:405-418) */
    int col = 0;
    for (int i = 0; i < p->n_vertices; ++i) {
        o_vertex* v  = &p->v[i];
        v->n_unfixed = 0;
        for (int c = 0; c < v->dim; ++c)
            if (!(v->fixed & (1u << c))) v->n_unfixed++;
        if (v->n_unfixed > 0) { v->col = col; col += v->n_unfixed; }
        else v->col = -1;
    }
    int n = col;

    p->x      = (double*)calloc(nv_store, sizeof(double));
    p->lb     = (double*)calloc(nv_store, sizeof(double));
    p->ub     = (double*)calloc(nv_store, sizeof(double));
    p->backup = (double*)calloc(nv_store, sizeof(double));
    p->x[nv + 2 * nu] = d->dt_ref; /* _u_prev = 0, _u_prev_dt = grid->getInitialDt() (structured_optimal_control_problem.cpp:67-71); _u_ref = uref = 0 */
    p->xref   = (double*)calloc(nx, sizeof(double));
    for (int i = 0; i < nx; ++i) { p->sq[i] = sqrt(d->q_diag[i]); p->sqf[i] = sqrt(d->qf_diag[i]); } /* quadratic_cost.cpp:59-67 */
    for (int i = 0; i < nu; ++i) p->sr[i] = sqrt(d->r_diag[i]);
    p->dt_weight = d->cost_nonlsq ? (double)(N - 1) : sqrt((double)(N - 1)); /* minimum_time.h:60 (single dt): sqrt(n - 1) in lsq form, n - 1 otherwise */

    /* ---- edges in creation order (finite_differences_grid.cpp:38-154 / multiple_shooting_grid.cpp:38-197,
     *      nlp_functions.cpp:70-132: state term, control term, dt term (twice!), ...) */
    int max_edges = 6 * N + 8;
    o_edge* lsq   = (o_edge*)calloc(max_edges, sizeof(o_edge));
    o_edge* eq    = (o_edge*)calloc(max_edges, sizeof(o_edge));
    o_edge* ineq  = (o_edge*)calloc(max_edges, sizeof(o_edge));
    o_edge* mix   = (o_edge*)calloc(max_edges, sizeof(o_edge)); /* getMixedEdgesRef(): (objective part, equality part) pairs */
    int n_lsq = 0, n_eq = 0, n_ineq = 0, n_mix = 0;
    int dt_vertex = 2 * (N - 1) + 1;
    for (int k = 0; k < N - 1; ++k) {
        int xk = 2 * k, uk = 2 * k + 1, xnext = 2 * (k + 1); /* x_next = (k < n-2) ? x_seq[k+1] : xf */
        /* MinTimeQuadratic (hybrid_cost.h:189-303) has all three terms, in this order */
        const int terms = CORBO_HIP_COST_TERMS(d->stage_cost);
        const int quad = (k >= d->quad_first_interval); /* MinTimeQuadratic::only_last_n, hybrid_cost.h:224-237 */
        const int nl = d->cost_nonlsq; /* QuadraticFormCost(.., lsq_form = false): one scalar term each (quadratic_cost.h: dimension 1) */
        if (d->cost_integral && d->grid == CORBO_HIP_GRID_MS) { /* multiple_shooting_grid.cpp:70-77: hasIntegralTerms(k) -> ONE mixed edge per interval instead of
                                                                 * the dynamics-only edge, on (s_k, u_k, dt, s_{k+1}); filed in the mixed list (last in every walk) */
            o_edge* eo = &mix[n_mix++]; eo->type = E_MS_MIXED_OBJ; eo->k = k; eo->dim = 1; eo->scale = 0; eo->nonlsq = 1; eo->nverts = 4;
            eo->vert[0] = xk; eo->vert[1] = uk; eo->vert[2] = dt_vertex; eo->vert[3] = xnext;
            o_edge* ee = &mix[n_mix++]; *ee = *eo; ee->type = E_MS_MIXED_EQ; ee->dim = nx; ee->scale = 1; ee->nonlsq = 0;
            continue;
        }
        if (d->cost_integral && (terms & 4) && k == 0) { /* MinTimeQuadratic in integral form: the NON-integral terms of an interval (its dt term, twice)
                                                          * are filed before the interval's integral edge (finite_differences_grid.cpp:58-77) */
            for (int rep = 0; rep < 2; ++rep) {
                o_edge* e = &lsq[n_lsq++]; e->type = E_DT_COST; e->k = k; e->nverts = 1; e->vert[0] = dt_vertex; e->dim = 1; e->scale = 0; e->nonlsq = d->cost_nonlsq;
            }
        }
        if (d->cost_integral && !quad) { /* MinTimeQuadratic::only_last_n: hasIntegralTerms(k) = k >= _quad_k_min (hybrid_cost.h:209) */ }
        else if (d->cost_integral) { /* QuadraticFormCost(Q, R, integral_form = true): no non-integral terms; one integral cost edge per interval
                                 * (finite_differences_grid.cpp:62-77), 1 = TrapezoidalRule, 2 = LeftSum */
            o_edge* e = &lsq[n_lsq++]; e->type = E_INTEGRAL_COST; e->k = k; e->dim = 1; e->scale = 0; e->nonlsq = 1;
            e->vert[0] = xk; e->vert[1] = uk;
            if (d->cost_integral == 1) { e->nverts = 4; e->vert[2] = xnext; e->vert[3] = dt_vertex; }
            else { e->nverts = 3; e->vert[2] = dt_vertex; }
        }
        else if ((terms & 1) && quad) {
            o_edge* e = &lsq[n_lsq++]; e->type = E_STATE_COST; e->k = k; e->nverts = 1; e->vert[0] = xk; e->dim = nl ? 1 : nx; e->scale = 0; e->nonlsq = nl;
        }
        if (!d->cost_integral && (terms & 2) && quad) {
            o_edge* e = &lsq[n_lsq++]; e->type = E_CONTROL_COST; e->k = k; e->nverts = 1; e->vert[0] = uk; e->dim = nl ? 1 : nu; e->scale = 0; e->nonlsq = nl;
        }
        if (!d->cost_integral && (terms & 4) && k == 0) {
            for (int rep = 0; rep < 2; ++rep) { /* duplicated dt edge, nlp_functions.cpp:91-107 */
                o_edge* e = &lsq[n_lsq++]; e->type = E_DT_COST; e->k = k; e->nverts = 1; e->vert[0] = dt_vertex; e->dim = 1; e->scale = 0; e->nonlsq = d->cost_nonlsq;
            }
        }
        if (d->stage_ineq != CORBO_HIP_INEQ_NONE && !d->stage_ineq_integral) {
            o_edge* e = &ineq[n_ineq++]; e->type = E_STAGE_INEQ; e->k = k; e->nverts = 1; e->vert[0] = xk; e->dim = 1; e->scale = 2;
        }
        if (d->stage_ineq_control) { /* nlp_functions.cpp:82-89: the control term's edge behind the state term's */
            o_edge* e = &ineq[n_ineq++]; e->type = E_U_INEQ; e->k = k; e->nverts = 1; e->vert[0] = uk; e->dim = 1; e->scale = 2;
        }
        if (d->ctrl_dev) { /* nlp_functions.cpp:117-123 (behind the state term of the same stage function): (u_k, u_prev, dt_prev);
                            * finite_differences_grid.cpp:51-53: u_prev = k > 0 ? u_seq[k-1] : _u_prev, dt_prev = k > 0 ? _dt : _u_prev_dt */
            o_edge* e = &ineq[n_ineq++]; e->type = E_CTRL_DEV; e->k = k; e->nverts = 3; e->dim = nu; e->scale = 2;
            e->vert[0] = uk; e->vert[1] = (k > 0) ? 2 * (k - 1) + 1 : v_uprev; e->vert[2] = (k > 0) ? dt_vertex : v_uprev_dt;
        }
        if (d->stage_eq && d->constraint_integration == 2) { /* finite_differences_grid.cpp:89-98: LeftSumEqualityEdge, then the system dynamics edge */
            o_edge* e = &eq[n_eq++]; e->type = E_INT_EQ_LEFT; e->k = k; e->nverts = 3; e->dim = 1; e->scale = 1;
            e->vert[0] = xk; e->vert[1] = uk; e->vert[2] = dt_vertex;
        }
        o_edge* e = &eq[n_eq++]; e->type = E_DEFECT; e->k = k; e->nverts = 4; e->dim = nx; e->scale = 1;
        e->vert[0] = xk; e->vert[1] = uk; e->vert[2] = xnext; e->vert[3] = dt_vertex;
        if (d->stage_eq && d->constraint_integration == 1) e->dim = nx + 1; /* :82-88: TrapezoidalIntegralEqualityDynamicsEdge */
        if (d->stage_ineq != CORBO_HIP_INEQ_NONE && d->stage_ineq_integral) { /* :107-122 */
            o_edge* ei = &ineq[n_ineq++]; ei->type = E_INT_INEQ; ei->k = k; ei->dim = 1; ei->scale = 2;
            ei->vert[0] = xk; ei->vert[1] = uk;
            if (d->constraint_integration == 1) { ei->nverts = 4; ei->vert[2] = xnext; ei->vert[3] = dt_vertex; }
            else { ei->nverts = 3; ei->vert[2] = dt_vertex; }
        }
    }
    if (p->v[2 * (N - 1)].n_unfixed > 0 && d->final_cost) { /* if (!_xf.isFixed()) ... getFinalStateCostEdge */
        o_edge* e = &lsq[n_lsq++]; e->type = E_FINAL_COST; e->k = N - 1; e->nverts = 1; e->vert[0] = 2 * (N - 1); e->dim = d->cost_nonlsq ? 1 : nx; e->scale = 0;
        e->nonlsq = d->cost_nonlsq;
    }
    if (p->v[2 * (N - 1)].n_unfixed > 0 && d->final_eq) { /* getFinalStateConstraintEdge, isEqualityConstraint() :136-141 */
        o_edge* e = &eq[n_eq++]; e->type = E_FINAL_EQ; e->k = N - 1; e->nverts = 1; e->vert[0] = 2 * (N - 1); e->dim = nx; e->scale = 1;
        if (d->final_eq_mask) { e->dim = 0; for (int i = 0; i < nx; ++i) e->dim += (d->final_eq_mask >> i) & 1u; } /* getNonIntegralStateTermDimension = _num_active */
    }
    if (p->v[2 * (N - 1)].n_unfixed > 0 && d->final_ineq == CORBO_HIP_FINAL_INEQ_TERMINAL_BALL) { /* getFinalStateConstraintEdge :136-143 */
        o_edge* e = &ineq[n_ineq++]; e->type = E_FINAL_INEQ; e->k = N - 1; e->nverts = 1; e->vert[0] = 2 * (N - 1); e->dim = 1; e->scale = 2;
    }
    if (d->ctrl_dev) { /* finite_differences_grid.cpp:145-153, nlp_functions.cpp:152-186: the control-deviation edge of the last control, index n,
                        * on (u_ref, u_seq.back(), dt) -- after the final-stage edges */
        o_edge* e = &ineq[n_ineq++]; e->type = E_CTRL_DEV; e->k = N; e->nverts = 3; e->dim = nu; e->scale = 2;
        e->vert[0] = v_uref; e->vert[1] = 2 * (N - 2) + 1; e->vert[2] = dt_vertex;
    }
    /* row indices: [lsq | eq | ineq | bounds] (edge_set.cpp:31-42, hyper_graph_optimization_problem_edge_based.cpp:1491-1493) */
    p->n_edges = n_lsq + n_eq + n_ineq + n_mix;
    p->e       = (o_edge*)calloc(p->n_edges, sizeof(o_edge));
    int row = 0, ne = 0;
    for (int i = 0; i < n_lsq; ++i) {
        if (lsq[i].nonlsq) { lsq[i].row = -1; p->has_nonlsq = 1; }   /* not a row of the LM residual (getLsqObjectiveDimension() == 0) */
        else { lsq[i].row = row; row += lsq[i].dim; }
        p->e[ne++] = lsq[i];
    }
    int dim_lsq = row;
    for (int i = 0; i < n_eq; ++i) { eq[i].row = row; row += eq[i].dim; p->e[ne++] = eq[i]; }
    /* equality indices run on through the mixed edges (edge_set.cpp:31-42: computeEdgeIndices(_mixed, ..) continues idx_eq) */
    for (int i = 0; i < n_mix; ++i)
        if (mix[i].type == E_MS_MIXED_EQ) { mix[i].row = row; row += mix[i].dim; }
    int dim_eq = row - dim_lsq;
    for (int i = 0; i < n_ineq; ++i) { ineq[i].row = row; row += ineq[i].dim; p->e[ne++] = ineq[i]; }
    int dim_ineq = row - dim_lsq - dim_eq;
    for (int i = 0; i < ne; ++i) p->e[i].partner = -1;
    for (int i = 0; i < n_mix; ++i) { /* stored behind every other edge: mixed edges are the last loop of every function of the reference */
        if (mix[i].type == E_MS_MIXED_OBJ) { mix[i].row = -1; mix[i].partner = ne + 1; p->has_nonlsq = 1; }
        else mix[i].partner = ne - 1;
        p->e[ne++] = mix[i];
    }
    free(lsq); free(eq); free(ineq); free(mix);

    /* default bounds from the descriptor (needed for the bound-row structure) */
    for (int k = 0; k < N - 1; ++k) {
        for (int i = 0; i < nx; ++i) { p->lb[k * s + i] = d->x_lb[i]; p->ub[k * s + i] = d->x_ub[i]; }
        for (int i = 0; i < nu; ++i) { p->lb[k * s + nx + i] = d->u_lb[i]; p->ub[k * s + nx + i] = d->u_ub[i]; }
    }
    for (int i = 0; i < nx; ++i) { p->lb[(N - 1) * s + i] = d->x_lb[i]; p->ub[(N - 1) * s + i] = d->x_ub[i]; }
    p->lb[(N - 1) * s + nx] = free_dt ? d->dt_lb : -CORBO_HIP_INF;
    p->ub[(N - 1) * s + nx] = free_dt ? d->dt_ub : CORBO_HIP_INF;
    p->x[(N - 1) * s + nx]  = d->dt_ref;

    /* ---- Jacobian blocks in the reference's sweep order (…edge_based.cpp:1495-1617): edge lists in order, attached
     *      vertices in order, skipping vertices without unfixed components */
    int max_blocks = 4 * p->n_edges;
    p->b           = (o_block*)calloc(max_blocks, sizeof(o_block));
    int nnz = 0, nb = 0;
    for (int i = 0; i < p->n_edges; ++i) {
        const o_edge* e = &p->e[i];
        if (e->nonlsq) continue;
        for (int vi = 0; vi < e->nverts; ++vi) {
            const o_vertex* v = &p->v[e->vert[vi]];
            if (v->n_unfixed == 0) continue;
            o_block* b = &p->b[nb++];
            b->edge = i; b->vtx_idx = vi; b->row0 = e->row; b->col0 = v->col; b->rows = e->dim; b->cols = v->n_unfixed; b->off = nnz;
            nnz += b->rows * b->cols;
        }
    }
    p->n_blocks        = nb;
    p->first_bound_nnz = nnz;
    /* bound rows: one per unfixed component with a finite bound, active-vertex order (…base.cpp:291-315) */
    p->bound_vert_off = (int*)calloc(n, sizeof(int));
    p->bound_col      = (int*)calloc(n, sizeof(int));
    p->param_off      = (int*)calloc(n, sizeof(int));
    int n_bounds = 0;
    for (int i = 0; i < p->n_vertices; ++i) {
        const o_vertex* v = &p->v[i];
        if (v->n_unfixed == 0) continue;
        int free_idx = 0;
        for (int c = 0; c < v->dim; ++c) {
            if (v->fixed & (1u << c)) continue;
            p->param_off[v->col + free_idx] = v->off + c;
            if (is_finite_bound(p->lb[v->off + c], p->ub[v->off + c])) {
                p->bound_vert_off[n_bounds] = v->off + c;
                p->bound_col[n_bounds]      = v->col + free_idx;
                ++n_bounds;
            }
            ++free_idx;
        }
    }
    nnz += n_bounds;

    p->dims.nv     = free_dt ? nv : nv - 1;
    p->dims.n      = n;
    p->dims.lsq    = dim_lsq;
    p->dims.eq     = dim_eq;
    p->dims.ineq   = dim_ineq;
    p->dims.bounds = n_bounds;
    p->dims.m      = dim_lsq + dim_eq + dim_ineq + n_bounds;
    p->dims.nnz    = nnz;
    finish_linear_algebra_setup(p);
    return p;
}

/* ------------------------------------------------------------------------------------------------------------ */
/* Callback problem: SimpleOptimizationProblemWithCallbacks (optimization/include/corbo-optimization/             */
/* simple_optimization_problem.h:203-310) with the interface's default, dense central-difference Jacobians.      */
/* It exists so that the SAME oracle_solve() that runs the OCPs can be pinned against the known-answer cases of   */
/* the reference's own solver test (optimization/test/test_levenberg_marquardt_sparse.cpp:71-371), which are      */
/* small non-OCP problems.  Jacobian value order: dense rows [lsq | eq | ineq] row-major, then one entry per bound. */

oracle_problem* oracle_create_generic(int n, int dim_lsq, int dim_eq, int dim_ineq, const double* lb, const double* ub, oracle_generic_fun f)
{
    if (n < 1 || dim_lsq < 0 || dim_eq < 0 || dim_ineq < 0 || !f) return NULL;
    oracle_problem* p = (oracle_problem*)calloc(1, sizeof(oracle_problem));
    p->gen = f;
    p->x      = (double*)calloc(n + 1, sizeof(double));
    p->backup = (double*)calloc(n + 1, sizeof(double));
    p->lb     = (double*)calloc(n + 1, sizeof(double));
    p->ub     = (double*)calloc(n + 1, sizeof(double));
    p->xref   = (double*)calloc(CORBO_HIP_MAX_NX, sizeof(double));
    p->param_off      = (int*)calloc(n, sizeof(int));
    p->bound_vert_off = (int*)calloc(n, sizeof(int));
    p->bound_col      = (int*)calloc(n, sizeof(int));
    int n_bounds = 0;
    for (int i = 0; i < n; ++i) {
        p->lb[i] = lb ? lb[i] : -CORBO_HIP_INF;
        p->ub[i] = ub ? ub[i] : CORBO_HIP_INF;
        p->param_off[i] = i;
        if (is_finite_bound(p->lb[i], p->ub[i])) { /* optimization_problem_interface.cpp:566-592 */
            p->bound_vert_off[n_bounds] = i;
            p->bound_col[n_bounds]      = i;
            ++n_bounds;
        }
    }
    int dense_rows = dim_lsq + dim_eq + dim_ineq;
    p->dims.nv = n; p->dims.n = n; p->dims.lsq = dim_lsq; p->dims.eq = dim_eq; p->dims.ineq = dim_ineq; p->dims.bounds = n_bounds;
    p->dims.m   = dense_rows + n_bounds;
    p->dims.nnz = dense_rows * n + n_bounds;
    p->first_bound_nnz = dense_rows * n;
    finish_linear_algebra_setup(p);
    return p;
}

void oracle_destroy(oracle_problem* p)
{
    if (!p) return;
    free(p->v); free(p->e); free(p->b); free(p->bound_vert_off); free(p->bound_col); free(p->param_off);
    free(p->x); free(p->lb); free(p->ub); free(p->xref); free(p->backup); free(p->refvec);
    free(p->csr_ptr); free(p->csr_col); free(p->csr_val); free(p->env_first); free(p->env_ptr);
    free(p->values); free(p->jac); free(p->H); free(p->L); free(p->rhs); free(p->delta); free(p->tmp);
    free(p);
}

int oracle_get_dims(const oracle_problem* p, corbo_hip_dims* dims)
{
    if (!p || !dims) return CORBO_HIP_ERR_INVALID;
    *dims = p->dims;
    return 0;
}

int oracle_get_structure(const oracle_problem* p, int32_t* rows, int32_t* cols)
{
    if (!p) return CORBO_HIP_ERR_INVALID;
    if (p->gen)
        for (int r = 0; r < p->dims.lsq + p->dims.eq + p->dims.ineq; ++r)
            for (int c = 0; c < p->dims.n; ++c) { rows[r * p->dims.n + c] = r; cols[r * p->dims.n + c] = c; }
    for (int i = 0; i < p->n_blocks; ++i) {
        const o_block* b = &p->b[i];
        for (int c = 0; c < b->cols; ++c)
            for (int r = 0; r < b->rows; ++r) {
                rows[b->off + c * b->rows + r] = b->row0 + r;
                cols[b->off + c * b->rows + r] = b->col0 + c;
            }
    }
    int bounds_row0 = p->dims.lsq + p->dims.eq + p->dims.ineq;
    for (int i = 0; i < p->dims.bounds; ++i) {
        rows[p->first_bound_nnz + i] = bounds_row0 + i;
        cols[p->first_bound_nnz + i] = p->bound_col[i];
    }
    return 0;
}

int oracle_init_trajectory(const corbo_hip_problem_desc* d, const double* x0, const double* xf, double* x_out)
{
    /* full_discretization_grid_base.cpp:134-179 (shooting_grid_base.cpp:141-214 is identical for 1 control/interval) */
    if (!validate(d)) return CORBO_HIP_ERR_INVALID;
    int nx = d->nx, nu = d->nu, N = d->N, s = nx + nu;
    int num_intervals = N - 1;
    double dir[CORBO_HIP_MAX_NX];
    double sq = 0;
    for (int i = 0; i < nx; ++i) { dir[i] = xf[i] - x0[i]; sq += dir[i] * dir[i]; }
    double dist = sqrt(sq);
    if (dist != 0)
        for (int i = 0; i < nx; ++i) dir[i] /= dist;
    double step = dist / num_intervals;
    for (int k = 0; k < num_intervals; ++k) {
        for (int i = 0; i < nx; ++i) x_out[k * s + i] = x0[i] + (double)k * step * dir[i];
        for (int i = 0; i < nu; ++i) x_out[k * s + nx + i] = 0.0; /* uref = ZeroReference */
    }
    for (int i = 0; i < nx; ++i) x_out[(N - 1) * s + i] = xf[i];
    if (dt_is_free(d)) x_out[(N - 1) * s + nx] = d->dt_ref;
    return 0;
}

int oracle_set_data(oracle_problem* p, const double* x, const double* lb, const double* ub, const double* xref)
{
    if (!p || !x) return CORBO_HIP_ERR_INVALID;
    int nvp = p->dims.nv;
    memcpy(p->x, x, nvp * sizeof(double));
    if (!dt_is_free(&p->d)) p->x[nvp] = p->d.dt_ref;
    if (lb) memcpy(p->lb, lb, nvp * sizeof(double));
    if (ub) memcpy(p->ub, ub, nvp * sizeof(double));
    for (int i = 0; i < p->d.nx; ++i) p->xref[i] = xref ? xref[i] : 0.0;
    return 0;
}

/* StructuredOptimalControlProblem::setPreviousControlInput (structured_optimal_control_problem.h:73-79): the values of the fixed vertices
 * _u_prev / _u_prev_dt; NULL / <= 0 = the defaults (zeros, dt_ref) */
int oracle_set_previous_control(oracle_problem* p, const double* u_prev, double dt_prev)
{
    if (!p || p->gen) return CORBO_HIP_ERR_INVALID;
    const int nvs = (p->d.N - 1) * (p->d.nx + p->d.nu) + p->d.nx + 1;
    for (int i = 0; i < p->d.nu; ++i) p->x[nvs + i] = u_prev ? u_prev[i] : 0.0;
    p->x[nvs + 2 * p->d.nu] = dt_prev > 0 ? dt_prev : p->d.dt_ref;
    return 0;
}

/* Start of a new moving-horizon run on the stored trajectory (FullDiscretizationGridBase::update with new_run,
 * optimal_control/src/structured_ocp/discretization_grids/full_discretization_grid_base.cpp:91-108):
 *   shift != 0 (grid->setWarmStart(true), fixed-dt grids only :133 / finite_differences_variable_grid.h:77):
 *       warmStartShifting(x0) :230-283 with findNearestState :285-317;
 *   always: x_seq.front() = x0 (:101) and the fixed goal components = xref (:103-106). */
/* one reference per vertex component (vertex layout, nv doubles); NULL = back to the static state reference */
int oracle_set_references(oracle_problem* p, const double* ref)
{
    if (!p || p->gen) return CORBO_HIP_ERR_INVALID;
    free(p->refvec);
    p->refvec = NULL;
    if (ref) {
        p->refvec = (double*)calloc(p->dims.nv + 2, sizeof(double));
        memcpy(p->refvec, ref, p->dims.nv * sizeof(double));
    }
    return 0;
}

int oracle_warm_start(oracle_problem* p, const double* x0, int shift)
{
    if (!p || !x0) return CORBO_HIP_ERR_INVALID;
    const int nx = p->d.nx, nu = p->d.nu, N = p->d.N, s = nx + nu;
    double* X = p->x;
#define XS(i) (X + (i) * s)            /* _x_seq[i], i <= N-2 */
#define US(i) (X + (i) * s + nx)       /* _u_seq[i] */
    double* xf = X + (N - 1) * s;
    if (shift && !dt_is_free(&p->d)) {
        /* findNearestState */
        int num_shift = 0;
        double first = 0;
        for (int c = 0; c < nx; ++c) { double d = x0[c] - XS(0)[c]; first += d * d; }
        first = sqrt(first);
        if (!(fabs(first) < 1e-12)) {
            int num_interv = N - 1, lookahead = num_interv - 1 < 20 ? num_interv - 1 : 20;
            double cache = first;
            for (int i = 1; i <= lookahead; ++i) {
                double dist = 0;
                for (int c = 0; c < nx; ++c) { double d = x0[c] - XS(i)[c]; dist += d * d; }
                dist = sqrt(dist);
                if (dist < cache) { cache = dist; num_shift = i; }
                else break;
            }
        }
        if (num_shift > 0 && num_shift <= N - 2) {
            for (int i = 0; i < N - num_shift; ++i) {
                int idx = i + num_shift;
                if (idx == N - 1) memcpy(XS(i), xf, nx * sizeof(double));
                else { memmove(XS(i), XS(idx), nx * sizeof(double)); memmove(US(i), US(idx), nu * sizeof(double)); }
            }
            int idx = N - num_shift;
            for (int i = 0; i < num_shift; ++i, ++idx) {
                double* dst = (i == num_shift - 1) ? xf : XS(idx);
                for (int c = 0; c < nx; ++c) dst[c] = XS(idx - 2)[c] + 2.0 * (XS(idx - 1)[c] - XS(idx - 2)[c]);
                memmove(US(idx - 1), US(idx - 2), nu * sizeof(double));
            }
        }
    }
    memcpy(XS(0), x0, nx * sizeof(double));
    for (int c = 0; c < nx; ++c)
        if (p->d.xf_fixed_mask & (1u << c)) xf[c] = p->xref[c];
#undef XS
#undef US
    return 0;
}

/* FullDiscretizationGridBase::resampleTrajectory(n_new), full_discretization_grid_base.cpp:397-474, on the vertex layout of a
 * free-dt grid: x_old = [x_0 u_0 | ... | x_{n-2} u_{n-2} | x_f | dt] (n grid points) -> x_new with n_new grid points.  The start
 * sample and x_f are not touched; interior states are interpolated linearly in time between the old samples, controls are held
 * (:433-447); dt_new = dt_old (n - 1) / (n_new - 1) (:421).  n_new == n: unchanged (:400). */
int oracle_resample_trajectory(int nx, int nu, int n, const double* x_old, int n_new, double* x_new)
{
    if (!x_old || !x_new || n < 2 || n_new < 2) return CORBO_HIP_ERR_INVALID;
    const int s = nx + nu;
    const double* xf = x_old + (n - 1) * s;
    const double dt_old = x_old[(n - 1) * s + nx];
    if (n == n_new) { memcpy(x_new, x_old, ((size_t)(n - 1) * s + nx + 1) * sizeof(double)); return 0; }
    const double dt_new = dt_old * (double)(n - 1) / (double)(n_new - 1);
    memcpy(x_new, x_old, s * sizeof(double));                          /* x_0, u_0 stay */
    int idx_old = 1;
    for (int idx_new = 1; idx_new < n_new - 1; ++idx_new) {
        const double t_new = dt_new * (double)idx_new;
        while (t_new > (double)idx_old * dt_old && idx_old < n) ++idx_old;
        const double t_old_p1 = (double)idx_old * dt_old;
        const double* x_prev = x_old + (idx_old - 1) * s;
        const double* x_cur  = (idx_old < n - 1) ? x_old + idx_old * s : xf;
        const double f = (t_new - (t_old_p1 - dt_old)) / dt_old;
        double* xn = x_new + idx_new * s;
        for (int c = 0; c < nx; ++c) xn[c] = x_prev[c] + f * (x_cur[c] - x_prev[c]);
        /* the old control sample idx_old - 1; the time series holds the last control once more behind the horizon (:548-560) */
        const double* u_prev = x_old + ((idx_old - 1 < n - 1) ? idx_old - 1 : n - 2) * s + nx;
        for (int c = 0; c < nu; ++c) xn[nx + c] = u_prev[c];
    }
    memcpy(x_new + (n_new - 1) * s, xf, nx * sizeof(double));
    x_new[(n_new - 1) * s + nx] = dt_new;
    return 0;
}

/* FiniteDifferencesVariableGrid::adaptGridTimeBasedSingleStep / ...AggressiveEstimate / ...SimpleShrinkingHorizon
 * (finite_differences_variable_grid.cpp:101-163): the number of grid points after the adaptation (== n: no change).
 * strategy: 1 = single step, 2 = aggressive estimate, 3 = simple shrinking horizon, 4 = aggressive estimate of the shooting grid
 * (MultipleShootingVariableGrid; its single-step and shrinking rules are 1 and 3: multiple_shooting_variable_grid.cpp:93-152). */
int oracle_sizeof_problem_desc(void) { return (int)sizeof(corbo_hip_problem_desc); }

int oracle_adapt_grid_n(int strategy, int n, double dt, double dt_ref, double hyst, int n_min, int n_max)
{
    if (strategy == 1) {
        if (dt > dt_ref * (1.0 + hyst) && n < n_max) return n + 1;
        if (dt < dt_ref * (1.0 - hyst) && n > n_min) return n - 1;
        return n;
    }
    if (strategy == 2) {
        if (dt >= dt_ref * (1.0 - hyst) && dt <= dt_ref * (1.0 + hyst)) return n;
        int new_n = (int)round((double)n * (dt / dt_ref));
        if (new_n > n_max) new_n = n_max;
        else if (new_n < n_min) new_n = n_min;
        return new_n;
    }
    if (strategy == 3) return (n > n_min) ? n - 1 : n;
    if (strategy == 4) { /* MultipleShootingVariableGrid::adaptGridTimeBasedAggressiveEstimate (multiple_shooting_variable_grid.cpp:115-141):
                          * the RATIO is rounded to an integer first -- dt < dt_ref / 2 collapses the grid to n_min */
        if (dt >= dt_ref * (1.0 - hyst) && dt <= dt_ref * (1.0 + hyst)) return n;
        int new_n = n * (int)round(dt / dt_ref);
        if (new_n > n_max) new_n = n_max;
        else if (new_n < n_min) new_n = n_min;
        return new_n;
    }
    return n;
}

int oracle_plant_step(const oracle_problem* p, int integrator, double dt, const double* disturbance, double* x_plant)
{
    if (!p || !x_plant || p->gen) return CORBO_HIP_ERR_INVALID;
    const corbo_hip_problem_desc* d = &p->d;
    const int nx = d->nx;
    const double* u = p->x + nx; /* first control of the sequence (simulated_plant.cpp:111) */
    double x2[CORBO_HIP_MAX_NX];
    if (integrator == CORBO_HIP_INTEGRATOR_RK4) { /* explicit_integrators.h:280-295 */
        double k1[CORBO_HIP_MAX_NX], k2[CORBO_HIP_MAX_NX], k3[CORBO_HIP_MAX_NX], k4[CORBO_HIP_MAX_NX], t[CORBO_HIP_MAX_NX];
        dynamics(d, x_plant, u, k1);
        for (int i = 0; i < nx; ++i) k1[i] *= dt;
        for (int i = 0; i < nx; ++i) t[i] = x_plant[i] + k1[i] / 2.0;
        dynamics(d, t, u, k2);
        for (int i = 0; i < nx; ++i) k2[i] *= dt;
        for (int i = 0; i < nx; ++i) t[i] = x_plant[i] + k2[i] / 2.0;
        dynamics(d, t, u, k3);
        for (int i = 0; i < nx; ++i) k3[i] *= dt;
        for (int i = 0; i < nx; ++i) t[i] = x_plant[i] + k3[i];
        dynamics(d, t, u, k4);
        for (int i = 0; i < nx; ++i) k4[i] *= dt;
        for (int i = 0; i < nx; ++i) x2[i] = x_plant[i] + (k1[i] + 2.0 * k2[i] + 2.0 * k3[i] + k4[i]) / 6.0;
    }
    else if (integrator == CORBO_HIP_INTEGRATOR_EULER) { /* explicit_integrators.h:66-72 */
        dynamics(d, x_plant, u, x2);
        for (int i = 0; i < nx; ++i) x2[i] *= dt;
        for (int i = 0; i < nx; ++i) x2[i] += x_plant[i];
    }
    else return CORBO_HIP_ERR_INVALID;
    for (int i = 0; i < nx; ++i) x_plant[i] = disturbance ? x2[i] + disturbance[i] : x2[i]; /* state disturbance, :141 */
    return 0;
}

int oracle_get_x(const oracle_problem* p, double* x_out)
{
    if (!p || !x_out) return CORBO_HIP_ERR_INVALID;
    memcpy(x_out, p->x, p->dims.nv * sizeof(double));
    return 0;
}

/* ------------------------------------------------------------------------------------------------------------ */
/* stacked residual and combined Jacobian                                                                         */

/* LevenbergMarquardtSparse::computeValues (optimization/src/solver/levenberg_marquardt_sparse.cpp:222-246) on top of
 * BaseHyperGraphOptimizationProblem::computeValues{LsqObjective,Equality,ActiveInequality},
 * computeDistanceFiniteCombinedBounds (hyper_graph_optimization_problem_base.cpp:106-125,162-180,278-315) */
static void compute_values(oracle_problem* p, double w_eq, double w_ineq, double w_b, double* values)
{
    if (p->gen) { /* computeValuesLsqObjective / Equality / ActiveInequality of the callback problem (:222-246) */
        int l = p->dims.lsq, q = p->dims.eq, g = p->dims.ineq;
        p->gen(p->x, values, values + l, values + l + q);
        for (int j = 0; j < q; ++j) values[l + j] *= w_eq;
        for (int j = 0; j < g; ++j) values[l + q + j] = (values[l + q + j] < 0) ? 0 : w_ineq * values[l + q + j]; /* optimization_problem_interface.cpp:118-130 */
    }
    for (int i = 0; i < p->n_edges; ++i) {
        const o_edge* e = &p->e[i];
        edge_values_at(p, e, values + e->row, e->row & 1);
        if (e->scale == 1)
            for (int j = 0; j < e->dim; ++j) values[e->row + j] *= w_eq;
        else if (e->scale == 2)
            for (int j = 0; j < e->dim; ++j) {
                if (values[e->row + j] < 0) values[e->row + j] = 0;
                else values[e->row + j] *= w_ineq;
            }
    }
    int row0 = p->dims.lsq + p->dims.eq + p->dims.ineq;
    for (int i = 0; i < p->dims.bounds; ++i) {
        int o = p->bound_vert_off[i];
        double v;
        if (p->x[o] < p->lb[o]) v = p->lb[o] - p->x[o];
        else if (p->x[o] > p->ub[o]) v = p->x[o] - p->ub[o];
        else v = 0;
        values[row0 + i] = v * w_b;
    }
}

/* HyperGraphOptimizationProblemEdgeBased::computeCombinedSparseJacobian
 * (optimization/src/hyper_graph/hyper_graph_optimization_problem_edge_based.cpp:1480-1753) */
/* OptimizationProblemInterface::computeCombinedSparseJacobian, default implementation (optimization_problem_interface.cpp:702-810):
 * one CentralDifferences::jacobian pass (numerics/include/corbo-numerics/finite_differences.hpp:167-188: x_i += d, f1, x_i -= 2d, f0,
 * col = (1/2d)(f1 - f0), x_i += d; d = 1e-9) per non-empty part -- lsq (:217-233), equalities, ACTIVE inequalities (max(0,c) is what gets
 * differentiated, :439-456) -- each scaled by its weight on insertion. */
static void generic_jacobian(oracle_problem* p, double w_eq, double w_ineq, double* jac)
{
    const double delta = 1e-9, ddelta = 2 * delta, scalar = 1.0 / ddelta;
    int n = p->dims.n, l = p->dims.lsq, q = p->dims.eq, g = p->dims.ineq, rows = l + q + g;
    double* f1 = (double*)calloc(2 * rows + 2, sizeof(double));
    double* f0 = f1 + rows + 1;
    int row0[3] = {0, l, l + q}, dim[3] = {l, q, g};
    for (int part = 0; part < 3; ++part) {
        if (dim[part] < 1) continue;
        for (int i = 0; i < n; ++i) {
            p->x[i] += delta;
            p->gen(p->x, f1, f1 + l, f1 + l + q);
            p->x[i] += -ddelta;
            p->gen(p->x, f0, f0 + l, f0 + l + q);
            for (int r = row0[part]; r < row0[part] + dim[part]; ++r) {
                double a = f1[r], b = f0[r];
                if (part == 2) { a = a < 0 ? 0 : a; b = b < 0 ? 0 : b; }
                double v = scalar * (a - b);
                if (part == 1) v = v * w_eq;
                if (part == 2) v = v * w_ineq;
                jac[r * n + i] = v;
            }
            p->x[i] += delta;
        }
    }
    free(f1);
}

static void compute_jacobian(oracle_problem* p, double w_eq, double w_ineq, double w_b, const double* values, double* jac)
{
    double block[CORBO_HIP_MAX_NX * CORBO_HIP_MAX_NX];
    if (p->gen) generic_jacobian(p, w_eq, w_ineq, jac);
    for (int i = 0; i < p->n_blocks; ++i) {
        const o_block* b = &p->b[i];
        const o_edge* e  = &p->e[b->edge];
        edge_jacobian(p, e, b->vtx_idx, block);
        for (int c = 0; c < b->cols; ++c)
            for (int r = 0; r < b->rows; ++r) {
                double v = block[c * b->rows + r];
                if (e->scale == 1) v = v * w_eq;                                        /* :1552 */
                else if (e->scale == 2) v = (values[e->row + r] > 0.0) ? v * w_ineq : 0.0; /* :1568-1610 */
                jac[b->off + c * b->rows + r] = v;
            }
    }
    for (int i = 0; i < p->dims.bounds; ++i) { /* :1721-1752 */
        int o = p->bound_vert_off[i];
        double v;
        if (p->x[o] < p->lb[o]) v = -w_b;
        else if (p->x[o] > p->ub[o]) v = w_b;
        else v = 0.0;
        jac[p->first_bound_nnz + i] = v;
    }
}

int oracle_eval(oracle_problem* p, double w_eq, double w_ineq, double w_b, double* values, double* jac)
{
    if (!p || !values) return CORBO_HIP_ERR_INVALID;
    if (p->has_nonlsq) return CORBO_HIP_ERR_UNSUPPORTED;
    compute_values(p, w_eq, w_ineq, w_b, values);
    if (jac) compute_jacobian(p, w_eq, w_ineq, w_b, values, jac);
    return 0;
}

/* ------------------------------------------------------------------------------------------------------------ */
/* Operators of the exact-Hessian path: what IpoptWrapper::eval_h asks the hypergraph for       */
/* (nlp_solver_ipopt_wrapper.cpp:249-271) and the two-side-bounded linear form of the QP interface.               */

#define O_MAX_BLOCK (CORBO_HIP_MAX_NX * CORBO_HIP_MAX_NX)
static double squared_norm(const double* v, int n);

/* BaseEdge::computeHessian / computeHessianInc with a precomputed Jacobian block (edge_interface.cpp:151-255): forward differences
 * (HESSIAN_DELTA = 1e-2, :32) of the central-difference Jacobian of vertex i with respect to the components of vertex j, the vertex
 * perturbed in place and reverted by a second addition.  block: n_i x n_j, column-major.  inc = 0: the first row assigns. */
static void edge_hessian(oracle_problem* p, const o_edge* e, int vi, int vj, const double* jac_i, double* block, const double* mult, double weight, int inc)
{
    const double delta = 1e-2;
    double scalar      = 1.0 / delta;
    if (weight != 1.0) scalar *= weight;
    const o_vertex* a = &p->v[e->vert[vi]];
    const o_vertex* b = &p->v[e->vert[vj]];
    double jac2[O_MAX_BLOCK];
    int cj = 0;
    for (int j = 0; j < b->dim; ++j) {
        if (b->fixed & (1u << j)) continue;
        p->x[b->off + j] += delta;
        edge_jacobian(p, e, vi, jac2);
        for (int r = 0; r < e->dim; ++r) {
            const double f = mult ? scalar * mult[r] : scalar;
            for (int c = 0; c < a->n_unfixed; ++c) {
                const double t = f * (jac2[c * e->dim + r] - jac_i[c * e->dim + r]);
                if (r == 0 && !inc) block[cj * a->n_unfixed + c] = t;
                else block[cj * a->n_unfixed + c] += t;
            }
        }
        p->x[b->off + j] += -delta;
        ++cj;
    }
}

/* one vertex-pair block of a Hessian list: entries behind position `at`, returns the new position */
static int hess_emit(const o_vertex* a, const o_vertex* b, int diag_lower, int with_values, const double* blk, int32_t* rows, int32_t* cols, double* vals, int at)
{
    const int ni = a->n_unfixed, nj = b->n_unfixed;
    if (diag_lower) { /* lower triangle, row by row (:3537-3546) */
        for (int i = 0; i < ni; ++i)
            for (int j = 0; j <= i; ++j, ++at) {
                if (rows) { rows[at] = a->col + i; cols[at] = b->col + j; }
                if (with_values) vals[at] = 0.0 + blk[j * ni + i];
            }
    }
    else { /* values: the block column-major (Eigen::Map<MatrixXd> on the value array, :3550-3552); the structure lists the
            * same entries ROW-major (:2993-3003) -- the reference's own mismatch for off-diagonal vertex pairs, kept */
        for (int i = 0; i < ni; ++i)
            for (int j = 0; j < nj; ++j)
                if (rows) { rows[at + i * nj + j] = a->col + i; cols[at + i * nj + j] = b->col + j; }
        if (with_values)
            for (int q = 0; q < ni * nj; ++q) vals[at + q] = 0.0 + blk[q];
        at += ni * nj;
    }
    return at;
}

/* BaseMixedEdge::computeJacobians (edge_interface.cpp:394-465) of a MultipleShootingEdgeSingleControl: ONE in-place perturbation cycle per
 * component serves the objective block (1 x n_unfixed) and the equality block (nx x n_unfixed). */
static void mixed_jacobians(oracle_problem* p, const o_edge* eo, const o_edge* ee, int vtx_idx, double* jac_obj, double* jac_eq)
{
    const double delta = 1e-9, neg2delta = -2 * delta, scalar = 1.0 / (2 * delta);
    const o_vertex* v = &p->v[eo->vert[vtx_idx]];
    double o1[1], o2[1], q1[CORBO_HIP_MAX_NX], q2[CORBO_HIP_MAX_NX];
    int col_idx = 0;
    for (int i = 0; i < v->dim; ++i) {
        if (v->fixed & (1u << i)) continue;
        p->x[v->off + i] += delta;
        edge_values(p, eo, o2);
        edge_values(p, ee, q2);
        p->x[v->off + i] += neg2delta;
        edge_values(p, eo, o1);
        edge_values(p, ee, q1);
        jac_obj[col_idx] = scalar * (o2[0] - o1[0]);
        for (int j = 0; j < ee->dim; ++j) jac_eq[col_idx * ee->dim + j] = scalar * (q2[j] - q1[j]);
        p->x[v->off + i] += delta;
        ++col_idx;
    }
}

/* which category of computeSparseHessians* an edge belongs to: 0 objective (lsq), 1 equalities, 2 inequalities */
static int hessian_walk(oracle_problem* p, int lower, int with_values, double mult_obj, const double* mult_eq, const double* mult_ineq,
                        int32_t* rows[3], int32_t* cols[3], double* vals[3], int nnz[3])
{
    nnz[0] = nnz[1] = nnz[2] = 0;
    const int row_eq0 = p->dims.lsq, row_ineq0 = p->dims.lsq + p->dims.eq;
    for (int ei = 0; ei < p->n_edges; ++ei) {
        const o_edge* e = &p->e[ei];
        if (e->type == E_MS_MIXED_EQ) continue; /* walked together with its objective part */
        if (e->type == E_MS_MIXED_OBJ) {
            /* a mixed edge with a plain objective part and an equality part, no inequality part: the last branch of the mixed loop
             * (:3941-3988) -- computeJacobians once per vertex i, then per vertex j computeObjectiveHessian[Inc](.., nullptr, multiplier_obj)
             * and computeEqualityHessian[Inc](.., mult_eq_part); both lists advance by the same blocks */
            const o_edge* q   = &p->e[e->partner];
            const double* meq = mult_eq ? mult_eq + (q->row - row_eq0) : NULL;
            for (int vi = 0; vi < e->nverts; ++vi) {
                const o_vertex* a = &p->v[e->vert[vi]];
                if (a->n_unfixed == 0) continue;
                double jo[O_MAX_BLOCK], je[O_MAX_BLOCK], blk[O_MAX_BLOCK];
                if (with_values) mixed_jacobians(p, e, q, vi, jo, je);
                const int vend = lower ? vi + 1 : e->nverts;
                for (int vj = 0; vj < vend; ++vj) {
                    const o_vertex* b = &p->v[e->vert[vj]];
                    if (b->n_unfixed == 0) continue;
                    const int diag_lower = lower && (e->vert[vi] == e->vert[vj]);
                    if (with_values) {
                        for (int z = 0; z < a->n_unfixed * b->n_unfixed; ++z) blk[z] = 0.0;
                        edge_hessian(p, e, vi, vj, jo, blk, NULL, mult_obj, diag_lower ? 0 : 1);
                    }
                    nnz[0] = hess_emit(a, b, diag_lower, with_values, blk, rows[0], cols[0], vals[0], nnz[0]);
                    if (with_values) {
                        for (int z = 0; z < a->n_unfixed * b->n_unfixed; ++z) blk[z] = 0.0;
                        edge_hessian(p, q, vi, vj, je, blk, meq, 1.0, diag_lower ? 0 : 1);
                    }
                    nnz[1] = hess_emit(a, b, diag_lower, with_values, blk, rows[1], cols[1], vals[1], nnz[1]);
                    /* the reference's quirk: the inequality list's mixed loop tests the PROBLEM's getInequalityDimension(), not the edge's
                     * (:3216, :3421) -- as soon as the problem has any inequality, every mixed edge gets the same blocks in the inequality list too;
                     * this branch of the values function never writes them (:3941-3988): zeros */
                    if (p->dims.ineq > 0) {
                        if (with_values) for (int z = 0; z < a->n_unfixed * b->n_unfixed; ++z) blk[z] = 0.0;
                        nnz[2] = hess_emit(a, b, diag_lower, with_values, blk, rows[2], cols[2], vals[2], nnz[2]);
                    }
                }
            }
            continue;
        }
        const int cat   = e->scale;
        const double* mult = (cat == 1) ? (mult_eq ? mult_eq + (e->row - row_eq0) : NULL) : (cat == 2) ? (mult_ineq ? mult_ineq + (e->row - row_ineq0) : NULL) : NULL;
        for (int vi = 0; vi < e->nverts; ++vi) {
            const o_vertex* a = &p->v[e->vert[vi]];
            if (a->n_unfixed == 0) continue;
            double jac1[O_MAX_BLOCK], jac2[O_MAX_BLOCK], blk[O_MAX_BLOCK];
            if (with_values) edge_jacobian(p, e, vi, jac1);
            const int vend = lower ? vi + 1 : e->nverts;
            for (int vj = 0; vj < vend; ++vj) {
                const o_vertex* b = &p->v[e->vert[vj]];
                if (b->n_unfixed == 0) continue;
                const int diag_lower = lower && (e->vert[vi] == e->vert[vj]);
                const int ni = a->n_unfixed, nj = b->n_unfixed;
                int at = nnz[cat];
                if (with_values) {
                    if (cat == 0 && e->nonlsq) { /* plain objective edge: BaseEdge::computeHessian[Inc] with weight = multiplier, no row multipliers (:2363-2410) */
                        for (int q = 0; q < ni * nj; ++q) blk[q] = 0.0;
                        edge_hessian(p, e, vi, vj, jac1, blk, NULL, mult_obj, diag_lower ? 0 : 1);
                    }
                    else if (cat == 0) { /* lsq objective edge: 2 * multiplier * J_i^T J_j (the Gauss-Newton block, :3566-3606) */
                        edge_jacobian(p, e, vj, jac2);
                        for (int c = 0; c < nj; ++c)
                            for (int r = 0; r < ni; ++r) {
                                double acc = 0.0;
                                /* Eigen: small products (rows + cols + depth < 20, EIGEN_GEMM_TO_COEFFBASED_THRESHOLD) are coefficient-based with
                                 * (2 m J_i^T) as the left factor -- every term scaled first; larger ones go through the GEMM kernel, which
                                 * applies the factor to the finished sum.  (The cost Jacobians are diagonal: one non-zero term per sum.) */
                                if (ni + nj + e->dim < 20) {
                                    for (int q = 0; q < e->dim; ++q) acc += ((2.0 * mult_obj) * jac1[r * e->dim + q]) * jac2[c * e->dim + q];
                                    blk[c * ni + r] = acc;
                                }
                                else {
                                    for (int q = 0; q < e->dim; ++q) acc += jac1[r * e->dim + q] * jac2[c * e->dim + q];
                                    blk[c * ni + r] = (2.0 * mult_obj) * acc;
                                }
                            }
                    }
                    else {
                        for (int q = 0; q < ni * nj; ++q) blk[q] = 0.0;
                        edge_hessian(p, e, vi, vj, jac1, blk, mult, 1.0, diag_lower ? 0 : 1);
                    }
                }
                nnz[cat] = hess_emit(a, b, diag_lower, with_values, blk, rows[cat], cols[cat], vals[cat], at);
            }
        }
    }
    return 0;
}

int oracle_get_param_offsets(const oracle_problem* p, int32_t* off_out)
{
    if (!p || p->gen || !off_out) return CORBO_HIP_ERR_INVALID;
    for (int i = 0; i < p->dims.n; ++i) off_out[i] = p->param_off[i];
    return 0;
}

int oracle_hessian_nnz(oracle_problem* p, int lower_part_only, int32_t nnz_out[3])
{
    if (!p || p->gen) return CORBO_HIP_ERR_INVALID;
    int32_t* none[3] = {NULL, NULL, NULL};
    double* nov[3]   = {NULL, NULL, NULL};
    int nnz[3];
    hessian_walk(p, lower_part_only, 0, 1.0, NULL, NULL, none, none, nov, nnz);
    for (int i = 0; i < 3; ++i) nnz_out[i] = nnz[i];
    return 0;
}

int oracle_hessian_structure(oracle_problem* p, int lower_part_only, int32_t* rows_obj, int32_t* cols_obj, int32_t* rows_eq, int32_t* cols_eq,
                             int32_t* rows_ineq, int32_t* cols_ineq)
{
    if (!p || p->gen) return CORBO_HIP_ERR_INVALID;
    int32_t* rows[3] = {rows_obj, rows_eq, rows_ineq};
    int32_t* cols[3] = {cols_obj, cols_eq, cols_ineq};
    double* nov[3]   = {NULL, NULL, NULL};
    int nnz[3];
    return hessian_walk(p, lower_part_only, 0, 1.0, NULL, NULL, rows, cols, nov, nnz);
}

/* computeSparseHessiansValues (hyper_graph_optimization_problem_edge_based.cpp:3491-3760) at the current x */
int oracle_hessian_values(oracle_problem* p, int lower_part_only, double mult_obj, const double* mult_eq, const double* mult_ineq,
                          double* vals_obj, double* vals_eq, double* vals_ineq)
{
    if (!p || p->gen) return CORBO_HIP_ERR_INVALID;
    int32_t* none[3] = {NULL, NULL, NULL};
    double* vals[3]  = {vals_obj, vals_eq, vals_ineq};
    int nnz[3];
    return hessian_walk(p, lower_part_only, 1, mult_obj, mult_eq, mult_ineq, none, none, vals, nnz);
}

/* computeSparseJacobianTwoSideBoundedLinearForm{NNZ,Structure,Values} with include_finite_bounds = true (:4762-4968) and
 * computeBoundsForTwoSideBoundedLinearForm (optimization_problem_interface.cpp:1141-1183; ubA of a bound row is x - ub there, kept).
 * rows / cols / vals may be NULL (nnz query); lbA / ubA: eq + ineq + bounds entries. */
int oracle_linear_form(oracle_problem* p, int32_t* nnz_out, int32_t* rows, int32_t* cols, double* vals, double* lbA, double* ubA)
{
    if (!p || p->gen) return CORBO_HIP_ERR_INVALID;
    const int row_eq0 = p->dims.lsq;
    int at = 0;
    double blk[O_MAX_BLOCK];
    for (int ei = 0; ei < p->n_edges; ++ei) {
        const o_edge* e = &p->e[ei];
        if (e->scale == 0) continue;
        for (int vi = 0; vi < e->nverts; ++vi) {
            const o_vertex* a = &p->v[e->vert[vi]];
            if (a->n_unfixed == 0) continue;
            if (vals) edge_jacobian(p, e, vi, blk);
            for (int c = 0; c < a->n_unfixed; ++c)
                for (int r = 0; r < e->dim; ++r, ++at) {
                    if (rows) { rows[at] = e->row - row_eq0 + r; cols[at] = a->col + c; }
                    if (vals) vals[at] = blk[c * e->dim + r];
                }
        }
    }
    const int rowb0 = p->dims.eq + p->dims.ineq;
    for (int i = 0; i < p->dims.bounds; ++i, ++at) {
        if (rows) { rows[at] = rowb0 + i; cols[at] = p->bound_col[i]; }
        if (vals) vals[at] = 1.0;
    }
    if (nnz_out) *nnz_out = at;
    if (lbA && ubA) {
        double* tmp = (double*)calloc(p->dims.m + 1, sizeof(double));
        for (int ei = 0; ei < p->n_edges; ++ei) {
            const o_edge* e = &p->e[ei];
            if (e->scale == 0) continue;
            edge_values(p, e, tmp);
            for (int r = 0; r < e->dim; ++r) {
                const int row = e->row - row_eq0 + r;
                if (e->scale == 1) { lbA[row] = tmp[r] * -1; ubA[row] = lbA[row]; }
                else { lbA[row] = -CORBO_HIP_INF; ubA[row] = tmp[r] * -1; }
            }
        }
        free(tmp);
        for (int i = 0; i < p->dims.bounds; ++i) {
            const int o = p->bound_vert_off[i];
            lbA[rowb0 + i] = p->lb[o] - p->x[o];
            ubA[rowb0 + i] = p->x[o] - p->ub[o];
        }
    }
    return 0;
}

/* The first-order callbacks of IpoptWrapper (nlp_solver_ipopt_wrapper.cpp:128-230) at the current x: computeGradientObjective
 * (hyper_graph_optimization_problem_edge_based.cpp:31-102: per least-squares edge the central-difference Jacobian block, THEN the
 * edge values -- at the point the in-place differences left -- and gradient += (2 values^T) J), then computeValueObjective
 * (hyper_graph_optimization_problem_base.cpp:127-161: sum of the edges' squared norms).  eval_g / eval_jac_g are the constraint values
 * and the unweighted constraint Jacobian: oracle_linear_form (lbA = -c_eq, ubA = -c_ineq, the value list without the bound rows). */
int oracle_objective_gradient(oracle_problem* p, double* grad, double* obj_out)
{
    if (!p || p->gen || !grad) return CORBO_HIP_ERR_INVALID;
    for (int i = 0; i < p->dims.n; ++i) grad[i] = 0.0;
    double blk[O_MAX_BLOCK], vals[CORBO_HIP_MAX_NX];
    for (int ei = 0; ei < p->n_edges; ++ei) {
        const o_edge* e = &p->e[ei];
        if (e->scale != 0) continue;
        for (int vi = 0; vi < e->nverts; ++vi) {
            const o_vertex* a = &p->v[e->vert[vi]];
            if (a->n_unfixed == 0) continue;
            edge_jacobian(p, e, vi, blk);
            if (e->nonlsq) { /* block_jacobian.colwise().sum() (:43-56); dimension 1: the entry itself */
                for (int c = 0; c < a->n_unfixed; ++c) {
                    double acc = 0.0;
                    for (int r = 0; r < e->dim; ++r) acc += blk[c * e->dim + r];
                    grad[a->col + c] += acc;
                }
                continue;
            }
            edge_values(p, e, vals);
            for (int c = 0; c < a->n_unfixed; ++c) {
                double acc = 0.0;
                for (int r = 0; r < e->dim; ++r) acc += (2.0 * vals[r]) * blk[c * e->dim + r];
                grad[a->col + c] += acc;
            }
        }
    }
    if (obj_out) {
        double value = 0.0;
        for (int ei = 0; ei < p->n_edges; ++ei) {
            const o_edge* e = &p->e[ei];
            if (e->scale != 0) continue;
            edge_values(p, e, vals);
            if (e->nonlsq) { for (int r = 0; r < e->dim; ++r) value += vals[r]; }   /* computeSumOfValues() */
            else value += squared_norm(vals, e->dim);
        }
        *obj_out = value;
    }
    return 0;
}

/* ------------------------------------------------------------------------------------------------------------ */
/* H = J^T J, rhs = J^T (-values)  (levenberg_marquardt_sparse.cpp:97-100): each H(i,j) is accumulated over the    */
/* rows of J in ascending order, which is the order Eigen's conservative sparse product produces.                 */

static void build_hessian_rhs(oracle_problem* p)
{
    int n = p->dims.n, m = p->dims.m;
    memset(p->H, 0, p->env_size * sizeof(double));
    memset(p->rhs, 0, n * sizeof(double));
    for (int r = 0; r < m; ++r) {
        double nv = -p->values[r];
        for (int a = p->csr_ptr[r]; a < p->csr_ptr[r + 1]; ++a) {
            int ca    = p->csr_col[a];
            double va = p->jac[p->csr_val[a]];
            p->rhs[ca] += va * nv;
            for (int b = p->csr_ptr[r]; b <= a; ++b) {
                int cb = p->csr_col[b]; /* cb <= ca */
                p->H[p->env_ptr[ca] + (cb - p->env_first[ca])] += va * p->jac[p->csr_val[b]];
            }
        }
    }
}

/* envelope Cholesky H = L L^T (natural order) and solve; replaces Eigen::SimplicialLLT (:140-148). */
static void factor_solve(oracle_problem* p)
{
    int n = p->dims.n;
    const int *first = p->env_first, *ptr = p->env_ptr;
    double* L = p->L;
    memcpy(L, p->H, p->env_size * sizeof(double));
    for (int i = 0; i < n; ++i) {
        double* Li = L + ptr[i] - first[i]; /* Li[j] = L(i,j) */
        for (int j = first[i]; j < i; ++j) {
            const double* Lj = L + ptr[j] - first[j];
            int k0   = first[i] > first[j] ? first[i] : first[j];
            double s = Li[j];
            for (int k = k0; k < j; ++k) s -= Li[k] * Lj[k];
            Li[j] = s / Lj[j];
        }
        double dsum = Li[i];
        for (int k = first[i]; k < i; ++k) dsum -= Li[k] * Li[k];
        Li[i] = sqrt(dsum);
    }
    double* y = p->delta;
    for (int i = 0; i < n; ++i) { /* L y = rhs */
        const double* Li = L + ptr[i] - first[i];
        double s = p->rhs[i];
        for (int k = first[i]; k < i; ++k) s -= Li[k] * y[k];
        y[i] = s / Li[i];
    }
    for (int i = n - 1; i >= 0; --i) { /* L^T x = y (column sweep) */
        const double* Li = L + ptr[i] - first[i];
        y[i] /= Li[i];
        for (int k = first[i]; k < i; ++k) y[k] -= Li[k] * y[i];
    }
}

static double squared_norm(const double* v, int n)
{
    double s = 0;
    for (int i = 0; i < n; ++i) s += v[i] * v[i];
    return s;
}

/* LevenbergMarquardtSparse::solve (optimization/src/solver/levenberg_marquardt_sparse.cpp:44-220) */
int oracle_solve(oracle_problem* p, const corbo_hip_lm_opts* o, int new_run, double* chi2_out, oracle_trace_entry* trace)
{
    if (!p || !o) return CORBO_HIP_SOLVER_ERROR;
    if (p->has_nonlsq) return CORBO_HIP_SOLVER_ERROR; /* isLeastSquaresProblem() false: SolverStatus::Error (levenberg_marquardt_sparse.cpp:48-55) */
    int n = p->dims.n, m = p->dims.m, nvs = p->dims.nv + (dt_is_free(&p->d) ? 0 : 1);
    if (chi2_out) *chi2_out = -1;
    /* adapt weights :83-86, :270-287 */
    if (new_run) { p->w_eq = o->weight_eq; p->w_ineq = o->weight_ineq; p->w_b = o->weight_bounds; }
    else {
        p->w_eq *= o->adapt_factor_eq;       if (p->w_eq > o->adapt_max_eq) p->w_eq = o->adapt_max_eq;
        p->w_ineq *= o->adapt_factor_ineq;   if (p->w_ineq > o->adapt_max_ineq) p->w_ineq = o->adapt_max_ineq;
        p->w_b *= o->adapt_factor_bounds;    if (p->w_b > o->adapt_max_bounds) p->w_b = o->adapt_max_bounds;
    }
    double w_eq = p->w_eq, w_ineq = p->w_ineq, w_b = p->w_b;

    compute_values(p, w_eq, w_ineq, w_b, p->values);                 /* :89 */
    compute_jacobian(p, w_eq, w_ineq, w_b, p->values, p->jac);      /* :92 */
    build_hessian_rhs(p);                                            /* :97-100 */

    const double eps1 = 1e-5, eps2 = 1e-5, eps3 = 1e-5, eps4 = 0;
    unsigned int v = 2;
    double tau = 1e-5;
    const double goodStepUpperScale = 2. / 3., goodStepLowerScale = 1. / 3.;

    double rhs_inf = 0;
    for (int i = 0; i < n; ++i) if (fabs(p->rhs[i]) > rhs_inf) rhs_inf = fabs(p->rhs[i]);
    int stop = (rhs_inf <= eps1);                                    /* :115 */
    double maxdiag = p->H[p->env_ptr[0] + (0 - p->env_first[0])];
    for (int i = 0; i < n; ++i) { double dgl = p->H[p->env_ptr[i] + (i - p->env_first[i])]; if (dgl > maxdiag) maxdiag = dgl; }
    double mu = tau * maxdiag;                                       /* :117 */
    if (mu < 0) mu = 0;
    double rho      = 0;
    double chi2_old = squared_norm(p->values, m);                    /* :125 */
    if (chi2_out) *chi2_out = chi2_old;

    for (int k = 0; k < o->iterations; ++k) {                        /* :129 */
        int inner = 0, accepted = 0;
        double dnorm = 0;
        do {
            ++inner;
            for (int i = 0; i < n; ++i) p->H[p->env_ptr[i] + (i - p->env_first[i])] += mu; /* :135-138 (never undone) */
            factor_solve(p);                                         /* :147-148 */
            dnorm = sqrt(squared_norm(p->delta, n));
            if (dnorm <= eps2) { stop = 1; }                         /* :151-154 */
            else {
                memcpy(p->backup, p->x, nvs * sizeof(double));       /* backupParameters :158 */
                for (int i = 0; i < n; ++i) p->x[p->param_off[i]] += p->delta[i]; /* applyIncrement :161 */
                compute_values(p, w_eq, w_ineq, w_b, p->values);     /* :164 */
                double chi2_new = squared_norm(p->values, m);
                double den = 0;
                for (int i = 0; i < n; ++i) den += p->delta[i] * (mu * p->delta[i] + p->rhs[i]);
                rho = (chi2_old - chi2_new) / den;                   /* :169 */
                if (rho > 0 && !isnan(chi2_new) && !isinf(chi2_new)) { /* :171 */
                    stop = (sqrt(chi2_old) - sqrt(chi2_new) < eps4 * sqrt(chi2_old));
                    accepted = 1;                                    /* discardBackupParameters :176 */
                    if (!stop && k < o->iterations - 1) {            /* :178 */
                        compute_jacobian(p, w_eq, w_ineq, w_b, p->values, p->jac);
                        build_hessian_rhs(p);
                        rhs_inf = 0;
                        for (int i = 0; i < n; ++i) if (fabs(p->rhs[i]) > rhs_inf) rhs_inf = fabs(p->rhs[i]);
                        stop = stop || (rhs_inf <= eps1);
                        double alpha       = fmin(goodStepUpperScale, 1 - pow((2 * rho - 1), 3));
                        double scaleFactor = fmax(goodStepLowerScale, alpha);
                        mu *= scaleFactor;
                        v = 2;
                    }
                    chi2_old = chi2_new;
                    if (chi2_out) *chi2_out = chi2_old;
                }
                else {
                    accepted = 0;
                    memcpy(p->x, p->backup, nvs * sizeof(double));   /* restoreBackupParameters(false) :207 */
                    mu = mu * v;                                     /* :211-212 */
                    v  = 2 * v;
                }
            }
            if (inner >= 64) break; /* guard (not in the reference, which would spin): documented in DESIGN.md */
        } while (rho <= 0 && !stop);
        stop = (sqrt(squared_norm(p->values, m)) <= eps3);           /* :216 */
        if (trace) {
            trace[k].k = k; trace[k].inner_passes = inner; trace[k].accepted = accepted; trace[k].mu = mu; trace[k].rho = rho;
            trace[k].chi2 = chi2_old; trace[k].delta_norm = dnorm;
        }
    }
    return (stop || rho <= 0) ? CORBO_HIP_SOLVER_CONVERGED : CORBO_HIP_SOLVER_EARLY_TERMINATED; /* :218 */
}

int oracle_solve_batch(const corbo_hip_problem_desc* desc, int batch, double* x, const double* xref, const corbo_hip_lm_opts* opts,
                       double* chi2_out, int32_t* status_out)
{
    oracle_problem* p = oracle_create(desc);
    if (!p) return CORBO_HIP_ERR_INVALID;
    int nv = p->dims.nv, nx = desc->nx;
    for (int b = 0; b < batch; ++b) {
        oracle_set_data(p, x + (size_t)b * nv, NULL, NULL, xref ? xref + (size_t)b * nx : NULL);
        double chi2;
        int st = oracle_solve(p, opts, 1, &chi2, NULL);
        oracle_get_x(p, x + (size_t)b * nv);
        if (chi2_out) chi2_out[b] = chi2;
        if (status_out) status_out[b] = st;
    }
    oracle_destroy(p);
    return 0;
}
