// structure.cpp -- see structure.hpp.  Host only (no HIP).
#include "structure.hpp"

#include <algorithm>
#include <cmath>
#include <cstring>
#include <map>

namespace corbo_hip {

// kind of a registered user stage function (csrc/stage_functions/_registry.inc): 0 state inequality, 1 control inequality; -1 = no such id (or nx below its nx_min)
static int user_stage_kind(int id, int nx)
{
    (void)nx;
#if __has_include("stage_functions/_registry.inc")
#define CORBO_HIP_USER_STAGE(NAME, SLOT, KIND_, NXMIN) if (id == CORBO_HIP_STAGE_FN_USER + SLOT) return nx >= NXMIN ? KIND_ : -1;
#include "stage_functions/_registry.inc"
#undef CORBO_HIP_USER_STAGE
#endif
    (void)id;
    return -1;
}

static bool finite_bound(double lb, double ub) { return lb > -CORBO_HIP_INF || ub < CORBO_HIP_INF; }  // vector_vertex.h:174-184

std::string validate_desc(const corbo_hip_problem_desc& d)
{
    if (d.nx < 1 || d.nx > CORBO_HIP_MAX_NX) return "nx out of range";
    if (d.nu < 1 || d.nu > CORBO_HIP_MAX_NU) return "nu out of range";
    if (d.N < 2) return "N must be >= 2";
    if (d.grid < CORBO_HIP_GRID_FD || d.grid > CORBO_HIP_GRID_MS_VARIABLE) return "unknown grid";
    if (d.defect < CORBO_HIP_DEFECT_FORWARD || d.defect > CORBO_HIP_DEFECT_RK4_SHOOTING) return "unknown defect";
    if ((d.grid == CORBO_HIP_GRID_MS || d.grid == CORBO_HIP_GRID_MS_VARIABLE) != (d.defect == CORBO_HIP_DEFECT_RK4_SHOOTING))
        return "multiple-shooting grid needs the RK4 shooting defect (and only it)";
    bool user_model = false;
#if __has_include("models/_registry.inc")
#define CORBO_HIP_USER_MODEL(NAME, SLOT, NX_, NU_, P0, P1, P2, P3) \
    if (d.dynamics == CORBO_HIP_DYN_USER + SLOT) {                                                                                              \
        if (d.nx != NX_ || d.nu != NU_) return "user model " #NAME ": wrong nx / nu";                                                           \
        if (NX_ > 4 && !big_family_dims(NX_, NU_)) return "user model " #NAME ": nx <= 4 (small-block families: nu <= 3) or the big-block family (5 <= nx <= 12, nu <= 4, nx + nu <= 16)"; \
        user_model = true;                                                                                                                       \
    }
#include "models/_registry.inc"
#undef CORBO_HIP_USER_MODEL
#endif
    if (!user_model)
    switch (d.dynamics) {
        case CORBO_HIP_DYN_VAN_DER_POL: if (d.nx != 2 || d.nu != 1) return "van der pol: nx=2 nu=1"; break;
        case CORBO_HIP_DYN_SERIAL_INTEGRATOR: if (d.nu != 1) return "serial integrator: nu=1"; break;
        case CORBO_HIP_DYN_UNICYCLE: if (d.nx != 3 || d.nu != 2) return "unicycle: nx=3 nu=2"; break;
        case CORBO_HIP_DYN_QUADROTOR: if (d.nx != 12 || d.nu != 4) return "quadrotor: nx=12 nu=4"; break;
        case CORBO_HIP_DYN_DUFFING:
        case CORBO_HIP_DYN_SIMPLE_PENDULUM:
        case CORBO_HIP_DYN_MASSLESS_PENDULUM:
        case CORBO_HIP_DYN_TOY_EXAMPLE:
        case CORBO_HIP_DYN_ARTSTEINS_CIRCLE: if (d.nx != 2 || d.nu != 1) return "this benchmark system has nx=2 nu=1"; break;
        case CORBO_HIP_DYN_FREE_SPACE_ROCKET: if (d.nx != 3 || d.nu != 1) return "free-space rocket: nx=3 nu=1"; break;
        case CORBO_HIP_DYN_CART_POLE: if (d.nx != 4 || d.nu != 1) return "cart-pole: nx=4 nu=1"; break;
        case CORBO_HIP_DYN_LINEAR_STATE_SPACE:
            if (!((d.nx == 2 && (d.nu == 1 || d.nu == 2)) || (d.nx == 3 && d.nu >= 1 && d.nu <= 3) || (d.nx == 4 && d.nu == 1)))
                return "linear state-space model: (nx, nu) in {(2,1), (2,2), (3,1), (3,2), (3,3), (4,1)}";
            break;
        case CORBO_HIP_DYN_PARALLEL_INTEGRATOR: if (d.nx != d.nu || d.nx < 2 || d.nx > 3) return "parallel integrators: nx=nu=2 or 3"; break;
        default: return "unknown dynamics";
    }
    if (d.stage_cost < CORBO_HIP_COST_NONE || d.stage_cost > CORBO_HIP_COST_MIN_TIME_QUADRATIC_LSQ) return "unknown stage cost";
    if (d.stage_cost > CORBO_HIP_COST_MIN_TIME_LSQ && (CORBO_HIP_COST_TERMS(d.stage_cost) & 4) && d.grid != CORBO_HIP_GRID_FD_VARIABLE &&
        d.grid != CORBO_HIP_GRID_MS_VARIABLE)
        return "a stage cost with a minimum-time term needs a grid with a free dt";
    if (d.cost_nonlsq != 0 && d.cost_nonlsq != 1) return "cost_nonlsq must be 0 or 1";
    if (d.cost_integral < 0 || d.cost_integral > 2) return "cost_integral must be 0, 1 (trapezoidal rule) or 2 (left sum)";
    {   // QuadraticFormCost(integral_form = true) on the FiniteDifferencesGrid / the MultipleShootingGrid; MinTimeQuadratic(integral_form = true) -- its
        // quadratic part integrated, next to its dt terms (hybrid_cost.h:189-303) -- on the FiniteDifferencesVariableGrid
        const bool quad_ok = d.stage_cost == CORBO_HIP_COST_QUADRATIC_LSQ && (d.grid == CORBO_HIP_GRID_FD || d.grid == CORBO_HIP_GRID_MS);
        const bool mtq_ok  = d.stage_cost == CORBO_HIP_COST_MIN_TIME_QUADRATIC_LSQ && d.grid == CORBO_HIP_GRID_FD_VARIABLE;
        if (d.cost_integral && (!d.cost_nonlsq || !(quad_ok || mtq_ok)))
            return "cost_integral: with cost_nonlsq = 1, a quadratic stage cost on the FiniteDifferencesGrid / the MultipleShootingGrid or MinTimeQuadratic on the FiniteDifferencesVariableGrid";
    }
    if (d.cost_integral && d.grid == CORBO_HIP_GRID_MS && (d.stage_ineq || (d.weights_dense & 3) || d.nx > 8))
        return "cost_integral on the MultipleShootingGrid (MultipleShootingEdgeSingleControl): diagonal Q / R, no stage inequality, nx <= 8";
    if (d.quad_first_interval < 0 || d.quad_first_interval > d.N - 1) return "quad_first_interval out of range";
    if (d.quad_first_interval != 0 && d.stage_cost != CORBO_HIP_COST_MIN_TIME_QUADRATIC_LSQ) return "quad_first_interval: MinTimeQuadratic only";
    if (!(d.stage_ineq >= CORBO_HIP_INEQ_NONE && d.stage_ineq <= CORBO_HIP_INEQ_BALL) && user_stage_kind(d.stage_ineq, d.nx) != 0) return "unknown stage inequality (user state functions: csrc/stage_functions/, kind=state_ineq, nx >= its nx_min)";
    if (d.stage_ineq_control != 0 && user_stage_kind(d.stage_ineq_control, d.nx) != 1) return "stage_ineq_control: not a registered control_ineq function (csrc/stage_functions/)";
    if (d.stage_ineq_control != 0 && (d.cost_nonlsq || d.cost_integral)) return "stage_ineq_control: Levenberg-Marquardt path only";
    if (d.stage_ineq == CORBO_HIP_INEQ_BALL && d.nx < 3) return "ball inequality needs nx >= 3";
    if (d.final_ineq < CORBO_HIP_FINAL_INEQ_NONE || d.final_ineq > CORBO_HIP_FINAL_INEQ_TERMINAL_BALL) return "unknown final-stage inequality";
    if (d.final_ineq != CORBO_HIP_FINAL_INEQ_NONE && d.nx > 4 && !big_family_dims(d.nx, d.nu)) return "terminal ball: families with nx <= 4, and the big-block family";
    if (d.final_eq != 0 && d.final_eq != 1) return "final_eq must be 0 or 1";
    if (d.final_eq && d.nx > 4 && !big_family_dims(d.nx, d.nu)) return "terminal equality constraint: families with nx <= 4, and the big-block family";
    if (d.final_eq && d.final_ineq != CORBO_HIP_FINAL_INEQ_NONE) return "one final-stage constraint only (setFinalStageConstraint)";
    if (d.final_eq_mask) {
        if (!d.final_eq) return "final_eq_mask without final_eq";
        if (d.nx > 4 && !big_family_dims(d.nx, d.nu)) return "partial terminal equality constraint: families with nx <= 4, and the big-block family";
        if (d.final_eq_mask >> d.nx) return "final_eq_mask has bits beyond nx";
    }
    if (d.shooting_integrator < 0 || d.shooting_integrator > 7 || d.shooting_integrator == 4) return "shooting_integrator: 0 (RK4), 1 (Euler), 2 (RK2), 3 (RK3), 5 / 6 / 7 (RK5 / RK6 / RK7)";
    // (Runge-Kutta 5 - 7 around a big-block model: LM path through the stage / chain kernels -- no band route, so no free dt for odd block sizes;
    //  the Hessian-path operators refuse it at the call)
    if (d.shooting_integrator >= 5 && d.nx > 4 && (d.grid == CORBO_HIP_GRID_MS_VARIABLE || d.grid == CORBO_HIP_GRID_FD_VARIABLE) && d.nx % 2 != 0)
        return "shooting_integrator 5 .. 7 with a free dt: families with nx <= 4 or an even number of states";
    if (d.shooting_integrator != 0 && d.defect != CORBO_HIP_DEFECT_RK4_SHOOTING) return "shooting_integrator: shooting grids only";
    if (d.weights_dense < 0 || d.weights_dense > 7) return "weights_dense: bits 0..2";
    if (d.weights_dense) {
        if (d.nx > 4 || d.nu > 4) return "non-diagonal weights: families with nx <= 4";
        if (d.cost_nonlsq) return "non-diagonal weights: least-squares form only (cost_nonlsq = 0)";
        if ((d.weights_dense & 3) && !(CORBO_HIP_COST_TERMS(d.stage_cost) & 3)) return "weights_dense bits 0 / 1 without a quadratic stage cost";
        if ((d.weights_dense & 4) && !d.final_cost) return "weights_dense bit 2 without a final cost";
        for (int i = 0; i < d.nx; ++i)
            for (int j = 0; j < i; ++j)
                if (((d.weights_dense & 1) && d.q_sqrt[i * d.nx + j] != 0.0) || ((d.weights_dense & 4) && d.qf_sqrt[i * d.nx + j] != 0.0))
                    return "q_sqrt / qf_sqrt must be upper triangular (Eigen::LLT<.., Upper>::matrixU())";
        for (int i = 0; i < d.nu; ++i)
            for (int j = 0; j < i; ++j)
                if ((d.weights_dense & 2) && d.r_sqrt[i * d.nu + j] != 0.0) return "r_sqrt must be upper triangular";
    }
    if (!(d.dt_ref > 0)) return "dt_ref must be > 0";
    {   // integral-form constraints / control-deviation term (user stage functions of the reference)
        const bool any = d.stage_ineq_integral || d.stage_eq || d.ctrl_dev || d.stage_ineq_control;
        if (d.constraint_integration < 0 || d.constraint_integration > 2) return "constraint_integration: 0, 1 (trapezoidal rule) or 2 (left sum)";
        if (d.stage_ineq_integral != 0 && d.stage_ineq_integral != 1) return "stage_ineq_integral must be 0 or 1";
        if (d.stage_ineq_integral && (d.stage_ineq == CORBO_HIP_INEQ_NONE || !d.constraint_integration)) return "stage_ineq_integral needs a stage inequality and a constraint_integration rule";
        if (d.stage_eq != CORBO_HIP_STAGE_EQ_NONE && d.stage_eq != CORBO_HIP_STAGE_EQ_LINEAR) return "unknown stage equality";
        if (d.stage_eq && !d.constraint_integration) return "stage_eq (integral form) needs a constraint_integration rule";
        if (d.ctrl_dev != CORBO_HIP_CTRL_DEV_NONE && d.ctrl_dev != CORBO_HIP_CTRL_DEV_RATE) return "unknown control-deviation term";
        if (any) {
            // The integral-form edges are classes of the finite-differences grids (finite_differences_collocation_edges.h:149-459); on the shooting grids
            // integral terms make the interval a MIXED edge (multiple_shooting_grid.cpp:70-77), which lives on the Hessian path (cost_integral).  The
            // control-deviation term is a non-integral term: every grid creates it (multiple_shooting_grid.cpp:62, 193-197).
            const bool fd_grid = (d.grid == CORBO_HIP_GRID_FD || d.grid == CORBO_HIP_GRID_FD_VARIABLE);
            if (!fd_grid && (d.stage_ineq_integral || d.stage_eq)) return "integral-form constraint edges: FiniteDifferencesGrid and FiniteDifferencesVariableGrid (the shooting grids take the control-deviation term only)";
            if (d.cost_nonlsq || d.cost_integral) return "integral-form constraints / control-deviation term: Levenberg-Marquardt path (least-squares costs) only";
            if (d.nx > 4 && !big_family_dims(d.nx, d.nu)) return "integral-form constraints / control-deviation term: families with nx <= 4, and the big-block family";
            if (d.N < 3 || d.N > 1024) return "integral-form constraints / control-deviation term: 3 <= N <= 1024";
            if (d.weights_dense && d.nx > 4) return "integral-form constraints / control-deviation term with non-diagonal weights: families with nx <= 4";   // (band route: the DENSE x XE sweep instantiation)
            if (d.shooting_integrator >= 5) return "integral-form constraints / control-deviation term: shooting integrators up to Runge-Kutta 4";
        }
    }
    return "";
}

std::string build_structure(const corbo_hip_problem_desc& d, Structure& S)
{
    std::string err = validate_desc(d);
    if (!err.empty()) return err;
    S       = Structure();
    S.desc  = d;
    S.nx    = d.nx;
    S.nu    = d.nu;
    S.N     = d.N;
    S.s     = d.nx + d.nu;
    S.dt_free = (d.grid == CORBO_HIP_GRID_FD_VARIABLE || d.grid == CORBO_HIP_GRID_MS_VARIABLE);
    const int nx = S.nx, nu = S.nu, N = S.N, s = S.s;
    S.off_xf = (N - 1) * s;
    S.off_dt = (N - 1) * s + nx;
    const int nv_all = S.off_dt + 1;
    S.nvs            = (nv_all + 1) & ~1;
    for (int i = 0; i < nx; ++i) { S.sq[i] = std::sqrt(d.q_diag[i]); S.sqf[i] = std::sqrt(d.qf_diag[i]); }
    for (int i = 0; i < nu; ++i) S.sr[i] = std::sqrt(d.r_diag[i]);
    if (d.cost_nonlsq) {   // plain objective edges x^T Q x: the kernels of the Hessian path get the weights themselves in these slots
        for (int i = 0; i < nx; ++i) { S.sq[i] = d.q_diag[i]; S.sqf[i] = d.qf_diag[i]; }
        for (int i = 0; i < nu; ++i) S.sr[i] = d.r_diag[i];
    }
    S.dt_weight = d.cost_nonlsq ? (double)(N - 1) : std::sqrt((double)(N - 1));  // minimum_time.h:60: sqrt(n - 1) in lsq form, n - 1 otherwise

    // ---- components: fixed flags and parameter indices (full_discretization_grid_base.cpp:514-527, vertex_set.cpp:405-418)
    S.comp.assign(S.nvs, CompInfo{1, -1, -1, -1, -1, -1, -1, -1});
    std::vector<double> lb(S.nvs, -CORBO_HIP_INF), ub(S.nvs, CORBO_HIP_INF);
    for (int k = 0; k < N - 1; ++k) {
        for (int i = 0; i < nx; ++i) { S.comp[k * s + i].fixed = (k == 0); lb[k * s + i] = d.x_lb[i]; ub[k * s + i] = d.x_ub[i]; }
        for (int i = 0; i < nu; ++i) { S.comp[k * s + nx + i].fixed = 0; lb[k * s + nx + i] = d.u_lb[i]; ub[k * s + nx + i] = d.u_ub[i]; }
    }
    int xf_unfixed = 0;
    for (int i = 0; i < nx; ++i) {
        bool fx = (d.xf_fixed_mask >> i) & 1u;
        S.comp[S.off_xf + i].fixed = fx;
        if (!fx) ++xf_unfixed;
        lb[S.off_xf + i] = d.x_lb[i];
        ub[S.off_xf + i] = d.x_ub[i];
    }
    S.comp[S.off_dt].fixed = S.dt_free ? 0 : 1;
    lb[S.off_dt] = d.dt_lb;
    ub[S.off_dt] = d.dt_ub;
    int n = 0;
    for (int v = 0; v < nv_all; ++v)
        if (!S.comp[v].fixed) { S.comp[v].param = n++; S.param_voff.push_back(v); }

    // ---- edges in creation order -> rows (finite_differences_grid.cpp:38-154, nlp_functions.cpp:70-132, edge_set.cpp:31-42)
    struct E { int kind, k, dim, scale; };
    std::vector<E> lsq, eq, ineq;
    for (int k = 0; k < N - 1; ++k) {
        const int terms = CORBO_HIP_COST_TERMS(d.stage_cost);   // nlp_functions.cpp:70-107: state term, control term, dt term twice
        // (cost_nonlsq: plain objective edges are no rows of the LM residual -- getLsqObjectiveDimension() == 0; the LM entries refuse such a handle)
        const bool quad = (k >= d.quad_first_interval) && !d.cost_nonlsq;   // MinTimeQuadratic::only_last_n (hybrid_cost.h:224-237)
        if ((terms & 1) && quad) lsq.push_back({EK_STATE_COST, k, nx, 0});
        if ((terms & 2) && quad) lsq.push_back({EK_CONTROL_COST, k, nu, 0});
        if ((terms & 4) && k == 0 && !d.cost_nonlsq) {   // MinimumTime on a single-dt grid: k = 0 only (minimum_time.h:49)
            lsq.push_back({EK_DT_COST, k, 1, 0});
            lsq.push_back({EK_DT_COST, k, 1, 0});  // duplicated edge
        }
        // (creation order inside an interval, finite_differences_grid.cpp:49-125: the non-integral terms of the stage functions -- inequalities: state
        //  term, then control-deviation term, nlp_functions.cpp:70-131 --, then the integral equality / dynamics edges, then the integral inequality)
        if (d.stage_ineq != CORBO_HIP_INEQ_NONE && !d.stage_ineq_integral) ineq.push_back({EK_STAGE_INEQ, k, 1, 2});
        if (d.stage_ineq_control) ineq.push_back({EK_U_INEQ, k, 1, 2});   // the control term's edge on u_k (nlp_functions.cpp:82-89)
        if (d.ctrl_dev) ineq.push_back({EK_CTRL_DEV, k, nu, 2});
        if (d.stage_eq && d.constraint_integration == 2) eq.push_back({EK_XI_EQ_LEFT, k, 1, 1});
        eq.push_back({EK_DEFECT, k, (d.stage_eq && d.constraint_integration == 1) ? nx + 1 : nx, 1});
        if (d.stage_ineq != CORBO_HIP_INEQ_NONE && d.stage_ineq_integral) ineq.push_back({EK_XI_INEQ, k, 1, 2});
    }
    S.eq_stride     = nx + (d.stage_eq ? 1 : 0);
    S.eq_defect_off = (d.stage_eq && d.constraint_integration == 2) ? 1 : 0;
    // TerminalEqualityConstraint: nx rows; TerminalPartialEqualityConstraint: one row per active component (final_state_constraints.h:219)
    const uint32_t feq_mask = d.final_eq_mask ? d.final_eq_mask : ((1u << nx) - 1u);
    int feq_dim = 0;
    for (int i = 0; i < nx; ++i) feq_dim += (feq_mask >> i) & 1u;
    if (xf_unfixed > 0 && d.final_eq) eq.push_back({EK_FINAL_EQ, N - 1, feq_dim, 1});   // finite_differences_grid.cpp:135-141
    if (xf_unfixed > 0 && d.final_cost && !d.cost_nonlsq) lsq.push_back({EK_FINAL_COST, N - 1, nx, 0});
    if (xf_unfixed > 0 && d.final_ineq != CORBO_HIP_FINAL_INEQ_NONE) ineq.push_back({EK_FINAL_INEQ, N - 1, 1, 2});  // finite_differences_grid.cpp:135-143
    if (d.ctrl_dev) ineq.push_back({EK_CTRL_DEV, N, nu, 2});   // the last control against u_ref, index n (finite_differences_grid.cpp:145-153)

    int row = 0, joff = 0;
    auto comp_of = [&](int kind, int k, int vi, int c) -> int {  // vertex-storage offset of component c of attached vertex vi
        switch (kind) {
            case EK_STATE_COST: case EK_STAGE_INEQ: return k * s + c;
            case EK_CONTROL_COST: return k * s + nx + c;
            case EK_FINAL_COST: case EK_FINAL_INEQ: case EK_FINAL_EQ: return S.off_xf + c;
            case EK_DT_COST: return S.off_dt;
            default:  // defect: (x_k, u_k, x_{k+1}, dt)
                if (vi == 0) return k * s + c;
                if (vi == 1) return k * s + nx + c;
                if (vi == 2) return (k + 1) * s + c;
                return S.off_dt;
        }
    };
    auto vert_dim = [&](int kind, int vi) -> int {
        switch (kind) {
            case EK_STATE_COST: case EK_FINAL_COST: case EK_STAGE_INEQ: case EK_FINAL_INEQ: case EK_FINAL_EQ: return nx;
            case EK_CONTROL_COST: return nu;
            case EK_DT_COST: return 1;
            default: return vi == 0 ? nx : vi == 1 ? nu : vi == 2 ? nx : 1;
        }
    };
    S.stage_cols.assign(N - 1, StageCols{});
    for (auto& sc : S.stage_cols) for (int& c : sc.col) c = -1;
    if (d.stage_ineq != CORBO_HIP_INEQ_NONE && !d.stage_ineq_integral) { S.ineq_cols.assign((size_t)(N - 1) * nx, -1); S.ineq_rows.assign(N - 1, -1); }
    int dt_cost_seen = 0;
    S.fin_row = -1;
    S.fin_eq_dim = (xf_unfixed > 0 && d.final_eq) ? feq_dim : 0;
    for (int& f : S.fin_joff) f = -1;
    // attached vertices of the extra edge kinds: (storage offset, dimension); offsets < 0 = the grid's always-fixed vertices (XEdge)
    struct XV { int voff, dim; };
    auto extra_verts = [&](const E& e, XV (&v)[4]) -> int {
        const int k = e.k;
        switch (e.kind) {
            case EK_XI_INEQ:
                v[0] = {k * s, nx}; v[1] = {k * s + nx, nu};
                if (d.constraint_integration == 1) { v[2] = {(k + 1) * s, nx}; v[3] = {S.off_dt, 1}; return 4; }
                v[2] = {S.off_dt, 1}; return 3;
            case EK_XI_EQ_LEFT: v[0] = {k * s, nx}; v[1] = {k * s + nx, nu}; v[2] = {S.off_dt, 1}; return 3;
            case EK_U_INEQ: v[0] = {k * s + nx, nu}; return 1;
            case EK_CTRL_DEV:   // (u_k, u_prev, dt_prev): finite_differences_grid.cpp:51-53; the last one on (u_ref, u_{N-2}, dt), :149
                if (k == N) { v[0] = {-2, nu}; v[1] = {(N - 2) * s + nx, nu}; v[2] = {S.off_dt, 1}; return 3; }
                v[0] = {k * s + nx, nu};
                if (k == 0) { v[1] = {-1, nu}; v[2] = {-3, 1}; }
                else { v[1] = {(k - 1) * s + nx, nu}; v[2] = {S.off_dt, 1}; }
                return 3;
            default:            // the dynamics edge (its appended integral row)
                v[0] = {k * s, nx}; v[1] = {k * s + nx, nu}; v[2] = {(k + 1) * s, nx}; v[3] = {S.off_dt, 1}; return 4;
        }
    };
    auto add_extra = [&](const E& e) {   // one edge of the kinds EK_XI_INEQ / EK_XI_EQ_LEFT / EK_CTRL_DEV: structure entries + its XEdge
        XEdge x{};
        XV v[4];
        x.kind = e.kind; x.k = e.k; x.row = row; x.dim = e.dim; x.edim = e.dim; x.rie = 0; x.scale = e.scale;
        x.nverts = extra_verts(e, v);
        for (int vi = 0; vi < x.nverts; ++vi) {
            x.voff[vi] = v[vi].voff; x.vdim[vi] = v[vi].dim; x.joff[vi] = -1; x.fixed[vi] = 0;
            for (int c = 0; c < v[vi].dim; ++c) {
                const bool fx = v[vi].voff < 0 || S.comp[v[vi].voff + c].fixed;
                if (fx) { x.fixed[vi] |= 1u << c; continue; }
                if (x.joff[vi] < 0) x.joff[vi] = joff;
                for (int r = 0; r < e.dim; ++r) { S.jac_rows.push_back(row + r); S.jac_cols.push_back(S.comp[v[vi].voff + c].param); }
                joff += e.dim;
            }
        }
        S.xedges.push_back(x);
        row += e.dim;
    };
    auto add_list = [&](const std::vector<E>& list) {
        for (const E& e : list) {
            if (e.kind == EK_XI_INEQ || e.kind == EK_XI_EQ_LEFT || e.kind == EK_CTRL_DEV || e.kind == EK_U_INEQ) { add_extra(e); continue; }
            if (e.kind == EK_DEFECT && e.dim > nx) {   // TrapezoidalIntegralEqualityDynamicsEdge: the appended row as an XEdge over the dynamics edge's blocks
                XEdge x{};
                XV v[4];
                x.kind = EK_XI_EQ_ROW; x.k = e.k; x.row = row + nx; x.dim = 1; x.edim = e.dim; x.rie = nx; x.scale = 1;
                x.nverts = extra_verts(e, v);
                int jo = joff;
                for (int vi = 0; vi < x.nverts; ++vi) {
                    x.voff[vi] = v[vi].voff; x.vdim[vi] = v[vi].dim; x.joff[vi] = -1; x.fixed[vi] = 0;
                    for (int c = 0; c < v[vi].dim; ++c) {
                        if (S.comp[v[vi].voff + c].fixed) { x.fixed[vi] |= 1u << c; continue; }
                        if (x.joff[vi] < 0) x.joff[vi] = jo;
                        jo += e.dim;
                    }
                }
                S.xedges.push_back(x);
            }
            if (e.kind == EK_STAGE_INEQ) S.ineq_rows[e.k] = row;
            if (e.kind == EK_FINAL_INEQ) S.fin_row = row;
            int nverts = (e.kind == EK_DEFECT) ? 4 : 1;
            for (int vi = 0; vi < nverts; ++vi) {
                int vd = vert_dim(e.kind, vi);
                for (int c = 0; c < vd; ++c) {
                    int voff = comp_of(e.kind, e.k, vi, c);
                    // the cost edge of a fixed vertex still contributes its value rows (no Jacobian column)
                    if (e.kind == EK_STATE_COST || e.kind == EK_CONTROL_COST || e.kind == EK_FINAL_COST) S.comp[voff].cost_row = row + c;
                    // second row of an x_f component: row idx = active components before it; an inactive component has no row
                    int feq_idx = 0;
                    for (int q = 0; q < c; ++q) feq_idx += (feq_mask >> q) & 1u;
                    const bool feq_active = (feq_mask >> c) & 1u;
                    if (e.kind == EK_FINAL_EQ) { S.comp[voff].cost2_row = feq_active ? row + feq_idx : -1; if (c == 0) S.fin_eq_row0 = row; }
                    if (S.comp[voff].fixed) continue;
                    // one column of the block: rows e.dim, parameter = comp.param
                    for (int r = 0; r < e.dim; ++r) { S.jac_rows.push_back(row + r); S.jac_cols.push_back(S.comp[voff].param); }
                    if (e.kind == EK_DEFECT) {
                        int local = (vi == 0) ? c : (vi == 1) ? nx + c : (vi == 2) ? s + c : s + nx;
                        S.stage_cols[e.k].col[local] = joff;
                    }
                    else if (e.kind == EK_STAGE_INEQ) S.ineq_cols[(size_t)e.k * nx + c] = joff;
                    else if (e.kind == EK_FINAL_INEQ) S.fin_joff[c] = joff;
                    else if (e.kind == EK_FINAL_EQ) S.comp[voff].cost2_joff = joff + (feq_active ? feq_idx : 0);   // active: the entry of its row; inactive (no row): the column's start (explicit zeros)
                    else if (e.kind == EK_DT_COST) {
                        if (dt_cost_seen == 0) { S.comp[voff].cost_joff = joff; S.comp[voff].cost_row = row; }
                        else { S.comp[voff].cost2_joff = joff; S.comp[voff].cost2_row = row; }
                    }
                    else {  // diagonal cost blocks: value of row c
                        S.comp[voff].cost_joff = joff + c;
                        S.comp[voff].cost_row  = row + c;
                    }
                    joff += e.dim;
                }
            }
            if (e.kind == EK_DT_COST) ++dt_cost_seen;
            row += e.dim;
        }
    };
    add_list(lsq);
    S.dims.lsq = row;
    S.eq_row0  = row;
    S.defect_joff0 = joff;
    add_list(eq);
    S.dims.eq   = row - S.eq_row0;
    S.ineq_row0 = row;
    add_list(ineq);
    S.dims.ineq = row - S.ineq_row0;
    S.bnd_row0  = row;
    // bound rows (hyper_graph_optimization_problem_base.cpp:291-315)
    for (int v = 0; v < nv_all; ++v) {
        if (S.comp[v].fixed) continue;
        if (!finite_bound(lb[v], ub[v])) continue;
        S.comp[v].bnd_joff = joff;
        S.comp[v].bnd_row  = row;
        S.jac_rows.push_back(row);
        S.jac_cols.push_back(S.comp[v].param);
        ++row;
        ++joff;
    }
    S.dims.bounds = row - S.bnd_row0;
    S.dims.m      = row;
    S.dims.nnz    = joff;
    S.dims.n      = n;
    S.dims.nv     = S.dt_free ? nv_all : nv_all - 1;
    return "";
}

void init_trajectory(const corbo_hip_problem_desc& d, int batch, const double* x0, const double* xf, double* x_out)
{
    // FullDiscretizationGridBase::initializeSequences (full_discretization_grid_base.cpp:134-179):
    //   dir = xf - x0; dist = |dir|; dir /= dist; step = dist / (N-1); x_k = x0 + k*step*dir; u_k = uref(k) = 0
    const int nx = d.nx, nu = d.nu, N = d.N, s = nx + nu;
    const bool dt_free = (d.grid == CORBO_HIP_GRID_FD_VARIABLE || d.grid == CORBO_HIP_GRID_MS_VARIABLE);
    const int nv = (N - 1) * s + nx + (dt_free ? 1 : 0);
    for (int b = 0; b < batch; ++b) {
        const double* a = x0 + (size_t)b * nx;
        const double* g = xf + (size_t)b * nx;
        double* o       = x_out + (size_t)b * nv;
        double dir[CORBO_HIP_MAX_NX];
        double sq = 0;
        for (int i = 0; i < nx; ++i) { dir[i] = g[i] - a[i]; sq += dir[i] * dir[i]; }
        double dist = std::sqrt(sq);
        if (dist != 0)
            for (int i = 0; i < nx; ++i) dir[i] /= dist;
        double step = dist / (N - 1);
        for (int k = 0; k < N - 1; ++k) {
            for (int i = 0; i < nx; ++i) o[k * s + i] = a[i] + (double)k * step * dir[i];
            for (int i = 0; i < nu; ++i) o[k * s + nx + i] = 0.0;
        }
        for (int i = 0; i < nx; ++i) o[(N - 1) * s + i] = g[i];
        if (dt_free) o[(N - 1) * s + nx] = d.dt_ref;
    }
}

}  // namespace corbo_hip

namespace corbo_hip {

// The walk of computeSparseHessians{NNZ,Structure} (hyper_graph_optimization_problem_edge_based.cpp:2087-2348, 2869-3050, 3172-3353):
// per edge, vertex pairs (i, j) in attachment order, j <= i for the lower part; a diagonal pair of the lower part lists its lower
// triangle row by row, every other pair its full block ROW-major (while the values are written column-major, :3550-3552 -- the
// reference's own mismatch for rectangular off-diagonal pairs, reproduced so that the lists line up with the reference's entry by entry).
void build_hessian_structure(const Structure& S, bool lower, HessianStructure& H)
{
    H = HessianStructure();
    const corbo_hip_problem_desc& d = S.desc;
    const int nx = S.nx, nu = S.nu, N = S.N, s = S.s;
    H.stage_off.assign((size_t)N * 6, -1);
    H.lin_off.assign((size_t)N * 2, -1);
    struct V { int voff, dim; };
    auto unfixed = [&](const V& v) { int n = 0; for (int i = 0; i < v.dim; ++i) n += S.comp[v.voff + i].fixed ? 0 : 1; return n; };
    auto col_of  = [&](const V& v) { for (int i = 0; i < v.dim; ++i) if (!S.comp[v.voff + i].fixed) return S.comp[v.voff + i].param; return -1; };
    auto walk = [&](int cat, const V* verts, int nverts) {
        for (int vi = 0; vi < nverts; ++vi) {
            const int ni = unfixed(verts[vi]);
            if (ni == 0) continue;
            const int vend = lower ? vi + 1 : nverts;
            for (int vj = 0; vj < vend; ++vj) {
                const int nj = unfixed(verts[vj]);
                if (nj == 0) continue;
                const int ci = col_of(verts[vi]), cj = col_of(verts[vj]);
                if (lower && vi == vj) {
                    for (int i = 0; i < ni; ++i)
                        for (int j = 0; j <= i; ++j) { H.rows[cat].push_back(ci + i); H.cols[cat].push_back(cj + j); }
                }
                else
                    for (int i = 0; i < ni; ++i)
                        for (int j = 0; j < nj; ++j) { H.rows[cat].push_back(ci + i); H.cols[cat].push_back(cj + j); }
            }
        }
    };
    auto lin_walk = [&](const V* verts, int nverts, int dim, int row0) {
        for (int vi = 0; vi < nverts; ++vi) {
            const V& v = verts[vi];
            for (int i = 0; i < v.dim; ++i) {
                if (S.comp[v.voff + i].fixed) continue;
                for (int r = 0; r < dim; ++r) { H.lin_rows.push_back(row0 + r); H.lin_cols.push_back(S.comp[v.voff + i].param); }
            }
        }
    };
    int xf_unfixed = 0;
    for (int i = 0; i < nx; ++i) xf_unfixed += S.comp[S.off_xf + i].fixed ? 0 : 1;
    const V xf{S.off_xf, nx}, dtv{S.off_dt, 1};
    if (d.cost_integral && d.grid == CORBO_HIP_GRID_MS) {
        // MultipleShootingGrid + integral-form cost: one MIXED edge per interval on (x_k, u_k, dt, x_{k+1}) (multiple_shooting_grid.cpp:70-77).
        // Mixed edges are the last loop of every function of the reference: their objective / equality blocks follow the final cost /
        // the terminal equality in the Hessian lists (per vertex pair one block in each list, :3721-3990), their equality rows follow the
        // terminal equality's (edge_set.cpp:31-42), their linear-form blocks follow the inequality edges' (:4944-4960).
        if (xf_unfixed > 0 && d.final_cost) { H.stage_off[(size_t)(N - 1) * 6 + 0] = (int32_t)H.rows[0].size(); walk(0, &xf, 1); }
        int eq_row = 0, ineq_row = 0;
        if (xf_unfixed > 0 && d.final_eq) {
            H.stage_off[(size_t)(N - 1) * 6 + 2] = (int32_t)H.rows[1].size();
            H.stage_off[(size_t)(N - 1) * 6 + 4] = eq_row;
            walk(1, &xf, 1);
            H.lin_off[(size_t)(N - 1) * 2 + 0] = (int32_t)H.lin_rows.size();
            lin_walk(&xf, 1, S.fin_eq_dim, eq_row);   // (TerminalPartialEqualityConstraint: its active components' rows)
            eq_row += S.fin_eq_dim;
        }
        const int eq_mixed0 = eq_row, eq_total = eq_row + (N - 1) * nx;
        for (int k = 0; k < N - 1; ++k) {
            const V verts[4] = {{k * s, nx}, {k * s + nx, nu}, dtv, {(k + 1) * s, nx}};
            H.stage_off[(size_t)k * 6 + 0] = (int32_t)H.rows[0].size();
            walk(0, verts, 4);
            H.stage_off[(size_t)k * 6 + 2] = (int32_t)H.rows[1].size();
            H.stage_off[(size_t)k * 6 + 4] = eq_mixed0 + k * nx;
            walk(1, verts, 4);
        }
        if (xf_unfixed > 0 && d.final_ineq != CORBO_HIP_FINAL_INEQ_NONE) {
            H.stage_off[(size_t)(N - 1) * 6 + 3] = (int32_t)H.rows[2].size();
            H.stage_off[(size_t)(N - 1) * 6 + 5] = ineq_row;
            walk(2, &xf, 1);
            H.lin_off[(size_t)(N - 1) * 2 + 1] = (int32_t)H.lin_rows.size();
            lin_walk(&xf, 1, 1, eq_total + ineq_row);
            ineq_row += 1;
            // the reference's quirk: the inequality list's mixed loop tests the PROBLEM's getInequalityDimension(), not the edge's (:3216, :3421) --
            // as soon as the problem has any inequality every mixed edge gets the same blocks in the inequality list too (never written: zeros)
            for (int k = 0; k < N - 1; ++k) {
                const V verts[4] = {{k * s, nx}, {k * s + nx, nu}, dtv, {(k + 1) * s, nx}};
                H.stage_off[(size_t)k * 6 + 3] = (int32_t)H.rows[2].size();
                walk(2, verts, 4);
            }
        }
        for (int k = 0; k < N - 1; ++k) {
            const V verts[4] = {{k * s, nx}, {k * s + nx, nu}, dtv, {(k + 1) * s, nx}};
            H.lin_off[(size_t)k * 2 + 0] = (int32_t)H.lin_rows.size();
            lin_walk(verts, 4, nx, eq_mixed0 + k * nx);
        }
        for (int c = 0; c < 3; ++c) H.nnz[c] = (int32_t)H.rows[c].size();
        H.lin_bounds0 = (int32_t)H.lin_rows.size();
        for (int v = 0; v <= S.off_dt; ++v) {
            if (S.comp[v].bnd_row < 0) continue;
            H.lin_rows.push_back(eq_total + ineq_row + (S.comp[v].bnd_row - S.bnd_row0));
            H.lin_cols.push_back(S.comp[v].param);
        }
        H.lin_nnz = (int32_t)H.lin_rows.size();
        return;
    }
    // objective (least-squares edges) and the per-stage offsets
    for (int k = 0; k < N - 1; ++k) {
        const V xk{k * s, nx}, uk{k * s + nx, nu};
        const int terms = CORBO_HIP_COST_TERMS(d.stage_cost);
        const bool quad = (k >= d.quad_first_interval) && !d.cost_integral;
        if (d.cost_integral && (terms & 4) && k == 0) {   // MinTimeQuadratic in integral form: an interval's non-integral terms (the dt term, twice) are
            H.dt_cost_off = (int32_t)H.rows[0].size();   // filed BEFORE its integral edge (finite_differences_grid.cpp:58-77)
            walk(0, &dtv, 1);
            walk(0, &dtv, 1);
        }
        if (d.cost_integral && k >= d.quad_first_interval) {   // one integral cost edge per interval instead of the per-vertex terms (finite_differences_grid.cpp:62-77;
            // MinTimeQuadratic::only_last_n: hasIntegralTerms(k) = k >= _quad_k_min, hybrid_cost.h:209)
            const V trap[4] = {xk, uk, {(k + 1) * s, nx}, dtv}, left[3] = {xk, uk, dtv};
            H.stage_off[(size_t)k * 6 + 0] = (int32_t)H.rows[0].size();
            if (d.cost_integral == 1) walk(0, trap, 4); else walk(0, left, 3);
        }
        if ((terms & 1) && quad) { H.stage_off[(size_t)k * 6 + 0] = (int32_t)H.rows[0].size(); walk(0, &xk, 1); }
        if ((terms & 2) && quad) { H.stage_off[(size_t)k * 6 + 1] = (int32_t)H.rows[0].size(); walk(0, &uk, 1); }
        if (!d.cost_integral && (terms & 4) && k == 0) {
            H.dt_cost_off = (int32_t)H.rows[0].size();
            walk(0, &dtv, 1);
            walk(0, &dtv, 1);   // duplicated edge (nlp_functions.cpp:91-107)
        }
    }
    if (xf_unfixed > 0 && d.final_cost) { H.stage_off[(size_t)(N - 1) * 6 + 0] = (int32_t)H.rows[0].size(); walk(0, &xf, 1); }
    // equalities
    int eq_row = 0;
    for (int k = 0; k < N - 1; ++k) {
        const V verts[4] = {{k * s, nx}, {k * s + nx, nu}, {(k + 1) * s, nx}, dtv};
        H.stage_off[(size_t)k * 6 + 2] = (int32_t)H.rows[1].size();
        H.stage_off[(size_t)k * 6 + 4] = eq_row;
        walk(1, verts, 4);
        H.lin_off[(size_t)k * 2 + 0] = (int32_t)H.lin_rows.size();
        lin_walk(verts, 4, nx, eq_row);
        eq_row += nx;
    }
    if (xf_unfixed > 0 && d.final_eq) {
        H.stage_off[(size_t)(N - 1) * 6 + 2] = (int32_t)H.rows[1].size();
        H.stage_off[(size_t)(N - 1) * 6 + 4] = eq_row;
        walk(1, &xf, 1);
        H.lin_off[(size_t)(N - 1) * 2 + 0] = (int32_t)H.lin_rows.size();
        lin_walk(&xf, 1, S.fin_eq_dim, eq_row);
        eq_row += S.fin_eq_dim;
    }
    // inequalities
    int ineq_row = 0;
    for (int k = 0; k < N - 1; ++k) {
        if (d.stage_ineq == CORBO_HIP_INEQ_NONE) break;
        const V xk{k * s, nx};
        H.stage_off[(size_t)k * 6 + 3] = (int32_t)H.rows[2].size();
        H.stage_off[(size_t)k * 6 + 5] = ineq_row;
        walk(2, &xk, 1);
        H.lin_off[(size_t)k * 2 + 1] = (int32_t)H.lin_rows.size();
        lin_walk(&xk, 1, 1, eq_row + ineq_row);
        ineq_row += 1;
    }
    if (xf_unfixed > 0 && d.final_ineq != CORBO_HIP_FINAL_INEQ_NONE) {
        H.stage_off[(size_t)(N - 1) * 6 + 3] = (int32_t)H.rows[2].size();
        H.stage_off[(size_t)(N - 1) * 6 + 5] = ineq_row;
        walk(2, &xf, 1);
        H.lin_off[(size_t)(N - 1) * 2 + 1] = (int32_t)H.lin_rows.size();
        lin_walk(&xf, 1, 1, eq_row + ineq_row);
        ineq_row += 1;
    }
    for (int c = 0; c < 3; ++c) H.nnz[c] = (int32_t)H.rows[c].size();
    // finite bounds: identity rows in parameter order
    H.lin_bounds0 = (int32_t)H.lin_rows.size();
    for (int v = 0; v <= S.off_dt; ++v) {
        if (S.comp[v].bnd_row < 0) continue;
        H.lin_rows.push_back(eq_row + ineq_row + (S.comp[v].bnd_row - S.bnd_row0));
        H.lin_cols.push_back(S.comp[v].param);
    }
    H.lin_nnz = (int32_t)H.lin_rows.size();
}

// ---- block-tridiagonal tables (structure.hpp BtTables)
bool build_bt_tables(const Structure& S, const std::vector<int32_t>& jmap, int nnz_pad, int m_pad, int threads, BtTables& out, std::string* why)
{
    auto no = [&](const std::string& w) { if (why) *why = w; return false; };
    const int s = S.s, NB = S.N, n = S.dims.n, m = S.dims.m, nnz = S.dims.nnz;
    const bool arrow = S.dt_free;
    const int zero = nnz_pad + m_pad;
    // parameter -> (block, component); the dt parameter: block -1
    std::vector<int> pblk(n), pcmp(n);
    std::vector<int> col_of((size_t)NB * s, -1);   // (block, component) -> parameter or -1 (fixed component / pad)
    int dt_col = -1;
    for (int c = 0; c < n; ++c) {
        const int v = S.param_voff[c];
        if (S.dt_free && v == S.off_dt) { pblk[c] = -1; pcmp[c] = 0; dt_col = c; continue; }
        if (v >= S.off_xf) { pblk[c] = NB - 1; pcmp[c] = v - S.off_xf; }
        else { pblk[c] = v / s; pcmp[c] = v % s; }
        if (pblk[c] < 0 || pblk[c] >= NB || pcmp[c] >= s) return no("parameter outside the stage blocks");
        col_of[(size_t)pblk[c] * s + pcmp[c]] = c;
    }
    if (arrow && dt_col < 0) return no("free dt without a dt parameter");
    std::vector<std::vector<std::pair<int, int>>> rows(m);   // per residual row: (column, operand index of the Jacobian value)
    for (int i = 0; i < nnz; ++i) rows[S.jac_rows[i]].push_back({S.jac_cols[i], jmap[i]});
    // products per entry of the lower part of H, per column of the right-hand side
    using Prod = std::pair<uint32_t, uint32_t>;   // byte offsets of the two operands in the LDS array [J | values | 0]
    std::map<std::pair<int, int>, std::vector<Prod>> ent;
    std::vector<std::vector<Prod>> rhs(n);
    // the rows of the dynamics-defect edges are NOT in the lists: the kernel assembles those edges from their dense local Jacobians (bt_factor.hpp, 2b)
    auto defect_row = [&](int r) {
        if (r < S.eq_row0 || S.eq_stride <= 0) return false;
        const int q = (r - S.eq_row0) / S.eq_stride, o = (r - S.eq_row0) % S.eq_stride;
        return q < NB - 1 && o >= S.eq_defect_off && o < S.eq_defect_off + S.nx;
    };
    for (int r = 0; r < m; ++r) {
        if (defect_row(r)) continue;
        for (const auto& a : rows[r]) {
            rhs[a.first].push_back({(uint32_t)a.second * 8u, (uint32_t)(nnz_pad + r) * 8u});
            for (const auto& b : rows[r]) {
                // key: (row parameter, column parameter) with the row's block >= the column's block; the dt parameter is the last row
                const int ba = pblk[a.first] < 0 ? NB : pblk[a.first], bb = pblk[b.first] < 0 ? NB : pblk[b.first];
                if (ba < bb || (ba == bb && pcmp[a.first] < pcmp[b.first])) continue;
                if (ba < NB && ba - bb > 1) return no("H = J^T J is not block tridiagonal in the stage blocks");
                ent[{a.first, b.first}].push_back({(uint32_t)a.second * 8u, (uint32_t)b.second * 8u});
            }
        }
    }
    const int S2 = s * s, oA = 0, oB = S2, oG = 2 * S2, oZ = 2 * S2 + s;
    const int szp = (2 * S2 + s + (arrow ? s : 0)) | 1;
    struct Entry { uint32_t target; const std::vector<Prod>* list; };
    static const std::vector<Prod> none;
    std::vector<Entry> E;
    auto list_of = [&](int ca, int cb) -> const std::vector<Prod>* {
        if (ca < 0 || cb < 0) return &none;
        auto it = ent.find({ca, cb});
        return it == ent.end() ? &none : &it->second;
    };
    // entries: every diagonal entry (its flag carries the damping / the identity row) and every parameter's right-hand side (its flag: the first
    // factorisation's |rhs|_inf); of the others those with products -- the kernel's stage lanes write every slot of every block before the lists are added
    for (int k = 0; k < NB; ++k) {
        const int base = k * szp;
        for (int i = 0; i < s; ++i) {
            const int ci = col_of[(size_t)k * s + i];
            for (int j = 0; j <= i; ++j) {
                const int cj = col_of[(size_t)k * s + j];
                const uint32_t kind = (i == j) ? (ci >= 0 ? 1u : 2u) : 0u;   // bit 28: diagonal of a parameter (+ damping), bit 29: identity row
                const auto* l = list_of(ci, cj);
                if (i == j || !l->empty()) E.push_back({(uint32_t)(base + oA + i * s + j) | (kind << 28), l});
            }
            if (ci >= 0) E.push_back({(uint32_t)(base + oG + i) | (4u << 28), &rhs[ci]});   // bit 30: right-hand side (negated sum)
            if (arrow && !list_of(dt_col, ci)->empty()) E.push_back({(uint32_t)(base + oZ + i), list_of(dt_col, ci)});
            if (k + 1 < NB)
                for (int c = 0; c < s; ++c) {   // F_k[i][c] = H((k + 1, i), (k, c))
                    const auto* l = list_of(col_of[(size_t)(k + 1) * s + i], col_of[(size_t)k * s + c]);
                    if (!l->empty()) E.push_back({(uint32_t)(base + oB + i * s + c), l});
                }
        }
    }
    if (arrow) {
        E.push_back({(uint32_t)(NB * szp) | (8u << 28), list_of(dt_col, dt_col)});   // bit 31: the corner H(dt, dt)
        E.push_back({(uint32_t)(NB * szp + 1) | (4u << 28), &rhs[dt_col]});
    }
    size_t used = 0;
    for (const auto& e : E) used += e.list->size();
    size_t all = 0;
    for (const auto& kv : ent) all += kv.second.size();
    for (const auto& r : rhs) all += r.size();
    if (used != all) return no("an entry of H = J^T J falls outside the block-tridiagonal pattern");
    std::stable_sort(E.begin(), E.end(), [](const Entry& a, const Entry& b) { return a.list->size() > b.list->size(); });
    out = BtTables{};
    out.S = s; out.NB = NB; out.szp = szp; out.threads = threads; out.arrow = arrow;
    // super-rounds of four rounds (4 * threads entries): lane t of super-round sr holds the entries (4 sr + c) * threads + t, c = 0 .. 3; their lists are
    // padded to the super-round's longest; step i of a super-round = the four operand pairs [i] of the lane's entries as one 16-byte word group
    const size_t per_sr = 4 * (size_t)threads;
    out.rounds = (int)((E.size() + per_sr - 1) / per_sr);
    out.off.assign(1, 0);
    out.target.assign((size_t)out.rounds * per_sr, (uint32_t)(NB * szp + 2));   // (no entry: the trash slot)
    const uint32_t zoff = (uint32_t)zero * 8u;
    for (int sr = 0; sr < out.rounds; ++sr) {
        size_t L = 0;
        for (size_t e = sr * per_sr; e < std::min(E.size(), (sr + 1) * per_sr); ++e) L = std::max(L, E[e].list->size());
        L = (L + 3) / 4 * 4;   // (the kernel's loop body is four steps: its window of loads in flight has static registers)
        const size_t o0 = (size_t)out.off.back();
        out.pairs.resize((o0 + L) * per_sr * 2, zoff);
        // Neighbours in the sorted list are the same entry of consecutive blocks: their operands sit a whole stage apart in the Jacobian -- 24 doubles for
        // the unicycle's defect blocks, i.e. FOUR distinct LDS banks for the 64 lanes of a wave (measured: 345 cycles per step, all of it bank conflicts).
        // The lanes of a super-round take its entries in a scattered order instead (a multiplicative permutation of the 4 * threads positions).
        const size_t P = (per_sr % 2 == 0) ? (size_t)(0.618 * per_sr) | 1 : 1;
        for (int c = 0; c < 4; ++c)
            for (int t = 0; t < threads; ++t) {
                const size_t e = sr * per_sr + (((size_t)c * threads + t) * P) % per_sr;
                if (e >= E.size()) continue;
                out.target[(sr * (size_t)threads + t) * 4 + c] = E[e].target;
                for (size_t i = 0; i < E[e].list->size(); ++i) {
                    out.pairs[((o0 + i) * threads + t) * 8 + 2 * c]     = (*E[e].list)[i].first;
                    out.pairs[((o0 + i) * threads + t) * 8 + 2 * c + 1] = (*E[e].list)[i].second;
                }
            }
        out.off.push_back((int32_t)(o0 + L));
    }
    out.pairs.resize(out.pairs.size() + 4 * per_sr * 2, zoff);     // (the kernel's window of loads runs four steps ahead of the last one)
    return true;
}

}  // namespace corbo_hip
