// graph_recogniser.cpp -- see the header.  Compiled against the reference's headers (inside the reference's build tree).
#include "graph_recogniser.h"

#include <corbo-numerics/explicit_integrators.h>
#include <corbo-numerics/finite_differences_collocation.h>
#include <corbo-optimal-control/structured_ocp/discretization_grids/finite_differences_grid.h>
#include <corbo-optimal-control/structured_ocp/discretization_grids/finite_differences_variable_grid.h>
#include <corbo-optimal-control/structured_ocp/discretization_grids/multiple_shooting_grid.h>
#include <corbo-optimal-control/structured_ocp/discretization_grids/multiple_shooting_variable_grid.h>
#include <corbo-optimal-control/structured_ocp/edges/finite_differences_collocation_edges.h>
#include <corbo-optimal-control/structured_ocp/edges/multiple_shooting_edges.h>
#include <corbo-optimization/hyper_graph/generic_edge.h>
#include <corbo-optimization/hyper_graph/scalar_vertex.h>
#include <corbo-optimization/hyper_graph/vector_vertex.h>
#include <corbo-systems/benchmark/linear_benchmark_systems.h>
#include <corbo-systems/benchmark/nonlinear_benchmark_systems.h>

#include <cmath>
#include <memory>
#include <mutex>
#include <algorithm>
#include <cstring>
#include <sstream>
#include <vector>

namespace corbo {
namespace {

// ---- private data members of reference classes that have no getter, reached without touching the reference's headers: the usual
//      access rules do not apply to the arguments of an explicit instantiation ([temp.spec]/6), and the instantiation publishes the
//      pointer-to-member through a static initialiser.
template <class Tag>
struct MemberOf
{
    static typename Tag::type ptr;
};
template <class Tag>
typename Tag::type MemberOf<Tag>::ptr;
template <class Tag, typename Tag::type P>
struct Publish
{
    struct Filler
    {
        Filler() { MemberOf<Tag>::ptr = P; }
    };
    static Filler filler;
};
template <class Tag, typename Tag::type P>
typename Publish<Tag, P>::Filler Publish<Tag, P>::filler;

#define CORBO_HIP_PRIVATE_MEMBER(TAG, CLASS, MEMBER_TYPE, MEMBER) \
    struct TAG { using type = MEMBER_TYPE CLASS::*; };           \
    template struct Publish<TAG, &CLASS::MEMBER>;

CORBO_HIP_PRIVATE_MEMBER(FdEdgeDynamics, FDCollocationEdge, SystemDynamicsInterface::Ptr, _dynamics)
CORBO_HIP_PRIVATE_MEMBER(FdEdgeScheme, FDCollocationEdge, FiniteDifferencesCollocationInterface::Ptr, _fd_eval)
CORBO_HIP_PRIVATE_MEMBER(TrapEqEdgeDynamics, TrapezoidalIntegralEqualityDynamicsEdge, SystemDynamicsInterface::Ptr, _dynamics)
CORBO_HIP_PRIVATE_MEMBER(TrapEqEdgeScheme, TrapezoidalIntegralEqualityDynamicsEdge, FiniteDifferencesCollocationInterface::Ptr, _fd_eval)
CORBO_HIP_PRIVATE_MEMBER(MsEdgeDynamics, MSVariableDynamicsOnlyEdge, SystemDynamicsInterface::Ptr, _dynamics)
CORBO_HIP_PRIVATE_MEMBER(MsEdgeIntegrator, MSVariableDynamicsOnlyEdge, NumericalIntegratorExplicitInterface::Ptr, _integrator)
CORBO_HIP_PRIVATE_MEMBER(MixedEdgeDynamics, MultipleShootingEdgeSingleControl, SystemDynamicsInterface::Ptr, _dynamics)
CORBO_HIP_PRIVATE_MEMBER(MixedEdgeIntegrator, MultipleShootingEdgeSingleControl, NumericalIntegratorExplicitInterface::Ptr, _integrator)
CORBO_HIP_PRIVATE_MEMBER(MixedEdgeStageCost, MultipleShootingEdgeSingleControl, StageCost::ConstPtr, _stage_cost)
CORBO_HIP_PRIVATE_MEMBER(DuffingDamping, DuffingOscillator, double, _damping)
CORBO_HIP_PRIVATE_MEMBER(DuffingAlpha, DuffingOscillator, double, _spring_alpha)
CORBO_HIP_PRIVATE_MEMBER(DuffingBeta, DuffingOscillator, double, _spring_beta)
CORBO_HIP_PRIVATE_MEMBER(PendulumM, SimplePendulum, double, _m)
CORBO_HIP_PRIVATE_MEMBER(PendulumL, SimplePendulum, double, _l)
CORBO_HIP_PRIVATE_MEMBER(PendulumG, SimplePendulum, double, _g)
CORBO_HIP_PRIVATE_MEMBER(PendulumRho, SimplePendulum, double, _rho)
CORBO_HIP_PRIVATE_MEMBER(MasslessOmega, MasslessPendulum, double, _omega0)
CORBO_HIP_PRIVATE_MEMBER(ToyMu, ToyExample, double, _mu)
CORBO_HIP_PRIVATE_MEMBER(CartMc, CartPole, double, _mc)
CORBO_HIP_PRIVATE_MEMBER(CartMp, CartPole, double, _mp)
CORBO_HIP_PRIVATE_MEMBER(CartL, CartPole, double, _l)
CORBO_HIP_PRIVATE_MEMBER(CartG, CartPole, double, _g)

template <class Tag, class Obj>
auto& member(Obj& o)
{
    return o.*(MemberOf<Tag>::ptr);
}

bool fail(std::string* reason, const std::string& why)
{
    if (reason) *reason = why;
    return false;
}

// temporarily changed vertex values: restored on scope exit (the graph is exactly as it was found when the recogniser returns)
struct VertexGuard
{
    VertexInterface* v;
    std::vector<double> keep;
    explicit VertexGuard(VertexInterface* vtx) : v(vtx), keep(vtx->getData(), vtx->getData() + vtx->getDimension()) {}
    ~VertexGuard() { std::memcpy(v->getDataRaw(), keep.data(), keep.size() * sizeof(double)); }
};

Eigen::VectorXd evalEdge(BaseEdge& e)
{
    Eigen::VectorXd v(e.getDimension());
    e.computeValues(v);
    return v;
}

// A row of a least-squares term  r_i = w_i (x_i - ref_i)  (diagonal weights, quadratic_cost.cpp:116-118, final_state_cost.cpp:76-92)
// identified EXACTLY through the edge's own evaluation: ref_i is the x_i at which the row is exactly zero (x - ref rounds to zero only
// for x == ref), and w_i = r_i / d at x_i = ref_i + d with d a power of two for which (ref_i + d) - ref_i == d holds in floating point
// (then w d is a pure exponent shift).  Off-diagonal responses must vanish.  false: not a diagonal affine term.
bool identifyDiagonalAffine(BaseEdge& e, VertexInterface* v, Eigen::VectorXd* w, Eigen::VectorXd* ref)
{
    const int n = v->getDimension();
    if (e.getDimension() != n) return false;
    VertexGuard guard(v);
    double* x = v->getDataRaw();
    w->resize(n);
    ref->resize(n);
    for (int i = 0; i < n; ++i)
    {
        const double x0 = x[i];
        const Eigen::VectorXd r0 = evalEdge(e);
        x[i] = x0 + 1.0;
        const Eigen::VectorXd r1 = evalEdge(e);
        for (int j = 0; j < n; ++j)
            if (j != i && r1[j] != r0[j]) return false;   // coupled components: not diagonal
        double wi = r1[i] - r0[i];
        if (wi == 0.0 || !std::isfinite(wi))
        {   // zero weight: the row is identically zero, any reference will do
            x[i] = x0 + 1024.0;
            if (evalEdge(e)[i] != 0.0 || r0[i] != 0.0) return false;
            (*w)[i] = 0.0; (*ref)[i] = 0.0;
            x[i] = x0;
            continue;
        }
        // root of the row: zero itself (terms without a reference), else Newton on an exactly linear function lands within an ulp or
        // two; then walk to the exact zero
        double xr = x0 - r0[i] / wi;
        bool found = false;
        x[i] = 0.0;
        if (evalEdge(e)[i] == 0.0) { xr = 0.0; found = true; }
        for (int it = 0; it < 8 && !found; ++it)
        {
            x[i] = xr;
            const double r = evalEdge(e)[i];
            if (r == 0.0) { found = true; break; }
            const double step = r / wi;
            double xn = xr - step;
            if (xn == xr) xn = std::nextafter(xr, (step > 0) ? -INFINITY : INFINITY);
            xr = xn;
        }
        if (!found) return false;
        (*ref)[i] = xr;
        // exact weight
        double d = std::ldexp(1.0, std::max(-20, std::min(20, (xr == 0.0) ? 0 : std::ilogb(xr))));
        bool ok = false;
        for (int t = 0; t < 40 && !ok; ++t, d *= 2.0)
        {
            volatile double xp = xr + d;
            if ((double)xp - xr == d)
            {
                x[i] = xp;
                (*w)[i] = evalEdge(e)[i] / d;
                ok = true;
            }
        }
        if (!ok) return false;
        // the model reproduces the edge at the original point bit for bit
        x[i] = x0;
        if (evalEdge(e)[i] != (*w)[i] * (x0 - (*ref)[i])) return false;
    }
    return true;
}

// U * xd the way the reference evaluates `cost.noalias() = _Q_sqrt * xd` (Eigen's column-major gemv into a zeroed destination: one running
// sum per row over the columns, a full block of four columns added pairwise) -- what the device and the oracle restate
Eigen::VectorXd upperTimes(const Eigen::MatrixXd& U, const Eigen::VectorXd& xd)
{
    const int n = (int)U.rows();
    Eigen::VectorXd out(n);
    for (int i = 0; i < n; ++i)
    {
        if (n == 4) { out[i] = 0.0 + ((U(i, 0) * xd[0] + U(i, 1) * xd[1]) + (U(i, 2) * xd[2] + U(i, 3) * xd[3])); continue; }
        double acc = 0.0;
        for (int j = 0; j < n; ++j) acc += U(i, j) * xd[j];
        out[i] = acc;
    }
    return out;
}

// A least-squares term with a NON-DIAGONAL weight,  r = U (x - ref)  with U the upper Cholesky factor the reference keeps
// (quadratic_cost.cpp:36-55, 116-118; final_state_cost.cpp:38-58, 88-90), identified EXACTLY through the edge's own evaluation: the
// references bottom-up (with x_j = ref_j for j > i row i is U_ii (x_i - ref_i) plus exact zeros: its root is ref_i), then column j of U
// from x = ref + d e_j with d a power of two for which (ref_j + d) - ref_j == d.  false: not such a term (lower-triangular responses,
// a singular diagonal, or the model does not reproduce the edge at the original point bit for bit).
bool identifyUpperAffine(BaseEdge& e, VertexInterface* v, Eigen::MatrixXd* U, Eigen::VectorXd* ref)
{
    const int n = v->getDimension();
    if (e.getDimension() != n || n > 4) return false;
    VertexGuard guard(v);
    double* x = v->getDataRaw();
    const std::vector<double> x0(x, x + n);
    const Eigen::VectorXd r0 = evalEdge(e);
    for (int j = 0; j < n; ++j)
    {   // structure: x_j moves rows 0 .. j only
        x[j] = x0[j] + 1.0;
        const Eigen::VectorXd r1 = evalEdge(e);
        x[j] = x0[j];
        for (int i = j + 1; i < n; ++i)
            if (r1[i] != r0[i]) return false;
        if (r1[j] == r0[j] || !std::isfinite(r1[j])) return false;   // a Cholesky factor has a positive diagonal
    }
    ref->resize(n);
    for (int i = n - 1; i >= 0; --i)
    {
        x[i] = x0[i];
        const double a0 = evalEdge(e)[i];
        x[i] = x0[i] + 1.0;
        const double wi = evalEdge(e)[i] - a0;
        double xr = x0[i] - a0 / wi;
        bool found = false;
        x[i] = 0.0;   // terms without a reference (controls): the root is zero itself
        if (evalEdge(e)[i] == 0.0) { xr = 0.0; found = true; }
        for (int it = 0; it < 8 && !found; ++it)
        {
            x[i] = xr;
            const double r = evalEdge(e)[i];
            if (r == 0.0) { found = true; break; }
            const double step = r / wi;
            double xn = xr - step;
            if (xn == xr) xn = std::nextafter(xr, (step > 0) ? -INFINITY : INFINITY);
            xr = xn;
        }
        if (!found) return false;
        (*ref)[i] = xr;
        x[i]      = xr;   // rows above see an exact zero difference from here on
    }
    *U = Eigen::MatrixXd::Zero(n, n);
    for (int j = 0; j < n; ++j)
    {
        const double xr = (*ref)[j];
        double d = std::ldexp(1.0, std::max(-20, std::min(20, (xr == 0.0) ? 0 : std::ilogb(xr))));
        bool ok = false;
        for (int t = 0; t < 40 && !ok; ++t, d *= 2.0)
        {
            volatile double xp = xr + d;
            if ((double)xp - xr == d)
            {
                x[j] = xp;
                const Eigen::VectorXd r = evalEdge(e);
                for (int i = 0; i <= j; ++i) (*U)(i, j) = r[i] / d;
                for (int i = j + 1; i < n; ++i)
                    if (r[i] != 0.0) return false;
                ok = true;
            }
        }
        x[j] = xr;
        if (!ok) return false;
    }
    // the model reproduces the edge at the original point bit for bit
    Eigen::VectorXd xd(n);
    for (int i = 0; i < n; ++i) { x[i] = x0[i]; xd[i] = x0[i] - (*ref)[i]; }
    const Eigen::VectorXd r = evalEdge(e), m = upperTimes(*U, xd);
    for (int i = 0; i < n; ++i)
        if (r[i] != m[i]) return false;
    return true;
}

// TerminalPartialEqualityConstraint (final_state_constraints.h:198-300): rows  x_i - ref_i  for the ACTIVE components of the vertex only, in
// component order.  Identified through the edge itself: component i is active iff moving it changes a row -- which must be the next row --;
// its reference is that row's exact root, its weight must be exactly 1.  *mask: bit i = active; ref: the active components' references.
bool identifyPartialIdentity(BaseEdge& e, VertexInterface* v, unsigned* mask, Eigen::VectorXd* ref)
{
    const int n = v->getDimension(), m = e.getDimension();
    if (m < 1 || m > n) return false;
    VertexGuard guard(v);
    double* x = v->getDataRaw();
    *mask = 0;
    *ref  = Eigen::VectorXd::Zero(n);
    int next_row = 0;
    for (int i = 0; i < n; ++i)
    {
        const double x0 = x[i];
        const Eigen::VectorXd r0 = evalEdge(e);
        x[i] = x0 + 1.0;
        const Eigen::VectorXd r1 = evalEdge(e);
        int hit = -1, hits = 0;
        for (int r = 0; r < m; ++r)
            if (r1[r] != r0[r]) { hit = r; ++hits; }
        if (hits == 0) { x[i] = x0; continue; }   // inactive component
        if (hits != 1 || hit != next_row) return false;
        // root of the row (Newton on an exactly linear function, then a walk to the exact zero), and the unit weight
        const double wi = r1[hit] - r0[hit];
        double xr = x0 - r0[hit] / wi;
        bool found = false;
        x[i] = 0.0;
        if (evalEdge(e)[hit] == 0.0) { xr = 0.0; found = true; }
        for (int it = 0; it < 8 && !found; ++it)
        {
            x[i] = xr;
            const double r = evalEdge(e)[hit];
            if (r == 0.0) { found = true; break; }
            double xn = xr - r / wi;
            if (xn == xr) xn = std::nextafter(xr, (r / wi > 0) ? -INFINITY : INFINITY);
            xr = xn;
        }
        if (!found) return false;
        double d = std::ldexp(1.0, std::max(-20, std::min(20, (xr == 0.0) ? 0 : std::ilogb(xr))));
        bool ok = false;
        for (int t = 0; t < 40 && !ok; ++t, d *= 2.0)
        {
            volatile double xp = xr + d;
            if ((double)xp - xr == d) { x[i] = xp; ok = (evalEdge(e)[hit] == d); break; }
        }
        x[i] = x0;
        if (!ok) return false;
        (*ref)[i] = xr;
        *mask |= 1u << i;
        ++next_row;
    }
    return next_row == m;
}

bool sameMatrix(const Eigen::MatrixXd& a, const Eigen::MatrixXd& b)
{
    return a.rows() == b.rows() && a.cols() == b.cols() && (a.array() == b.array()).all();
}

bool sameVector(const Eigen::VectorXd& a, const Eigen::VectorXd& b)
{
    return a.size() == b.size() && (a.array() == b.array()).all();
}

// ---- dynamics object -> device dynamics id + parameters
bool describeDynamics(SystemDynamicsInterface& dyn, corbo_hip_problem_desc& d, std::string* reason)
{
    const int nx = dyn.getStateDimension(), nu = dyn.getInputDimension();
    for (double& p : d.dyn_params) p = 0.0;
    if (auto* s = dynamic_cast<VanDerPolOscillator*>(&dyn)) { d.dynamics = CORBO_HIP_DYN_VAN_DER_POL; d.dyn_params[0] = s->getDampingCoefficient(); return true; }
    if (auto* s = dynamic_cast<SerialIntegratorSystem*>(&dyn)) { d.dynamics = CORBO_HIP_DYN_SERIAL_INTEGRATOR; d.dyn_params[0] = s->getTimeConstant(); return true; }
    if (auto* s = dynamic_cast<ParallelIntegratorSystem*>(&dyn)) { d.dynamics = CORBO_HIP_DYN_PARALLEL_INTEGRATOR; d.dyn_params[0] = s->getTimeConstant(); return true; }
    if (auto* s = dynamic_cast<DuffingOscillator*>(&dyn))
    {
        d.dynamics = CORBO_HIP_DYN_DUFFING;
        d.dyn_params[0] = member<DuffingDamping>(*s); d.dyn_params[1] = member<DuffingAlpha>(*s); d.dyn_params[2] = member<DuffingBeta>(*s);
        return true;
    }
    if (dynamic_cast<FreeSpaceRocket*>(&dyn)) { d.dynamics = CORBO_HIP_DYN_FREE_SPACE_ROCKET; return true; }
    if (auto* s = dynamic_cast<SimplePendulum*>(&dyn))
    {
        d.dynamics = CORBO_HIP_DYN_SIMPLE_PENDULUM;
        d.dyn_params[0] = member<PendulumM>(*s); d.dyn_params[1] = member<PendulumL>(*s); d.dyn_params[2] = member<PendulumG>(*s);
        d.dyn_params[3] = member<PendulumRho>(*s);
        return true;
    }
    if (auto* s = dynamic_cast<MasslessPendulum*>(&dyn)) { d.dynamics = CORBO_HIP_DYN_MASSLESS_PENDULUM; d.dyn_params[0] = member<MasslessOmega>(*s); return true; }
    if (auto* s = dynamic_cast<ToyExample*>(&dyn)) { d.dynamics = CORBO_HIP_DYN_TOY_EXAMPLE; d.dyn_params[0] = member<ToyMu>(*s); return true; }
    if (dynamic_cast<ArtsteinsCircle*>(&dyn)) { d.dynamics = CORBO_HIP_DYN_ARTSTEINS_CIRCLE; return true; }
    if (auto* s = dynamic_cast<CartPole*>(&dyn))
    {
        if (member<CartMc>(*s) != 1.0 || member<CartMp>(*s) != 0.3 || member<CartL>(*s) != 0.5 || member<CartG>(*s) != 9.81)
            return fail(reason, "CartPole with non-default parameters (the device model has the reference's defaults built in)");
        d.dynamics = CORBO_HIP_DYN_CART_POLE;
        return true;
    }
    Eigen::VectorXd x = Eigen::VectorXd::Zero(nx), u = Eigen::VectorXd::Zero(nu), f(nx);
    if (dynamic_cast<LinearStateSpaceModel*>(&dyn))
    {   // f = A x + B u is exact on unit vectors (every other term of a row sum is an exact zero)
        if (nx * nx > 16 || nx * nu > 12) return fail(reason, "LinearStateSpaceModel too large for the device table");
        d.dynamics = CORBO_HIP_DYN_LINEAR_STATE_SPACE;
        for (int j = 0; j < nx; ++j)
        {
            x.setZero(); x[j] = 1.0;
            dyn.dynamics(x, u, f);
            for (int i = 0; i < nx; ++i) d.lin_a[i * nx + j] = f[i];
        }
        x.setZero();
        for (int j = 0; j < nu; ++j)
        {
            u.setZero(); u[j] = 1.0;
            dyn.dynamics(x, u, f);
            for (int i = 0; i < nx; ++i) d.lin_b[i * nu + j] = f[i];
        }
        return true;
    }
    // user systems: matched against the device library's plug-in models by evaluating them (corbo_hip_eval_dynamics runs the device's
    // formula on the GPU) at deterministic probe points; sin / cos may differ from the host's by an ulp
    struct Candidate { int id; int nx, nu; double prm[5]; const char* name; };
    const Candidate cands[] = {{CORBO_HIP_DYN_UNICYCLE, 3, 2, {0, 0, 0, 0, 0}, "unicycle"},
                               {CORBO_HIP_DYN_QUADROTOR, 12, 4, {9.81, 1.0, 0.01, 0.01, 0.02}, "quadrotor (g = 9.81, m = 1, I = 0.01 / 0.01 / 0.02)"},
// the user models dropped into csrc/models/ (registry generated by the build), each with the parameter values its header states
#if __has_include("../csrc/models/_registry.inc")
#define CORBO_HIP_USER_MODEL(NAME, SLOT, NX_, NU_, P0, P1, P2, P3) {CORBO_HIP_DYN_USER + SLOT, NX_, NU_, {P0, P1, P2, P3, 0}, "user model " #NAME},
#include "../csrc/models/_registry.inc"
#undef CORBO_HIP_USER_MODEL
#endif
    };
    // the device's answers at the probe points do not change while the process lives: ONE device evaluation per candidate and process, kept here
    // (the adapter's model tracking walks through this function once per new run -- a kernel launch and a synchronisation each time otherwise)
    static std::mutex cache_mutex;
    static std::vector<std::vector<double>> device_values(sizeof(cands) / sizeof(cands[0]));
    for (const Candidate& c : cands)
    {
        if (c.nx != nx || c.nu != nu) continue;
        corbo_hip_problem_desc t = d;
        t.dynamics = c.id; t.nx = nx; t.nu = nu;
        for (int i = 0; i < 5; ++i) t.dyn_params[i] = c.prm[i];
        const int P = 6;
        std::vector<double> xs(P * nx), us(P * nu), fd(P * nx);
        for (int p = 0; p < P; ++p)
        {
            for (int i = 0; i < nx; ++i) xs[p * nx + i] = 0.37 * std::sin(1.0 + 1.7 * i + 0.9 * p) + 0.05 * p;
            for (int i = 0; i < nu; ++i) us[p * nu + i] = 0.8 * std::cos(0.3 + 2.1 * i + 1.3 * p) + ((c.id == CORBO_HIP_DYN_QUADROTOR && i == 0) ? 9.0 : 0.0);   // (|u| < 0.8: inside tan()'s first branch for a steering angle)
        }
        {
            std::lock_guard<std::mutex> lock(cache_mutex);
            std::vector<double>& kept = device_values[&c - cands];
            if (kept.empty())
            {
                if (corbo_hip_eval_dynamics(&t, P, xs.data(), us.data(), fd.data()) != CORBO_HIP_OK) continue;
                kept = fd;
            }
            else fd = kept;
        }
        bool same = true;
        for (int p = 0; p < P && same; ++p)
        {
            dyn.dynamics(Eigen::Map<Eigen::VectorXd>(&xs[p * nx], nx), Eigen::Map<Eigen::VectorXd>(&us[p * nu], nu), f);
            for (int i = 0; i < nx; ++i)
                if (!(std::abs(f[i] - fd[p * nx + i]) <= 1e-14 * (1.0 + std::abs(f[i])))) same = false;
        }
        if (same)
        {
            d.dynamics = c.id;
            for (int i = 0; i < 5; ++i) d.dyn_params[i] = c.prm[i];
            return true;
        }
    }
    return fail(reason, "system dynamics class is neither one of the reference's benchmark systems nor one of the device library's plug-in models "
                        "(setDeviceModel() states a model explicitly)");
}

// ---- integral-form constraint edges and control-deviation edges (user stage functions): the integrand / term is probed THROUGH the edge.
// An integral edge evaluates 0.5 dt (f(x1, u) + f(x2, u)) resp. f(x1, u) dt (finite_differences_collocation_edges.h:149-459): with dt = 1 and
// x1 = x2 both are f(x, u) exactly (f + f and the halving are exact).  `row`: the value row of the integrand inside the edge.
struct IntegrandProbe
{
    BaseEdge& e;
    std::vector<VertexInterface*> xv;   // the state vertices of the edge (one or two), all set to the probe point
    VertexInterface* u;
    VertexInterface* dt;
    int row;
    std::vector<VertexGuard> guards;
    IntegrandProbe(BaseEdge& edge, std::vector<VertexInterface*> xs_, VertexInterface* u_, VertexInterface* dt_, int row_) : e(edge), xv(std::move(xs_)), u(u_), dt(dt_), row(row_)
    {
        guards.reserve(xv.size() + 2);
        for (VertexInterface* v : xv) guards.emplace_back(v);
        guards.emplace_back(u);
        guards.emplace_back(dt);
        dt->getDataRaw()[0] = 1.0;
    }
    double operator()(const double* x, const double* uu)
    {
        for (VertexInterface* v : xv) std::memcpy(v->getDataRaw(), x, v->getDimension() * sizeof(double));
        std::memcpy(u->getDataRaw(), uu, u->getDimension() * sizeof(double));
        return evalEdge(e)[row];
    }
};

// LinearIntegralEquality-like integrand  a^T x + b^T u - c: prm = a (nx), b (nu), c.  `known`: check against these parameters only (later intervals).
bool identifyLinearIntegrand(IntegrandProbe& f, int nx, int nu, double* prm, bool known)
{
    std::vector<double> x(nx, 0.0), u(nu, 0.0);
    auto mine = [&](const double* p) {
        double acc = 0.0;
        for (int i = 0; i < nx; ++i) acc += p[i] * x[i];
        for (int i = 0; i < nu; ++i) acc += p[nx + i] * u[i];
        return acc - p[nx + nu];
    };
    if (!known)
    {
        const double c = -f(x.data(), u.data());   // f(0, 0) = 0 - c exactly
        for (int i = 0; i < nx; ++i) { x[i] = 1.0; prm[i] = f(x.data(), u.data()) + c; x[i] = 0.0; }
        for (int i = 0; i < nu; ++i) { u[i] = 1.0; prm[nx + i] = f(x.data(), u.data()) + c; u[i] = 0.0; }
        prm[nx + nu] = c;
    }
    for (int p = 0; p < (known ? 1 : 4); ++p)
    {   // the device formula (kernels.hip xedge_values) at probe points
        for (int i = 0; i < nx; ++i) x[i] = 0.4 * std::sin(0.3 + 1.1 * i + 1.7 * p);
        for (int i = 0; i < nu; ++i) u[i] = 0.5 * std::cos(0.9 + 0.7 * i + 2.3 * p);
        const double m = mine(prm);
        if (!(std::abs(f(x.data(), u.data()) - m) <= 1e-13 * (1.0 + std::abs(m)))) return false;
    }
    return true;
}

// keep-out ball as an integrand (independent of u): prm = cx, cy, cz, r
bool identifyBallIntegrand(IntegrandProbe& f, int nx, int nu, double* prm, bool known)
{
    if (nx < 3) return false;
    std::vector<double> x(nx, 0.0), u(nu, 0.0);
    if (!known)
    {
        double ctr[3];
        for (int i = 0; i < 3; ++i)
        {
            x[i] = 1.0;  const double cp = f(x.data(), u.data());
            x[i] = -1.0; const double cm = f(x.data(), u.data());
            x[i] = 0.0;
            ctr[i] = (cp - cm) / 4.0;
        }
        for (int i = 0; i < 3; ++i) x[i] = ctr[i];
        const double r2 = f(x.data(), u.data());
        if (!(r2 > 0)) return false;
        prm[0] = ctr[0]; prm[1] = ctr[1]; prm[2] = ctr[2]; prm[3] = std::sqrt(r2);
    }
    for (int p = 0; p < (known ? 1 : 4); ++p)
    {
        for (int i = 0; i < nx; ++i) x[i] = 0.3 * std::sin(0.7 + 1.3 * i + 2.1 * p);
        for (int i = 0; i < nu; ++i) u[i] = 0.5 * std::cos(0.2 + 0.9 * i + 1.3 * p);
        const double dx = x[0] - prm[0], dy = x[1] - prm[1], dz = x[2] - prm[2];
        const double m = prm[3] * prm[3] - (dx * dx + dy * dy + dz * dz);
        const double v0 = f(x.data(), u.data());
        if (!(std::abs(v0 - m) <= 1e-13 * (1.0 + std::abs(m)))) return false;
        for (int i = 0; i < nu; ++i) u[i] += 1.0;   // must not depend on the control
        if (f(x.data(), u.data()) != v0) return false;
    }
    return true;
}

// the control-deviation edge type the grids create (nlp_functions.cpp:117-131, 152-186)
using ControlDeviationEdge = TernaryVectorScalarVertexEdge<StageFunction, &StageFunction::computeNonIntegralControlDeviationTerm>;

// input-rate limit  ((u_k - u_prev) / dt_prev)^2 - r_max^2  per control: prm = r_max (nu)
bool identifyRateLimit(BaseEdge& e, VertexInterface* uk, VertexInterface* up, VertexInterface* dtp, int nu, double* prm, bool known)
{
    if (e.getDimension() != nu) return false;
    VertexGuard g0(uk), g1(up), g2(dtp);
    double *a = uk->getDataRaw(), *b = up->getDataRaw(), *t = dtp->getDataRaw();
    if (!known)
    {
        for (int i = 0; i < nu; ++i) { a[i] = 0.0; b[i] = 0.0; }
        t[0] = 1.0;
        const Eigen::VectorXd v = evalEdge(e);   // - r_max^2
        for (int i = 0; i < nu; ++i)
        {
            if (!(v[i] < 0)) return false;
            prm[i] = std::sqrt(-v[i]);
        }
    }
    for (int p = 0; p < (known ? 1 : 3); ++p)
    {
        for (int i = 0; i < nu; ++i) { a[i] = 0.6 * std::sin(0.4 + 1.9 * i + 1.1 * p); b[i] = 0.5 * std::cos(1.2 + 0.8 * i + 2.9 * p); }
        t[0] = 0.05 + 0.07 * (p + 1);
        const Eigen::VectorXd v = evalEdge(e);
        for (int i = 0; i < nu; ++i)
        {
            const double dd = (a[i] - b[i]) / t[0];
            const double m  = dd * dd - prm[i] * prm[i];
            if (!(std::abs(v[i] - m) <= 1e-13 * (1.0 + std::abs(m)))) return false;
        }
    }
    return true;
}

// keep-out ball  c(x) = r^2 - |x[0:3] - centre|^2  (stage inequality of cfg 5), identified from evaluations and verified at probes
bool identifyBall(BaseEdge& e, VertexInterface* v, double* prm /*cx, cy, cz, r*/)
{
    const int n = v->getDimension();
    if (e.getDimension() != 1 || n < 3) return false;
    VertexGuard guard(v);
    double* x = v->getDataRaw();
    for (int i = 0; i < n; ++i) x[i] = 0.0;
    const double c0 = evalEdge(e)[0];
    double ctr[3];
    for (int i = 0; i < 3; ++i)
    {   // c(e_i) - c(-e_i) = 4 centre_i   (exact for moderate centres)
        x[i] = 1.0;  const double cp = evalEdge(e)[0];
        x[i] = -1.0; const double cm = evalEdge(e)[0];
        x[i] = 0.0;
        ctr[i] = (cp - cm) / 4.0;
    }
    for (int i = 0; i < 3; ++i) x[i] = ctr[i];
    const double r2 = evalEdge(e)[0];   // value at the centre
    if (!(r2 > 0)) return false;
    prm[0] = ctr[0]; prm[1] = ctr[1]; prm[2] = ctr[2]; prm[3] = std::sqrt(r2);
    // verify: the device formula (model.hpp ineq_ball) at probe points, components >= 3 do not matter
    (void)c0;
    for (int p = 0; p < 4; ++p)
    {
        for (int i = 0; i < n; ++i) x[i] = 0.3 * std::sin(0.7 + 1.3 * i + 2.1 * p);
        const double dx = x[0] - prm[0], dy = x[1] - prm[1], dz = x[2] - prm[2];
        const double mine = prm[3] * prm[3] - (dx * dx + dy * dy + dz * dz);
        if (!(std::abs(evalEdge(e)[0] - mine) <= 1e-13 * (1.0 + std::abs(mine)))) return false;
    }
    return true;
}

// A USER stage function of csrc/stage_functions/ (kind 0: the stage inequalities' state term on x_k, kind 1: their control term on u_k): the edge's value against
// the device's own formula -- corbo_hip_eval_stage_function evaluates the very template the kernels compile -- at probe points.  Parameters: the shipped
// examples have the form f(v) - prm[0]^2, so prm[0] = sqrt(-c(0)) (README.md there: a function with other parameters is matched with the registered defaults).
bool identifyUserStageFunction(BaseEdge& e, VertexInterface* v, int kind, int32_t& id_out, double* prm /*[8]*/)
{
    const int n = v->getDimension();
    if (e.getDimension() != 1) return false;
    VertexGuard guard(v);
    double* x = v->getDataRaw();
    for (int i = 0; i < n; ++i) x[i] = 0.0;
    const double c0 = evalEdge(e)[0];
    for (int slot = 0; slot < 16; ++slot)
    {
        const int id = CORBO_HIP_STAGE_FN_USER + slot;
        if (corbo_hip_stage_function_kind(id) != kind) continue;
        double p[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        if (c0 < 0) p[0] = std::sqrt(-c0);
        bool ok = true;
        for (int q = 0; q < 5 && ok; ++q)
        {
            for (int i = 0; i < n; ++i) x[i] = (q == 0) ? 0.0 : 0.3 * std::sin(0.7 + 1.3 * i + 2.1 * q);
            double mine = 0.0;
            if (corbo_hip_eval_stage_function(id, n, 1, x, p, &mine) != CORBO_HIP_OK) { ok = false; break; }
            ok = std::abs(evalEdge(e)[0] - mine) <= 1e-13 * (1.0 + std::abs(mine));
        }
        if (ok) { id_out = id; std::memcpy(prm, p, sizeof(p)); return true; }
    }
    return false;
}

// A scalar term  scale * sum_i q_i (x_i - ref_i)^2  (a cost in plain, non-least-squares form: quadratic_cost.cpp:133-138; the integrand of the
// integral cost edges, finite_differences_collocation_edges.h:98-152, 323-368), diagonal and non-negative, evaluated through the edge itself.
// The same term checked against a KNOWN model (the resident device model of the previous run): n + 1 evaluations instead of ~ 7 n.  The edge must
// reproduce  w_i (x_i - ref_i)  bit for bit at the current point and at x_i + 1 for every component (two points fix an affine row), off-diagonal
// responses must vanish, rows with a zero weight must be identically zero.  false: identify from scratch.
bool verifyDiagonalAffine(BaseEdge& e, VertexInterface* v, const double* wh, const double* rh, Eigen::VectorXd* w, Eigen::VectorXd* ref)
{
    const int n = v->getDimension();
    if (e.getDimension() != n || n > CORBO_HIP_MAX_NX) return false;
    VertexGuard guard(v);
    double* x = v->getDataRaw();
    double b0[CORBO_HIP_MAX_NX], b1[CORBO_HIP_MAX_NX];   // (no heap on this path: it runs once per cost edge and control step)
    Eigen::Map<Eigen::VectorXd> r0(b0, n), r1(b1, n);
    e.computeValues(r0);
    for (int i = 0; i < n; ++i)
        if (r0[i] != ((wh[i] == 0.0) ? 0.0 : wh[i] * (x[i] - rh[i]))) return false;
    for (int i = 0; i < n; ++i)
    {
        const double x0 = x[i];
        volatile double xp = x0 + 1.0;
        x[i] = xp;
        e.computeValues(r1);
        x[i] = x0;
        for (int j = 0; j < n; ++j)
            if (j != i && r1[j] != r0[j]) return false;
        if (r1[i] != ((wh[i] == 0.0) ? 0.0 : wh[i] * ((double)xp - rh[i]))) return false;
        if (wh[i] != 0.0 && r1[i] == r0[i]) return false;
    }
    w->resize(n);
    ref->resize(n);
    for (int i = 0; i < n; ++i) { (*w)[i] = wh[i]; (*ref)[i] = (wh[i] == 0.0) ? 0.0 : rh[i]; }   // (a row without a weight reports reference 0, like the identification)
    return true;
}

// `twins`: vertices that receive the same values (x_k and x_{k+1} of a trapezoidal edge: 0.5 dt (c + c) = dt c exactly).  The reference is
// the point where the term is EXACTLY zero (three-point estimate, then a walk over neighbouring floating-point numbers); the weights from
// x_i = ref_i + d with d a power of two ((d q_i) d is a pure exponent shift; divided by `scale`, which costs an ulp unless scale == 1).
bool identifyDiagonalQuadratic(BaseEdge& e, const std::vector<VertexInterface*>& twins, double scale, Eigen::VectorXd* q, Eigen::VectorXd* ref)
{
    if (e.getDimension() != 1 || twins.empty() || !(scale > 0.0)) return false;
    const int n = twins[0]->getDimension();
    std::vector<std::unique_ptr<VertexGuard>> guards;
    for (VertexInterface* v : twins)
    {
        if (v->getDimension() != n) return false;
        guards.emplace_back(new VertexGuard(v));
    }
    Eigen::VectorXd x = Eigen::VectorXd::Zero(n);
    auto f = [&]() {
        for (VertexInterface* v : twins) std::memcpy(v->getDataRaw(), x.data(), n * sizeof(double));
        return evalEdge(e)[0];
    };
    q->setZero(n);
    ref->setZero(n);
    const double f0 = f();
    std::vector<bool> active(n, false);
    for (int i = 0; i < n; ++i)
    {
        x[i] = 1.0;  const double fp = f();
        x[i] = -1.0; const double fm = f();
        x[i] = 0.0;
        const double qi = 0.5 * (fp + fm - 2.0 * f0);
        if (!std::isfinite(qi) || qi < 0.0) return false;
        if (qi == 0.0) { if (fp != f0 || fm != f0) return false; continue; }   // no weight on this component: any reference will do
        active[i] = true;
        (*ref)[i] = (fm - fp) / (4.0 * qi);
    }
    for (int i = 0; i < n; ++i) x[i] = active[i] ? (*ref)[i] : 0.0;
    // The estimates above come from values of size sum_j q_j ref_j^2: a component with a small weight next to others with large ones is off
    // by hundreds of ulps (q = 0.1 beside q = 1 and references of size 2: ~1e-13).  Refine around the estimate, where the OTHER terms are
    // (nearly) zero and every value is of size q h^2: the correction (f(r - h) - f(r + h)) / (4 q h) is then good to a fraction of an ulp.
    for (int round = 0; round < 2; ++round)
        for (int i = 0; i < n; ++i)
        {
            if (!active[i]) continue;
            const double r = x[i], h = std::ldexp(1.0, -10) * std::max(1.0, std::abs(r));
            const double fc = f();
            x[i] = r + h; const double fp = f();
            x[i] = r - h; const double fm = f();
            const double qi = 0.5 * (fp + fm - 2.0 * fc) / (h * h);
            x[i] = (qi > 0.0 && std::isfinite(qi)) ? r + (fm - fp) / (4.0 * qi * h) : r;
        }
    double fx = f();
    for (int sweep = 0; sweep < 4 && fx != 0.0; ++sweep)
        for (int i = 0; i < n; ++i)
        {
            if (!active[i]) continue;
            for (int dir = -1; dir <= 1; dir += 2)
                for (int it = 0; it < 256; ++it)
                {
                    const double keep = x[i];
                    x[i] = std::nextafter(keep, dir > 0 ? INFINITY : -INFINITY);
                    const double ft = f();
                    if (ft < fx) fx = ft;
                    else { x[i] = keep; break; }
                }
        }
    if (f() != 0.0) return false;   // not a sum of squares around one point
    *ref = x;
    for (int i = 0; i < n; ++i)
    {
        if (!active[i]) { (*ref)[i] = 0.0; continue; }
        const double r = x[i];
        // a power of two d with (r + d) - r == d in floating point: r + d must not need a bit more than r has (an arbitrary reference -- a sample
        // of a time-varying trajectory -- plus 2^ilogb(r) crosses into the next binade and loses its last bit half of the time): start well below
        // r's own exponent and take the first d that survives the round trip
        double d = std::ldexp(1.0, std::max(-20, std::min(20, (r == 0.0) ? 0 : std::ilogb(r))) - ((r == 0.0) ? 0 : 12));
        bool ok = false;
        for (int t = 0; t < 24 && !ok; ++t, d *= 0.5)
        {
            volatile double xp = r + d;
            ok = ((double)xp - r == d);
            if (ok) x[i] = xp;
        }
        if (!ok) { x[i] = r; return false; }
        d = x[i] - r;
        (*q)[i] = f() / (d * d) / scale;
        x[i] = r;
    }
    return true;
}

// TerminalBall, diagonal S:  c(x_f) = (x_f - ref)^T S (x_f - ref) - gamma   (final_state_constraints.cpp:60-80)
bool identifyTerminalBall(BaseEdge& e, VertexInterface* v, const Eigen::VectorXd& ref, double* prm /*S_11..S_nn, gamma*/)
{
    const int n = v->getDimension();
    if (e.getDimension() != 1 || ref.size() != n) return false;
    VertexGuard guard(v);
    double* x = v->getDataRaw();
    for (int i = 0; i < n; ++i) x[i] = ref[i];
    const double gamma = -evalEdge(e)[0];   // 0 - gamma, exact
    for (int i = 0; i < n; ++i)
    {
        double d = 1024.0;
        volatile double xp = ref[i] + d;
        if ((double)xp - ref[i] != d) return false;
        x[i] = xp;
        const double s_est = (evalEdge(e)[0] + gamma) / (d * d);
        // candidates around the estimate: the one that reproduces the edge at several distances wins
        double best = s_est;
        int best_hits = -1;
        for (int c = -2; c <= 2; ++c)
        {
            double s = s_est;
            for (int t = 0; t < std::abs(c); ++t) s = std::nextafter(s, (c > 0) ? INFINITY : -INFINITY);
            int hits = 0;
            for (double dd : {1.0, 2.0, 0.5, 1024.0})
            {
                volatile double xq = ref[i] + dd;
                if ((double)xq - ref[i] != dd) continue;
                x[i] = xq;
                if (evalEdge(e)[0] == (dd * s) * dd - gamma) ++hits;
            }
            if (hits > best_hits) { best_hits = hits; best = s; }
        }
        prm[i] = best;
        x[i]   = ref[i];
    }
    prm[n] = gamma;
    for (int p = 0; p < 3; ++p)
    {   // cross terms must vanish: verify the diagonal model away from the axes
        double acc = 0.0;
        for (int i = 0; i < n; ++i)
        {
            x[i] = ref[i] + 0.25 * std::sin(0.4 + 1.1 * i + 1.9 * p);
            const double xd = x[i] - ref[i];
            acc += (xd * prm[i]) * xd;
        }
        const double mine = acc - gamma;
        if (!(std::abs(evalEdge(e)[0] - mine) <= 1e-13 * (1.0 + std::abs(mine)))) return false;
    }
    return true;
}

struct GridView
{
    int kind = -1;   // corbo_hip_grid
    int N = 0, nx = 0, nu = 0;
    std::vector<VertexInterface*> xs, us;   // x_0 .. x_{N-2}, u_0 .. u_{N-2}
    VertexInterface* xf = nullptr;
    VertexInterface* dt = nullptr;
};

bool viewGrid(BaseHyperGraphOptimizationProblem& hg, GridView* g, std::string* reason)
{
    if (!hg.getGraph().hasVertexSet()) return fail(reason, "the hypergraph has no vertex set");
    VertexSetInterface* vs = hg.getGraph().getVertexSetRaw();
    if (dynamic_cast<FiniteDifferencesVariableGrid*>(vs)) g->kind = CORBO_HIP_GRID_FD_VARIABLE;
    else if (dynamic_cast<FiniteDifferencesGrid*>(vs)) g->kind = CORBO_HIP_GRID_FD;
    else if (dynamic_cast<MultipleShootingVariableGrid*>(vs)) g->kind = CORBO_HIP_GRID_MS_VARIABLE;
    else if (dynamic_cast<MultipleShootingGrid*>(vs)) g->kind = CORBO_HIP_GRID_MS;
    else return fail(reason, "vertex set is not a FiniteDifferencesGrid, FiniteDifferencesVariableGrid, MultipleShootingGrid or MultipleShootingVariableGrid");
    std::vector<VertexInterface*> vtx;
    vs->getVertices(vtx);
    // x_0..x_{N-2}, u_0..u_{N-2}, x_f, dt, (u_prev, u_ref, u_prev_dt)  (full_discretization_grid_base.cpp:499-512); ShootingGridBase
    // interleaves states and controls (shooting_grid_base.cpp:567-581)
    if ((int)vtx.size() < 7 || ((int)vtx.size() - 5) % 2 != 0) return fail(reason, "unexpected vertex list of the grid");
    g->N = ((int)vtx.size() - 5) / 2 + 1;
    const bool interleaved = (g->kind == CORBO_HIP_GRID_MS || g->kind == CORBO_HIP_GRID_MS_VARIABLE);
    for (int k = 0; k < g->N - 1; ++k)
    {
        g->xs.push_back(interleaved ? vtx[2 * k] : vtx[k]);
        g->us.push_back(interleaved ? vtx[2 * k + 1] : vtx[g->N - 1 + k]);
    }
    g->xf = vtx[2 * (g->N - 1)];
    g->dt = vtx[2 * (g->N - 1) + 1];
    g->nx = g->xf->getDimension();
    g->nu = g->us[0]->getDimension();
    if (g->nx > CORBO_HIP_MAX_NX || g->nu > CORBO_HIP_MAX_NU) return fail(reason, "state / control dimension beyond the device limits");
    for (int k = 0; k < g->N - 1; ++k)
        if (g->xs[k]->getDimension() != g->nx || g->us[k]->getDimension() != g->nu) return fail(reason, "vertex dimensions vary along the horizon");
    if (g->dt->getDimension() != 1) return fail(reason, "dt vertex is not scalar");
    return true;
}

int indexOf(const std::vector<VertexInterface*>& list, const VertexInterface* v)
{
    for (size_t i = 0; i < list.size(); ++i)
        if (list[i] == v) return (int)i;
    return -1;
}

}  // namespace

bool readStateReferenceForHip(BaseHyperGraphOptimizationProblem& hg, int nx, Eigen::VectorXd* xref)
{
    GridView g;
    std::string why;
    if (!viewGrid(hg, &g, &why) || g.nx != nx) return false;
    for (const BaseEdge::Ptr& e : hg.getGraph().getEdgeSetRaw()->getLsqObjectiveEdges())
    {
        if (e->getNumVertices() != 1 || e->getDimension() != nx) continue;
        VertexInterface* v = e->getVertexRaw(0);
        if (v != g.xf && indexOf(g.xs, v) < 0) continue;
        Eigen::VectorXd w, ref;
        if (!identifyDiagonalAffine(*e, v, &w, &ref)) return false;
        if ((w.array() == 0.0).any()) continue;   // a zero weight hides the reference of that component: try another edge
        *xref = ref;
        return true;
    }
    return false;
}

bool readPreviousControlForHip(BaseHyperGraphOptimizationProblem& hg, int nu, Eigen::VectorXd* u_prev, double* dt_prev)
{
    for (const BaseEdge::Ptr& e : hg.getGraph().getEdgeSetRaw()->getInequalityEdges())
    {
        if (!dynamic_cast<ControlDeviationEdge*>(e.get()) || e->getNumVertices() != 3) continue;
        if (e->getVertexRaw(1)->getDimension() != nu || !e->getVertexRaw(1)->isFixed()) return false;   // the first one is interval 0's: (u_0, _u_prev, _u_prev_dt)
        *u_prev  = Eigen::Map<const Eigen::VectorXd>(e->getVertexRaw(1)->getData(), nu);
        *dt_prev = e->getVertexRaw(2)->getData()[0];
        return true;
    }
    return false;
}

bool readStateReferenceTrajectoryForHip(BaseHyperGraphOptimizationProblem& hg, int nx, Eigen::MatrixXd* traj)
{
    GridView g;
    std::string why;
    if (!viewGrid(hg, &g, &why) || g.nx != nx || traj->rows() != g.N || traj->cols() != nx) return false;
    int seen = 0;
    for (const BaseEdge::Ptr& e : hg.getGraph().getEdgeSetRaw()->getLsqObjectiveEdges())
    {
        if (e->getNumVertices() != 1 || e->getDimension() != nx) continue;
        VertexInterface* v = e->getVertexRaw(0);
        const int k = (v == g.xf) ? g.N - 1 : indexOf(g.xs, v);
        if (k < 0) continue;
        Eigen::VectorXd w, ref;
        if (!identifyDiagonalAffine(*e, v, &w, &ref)) return false;
        traj->row(k) = ref.transpose();
        ++seen;
    }
    return seen >= g.N - 1;   // (without a final cost term row N-1 keeps what the caller put there)
}

bool recogniseHyperGraphForHip(BaseHyperGraphOptimizationProblem& hg, HipRecognisedModel* model, std::string* reason, const HipRecognisedModel* hint)
{
    GridView g;
    if (!viewGrid(hg, &g, reason)) return false;
    corbo_hip_problem_desc& d = model->desc;
    std::memset(&d, 0, sizeof(d));
    d.grid = g.kind; d.nx = g.nx; d.nu = g.nu; d.N = g.N;
    const OptimizationEdgeSet* es = hg.getGraph().getEdgeSetRaw();
    // plain objective edges (costs with lsq_form = false, integral-form costs): the IPOPT-style configuration.  Not a least-squares problem --
    // solve() will refuse it like the reference's solver -- but the Hessian-path operators work on it; identified below
    const bool plain_costs = !es->getObjectiveEdges().empty();
    if (plain_costs && !es->getLsqObjectiveEdges().empty()) return fail(reason, "cost terms in least-squares form and in plain form in one graph");
    // mixed edges: a MultipleShootingGrid whose stage cost has integral terms files ONE MultipleShootingEdgeSingleControl per interval INSTEAD of
    // the dynamics-only edge (multiple_shooting_grid.cpp:70-77): objective part = the cost integrated along the shooting step, equality part = the defect
    const std::vector<BaseMixedEdge::Ptr>& mixed = es->getMixedEdges();
    const bool ms_mixed = !mixed.empty();
    auto integratorId = [](NumericalIntegratorExplicitInterface* in) {
        if (dynamic_cast<IntegratorExplicitRungeKutta4*>(in)) return 0;
        if (dynamic_cast<IntegratorExplicitEuler*>(in)) return 1;
        if (dynamic_cast<IntegratorExplicitRungeKutta2*>(in)) return 2;
        if (dynamic_cast<IntegratorExplicitRungeKutta3*>(in)) return 3;
        if (dynamic_cast<IntegratorExplicitRungeKutta5*>(in)) return 5;
        if (dynamic_cast<IntegratorExplicitRungeKutta6*>(in)) return 6;
        if (dynamic_cast<IntegratorExplicitRungeKutta7*>(in)) return 7;
        return -1;
    };

    // ---- equality edges: one defect edge per interval, in order; then optionally the terminal equality constraint
    const std::vector<BaseEdge::Ptr>& eqs = es->getEqualityEdges();
    int n_defect_eq = ms_mixed ? 0 : g.N - 1;   // (mixed edges carry the defects themselves); + the LeftSumEqualityEdges found below
    if ((int)eqs.size() < n_defect_eq) return fail(reason, "fewer equality edges than grid intervals");
    int integral_rule = 0;   // 1 / 2: trapezoidal / left-sum integral constraint edges were found (the grid's one integration rule)
    auto set_rule = [&](int r) { if (integral_rule && integral_rule != r) return false; integral_rule = r; return true; };
    SystemDynamicsInterface* dyn = nullptr;
    const StageCost* mixed_cost = nullptr;
    if (ms_mixed)
    {
        if (g.kind != CORBO_HIP_GRID_MS) return fail(reason, "mixed edges on a grid other than the MultipleShootingGrid with a fixed dt");
        if ((int)mixed.size() != g.N - 1) return fail(reason, "mixed edges: not one per interval");
        if (!es->getLsqObjectiveEdges().empty()) return fail(reason, "mixed edges next to least-squares cost terms");
        for (int k = 0; k < g.N - 1; ++k)
        {
            auto* me = dynamic_cast<MultipleShootingEdgeSingleControl*>(mixed[k].get());
            VertexInterface* x2 = (k + 1 < g.N - 1) ? g.xs[k + 1] : g.xf;
            if (!me) return fail(reason, "mixed edge " + std::to_string(k) + " is not a MultipleShootingEdgeSingleControl");
            if (me->getNumVertices() != 4 || me->getVertexRaw(0) != g.xs[k] || me->getVertexRaw(1) != g.us[k] || me->getVertexRaw(2) != g.dt || me->getVertexRaw(3) != x2)
                return fail(reason, "mixed edge " + std::to_string(k) + " is not on (x_k, u_k, dt, x_{k+1})");
            if (me->getObjectiveDimension() != 1 || me->getEqualityDimension() != g.nx || me->getInequalityDimension() != 0)
                return fail(reason, "mixed edge with integral equality / inequality terms");
            SystemDynamicsInterface* dk = member<MixedEdgeDynamics>(*me).get();
            const int integ = integratorId(member<MixedEdgeIntegrator>(*me).get());
            const StageCost* ck = member<MixedEdgeStageCost>(*me).get();
            if (integ < 0) return fail(reason, "shooting integrator other than IntegratorExplicitEuler / RungeKutta2 ... RungeKutta7");
            if (k == 0) { dyn = dk; mixed_cost = ck; d.shooting_integrator = integ; d.defect = CORBO_HIP_DEFECT_RK4_SHOOTING; }
            else if (dk != dyn || ck != mixed_cost || integ != d.shooting_integrator) return fail(reason, "dynamics / stage cost / integrator of the mixed edges vary along the horizon");
        }
        if (!mixed_cost) return fail(reason, "mixed edges without a stage cost");
    }
    const int n_intervals_eq = n_defect_eq;
    for (int k = 0, qi = 0; k < n_intervals_eq; ++k, ++qi)
    {
        if (qi >= (int)eqs.size()) return fail(reason, "fewer equality edges than grid intervals");
        BaseEdge* e = eqs[qi].get();
        VertexInterface* x2 = (k + 1 < g.N - 1) ? g.xs[k + 1] : g.xf;
        if (dynamic_cast<LeftSumEqualityEdge*>(e))
        {   // integral stage equality, left sum: its own edge on (x_k, u_k, dt) in front of the dynamics edge (finite_differences_grid.cpp:89-98)
            if (e->getNumVertices() != 3 || e->getVertexRaw(0) != g.xs[k] || e->getVertexRaw(1) != g.us[k] || e->getVertexRaw(2) != g.dt || e->getDimension() != 1)
                return fail(reason, "LeftSumEqualityEdge " + std::to_string(k) + ": not a one-row integrand on (x_k, u_k, dt) of a family with nx <= 4");
            IntegrandProbe f(*e, {g.xs[k]}, g.us[k], g.dt, 0);
            if (!set_rule(2) || !identifyLinearIntegrand(f, g.nx, g.nu, d.stage_eq_params, k > 0))
                return fail(reason, "integral stage equality " + std::to_string(k) + " is not a^T x + b^T u - c (the device's plug-in), or varies along the horizon");
            d.stage_eq = CORBO_HIP_STAGE_EQ_LINEAR;
            ++qi; ++n_defect_eq;
            if (qi >= (int)eqs.size()) return fail(reason, "LeftSumEqualityEdge without a dynamics edge behind it");
            e = eqs[qi].get();
        }
        if (e->getNumVertices() != 4 || e->getVertexRaw(0) != g.xs[k] || e->getVertexRaw(1) != g.us[k] || e->getVertexRaw(2) != x2 || e->getVertexRaw(3) != g.dt)
            return fail(reason, "equality edge " + std::to_string(k) + " is not a dynamics defect on (x_k, u_k, x_{k+1}, dt)");
        SystemDynamicsInterface* dk = nullptr;
        int defect = -1;
        if (auto* fd = dynamic_cast<FDCollocationEdge*>(e))
        {
            dk = member<FdEdgeDynamics>(*fd).get();
            FiniteDifferencesCollocationInterface* sch = member<FdEdgeScheme>(*fd).get();
            if (dynamic_cast<ForwardDiffCollocation*>(sch)) defect = CORBO_HIP_DEFECT_FORWARD;
            else if (dynamic_cast<BackwardDiffCollocation*>(sch)) defect = CORBO_HIP_DEFECT_BACKWARD;
            else if (dynamic_cast<MidpointDiffCollocation*>(sch)) defect = CORBO_HIP_DEFECT_MIDPOINT;
            else if (dynamic_cast<CrankNicolsonDiffCollocation*>(sch)) defect = CORBO_HIP_DEFECT_CRANK_NICOLSON;
            else return fail(reason, "unknown finite-differences collocation scheme");
        }
        else if (auto* tq = dynamic_cast<TrapezoidalIntegralEqualityDynamicsEdge*>(e))
        {   // dynamics + the trapezoidal integral of the stage equalities in one edge (finite_differences_grid.cpp:82-88)
            dk = member<TrapEqEdgeDynamics>(*tq).get();
            FiniteDifferencesCollocationInterface* sch = member<TrapEqEdgeScheme>(*tq).get();
            if (dynamic_cast<ForwardDiffCollocation*>(sch)) defect = CORBO_HIP_DEFECT_FORWARD;
            else if (dynamic_cast<BackwardDiffCollocation*>(sch)) defect = CORBO_HIP_DEFECT_BACKWARD;
            else if (dynamic_cast<MidpointDiffCollocation*>(sch)) defect = CORBO_HIP_DEFECT_MIDPOINT;
            else if (dynamic_cast<CrankNicolsonDiffCollocation*>(sch)) defect = CORBO_HIP_DEFECT_CRANK_NICOLSON;
            else return fail(reason, "unknown finite-differences collocation scheme");
            if (e->getDimension() != g.nx + 1) return fail(reason, "TrapezoidalIntegralEqualityDynamicsEdge: one integrand row");
            IntegrandProbe f(*e, {g.xs[k], x2}, g.us[k], g.dt, g.nx);
            if (!set_rule(1) || !identifyLinearIntegrand(f, g.nx, g.nu, d.stage_eq_params, k > 0))
                return fail(reason, "integral stage equality " + std::to_string(k) + " is not a^T x + b^T u - c (the device's plug-in), or varies along the horizon");
            d.stage_eq = CORBO_HIP_STAGE_EQ_LINEAR;
        }
        else if (auto* ms = dynamic_cast<MSVariableDynamicsOnlyEdge*>(e))
        {
            dk = member<MsEdgeDynamics>(*ms).get();
            const int integ = integratorId(member<MsEdgeIntegrator>(*ms).get());
            if (integ < 0) return fail(reason, "shooting integrator other than IntegratorExplicitEuler / RungeKutta2 ... RungeKutta7");
            if (k > 0 && integ != d.shooting_integrator) return fail(reason, "shooting integrator varies along the horizon");
            d.shooting_integrator = integ;
            defect = CORBO_HIP_DEFECT_RK4_SHOOTING;
        }
        else return fail(reason, "equality edge " + std::to_string(k) + " is neither an FDCollocationEdge nor an MSVariableDynamicsOnlyEdge");
        if (k == 0) { dyn = dk; d.defect = defect; }
        else if (dk != dyn || defect != d.defect) return fail(reason, "dynamics object / defect formula varies along the horizon");
    }
    if (!dyn) return fail(reason, "no dynamics object");
    if (dyn->getStateDimension() != g.nx || dyn->getInputDimension() != g.nu) return fail(reason, "dynamics dimensions do not match the vertices");

    // ---- least-squares objective edges, in the grid's creation order (nlp_functions.cpp:70-132, finite_differences_grid.cpp:38-154)
    Eigen::VectorXd sq, sr, sqf, xref_state, xref_final;
    Eigen::MatrixXd Uq, Ur, Uqf;   // upper Cholesky factors of non-diagonal weights (empty: diagonal)
    std::vector<Eigen::VectorXd> stage_refs(g.N - 1);   // reference of the state cost term of every interval (getReferenceCached(k))
    bool refs_vary = false;
    int n_state = 0, n_ctrl = 0, n_final = 0, n_dt = 0;
    int k_first_state = g.N, k_first_ctrl = g.N;   // first interval that carries a state / control term (MinTimeQuadratic::only_last_n)
    double dt_weight = 0.0;
    Eigen::VectorXd w, ref;   // (outside the loop: no allocation per edge once they have their size)
    for (const BaseEdge::Ptr& ep : es->getLsqObjectiveEdges())
    {
        BaseEdge* e = ep.get();
        if (e->getNumVertices() != 1) return fail(reason, "least-squares edge on more than one vertex (control deviation / integral term)");
        VertexInterface* v = e->getVertexRaw(0);
        if (v == g.dt)
        {   // MinimumTime(lsq): weight * dt, created twice (nlp_functions.cpp:91-107)
            if (e->getDimension() != 1) return fail(reason, "dt cost term of dimension > 1");
            if (!identifyDiagonalAffine(*e, v, &w, &ref) || ref[0] != 0.0) return fail(reason, "dt cost term is not weight * dt");
            if (n_dt > 0 && w[0] != dt_weight) return fail(reason, "dt cost terms with different weights");
            dt_weight = w[0];
            ++n_dt;
            continue;
        }
        Eigen::MatrixXd Ud;   // non-empty: the term has a non-diagonal weight, r = U (v - ref)
        // a resident model of the same graph (the adapter's model tracking, one call per new run): check the term against it first -- n + 1
        // evaluations of the edge instead of the full identification; any surprise falls through to the identification from scratch
        bool hinted = false;
        if (hint && !hint->desc.weights_dense && !hint->desc.cost_nonlsq && hint->desc.nx == g.nx && hint->desc.nu == g.nu)
        {
            const bool traj = hint->xref_traj.rows() == g.N && hint->xref_traj.cols() == g.nx;
            double wh[CORBO_HIP_MAX_NX], rh[CORBO_HIP_MAX_NX];
            int nh = 0;
            const int ks = indexOf(g.xs, v);
            if (v == g.xf && hint->desc.final_cost && hint->xref.size() == g.nx)
            {
                nh = g.nx;
                for (int i = 0; i < g.nx; ++i) { wh[i] = std::sqrt(hint->desc.qf_diag[i]); rh[i] = traj ? hint->xref_traj(g.N - 1, i) : hint->xref[i]; }
            }
            else if (ks >= 0 && (CORBO_HIP_COST_TERMS(hint->desc.stage_cost) & 1) && hint->xref.size() == g.nx)
            {
                nh = g.nx;
                for (int i = 0; i < g.nx; ++i) { wh[i] = std::sqrt(hint->desc.q_diag[i]); rh[i] = traj ? hint->xref_traj(ks, i) : hint->xref[i]; }
            }
            else if (indexOf(g.us, v) >= 0 && (CORBO_HIP_COST_TERMS(hint->desc.stage_cost) & 2))
            {
                nh = g.nu;
                for (int i = 0; i < g.nu; ++i) { wh[i] = std::sqrt(hint->desc.r_diag[i]); rh[i] = 0.0; }
            }
            hinted = nh > 0 && nh == v->getDimension() && verifyDiagonalAffine(*e, v, wh, rh, &w, &ref);
        }
        if (!hinted && !identifyDiagonalAffine(*e, v, &w, &ref))
        {
            if (g.nx > 4 || g.nu > 4 || !identifyUpperAffine(*e, v, &Ud, &ref))
                return fail(reason, "a least-squares term is neither sqrt(W_diag) (v - ref) nor U (v - ref) with an upper Cholesky factor U (nx <= 4; with a ZERO "
                                    "reference the reference's non-diagonal branch assigns a scalar to the vector, quadratic_cost.cpp:108-111 -- nothing to reproduce)");
            w = Ud.diagonal();
        }
        if (v == g.xf)
        {
            if (n_final++ > 0) return fail(reason, "more than one least-squares term on x_f");
            sqf = w; xref_final = ref; Uqf = Ud;
        }
        else if (indexOf(g.xs, v) >= 0)
        {
            stage_refs[indexOf(g.xs, v)] = ref;
            k_first_state = std::min(k_first_state, indexOf(g.xs, v));
            if (n_state++ == 0) { sq = w; xref_state = ref; Uq = Ud; }
            else if (!sameVector(w, sq) || !sameMatrix(Ud, Uq)) return fail(reason, "state cost weights vary along the horizon");
            else if (!sameVector(ref, xref_state)) refs_vary = true;   // a time-varying reference trajectory
        }
        else if (indexOf(g.us, v) >= 0)
        {
            if ((ref.array() != 0.0).any()) return fail(reason, "non-zero control reference");
            k_first_ctrl = std::min(k_first_ctrl, indexOf(g.us, v));
            if (n_ctrl++ == 0) { sr = w; Ur = Ud; }
            else if (!sameVector(w, sr) || !sameMatrix(Ud, Ur)) return fail(reason, "control cost weights vary along the horizon");
        }
        else return fail(reason, "least-squares term on an unexpected vertex");
    }
    // the stage cost is identified by the TERMS it created (nlp_functions.cpp:70-107), not by its class: QuadraticFormCost = state + control,
    // MinimumTime = dt (twice), MinTimeQuadratic = all three (hybrid_cost.h:189-303) -- a user's own StageCost with the same terms maps as
    // well.  (QuadraticStateCost / QuadraticControlCost / MinTimeQuadraticStates / ...Controls with a diagonal weight create no
    // least-squares term at all -- quadratic_state_cost.cpp:33-62 leaves _Q empty -- and arrive here as what is left of them.)
    if (n_dt > 0)
    {
        if (n_dt != 2) return fail(reason, "minimum-time term that was not created twice at k = 0 (a grid with one dt per interval?)");
        if (dt_weight != std::sqrt((double)(g.N - 1))) return fail(reason, "minimum-time weight is not sqrt(N - 1)");
    }
    if ((n_state != 0) != (n_ctrl != 0)) return fail(reason, "quadratic stage cost with a state term but no control term, or the other way round (non-diagonal weights?)");
    if (n_state)
    {   // on every interval -- or, next to a minimum-time term, on the last intervals only (MinTimeQuadratic::only_last_n, hybrid_cost.h:224-237)
        const int k0 = k_first_state;
        if (k_first_ctrl != k0 || n_state != g.N - 1 - k0 || n_ctrl != g.N - 1 - k0) return fail(reason, "quadratic cost terms on a set of intervals that is not a tail of the horizon");
        if (k0 != 0 && !n_dt) return fail(reason, "quadratic cost terms on the last intervals only, without a minimum-time term");
        d.quad_first_interval = k0;
    }
    d.stage_cost = n_dt ? (n_state ? CORBO_HIP_COST_MIN_TIME_QUADRATIC_LSQ : CORBO_HIP_COST_MIN_TIME_LSQ) : (n_state ? CORBO_HIP_COST_QUADRATIC_LSQ : CORBO_HIP_COST_NONE);
    if (n_state)
    {
        for (int i = 0; i < g.nx; ++i) d.q_diag[i] = sq[i] * sq[i];
        for (int i = 0; i < g.nu; ++i) d.r_diag[i] = sr[i] * sr[i];
    }
    d.final_cost = n_final ? 1 : 0;
    if (n_final)
        for (int i = 0; i < g.nx; ++i) d.qf_diag[i] = sqf[i] * sqf[i];
    // one static reference for every term -- or one per grid point (time-varying ReferenceTrajectoryInterface): then the final-stage terms
    // have their own (the reference at the last grid point) and model->xref is that one
    model->xref_traj.resize(0, 0);
    model->xref = Eigen::VectorXd::Zero(g.nx);
    if (n_state) model->xref = xref_state;
    if (refs_vary)
    {
        if ((sq.array() == 0.0).any()) return fail(reason, "time-varying state reference with a zero state weight (the reference of that component cannot be identified)");
        if (n_final) model->xref = xref_final;
        else model->xref = stage_refs[g.N - 2];
    }
    else if (n_final)
    {
        if (n_state)
        {
            for (int i = 0; i < g.nx; ++i)
                if (sq[i] != 0.0 && sqf[i] != 0.0 && xref_state[i] != xref_final[i]) return fail(reason, "stage and final cost use different state references");
            for (int i = 0; i < g.nx; ++i)
                if (sq[i] == 0.0) model->xref[i] = xref_final[i];
        }
        else model->xref = xref_final;
    }
    // The device takes the weights as Q / R / Qf diagonals and forms sqrt() itself (structure.cpp): the round trip sqrt(w * w) == w
    // must hold for the identified sqrt-weights, otherwise the residual rows would differ in the last bit
    for (int i = 0; i < g.nx; ++i)
        if ((n_state && !Uq.size() && std::sqrt(d.q_diag[i]) != sq[i]) || (n_final && !Uqf.size() && std::sqrt(d.qf_diag[i]) != sqf[i]))
            return fail(reason, "state weight does not survive the square / square-root round trip");
    for (int i = 0; i < g.nu; ++i)
        if (n_ctrl && !Ur.size() && std::sqrt(d.r_diag[i]) != sr[i]) return fail(reason, "control weight does not survive the square / square-root round trip");
    // non-diagonal weights travel as the factors themselves (corbo_hip_problem_desc::weights_dense)
    if (Uq.size()) { d.weights_dense |= 1; for (int i = 0; i < g.nx; ++i) for (int j = 0; j < g.nx; ++j) d.q_sqrt[i * g.nx + j] = Uq(i, j); }
    if (Ur.size()) { d.weights_dense |= 2; for (int i = 0; i < g.nu; ++i) for (int j = 0; j < g.nu; ++j) d.r_sqrt[i * g.nu + j] = Ur(i, j); }
    if (Uqf.size()) { d.weights_dense |= 4; for (int i = 0; i < g.nx; ++i) for (int j = 0; j < g.nx; ++j) d.qf_sqrt[i * g.nx + j] = Uqf(i, j); }
    if (d.weights_dense && plain_costs) return fail(reason, "non-diagonal weights next to plain objective edges");

    if (plain_costs || ms_mixed)
    {   // per interval: a state term and a control term (QuadraticFormCost(.., lsq_form = false)), or ONE integral cost edge
        // (integral_form = true: TrapezoidalIntegralCostEdge on (x_k, u_k, x_{k+1}, dt) / LeftSumCostEdge on (x_k, u_k, dt)); then the final cost
        Eigen::VectorXd q, r, qf, ref, rf, uz;
        int ns = 0, nc = 0, nf = 0, ni = 0, ndt = 0, integral = 0;
        int k_first_integral = ms_mixed ? 0 : g.N;   // first interval with an integral cost edge (MinTimeQuadratic::only_last_n in integral form)
        const double dtv = g.dt->getData()[0];
        if (ms_mixed)
        {   // the integrand c(x, u) of the mixed edges' objective part, probed through the stage cost's own computeIntegralStateControlTerm
            // (what MultipleShootingEdgeSingleControl::configureIntegrand calls, multiple_shooting_edges.h:251-263) on x_k / u_k of an interval:
            // a diagonal quadratic form around one state reference and a zero control reference, the same on the first and the last interval
            for (int k = 0; k < g.N - 1; ++k)
            {
                const StageCost* sc = mixed_cost;
                EdgeGenericScalarFun<VectorVertex, VectorVertex> probe(
                    [sc, k](const EdgeGenericScalarFun<VectorVertex, VectorVertex>::VertexContainer& vs) {
                        Eigen::VectorXd c(1);
                        c[0] = 0.0;
                        sc->computeIntegralStateControlTerm(k, static_cast<const VectorVertex*>(vs[0])->values(), static_cast<const VectorVertex*>(vs[1])->values(), c);
                        return c[0];
                    },
                    false, *static_cast<VectorVertex*>(g.xs[k]), *static_cast<VectorVertex*>(g.us[k]));
                Eigen::VectorXd w, rr, wu, ru;
                {
                    VertexGuard gu(g.us[k]);
                    std::memset(g.us[k]->getDataRaw(), 0, g.nu * sizeof(double));
                    if (!identifyDiagonalQuadratic(probe, {g.xs[k]}, 1.0, &w, &rr)) return fail(reason, "mixed edges: the integrand is not a diagonal quadratic form in the state");
                }
                {
                    VertexGuard gx(g.xs[k]);
                    std::memcpy(g.xs[k]->getDataRaw(), rr.data(), g.nx * sizeof(double));
                    if (!identifyDiagonalQuadratic(probe, {g.us[k]}, 1.0, &wu, &ru) || (ru.array() != 0.0).any())
                        return fail(reason, "mixed edges: the integrand is not a diagonal quadratic form in the control (zero reference)");
                }
                stage_refs[k] = rr;
                if (k == 0) { q = w; ref = rr; r = wu; }
                else if (!sameVector(w, q) || !sameVector(wu, r)) return fail(reason, "mixed edges: the integrand's weights vary along the horizon");
                else if (!sameVector(rr, ref)) refs_vary = true;   // a time-varying reference trajectory: interval k integrates against reference k
            }
            integral = 1;
            ni = g.N - 1;
        }
        for (const BaseEdge::Ptr& ep : es->getObjectiveEdges())
        {
            BaseEdge* e = ep.get();
            Eigen::VectorXd w, rr;
            const bool trap = dynamic_cast<TrapezoidalIntegralCostEdge*>(e) != nullptr, left = dynamic_cast<LeftSumCostEdge*>(e) != nullptr;
            if (trap || left)
            {
                if (g.kind != CORBO_HIP_GRID_FD && g.kind != CORBO_HIP_GRID_FD_VARIABLE) return fail(reason, "integral cost edges on a grid other than the FiniteDifferencesGrid / FiniteDifferencesVariableGrid");
                const int k = indexOf(g.xs, e->getVertexRaw(0));
                if (k >= 0 && k < k_first_integral) k_first_integral = k;
                VertexInterface* x2 = (k + 1 < g.N - 1) ? g.xs[k + 1] : g.xf;
                if (k < 0 || e->getVertexRaw(1) != g.us[k] || (trap && (e->getVertexRaw(2) != x2 || e->getVertexRaw(3) != g.dt)) || (left && e->getVertexRaw(2) != g.dt))
                    return fail(reason, "integral cost edge on unexpected vertices");
                if (integral && integral != (trap ? 1 : 2)) return fail(reason, "trapezoidal and left-sum cost edges in one graph");
                integral = trap ? 1 : 2;
                VertexGuard gu(g.us[k]);
                std::memset(g.us[k]->getDataRaw(), 0, g.nu * sizeof(double));   // the control part of the integrand is (u r) u: exactly zero at u = 0
                std::vector<VertexInterface*> tw = {g.xs[k]};
                if (trap) tw.push_back(x2);
                if (!identifyDiagonalQuadratic(*e, tw, dtv, &w, &rr)) return fail(reason, "integral cost edge whose integrand is not a diagonal quadratic form in the state");
                Eigen::VectorXd wu, ru;
                {   // the control part, with the states at their reference (state part exactly zero)
                    std::vector<std::unique_ptr<VertexGuard>> gs;
                    for (VertexInterface* v : tw) { gs.emplace_back(new VertexGuard(v)); std::memcpy(v->getDataRaw(), rr.data(), g.nx * sizeof(double)); }
                    if (!identifyDiagonalQuadratic(*e, {g.us[k]}, trap ? dtv : dtv, &wu, &ru) || (ru.array() != 0.0).any())
                        return fail(reason, "integral cost edge whose integrand is not a diagonal quadratic form in the control (zero reference)");
                }
                stage_refs[k] = rr;
                if (ni++ == 0) { q = w; ref = rr; r = wu; }
                else if ((w - q).cwiseAbs().maxCoeff() > 1e-12 * (1.0 + q.cwiseAbs().maxCoeff())) return fail(reason, "integral cost weights vary along the horizon");
                else if (!sameVector(rr, ref)) refs_vary = true;   // time-varying reference: both ends of interval k use reference k
                continue;
            }
            if (e->getNumVertices() != 1) return fail(reason, "plain objective edge on more than one vertex that is not an integral cost edge");
            VertexInterface* v = e->getVertexRaw(0);
            if (v == g.dt)
            {   // MinimumTime(lsq_form = false): (N - 1) dt (minimum_time.h:60), created twice (nlp_functions.cpp:91-107)
                VertexGuard gd(g.dt);
                double* t = g.dt->getDataRaw();
                t[0] = 0.0; const double a0 = evalEdge(*e)[0];
                t[0] = 1.0; const double a1 = evalEdge(*e)[0];
                t[0] = 2.0; const double a2 = evalEdge(*e)[0];
                if (e->getDimension() != 1 || a0 != 0.0 || a1 != (double)(g.N - 1) || a2 != 2.0 * a1) return fail(reason, "plain dt term that is not (N - 1) dt");
                ++ndt;
                continue;
            }
            if (!identifyDiagonalQuadratic(*e, {v}, 1.0, &w, &rr)) return fail(reason, "plain objective edge that is not a diagonal quadratic form around one reference");
            if (v == g.xf) { if (nf++ > 0) return fail(reason, "more than one plain term on x_f"); qf = w; rf = rr; }
            else if (indexOf(g.xs, v) >= 0)
            {
                stage_refs[indexOf(g.xs, v)] = rr;
                if (ns++ == 0) { q = w; ref = rr; }
                else if (!sameVector(w, q)) return fail(reason, "plain state cost weights vary along the horizon");
                else if (!sameVector(rr, ref)) refs_vary = true;   // a time-varying reference trajectory
            }
            else if (indexOf(g.us, v) >= 0)
            {
                if ((rr.array() != 0.0).any()) return fail(reason, "non-zero control reference");
                if (nc++ == 0) r = w;
                else if (!sameVector(w, r)) return fail(reason, "plain control cost varies along the horizon");
            }
            else return fail(reason, "plain objective edge on an unexpected vertex");
        }
        const bool quad = integral || ns || nc;
        if (quad && (integral ? (ni != g.N - 1 - k_first_integral || ns || nc) : (ns != g.N - 1 || nc != g.N - 1)))
            return fail(reason, "plain cost terms that are not one state and one control term (or one integral edge) per interval -- or, integral edges, per interval of a tail of the horizon");
        if (ndt != 0 && ndt != 2) return fail(reason, "plain minimum-time term that was not created twice at k = 0");
        // integral edges next to the dt terms: MinTimeQuadratic(integral_form = true), also with only_last_n (hybrid_cost.h:209); without dt terms
        // they are QuadraticFormCost's, on every interval of a fixed-dt grid
        if (integral && !ms_mixed && ((ndt == 2) != (g.kind == CORBO_HIP_GRID_FD_VARIABLE) || (k_first_integral != 0 && !ndt)))
            return fail(reason, "integral cost edges: QuadraticFormCost on a fixed-dt grid or MinTimeQuadratic on the FiniteDifferencesVariableGrid");
        if (integral && !ms_mixed) d.quad_first_interval = k_first_integral;
        d.cost_nonlsq = 1;
        d.cost_integral = integral;
        d.stage_cost = ndt ? (quad ? CORBO_HIP_COST_MIN_TIME_QUADRATIC_LSQ : CORBO_HIP_COST_MIN_TIME_LSQ) : (quad ? CORBO_HIP_COST_QUADRATIC_LSQ : CORBO_HIP_COST_NONE);
        if (!quad) { q = Eigen::VectorXd::Zero(g.nx); r = Eigen::VectorXd::Zero(g.nu); ref = nf ? rf : Eigen::VectorXd::Zero(g.nx); }
        for (int i = 0; i < g.nx; ++i) d.q_diag[i] = q[i];
        for (int i = 0; i < g.nu; ++i) d.r_diag[i] = r[i];
        d.final_cost = nf ? 1 : 0;
        if (nf) for (int i = 0; i < g.nx; ++i) d.qf_diag[i] = qf[i];
        model->xref = ref;
        if (refs_vary)
        {   // one reference per grid point: model->xref is the final-stage terms' (the last sample)
            if ((q.array() == 0.0).any()) return fail(reason, "time-varying state reference with a zero state weight (the reference of that component cannot be identified)");
            model->xref = nf ? rf : stage_refs[g.N - 2];
        }
        else if (nf)
            for (int i = 0; i < g.nx; ++i)
            {
                if (q[i] != 0.0 && qf[i] != 0.0 && ref[i] != rf[i]) return fail(reason, "stage and final cost use different state references");
                if (q[i] == 0.0) model->xref[i] = rf[i];
            }
    }

    // ---- terminal equality constraint: x_f - xref (final_state_constraints.h:130-160)
    if ((int)eqs.size() == n_defect_eq + 1)
    {
        BaseEdge* e = eqs[n_defect_eq].get();
        Eigen::VectorXd w, ref;
        if (e->getNumVertices() != 1 || e->getVertexRaw(0) != g.xf) return fail(reason, "extra equality edge is not on x_f");
        if (e->getDimension() < g.nx)
        {   // TerminalPartialEqualityConstraint: rows for a subset of the components
            unsigned mask = 0;
            if ((g.nx > 4 && !(g.nx <= 12 && g.nu <= 4 && g.nx + g.nu <= 16)) || !identifyPartialIdentity(*e, g.xf, &mask, &ref)) return fail(reason, "extra equality edge is not a TerminalPartialEqualityConstraint (x_f - xref)_active");
            for (int i = 0; i < g.nx; ++i)
                if ((mask >> i) & 1u)
                {
                    if ((n_final || (n_state && !refs_vary)) && ref[i] != model->xref[i]) return fail(reason, "partial terminal equality constraint uses a different reference");
                    model->xref[i] = ref[i];
                }
            d.final_eq = 1;
            d.final_eq_mask = mask;
        }
        else
        {
        if (!identifyDiagonalAffine(*e, g.xf, &w, &ref) || (w.array() != 1.0).any())
            return fail(reason, "extra equality edge is not a TerminalEqualityConstraint x_f - xref");
        if ((n_final || (n_state && !refs_vary)) && !sameVector(ref, model->xref)) return fail(reason, "terminal equality constraint uses a different reference");
        model->xref = ref;
        d.final_eq = 1;
        }
    }
    else if ((int)eqs.size() > n_defect_eq + 1) return fail(reason, "unexpected additional equality edges");

    // ---- inequality edges, in the grid's creation order (finite_differences_grid.cpp:49-153): per interval the stage inequality on x_k (keep-out
    //      ball), the control-deviation edge (input-rate limit), the integral inequality edge (the ball as integrand); then the TerminalBall on x_f;
    //      then the control-deviation edge of the last control against u_ref
    const std::vector<BaseEdge::Ptr>& ins = es->getInequalityEdges();
    size_t at = 0;
    {
        double prm_k[8];
        int n_ball = 0, n_dev = 0, n_int = 0, n_uctl = 0;
        int32_t state_id = CORBO_HIP_INEQ_BALL;
        for (int k = 0; k < g.N - 1; ++k)
        {
            VertexInterface* x2 = (k + 1 < g.N - 1) ? g.xs[k + 1] : g.xf;
            if (at < ins.size() && ins[at]->getNumVertices() == 1 && ins[at]->getVertexRaw(0) == g.xs[k])
            {   // the state term: the keep-out ball, or a user function dropped into csrc/stage_functions/ (kind state_ineq)
                BaseEdge* e = ins[at].get();
                double* into = (n_ball == 0) ? d.ineq_params : prm_k;
                for (int i = 0; i < 8; ++i) into[i] = 0.0;
                int32_t id_k = CORBO_HIP_INEQ_BALL;
                if (!identifyBall(*e, g.xs[k], into))
                {
                    for (int i = 0; i < 8; ++i) into[i] = 0.0;
                    if (!identifyUserStageFunction(*e, g.xs[k], 0, id_k, into))
                        return fail(reason, "stage inequality " + std::to_string(k) + " is neither a keep-out ball on the first three state components nor a registered user state function (csrc/stage_functions/)");
                }
                if (n_ball == 0) state_id = id_k;
                else if (id_k != state_id || std::memcmp(d.ineq_params, prm_k, sizeof(prm_k)) != 0) return fail(reason, "stage inequality varies along the horizon");
                ++n_ball; ++at;
            }
            if (at < ins.size() && ins[at]->getNumVertices() == 1 && ins[at]->getVertexRaw(0) == g.us[k])
            {   // the control term (nlp_functions.cpp:82-89): a user function of kind control_ineq
                BaseEdge* e = ins[at].get();
                int32_t id_k = 0;
                double* into = (n_uctl == 0) ? d.ineq_control_params : prm_k;
                if (!identifyUserStageFunction(*e, g.us[k], 1, id_k, into))
                    return fail(reason, "inequality edge on u_" + std::to_string(k) + " is not a registered user control function (csrc/stage_functions/, kind=control_ineq)");
                if (n_uctl == 0) d.stage_ineq_control = id_k;
                else if (id_k != d.stage_ineq_control || std::memcmp(d.ineq_control_params, prm_k, sizeof(prm_k)) != 0) return fail(reason, "the stage inequalities' control term varies along the horizon");
                ++n_uctl; ++at;
            }
            if (at < ins.size() && dynamic_cast<ControlDeviationEdge*>(ins[at].get()))
            {
                BaseEdge* e = ins[at].get();
                if (e->getNumVertices() != 3 || e->getVertexRaw(0) != g.us[k] || (k > 0 && (e->getVertexRaw(1) != g.us[k - 1] || e->getVertexRaw(2) != g.dt)) ||
                    (k == 0 && (!e->getVertexRaw(1)->isFixed() || !e->getVertexRaw(2)->isFixed())))
                    return fail(reason, "control-deviation edge " + std::to_string(k) + " is not on (u_k, u_{k-1}, dt) resp. (u_0, previous control, its age)");
                if (!identifyRateLimit(*e, e->getVertexRaw(0), e->getVertexRaw(1), e->getVertexRaw(2), g.nu, d.ctrl_dev_params, n_dev > 0))
                    return fail(reason, "control-deviation term " + std::to_string(k) + " is not the input-rate limit ((u_k - u_prev) / dt)^2 - r_max^2 (the device's plug-in), or varies along the horizon");
                if (k == 0)
                {   // the previously applied control and its age: the values of the two fixed vertices
                    model->u_prev = Eigen::Map<const Eigen::VectorXd>(e->getVertexRaw(1)->getData(), g.nu);
                    model->u_prev_dt = e->getVertexRaw(2)->getData()[0];
                }
                ++n_dev; ++at;
            }
            if (at < ins.size() && (dynamic_cast<TrapezoidalIntegralInequalityEdge*>(ins[at].get()) || dynamic_cast<LeftSumInequalityEdge*>(ins[at].get())))
            {
                BaseEdge* e = ins[at].get();
                const bool trap = dynamic_cast<TrapezoidalIntegralInequalityEdge*>(e) != nullptr;
                if (e->getDimension() != 1 || e->getNumVertices() != (trap ? 4 : 3) || e->getVertexRaw(0) != g.xs[k] || e->getVertexRaw(1) != g.us[k] ||
                    e->getVertexRaw(trap ? 3 : 2) != g.dt || (trap && e->getVertexRaw(2) != x2))
                    return fail(reason, "integral inequality edge " + std::to_string(k) + ": not a one-row integrand on (x_k, u_k[, x_{k+1}], dt)");
                IntegrandProbe f(*e, trap ? std::vector<VertexInterface*>{g.xs[k], x2} : std::vector<VertexInterface*>{g.xs[k]}, g.us[k], g.dt, 0);
                if (!set_rule(trap ? 1 : 2) || !identifyBallIntegrand(f, g.nx, g.nu, d.ineq_params, n_int > 0))
                    return fail(reason, "integral stage inequality " + std::to_string(k) + " is not the keep-out ball on the first three state components, or varies along the horizon");
                ++n_int; ++at;
            }
        }
        if ((n_ball && n_ball != g.N - 1) || (n_dev && n_dev != g.N - 1) || (n_int && n_int != g.N - 1) || (n_ball && n_int) || (n_uctl && n_uctl != g.N - 1))
            return fail(reason, "stage inequality terms that are not created on every interval (or both a non-integral and an integral state term)");
        if (n_ball) d.stage_ineq = state_id;
        else if (n_int) d.stage_ineq = CORBO_HIP_INEQ_BALL;
        d.stage_ineq_integral = n_int ? 1 : 0;
        d.ctrl_dev = n_dev ? CORBO_HIP_CTRL_DEV_RATE : CORBO_HIP_CTRL_DEV_NONE;
    }
    if (at < ins.size() && ins[at]->getNumVertices() == 1 && ins[at]->getVertexRaw(0) == g.xf)
    {
        BaseEdge* e = ins[at].get();
        if (!identifyTerminalBall(*e, g.xf, model->xref, d.final_ineq_params))
            return fail(reason, "inequality edge on x_f that is not a TerminalBall (diagonal S) around the cost reference");
        d.final_ineq = CORBO_HIP_FINAL_INEQ_TERMINAL_BALL;
        ++at;
    }
    if (d.ctrl_dev)
    {   // finite_differences_grid.cpp:145-153: (u_ref, u_{N-2}, dt); u_ref = the control reference of the last interval, zero on this path
        BaseEdge* e = (at < ins.size()) ? ins[at].get() : nullptr;
        if (!e || !dynamic_cast<ControlDeviationEdge*>(e) || e->getNumVertices() != 3 || !e->getVertexRaw(0)->isFixed() || e->getVertexRaw(1) != g.us[g.N - 2] || e->getVertexRaw(2) != g.dt)
            return fail(reason, "control-deviation term without its final edge on (u_ref, u_{N-2}, dt)");
        for (int i = 0; i < g.nu; ++i)
            if (e->getVertexRaw(0)->getData()[i] != 0.0) return fail(reason, "control-deviation term with a non-zero control reference");
        double prm_f[CORBO_HIP_MAX_NU];
        std::memcpy(prm_f, d.ctrl_dev_params, sizeof(prm_f));
        if (!identifyRateLimit(*e, e->getVertexRaw(0), e->getVertexRaw(1), e->getVertexRaw(2), g.nu, prm_f, true))
            return fail(reason, "final control-deviation edge is not the input-rate limit of the intervals");
        ++at;
    }
    if (at != ins.size()) return fail(reason, "inequality edges the device cannot describe (kinds: keep-out ball / a registered user function on x_k, a registered user function on u_k, the ball as integrand, input-rate limit, TerminalBall)");
    d.constraint_integration = integral_rule;
    // (the control-deviation term is a non-integral term: the shooting grids create its edges too, multiple_shooting_grid.cpp:62, 193-197)
    if ((d.stage_eq || d.stage_ineq_integral) && (g.kind != CORBO_HIP_GRID_FD && g.kind != CORBO_HIP_GRID_FD_VARIABLE))
        return fail(reason, "integral-form constraint edges on a grid other than the finite-differences grids");
    if (refs_vary)
    {   // rows 0 .. N-2: the stage references, row N-1: the reference of the final-stage terms
        model->xref_traj.resize(g.N, g.nx);
        // (MinTimeQuadratic::only_last_n: the intervals before quad_first_interval carry no state term and no reference -- their rows are never
        //  read by a cost row; they get the first identified one)
        for (int k = 0; k < g.N - 1; ++k) model->xref_traj.row(k) = stage_refs[std::max(k, (int)d.quad_first_interval)].transpose();
        model->xref_traj.row(g.N - 1) = model->xref.transpose();
    }
    // ---- the dynamics object last (user systems are matched on the device)
    return describeDynamics(*dyn, d, reason);
}

}  // namespace corbo
