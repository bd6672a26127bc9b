// model.hpp -- device-side edge models: system dynamics f(x,u) and the dynamics-defect formulas.
//
// Everything here is evaluated with floating-point contraction OFF so that every operation is the same IEEE fp64
// operation, in the same order, as the reference's scalar C++ (which is compiled for baseline x86-64, no FMA).  The only
// arithmetic that can differ from the CPU is libm-vs-OCML sin/cos (<= 1-2 ulp), see DESIGN.md "numerical fidelity".
#pragma once

#include <hip/hip_runtime.h>

#include "../../include/corbo_hip.h"

namespace corbo_hip {

struct ModelParams {
    double dyn[8];
    double ineq[8];
    int32_t ineq_id;               // corbo_hip_problem_desc::stage_ineq: CORBO_HIP_INEQ_BALL or a user stage function (stage_ineq_state below)
    int32_t ineq_ctrl_id;          // corbo_hip_problem_desc::stage_ineq_control (stage_ineq_control below); its parameters travel with the extra edges' (SweepParams::xparams)
    double sq[CORBO_HIP_MAX_NX];   // sqrt(Q_ii)   (QuadraticFormCost::setWeightQ, quadratic_cost.cpp:59-67)
    double sr[CORBO_HIP_MAX_NU];   // sqrt(R_ii)
    double sqf[CORBO_HIP_MAX_NX];  // sqrt(Qf_ii)  (final_state_cost.cpp:60-68)
    double dt_weight;              // sqrt(N-1)    (minimum_time.h:60)
    double fin[CORBO_HIP_MAX_NX + 1];  // final-stage inequality: S_11 .. S_nn, gamma (TerminalBall)
    const double* wdense;          // or null: non-diagonal weights, [Uq 16 | Ur 16 | Uqf 16] row-major upper Cholesky factors (corbo_hip_problem_desc::q_sqrt ...)
    int32_t wdense_mask;           // bit 0 Q, bit 1 R, bit 2 Qf are dense (sq / sr / sqf of that class are then unused)
    int32_t fin_eq_mask;           // TerminalPartialEqualityConstraint: the active components (0 = all of them); read by the Hessian-path edges
};

#pragma clang fp contract(off)

// ---- SystemDynamicsInterface::dynamics -----------------------------------------------------------------------------
// Every model is split as  f(x,u) = eval(x, prepare(x), u):  prepare() holds the part that depends on the state only
// (the transcendental functions), so a sweep evaluates it once per grid state and once per perturbed state instead of
// once per finite-difference column.  eval(x, prepare(x), u) is, operation for operation, the reference formula, so
// the split changes no bit of any result.  CACHE_XMASK: bit i set = prepare() reads x[i].
template <int DYN> struct Dynamics;

template <> struct Dynamics<CORBO_HIP_DYN_VAN_DER_POL> {  // nonlinear_benchmark_systems.h:52-60
    static constexpr int NX = 2, NU = 1, NC = 1;
    static constexpr unsigned CACHE_XMASK = 0u;
    static constexpr unsigned RK4_CACHE_DEP_COLS = 0u, RK4_GROUP1_COLS = 0b11000u;
    __device__ static __forceinline__ void prepare(const double*, const double*, double* c) { c[0] = 0.0; }
    __device__ static __forceinline__ void eval(const double* x, const double*, const double* u, const double* prm, double* f)
    {
        const double a = prm[0];
        f[0]           = x[1];
        f[1]           = -a * (x[0] * x[0] - 1) * x[1] - x[0] + u[0];
    }
};

template <> struct Dynamics<CORBO_HIP_DYN_SERIAL_INTEGRATOR> {  // linear_benchmark_systems.h:72-83, order 2 (double integrator)
    static constexpr int NX = 2, NU = 1, NC = 1;
    static constexpr unsigned CACHE_XMASK = 0u;
    static constexpr unsigned RK4_CACHE_DEP_COLS = 0u, RK4_GROUP1_COLS = 0b11000u;
    __device__ static __forceinline__ void prepare(const double*, const double*, double* c) { c[0] = 0.0; }
    __device__ static __forceinline__ void eval(const double* x, const double*, const double* u, const double* prm, double* f)
    {
        f[0] = x[1];
        f[1] = u[0] / prm[0];
    }
};

// the same system of order 3 (public id CORBO_HIP_DYN_SERIAL_INTEGRATOR with nx = 3; internal template id)
constexpr int DYN_SERIAL_INTEGRATOR3 = 100;
template <> struct Dynamics<DYN_SERIAL_INTEGRATOR3> {
    static constexpr int NX = 3, NU = 1, NC = 1;
    static constexpr unsigned CACHE_XMASK = 0u;
    static constexpr unsigned RK4_CACHE_DEP_COLS = 0u, RK4_GROUP1_COLS = 0b1110000u;
    __device__ static __forceinline__ void prepare(const double*, const double*, double* c) { c[0] = 0.0; }
    __device__ static __forceinline__ void eval(const double* x, const double*, const double* u, const double* prm, double* f)
    {
        f[0] = x[1];
        f[1] = x[2];
        f[2] = u[0] / prm[0];
    }
};

// ---- the reference's other benchmark systems (nonlinear_benchmark_systems.h), each formula in the reference's
//      operation order (C++ evaluates a - b - c + d and a * b * c left to right)
template <> struct Dynamics<CORBO_HIP_DYN_DUFFING> {  // :108-115, prm = damping, spring_alpha, spring_beta
    static constexpr int NX = 2, NU = 1, NC = 1;
    static constexpr unsigned CACHE_XMASK = 0u;
    static constexpr unsigned RK4_CACHE_DEP_COLS = 0u, RK4_GROUP1_COLS = 0b11000u;
    __device__ static __forceinline__ void prepare(const double*, const double*, double* c) { c[0] = 0.0; }
    __device__ static __forceinline__ void eval(const double* x, const double*, const double* u, const double* prm, double* f)
    {
        f[0] = x[1];
        f[1] = -prm[0] * x[1] - prm[1] * x[0] - prm[2] * x[0] * x[0] * x[0] + u[0];
    }
};

template <> struct Dynamics<CORBO_HIP_DYN_FREE_SPACE_ROCKET> {  // :174-183
    static constexpr int NX = 3, NU = 1, NC = 1;
    static constexpr unsigned CACHE_XMASK = 0u;
    static constexpr unsigned RK4_CACHE_DEP_COLS = 0u, RK4_GROUP1_COLS = 0b1110000u;
    __device__ static __forceinline__ void prepare(const double*, const double*, double* c) { c[0] = 0.0; }
    __device__ static __forceinline__ void eval(const double* x, const double*, const double* u, const double*, double* f)
    {
        f[0] = x[1];
        f[1] = (u[0] - 0.02 * x[1] * x[1]) / x[2];
        f[2] = -0.01 * u[0] * u[0];
    }
};

template <> struct Dynamics<CORBO_HIP_DYN_SIMPLE_PENDULUM> {  // :207-215, prm = m, l, g, rho
    static constexpr int NX = 2, NU = 1, NC = 1;
    static constexpr unsigned CACHE_XMASK = 0b01u;                                   // sin(phi)
    static constexpr unsigned RK4_CACHE_DEP_COLS = 0b00111u, RK4_GROUP1_COLS = 0b11100u;   // phi at the later stages depends on phi, phidot, u
    __device__ static __forceinline__ void prepare(const double* x, const double*, double* c) { c[0] = sin(x[0]); }
    __device__ static __forceinline__ void eval(const double* x, const double* c, const double* u, const double* prm, double* f)
    {
        const double m = prm[0], l = prm[1], g = prm[2], rho = prm[3];
        f[0] = x[1];
        f[1] = u[0] - rho / (m * l * l) * x[1] - g / l * c[0];
    }
};

template <> struct Dynamics<CORBO_HIP_DYN_MASSLESS_PENDULUM> {  // :281-289, prm[0] = omega0
    static constexpr int NX = 2, NU = 1, NC = 1;
    static constexpr unsigned CACHE_XMASK = 0b01u;
    static constexpr unsigned RK4_CACHE_DEP_COLS = 0b00111u, RK4_GROUP1_COLS = 0b11100u;
    __device__ static __forceinline__ void prepare(const double* x, const double*, double* c) { c[0] = sin(x[0]); }
    __device__ static __forceinline__ void eval(const double* x, const double* c, const double* u, const double* prm, double* f)
    {
        f[0] = x[1];
        f[1] = u[0] - prm[0] * c[0];
    }
};

template <> struct Dynamics<CORBO_HIP_DYN_TOY_EXAMPLE> {  // :426-436, prm[0] = mu
    static constexpr int NX = 2, NU = 1, NC = 1;
    static constexpr unsigned CACHE_XMASK = 0u;
    static constexpr unsigned RK4_CACHE_DEP_COLS = 0u, RK4_GROUP1_COLS = 0b11000u;
    __device__ static __forceinline__ void prepare(const double*, const double*, double* c) { c[0] = 0.0; }
    __device__ static __forceinline__ void eval(const double* x, const double*, const double* u, const double* prm, double* f)
    {
        const double mu = prm[0];
        f[0] = x[1] + u[0] * (mu + (1.0 - mu) * x[0]);
        f[1] = x[0] + u[0] * (mu - 4.0 * (1.0 - mu) * x[1]);
    }
};

template <> struct Dynamics<CORBO_HIP_DYN_ARTSTEINS_CIRCLE> {  // :483-491
    static constexpr int NX = 2, NU = 1, NC = 1;
    static constexpr unsigned CACHE_XMASK = 0u;
    static constexpr unsigned RK4_CACHE_DEP_COLS = 0u, RK4_GROUP1_COLS = 0b11000u;
    __device__ static __forceinline__ void prepare(const double*, const double*, double* c) { c[0] = 0.0; }
    __device__ static __forceinline__ void eval(const double* x, const double*, const double* u, const double*, double* f)
    {
        f[0] = (x[0] * x[0] - x[1] * x[1]) * u[0];
        f[1] = 2 * x[0] * x[1] * u[0];
    }
};

// ParallelIntegratorSystem of dimension 2 / 3 (linear_benchmark_systems.h:142-148: f = T u; public id CORBO_HIP_DYN_PARALLEL_INTEGRATOR
// with nx = nu = p; internal template ids)
constexpr int DYN_PARALLEL_INTEGRATOR2 = 101, DYN_PARALLEL_INTEGRATOR3 = 102;
template <int P> struct ParallelIntegratorDynamics {
    static constexpr int NX = P, NU = P, NC = 1;
    static constexpr unsigned CACHE_XMASK = 0u;
    static constexpr unsigned RK4_CACHE_DEP_COLS = 0u;
    static constexpr unsigned RK4_GROUP1_COLS = (P == 2) ? 0b111000u : 0b111100000u;   // the last control and x_{k+1}
    __device__ static __forceinline__ void prepare(const double*, const double*, double* c) { c[0] = 0.0; }
    __device__ static __forceinline__ void eval(const double*, const double*, const double* u, const double* prm, double* f)
    {
#pragma unroll
        for (int i = 0; i < P; ++i) f[i] = prm[0] * u[i];
    }
};
template <> struct Dynamics<DYN_PARALLEL_INTEGRATOR2> : ParallelIntegratorDynamics<2> {};
template <> struct Dynamics<DYN_PARALLEL_INTEGRATOR3> : ParallelIntegratorDynamics<3> {};

// LinearStateSpaceModel (linear_benchmark_systems.h:206-213: f = A x + B u; public id CORBO_HIP_DYN_LINEAR_STATE_SPACE, one internal
// template id per (nx, nu) family).  Eigen evaluates `f = A * x + B * u` as ONE running sum per row -- f_i = 0, += A_i0 x_0, += A_i1 x_1,
// ..., += B_i0 u_0, ... (dst = A x; dst += B u, column-major gemv accumulating column by column; a full block of FOUR columns is added
// pairwise, (c0 + c1) + (c2 + c3) -- measured against the compiled reference) -- restated here.  The matrices do not
// fit the 8 model parameters: prm[0] carries, as a bit pattern, the device address of the row-major table [A | B] (set by the host
// side of the C-ABI; uniform, so the entries arrive through scalar loads).
constexpr int dyn_linear_id(int nx, int nu) { return 200 + 10 * nx + nu; }
template <int NXv, int NUv> struct LinearDynamics {
    static constexpr int NX = NXv, NU = NUv, NC = 1;
    static constexpr unsigned CACHE_XMASK = 0u;
    static constexpr unsigned RK4_CACHE_DEP_COLS = 0u;
    // group 1: the last ceil(NU / 2) controls' worth of columns and x_{k+1} (the same split as the models of equal shape above)
    static constexpr unsigned RK4_GROUP1_COLS = (((1u << (2 * NXv + NUv)) - 1u) >> (NXv + (NUv + 1) / 2)) << (NXv + (NUv + 1) / 2);
    __device__ static __forceinline__ void prepare(const double*, const double*, double* c) { c[0] = 0.0; }
    __device__ static __forceinline__ void eval(const double* x, const double*, const double* u, const double* prm, double* f)
    {
        const double* ab = reinterpret_cast<const double*>(__double_as_longlong(prm[0]));
#pragma unroll
        for (int i = 0; i < NXv; ++i) {
            double acc = 0.0;
            if constexpr (NXv == 4) {   // a full block of four columns: Eigen's gemv kernel adds it pairwise
                const double* a = ab + i * 4;
                acc = (a[0] * x[0] + a[1] * x[1]) + (a[2] * x[2] + a[3] * x[3]);
            }
            else {
#pragma unroll
                for (int j = 0; j < NXv; ++j) acc += ab[i * NXv + j] * x[j];
            }
#pragma unroll
            for (int j = 0; j < NUv; ++j) acc += ab[NXv * NXv + i * NUv + j] * u[j];
            f[i] = acc;
        }
    }
};
template <> struct Dynamics<dyn_linear_id(2, 1)> : LinearDynamics<2, 1> {};
template <> struct Dynamics<dyn_linear_id(2, 2)> : LinearDynamics<2, 2> {};
template <> struct Dynamics<dyn_linear_id(3, 1)> : LinearDynamics<3, 1> {};
template <> struct Dynamics<dyn_linear_id(3, 2)> : LinearDynamics<3, 2> {};
template <> struct Dynamics<dyn_linear_id(3, 3)> : LinearDynamics<3, 3> {};
template <> struct Dynamics<dyn_linear_id(4, 1)> : LinearDynamics<4, 1> {};

template <> struct Dynamics<CORBO_HIP_DYN_CART_POLE> {  // :337-355; state [x phi xdot phidot]; the reference's fixed parameters
    static constexpr int NX = 4, NU = 1, NC = 2;
    static constexpr unsigned CACHE_XMASK = 0b0010u;                 // sin(phi), cos(phi)
    // phi at the later Runge-Kutta stages depends on phi, phidot and (through phidot') u; never on x or xdot
    static constexpr unsigned RK4_CACHE_DEP_COLS = 0b000011010u;
    static constexpr unsigned RK4_GROUP1_COLS    = 0b111111000u;    // phidot, u, x_{k+1}
    __device__ static __forceinline__ void prepare(const double* x, const double*, double* c) { sincos(x[1], &c[0], &c[1]); }
    __device__ static __forceinline__ void eval(const double* x, const double* c, const double* u, const double*, double* f)
    {
        const double mc = 1.0, mp = 0.3, l = 0.5, g = 9.81;
        const double s = c[0], co = c[1];
        const double sin_phi_phidot_sq = s * x[3] * x[3];
        const double denum             = mc + mp * (1 - co * co);    // std::pow(cos, 2) = cos * cos exactly
        f[0] = x[2];
        f[1] = x[3];
        f[2] = (l * mp * sin_phi_phidot_sq + u[0] + mp * g * co * s) / denum;
        f[3] = -(l * mp * co * sin_phi_phidot_sq + u[0] * co + (mp + mc) * g * s) / (l * denum);
    }
};

template <> struct Dynamics<CORBO_HIP_DYN_UNICYCLE> {  // user plug-in: xdot = u1 cos(th), ydot = u1 sin(th), thdot = u2
    static constexpr int NX = 3, NU = 2, NC = 2;
    static constexpr unsigned CACHE_XMASK = 0b100u;
    static constexpr unsigned RK4_CACHE_DEP_COLS = 0b10100u;      // theta, u2 (theta' = u2)
    static constexpr unsigned RK4_GROUP1_COLS    = 0b11110000u;   // u2, x_{k+1}
    __device__ static __forceinline__ void prepare(const double* x, const double*, double* c) { sincos(x[2], &c[0], &c[1]); }
    __device__ static __forceinline__ void eval(const double*, const double* c, const double* u, const double*, double* f)
    {
        f[0] = u[0] * c[1];
        f[1] = u[0] * c[0];
        f[2] = u[1];
    }
};

template <> struct Dynamics<CORBO_HIP_DYN_QUADROTOR> {  // user plug-in (DESIGN.md "quadrotor"); prm = g, m, Ixx, Iyy, Izz
    static constexpr int NX = 12, NU = 4, NC = 6;
    static constexpr unsigned CACHE_XMASK = 0b111000000u;  // roll, pitch, yaw
    // the angles at the later Runge-Kutta stages depend on the angles, the body rates and (through the rates) the torques only
    static constexpr unsigned RK4_CACHE_DEP_COLS = 0b1110111111000000u;                        // x[6..11], u[1..3]
    // two lanes per stage: group 1 takes x[3..5], x[11], the controls and the (trivial) x_{k+1} columns -- 4 columns that
    // re-evaluate sin/cos and 4 that do not; group 0 the other 5 + 3
    static constexpr unsigned RK4_GROUP1_COLS = 0b1111111111111111100000111000u;
    __device__ static __forceinline__ void prepare(const double* x, const double*, double* c)
    {
        sincos(x[6], &c[0], &c[1]);  // sphi, cphi
        sincos(x[7], &c[2], &c[3]);  // sth, cth
        sincos(x[8], &c[4], &c[5]);  // spsi, cpsi
    }
    __device__ static __forceinline__ void eval(const double* x, const double* c, const double* u, const double* prm, double* f)
    {
        const double g = prm[0], m = prm[1], Ixx = prm[2], Iyy = prm[3], Izz = prm[4];
        const double sphi = c[0], cphi = c[1], sth = c[2], cth = c[3], spsi = c[4], cpsi = c[5];
        const double tm = u[0] / m;
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

template <int DYN>
__device__ __forceinline__ void dyn_full(const double* x, const double* u, const double* prm, double* f)
{
    double c[Dynamics<DYN>::NC];
    Dynamics<DYN>::prepare(x, prm, c);
    Dynamics<DYN>::eval(x, c, u, prm, f);
}

// defects whose dynamics are evaluated AT the grid states x1 / x2 can reuse per-state caches
template <int DEFECT> struct DefectTraits {
    static constexpr bool cached = (DEFECT == CORBO_HIP_DEFECT_FORWARD || DEFECT == CORBO_HIP_DEFECT_BACKWARD ||
                                    DEFECT == CORBO_HIP_DEFECT_CRANK_NICOLSON);
};

// RK4_CACHE_DEP_COLS (local columns of a defect edge: x_k | u_k): bit set = perturbing that component can change what prepare()
// sees at SOME Runge-Kutta stage; for every other column the caches of the unperturbed evaluation are bit-identical and reused.
// RK4_GROUP1_COLS (x_k | u_k | x_{k+1}): which of the two lanes of a stage takes the column (balances the expensive ones).

// The cached defects are combinations of three parts: q = (x2 - x1)/dt, f1 = f(x1,u1), f2 = f(x2,u1).  A finite-difference
// column only re-evaluates the parts that depend on the perturbed component; the others are bit-identical by construction.
template <int DEFECT> struct DefectParts {
    static constexpr bool uses_f1 = (DEFECT == CORBO_HIP_DEFECT_FORWARD || DEFECT == CORBO_HIP_DEFECT_CRANK_NICOLSON);
    static constexpr bool uses_f2 = (DEFECT == CORBO_HIP_DEFECT_BACKWARD || DEFECT == CORBO_HIP_DEFECT_CRANK_NICOLSON);
};
template <int NX, int DEFECT>
__device__ __forceinline__ void defect_combine(const double* q, const double* f1, const double* f2, double* err)
{
#pragma unroll
    for (int i = 0; i < NX; ++i) {
        if constexpr (DEFECT == CORBO_HIP_DEFECT_FORWARD) err[i] = f1[i] - q[i];        // err = f(x1,u1); err -= (x2-x1)/dt
        else if constexpr (DEFECT == CORBO_HIP_DEFECT_BACKWARD) err[i] = f2[i] - q[i];  // err = f(x2,u1); err -= (x2-x1)/dt
        else err[i] = q[i] - 0.5 * (f1[i] + f2[i]);                                     // (x2-x1)/dt - 0.5*(f1 + f2)
    }
}

// defect from cached states: c1 = prepare(x1), c2 = prepare(x2)
template <int DYN, int DEFECT>
__device__ __forceinline__ void defect_eval_cached(const double* x1, const double* c1, const double* u1, const double* x2, const double* c2,
                                                   double dt, const double* prm, double* err)
{
    using D          = Dynamics<DYN>;
    constexpr int NX = D::NX;
    if constexpr (DEFECT == CORBO_HIP_DEFECT_FORWARD) {  // finite_differences_collocation.h:126-134
        D::eval(x1, c1, u1, prm, err);
#pragma unroll
        for (int i = 0; i < NX; ++i) err[i] -= (x2[i] - x1[i]) / dt;
    }
    else if constexpr (DEFECT == CORBO_HIP_DEFECT_BACKWARD) {  // :160-168
        D::eval(x2, c2, u1, prm, err);
#pragma unroll
        for (int i = 0; i < NX; ++i) err[i] -= (x2[i] - x1[i]) / dt;
    }
    else {  // Crank-Nicolson :228-238
        double f1[NX];
        D::eval(x1, c1, u1, prm, f1);
        D::eval(x2, c2, u1, prm, err);
#pragma unroll
        for (int i = 0; i < NX; ++i) err[i] = (x2[i] - x1[i]) / dt - 0.5 * (f1[i] + err[i]);
    }
}

// IntegratorExplicitRungeKutta5 / 6 / 7 on the shooting grids (explicit_integrators.h:371-394, :479-503, :600-628; shooting_integrator = 5, 6, 7):
// 6 / 8 / 11 stages k_s = dt f(x1 + combination of the earlier ones), every combination written with the reference's association (Eigen
// evaluates the expression coefficient-wise, left to right).  A rarely used option with up to eleven live stage vectors: it is NOT a
// branch of rk4_end_state (that would raise the register peak of every kernel that integrates with Runge-Kutta 4; an out-of-line call
// halved their occupancy just the same) but a defect formula of its own, DEFECT_SHOOTING_HIGH, instantiated for the stand-alone sweep
// and Hessian kernels of the small-block families only (such handles run the phases as separate launches).
constexpr int DEFECT_SHOOTING_HIGH = CORBO_HIP_DEFECT_RK4_SHOOTING + 1;   // internal: RK4_SHOOTING with shooting_integrator >= 5
template <int DYN>
__device__ __forceinline__ void rk_high_order_end_state(const double* x1, const double* u1, double dt, const double* prm, int order, double* xe)
{
    using D          = Dynamics<DYN>;
    constexpr int NX = D::NX;
    double k1[NX], k2[NX], k3[NX], k4[NX], k5[NX], k6[NX], k7[NX], k8[NX], t[NX];
#define RK_STAGE(K, EXPR)                                          \
    {                                                              \
        _Pragma("unroll") for (int i = 0; i < NX; ++i) t[i] = EXPR; \
        dyn_full<DYN>(t, u1, prm, K);                              \
        _Pragma("unroll") for (int i = 0; i < NX; ++i) K[i] *= dt; \
    }
    RK_STAGE(k1, x1[i])
    if (order == 5) {
        const double s6 = 2.449489742783178;   // std::sqrt(6.0), correctly rounded (0x1.3988e1409212ep+1)
        RK_STAGE(k2, x1[i] + 4.0 * k1[i] / 11.0)
        RK_STAGE(k3, x1[i] + (9.0 * k1[i] + 11.0 * k2[i]) / 50.0)
        RK_STAGE(k4, x1[i] + (-11.0 * k2[i] + 15.0 * k3[i]) / 4.0)
        RK_STAGE(k5, x1[i] + ((81.0 + 9.0 * s6) * k1[i] + (255.0 - 55.0 * s6) * k3[i] + (24.0 - 14.0 * s6) * k4[i]) / 600.0)
        RK_STAGE(k6, x1[i] + ((81.0 - 9.0 * s6) * k1[i] + (255.0 + 55.0 * s6) * k3[i] + (24.0 + 14.0 * s6) * k4[i]) / 600.0)
#pragma unroll
        for (int i = 0; i < NX; ++i) xe[i] = x1[i] + (4.0 * k1[i] + (16.0 + s6) * k5[i] + (16.0 - s6) * k6[i]) / 36.0;
    }
    else if (order == 6) {
        RK_STAGE(k2, x1[i] + 2.0 * k1[i] / 33.0)
        RK_STAGE(k3, x1[i] + 4.0 * k2[i] / 33.0)
        RK_STAGE(k4, x1[i] + (k1[i] + 3.0 * k3[i]) / 22.0)
        RK_STAGE(k5, x1[i] + (43.0 * k1[i] - 165.0 * k3[i] + 144.0 * k4[i]) / 64.0)
        RK_STAGE(k6, x1[i] + (-4053483.0 * k1[i] + 16334703.0 * k3[i] - 12787632.0 * k4[i] + 1057536.0 * k5[i]) / 826686.0)
        RK_STAGE(k7, x1[i] + (169364139.0 * k1[i] - 663893307.0 * k3[i] + 558275718.0 * k4[i] - 29964480.0 * k5[i] + 35395542.0 * k6[i]) / 80707214.0)
        RK_STAGE(k8, x1[i] + (-733.0 * k1[i] + 3102.0 * k3[i]) / 176.0 - (335763.0 * k4[i] / 23296.0) + (216.0 * k5[i] / 77.0) - (4617.0 * k6[i] / 2816.0) + (7203.0 * k7[i] / 9152.0))
#pragma unroll
        for (int i = 0; i < NX; ++i)
            xe[i] = x1[i] + (336336.0 * k1[i] + 1771561.0 * k4[i] + 1916928.0 * k5[i] + 597051.0 * k6[i] + 1411788.0 * k7[i] + 256256.0 * k8[i]) / 6289920.0;
    }
    else {
        double k9[NX], k10[NX], k11[NX];
        RK_STAGE(k2, x1[i] + 2.0 * k1[i] / 27.0)
        RK_STAGE(k3, x1[i] + (k1[i] + 3.0 * k2[i]) / 36.0)
        RK_STAGE(k4, x1[i] + (k1[i] + 3.0 * k3[i]) / 24.0)
        RK_STAGE(k5, x1[i] + (80.0 * k1[i] - 300.0 * k3[i] + 300.0 * k4[i]) / 192.0)
        RK_STAGE(k6, x1[i] + (k1[i] + 5.0 * k4[i] + 4.0 * k5[i]) / 20.0)
        RK_STAGE(k7, x1[i] + (-25.0 * k1[i] + 125.0 * k4[i] - 260.0 * k5[i] + 250.0 * k6[i]) / 108.0)
        RK_STAGE(k8, x1[i] + (93.0 * k1[i] + 244.0 * k5[i] - 200.0 * k6[i] + 13.0 * k7[i]) / 900.0)
        RK_STAGE(k9, x1[i] + (12.0 * k1[i] - 53.0 * k4[i]) / 6.0 + (1408.0 * k5[i] - 1070.0 * k6[i] + 67.0 * k7[i] + 270.0 * k8[i]) / 90.0)
        RK_STAGE(k10, x1[i] + (-12285.0 * k1[i] + 3105.0 * k4[i] - 105408.0 * k5[i] + 83970.0 * k6[i] - 4617.0 * k7[i] + 41310.0 * k8[i] - 1215.0 * k9[i]) / 14580.0)
        RK_STAGE(k11, x1[i] + (2383.0 * k1[i] - 8525.0 * k4[i] + 17984.0 * k5[i] - 15050.0 * k6[i] + 2133.0 * k7[i] + 2250.0 * k8[i] + 1125.0 * k9[i] + 1800.0 * k10[i]) / 4100.0)
#pragma unroll
        for (int i = 0; i < NX; ++i)
            xe[i] = x1[i] + (41.0 * k1[i] + 272.0 * k6[i] + 216.0 * k7[i] + 216.0 * k8[i] + 27.0 * k9[i] + 27.0 * k10[i] + 41.0 * k11[i]) / 840.0;
    }
#undef RK_STAGE
}

template <int DYN, bool REUSE>
__device__ __forceinline__ void rk4_end_state(const double* x1, const double* u1, double dt, const double* prm, double (&ck)[4][Dynamics<DYN>::NC], double* xe);

// ---- dynamics defect of the equality edge (x1, u1, x2, dt) ----------------------------------------------------------
template <int DYN, int DEFECT>
__device__ __forceinline__ void defect_eval(const double* x1, const double* u1, const double* x2, double dt, const double* prm, double* err)
{
    using D          = Dynamics<DYN>;
    constexpr int NX = D::NX;
    if constexpr (DEFECT == CORBO_HIP_DEFECT_FORWARD) {  // finite_differences_collocation.h:126-134
        dyn_full<DYN>(x1, u1, prm, err);
#pragma unroll
        for (int i = 0; i < NX; ++i) err[i] -= (x2[i] - x1[i]) / dt;
    }
    else if constexpr (DEFECT == CORBO_HIP_DEFECT_BACKWARD) {  // :160-168
        dyn_full<DYN>(x2, u1, prm, err);
#pragma unroll
        for (int i = 0; i < NX; ++i) err[i] -= (x2[i] - x1[i]) / dt;
    }
    else if constexpr (DEFECT == CORBO_HIP_DEFECT_MIDPOINT) {  // :194-202
        double t[NX];
#pragma unroll
        for (int i = 0; i < NX; ++i) t[i] = 0.5 * (x1[i] + x2[i]);
        dyn_full<DYN>(t, u1, prm, err);
#pragma unroll
        for (int i = 0; i < NX; ++i) err[i] -= (x2[i] - x1[i]) / dt;
    }
    else if constexpr (DEFECT == CORBO_HIP_DEFECT_CRANK_NICOLSON) {  // :228-238
        double f1[NX];
        dyn_full<DYN>(x1, u1, prm, f1);
        dyn_full<DYN>(x2, u1, prm, err);
#pragma unroll
        for (int i = 0; i < NX; ++i) err[i] = (x2[i] - x1[i]) / dt - 0.5 * (f1[i] + err[i]);
    }
    else if constexpr (DEFECT == DEFECT_SHOOTING_HIGH) {   // shooting with Runge-Kutta 5 / 6 / 7
        double xe[NX];
        rk_high_order_end_state<DYN>(x1, u1, dt, prm, (int)prm[7], xe);
#pragma unroll
        for (int i = 0; i < NX; ++i) { err[i] = xe[i]; err[i] -= x2[i]; }
    }
    else {  // shooting: x_{k+1}(integrator) - x2  (explicit_integrators.h + integrator_interface.h:217-222); the step itself: rk4_end_state below
        double ck[4][D::NC], xe[NX];
        rk4_end_state<DYN, false>(x1, u1, dt, prm, ck, xe);
#pragma unroll
        for (int i = 0; i < NX; ++i) { err[i] = xe[i]; err[i] -= x2[i]; }
    }
}

// End state of one explicit Runge-Kutta-4 step (explicit_integrators.h:280-295), operation for operation the sequence inside
// defect_eval above.  REUSE = false: the per-stage caches prepare(x_stage) are computed and returned in ck; true: taken from ck.
template <int DYN, bool REUSE>
__device__ __forceinline__ void rk4_end_state(const double* x1, const double* u1, double dt, const double* prm,
                                              double (&ck)[4][Dynamics<DYN>::NC], double* xe)
{
    using D          = Dynamics<DYN>;
    constexpr int NX = D::NX;
    // The weighted sum ((k1 + 2 k2) + 2 k3) + k4 is formed left to right AS the stages complete -- the reference's association, the same
    // bits -- so that only one stage vector is live at a time (the four of the textbook form cost the stage kernel 50 registers).
    double k[NX], sum[NX], t[NX];
    if constexpr (!REUSE) D::prepare(x1, prm, ck[0]);
    D::eval(x1, ck[0], u1, prm, k);
    // the other explicit integrators of the shooting grids (corbo_hip_problem_desc::shooting_integrator, carried in prm[7]; uniform branch)
    const int integ = (int)prm[7];
    if (integ != 0) {
        if (integ == 1) {   // IntegratorExplicitEuler (explicit_integrators.h:66-72): x2 = f; x2 *= dt; x2 += x1
#pragma unroll
            for (int i = 0; i < NX; ++i) xe[i] = k[i] * dt + x1[i];
            return;
        }
        double k1[NX];
#pragma unroll
        for (int i = 0; i < NX; ++i) { k[i] *= dt; k1[i] = k[i]; }
        if (integ == 2) {   // IntegratorExplicitRungeKutta2 (:127-138): k2 = dt f(x1 + k1); x2 = x1 + (k1 + k2) / 2
#pragma unroll
            for (int i = 0; i < NX; ++i) t[i] = x1[i] + k1[i];
            if constexpr (!REUSE) D::prepare(t, prm, ck[1]);
            D::eval(t, ck[1], u1, prm, k);
#pragma unroll
            for (int i = 0; i < NX; ++i) { k[i] *= dt; xe[i] = x1[i] + (k1[i] + k[i]) / 2.0; }
            return;
        }
        // IntegratorExplicitRungeKutta3 (:200-213): k2 = dt f(x1 + k1 / 2); k3 = dt f(x1 - k1 + 2 k2); x2 = x1 + (k1 + 4 k2 + k3) / 6
        double k2[NX];
#pragma unroll
        for (int i = 0; i < NX; ++i) t[i] = x1[i] + (k1[i] / 2.0);
        if constexpr (!REUSE) D::prepare(t, prm, ck[1]);
        D::eval(t, ck[1], u1, prm, k);
#pragma unroll
        for (int i = 0; i < NX; ++i) { k[i] *= dt; k2[i] = k[i]; t[i] = (x1[i] - k1[i]) + 2.0 * k2[i]; }
        if constexpr (!REUSE) D::prepare(t, prm, ck[2]);
        D::eval(t, ck[2], u1, prm, k);
#pragma unroll
        for (int i = 0; i < NX; ++i) { k[i] *= dt; xe[i] = x1[i] + ((k1[i] + 4.0 * k2[i]) + k[i]) / 6.0; }
        return;
    }
#pragma unroll
    for (int i = 0; i < NX; ++i) { k[i] *= dt; sum[i] = k[i]; t[i] = x1[i] + k[i] / 2.0; }
    if constexpr (!REUSE) D::prepare(t, prm, ck[1]);
    D::eval(t, ck[1], u1, prm, k);
#pragma unroll
    for (int i = 0; i < NX; ++i) { k[i] *= dt; sum[i] = sum[i] + 2.0 * k[i]; t[i] = x1[i] + k[i] / 2.0; }
    if constexpr (!REUSE) D::prepare(t, prm, ck[2]);
    D::eval(t, ck[2], u1, prm, k);
#pragma unroll
    for (int i = 0; i < NX; ++i) { k[i] *= dt; sum[i] = sum[i] + 2.0 * k[i]; t[i] = x1[i] + k[i]; }
    if constexpr (!REUSE) D::prepare(t, prm, ck[3]);
    D::eval(t, ck[3], u1, prm, k);
#pragma unroll
    for (int i = 0; i < NX; ++i) { k[i] *= dt; xe[i] = x1[i] + (sum[i] + k[i]) / 6.0; }
}

// final-stage inequality TerminalBall, diagonal S, non-zero reference (final_state_constraints.cpp:72-76):
//   xd = x - xref;  c = xd^T * S_diag * xd - gamma   (row vector times diagonal, then the inner product)
// Sum of the n terms of a quadratic form x^T W_diag x in the order of the reference's arithmetic: Eigen's vectorised reduction of the inner product
// (Redux.h, LinearVectorizedTraversal: packets of two doubles on the reference's baseline x86-64 build, two packet accumulators) -- lane sums
// (t0 + t2 [+ t4 ..]) and (t1 + t3 [+ t5 ..]), added, then an odd last term.  Up to three terms the plain left-to-right sum; from four on the
// rounding differs (oracle/corbo_oracle.c eigen_sum, pinned bit for bit by the nx = 6 / 12 fixtures of the Hessian path).
template <int N>
__host__ __device__ __forceinline__ double eigen_sum(const double (&t)[N])
{
    constexpr int AS2 = (N / 4) * 4, AS1 = (N / 2) * 2;
    if constexpr (N == 1) return t[0];
    else {
        double a0 = t[0], a1 = t[1];
        if constexpr (AS1 > 2) {
            double b0 = t[2], b1 = t[3];
#pragma unroll
            for (int i = 4; i < AS2; i += 4) { a0 += t[i]; a1 += t[i + 1]; b0 += t[i + 2]; b1 += t[i + 3]; }
            a0 += b0; a1 += b1;
            if constexpr (AS1 > AS2) { a0 += t[AS2]; a1 += t[AS2 + 1]; }
        }
        double res = a0 + a1;
        if constexpr (N > AS1) res += t[N - 1];
        return res;
    }
}

template <int NX>
__device__ __forceinline__ double terminal_ball(const double* x, const double* xref, const double* prm)
{
    double t[NX];
#pragma unroll
    for (int i = 0; i < NX; ++i) { const double xd = x[i] - xref[i]; t[i] = (xd * prm[i]) * xd; }
    return eigen_sum<NX>(t) - prm[NX];
}

// stage inequality on x_k (keep-out ball, cfg 5): c = r^2 - |pos - center|^2  (<= 0 feasible)
__host__ __device__ __forceinline__ double ineq_ball(const double* x, const double* prm)
{
    double dx = x[0] - prm[0], dy = x[1] - prm[1], dz = x[2] - prm[2];
    return prm[3] * prm[3] - (dx * dx + dy * dy + dz * dz);
}

// ---- user stage functions dropped into csrc/stage_functions/ (generated include list and registry: __graft_entry__.build(); README.md there):
//      StageFunction<slot>::value<NV>(vertex, prm), public id CORBO_HIP_STAGE_FN_USER + slot
constexpr int CORBO_HIP_STAGE_FN_STATE_INEQ = 0, CORBO_HIP_STAGE_FN_CONTROL_INEQ = 1;
template <int SLOT> struct StageFunction;
#if __has_include("stage_functions/_includes.inc")
#include "stage_functions/_includes.inc"
#endif
// the stage inequalities' non-integral STATE term c(x_k) by descriptor id (corbo_hip_problem_desc::stage_ineq): the keep-out ball or a user function
template <int NX>
__host__ __device__ __forceinline__ double stage_ineq_state(int id, const double* x, const double* prm)
{
    if (id == CORBO_HIP_INEQ_BALL) {
        if constexpr (NX >= 3) return ineq_ball(x, prm);
        else return 0.0;
    }
#if __has_include("stage_functions/_registry.inc")
    switch (id - CORBO_HIP_STAGE_FN_USER) {
#define CORBO_HIP_USER_STAGE(NAME, SLOT, KIND_, NXMIN) \
        case SLOT: if constexpr (KIND_ == CORBO_HIP_STAGE_FN_STATE_INEQ) return StageFunction<SLOT>::template value<NX>(x, prm); else break;
#include "stage_functions/_registry.inc"
#undef CORBO_HIP_USER_STAGE
        default: break;
    }
#endif
    return 0.0;
}
// is there a registered user state function this state dimension can carry?  (kernels that special-case the keep-out ball compile their general path only then)
template <int NX>
constexpr bool has_user_state_ineq()
{
    bool any = false;
#if __has_include("stage_functions/_registry.inc")
#define CORBO_HIP_USER_STAGE(NAME, SLOT, KIND_, NXMIN) any = any || (KIND_ == CORBO_HIP_STAGE_FN_STATE_INEQ && NX >= NXMIN);
#include "stage_functions/_registry.inc"
#undef CORBO_HIP_USER_STAGE
#endif
    return any;
}
// ... and their non-integral CONTROL term c(u_k) (corbo_hip_problem_desc::stage_ineq_control)
template <int NU>
__host__ __device__ __forceinline__ double stage_ineq_control(int id, const double* u, const double* prm)
{
#if __has_include("stage_functions/_registry.inc")
    switch (id - CORBO_HIP_STAGE_FN_USER) {
#define CORBO_HIP_USER_STAGE(NAME, SLOT, KIND_, NXMIN) \
        case SLOT: if constexpr (KIND_ == CORBO_HIP_STAGE_FN_CONTROL_INEQ) return StageFunction<SLOT>::template value<NU>(u, prm); else break;
#include "stage_functions/_registry.inc"
#undef CORBO_HIP_USER_STAGE
        default: break;
    }
#endif
    return 0.0;
}

// ---- user models dropped into csrc/models/ (generated include list: __graft_entry__.build(); see models/README.md)
#if __has_include("models/_includes.inc")
#include "models/_includes.inc"
#endif

#pragma clang fp contract(fast)

}  // namespace corbo_hip
