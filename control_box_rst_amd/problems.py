"""Descriptors and synthetic instance generators for the BASELINE.json configurations.

Each builder returns the POD ``ProblemDesc`` the C-ABI consumes; it encodes exactly what the reference's C++ setters
would configure (FiniteDifferencesGrid::setNRef/setDtRef, QuadraticFormCost(Q, R, false, true),
QuadraticFinalStateCost(Qf, true), StructuredOptimalControlProblem::setBounds, ...).
"""
from __future__ import annotations

import math

import numpy as np

from . import capi
from .capi import INF, ProblemDesc


def _fill(arr, values, default=0.0):
    for i in range(len(arr)):
        arr[i] = default
    for i, v in enumerate(values):
        arr[i] = float(v)


def make_desc(*, grid, defect, dynamics, nx, nu, N, dt, stage_cost=capi.COST_QUADRATIC_LSQ, final_cost=1,
              stage_ineq=capi.INEQ_NONE, q=(), r=(), qf=(), x_lb=(), x_ub=(), u_lb=(), u_ub=(), xf_fixed_mask=0,
              dt_lb=0.0, dt_ub=INF, dyn_params=(), ineq_params=(), final_ineq=capi.FINAL_INEQ_NONE, final_ineq_params=(), final_eq=0) -> ProblemDesc:
    d = ProblemDesc()
    d.grid, d.defect, d.dynamics = grid, defect, dynamics
    d.stage_cost, d.final_cost, d.stage_ineq = stage_cost, final_cost, stage_ineq
    d.nx, d.nu, d.N = nx, nu, N
    d.xf_fixed_mask = xf_fixed_mask
    d.dt_ref, d.dt_lb, d.dt_ub = dt, dt_lb, dt_ub
    _fill(d.x_lb, x_lb, -INF)
    _fill(d.x_ub, x_ub, INF)
    _fill(d.u_lb, u_lb, -INF)
    _fill(d.u_ub, u_ub, INF)
    _fill(d.q_diag, q)
    _fill(d.r_diag, r)
    _fill(d.qf_diag, qf)
    _fill(d.dyn_params, dyn_params)
    _fill(d.ineq_params, ineq_params)
    d.final_ineq = final_ineq
    d.final_eq = final_eq
    _fill(d.final_ineq_params, final_ineq_params)
    return d


# ---- cfg 3 / 4 (headline): unicycle point-to-point, FiniteDifferencesGrid N=100, Crank-Nicolson -----------------
def unicycle_desc(N=100, dt=0.1, defect=capi.DEFECT_CRANK_NICOLSON, terminal_ball=None) -> ProblemDesc:
    """terminal_ball = (S_diag, gamma): TerminalBall final-stage constraint (x_f - xref)^T S (x_f - xref) <= gamma."""
    q = (1.0, 1.0, 0.1)
    extra = {}
    if terminal_ball is not None:
        extra = dict(final_ineq=capi.FINAL_INEQ_TERMINAL_BALL, final_ineq_params=tuple(terminal_ball[0]) + (terminal_ball[1],))
    return make_desc(grid=capi.GRID_FD, defect=defect, dynamics=capi.DYN_UNICYCLE, nx=3, nu=2, N=N, dt=dt,
                     q=q, r=(0.1, 0.05), qf=tuple(10.0 * v for v in q),
                     x_lb=(-10.0,) * 3, x_ub=(10.0,) * 3, u_lb=(-1.0,) * 2, u_ub=(1.0,) * 2, **extra)


UNICYCLE_WEIGHTS = (10.0, 10.0, 10.0)


def kinematic_car_desc(N=100, dt=0.1, defect=capi.DEFECT_CRANK_NICOLSON, wheelbase=2.5) -> ProblemDesc:
    """The user-model example csrc/models/kinematic_car.hpp (public dynamics id DYN_USER + 0) inside the unicycle's OCP (same cost and bounds)."""
    d = unicycle_desc(N=N, dt=dt, defect=defect)
    d.dynamics = capi.DYN_USER + 0
    d.dyn_params[0] = wheelbase
    return d


def unicycle_instances(batch: int, seed: int = 20260928, first: int = 0):
    """x0 = (U(-1,1), U(-1,1), U(-pi/4,pi/4)), xf = (2,1,0.5)+U(-0.5,0.5)^3 with default_rng(seed + i).

    ``first`` = global index of the first instance (rank offset for batch sharding)."""
    x0 = np.empty((batch, 3))
    xf = np.empty((batch, 3))
    for b in range(batch):
        rng = np.random.default_rng(seed + first + b)
        x0[b] = (rng.uniform(-1, 1), rng.uniform(-1, 1), rng.uniform(-math.pi / 4, math.pi / 4))
        xf[b] = np.array([2.0, 1.0, 0.5]) + rng.uniform(-0.5, 0.5, 3)
    return x0, xf


# ---- cfg 1: Van-der-Pol regulator, FiniteDifferencesGrid N=20 --------------------------------------------------
def vdp_desc(N=20, dt=0.1, defect=capi.DEFECT_CRANK_NICOLSON, terminal_ball=None) -> ProblemDesc:
    extra = {}
    if terminal_ball is not None:
        extra = dict(final_ineq=capi.FINAL_INEQ_TERMINAL_BALL, final_ineq_params=tuple(terminal_ball[0]) + (terminal_ball[1],))
    return make_desc(grid=capi.GRID_FD, defect=defect, dynamics=capi.DYN_VAN_DER_POL, nx=2, nu=1, N=N, dt=dt,
                     q=(1.0, 1.0), r=(0.1,), qf=(10.0, 10.0), u_lb=(-1.0,), u_ub=(1.0,), dyn_params=(1.0,), **extra)


VDP_WEIGHTS = (2.0, 2.0, 2.0)


# ---- cfg 2: time-optimal double integrator, FiniteDifferencesVariableGrid N=50, x_f fixed, MinimumTime(lsq) -----
def dint_desc(N=50, dt=0.1, shooting=False) -> ProblemDesc:
    """shooting=True: the same problem on a MultipleShootingVariableGrid with RK4 defects."""
    grid, defect = (capi.GRID_MS_VARIABLE, capi.DEFECT_RK4_SHOOTING) if shooting else (capi.GRID_FD_VARIABLE, capi.DEFECT_CRANK_NICOLSON)
    return make_desc(grid=grid, defect=defect, dynamics=capi.DYN_SERIAL_INTEGRATOR,
                     nx=2, nu=1, N=N, dt=dt, stage_cost=capi.COST_MIN_TIME_LSQ, final_cost=0,
                     u_lb=(-1.0,), u_ub=(1.0,), xf_fixed_mask=0b11, dt_lb=0.01, dt_ub=10.0, dyn_params=(1.0,))


DINT_WEIGHTS = (100.0, 100.0, 100.0)


def hessian_path_cost_form(d: ProblemDesc, integral: str = "") -> ProblemDesc:
    """The cost forms of the reference's IPOPT / QP callers (Hessian-path operators only; corbo_hip_solve refuses such a handle like
    LevenbergMarquardtSparse does): every cost term with lsq_form = False -- scalar terms x^T Q x, plain objective edges -- and, with
    integral = "trapezoidal" | "left_sum", QuadraticFormCost(integral_form = True): one TrapezoidalIntegralCostEdge / LeftSumCostEdge per
    interval of a FiniteDifferencesGrid instead of the per-vertex terms.  On a MultipleShootingGrid (set d.grid / d.defect afterwards) either
    value stands for the same thing: ONE MultipleShootingEdgeSingleControl per interval -- a mixed edge that integrates the cost along the
    shooting step with the grid's integrator and carries the defect as its equality part (multiple_shooting_grid.cpp:70-77)."""
    d.cost_nonlsq = 1
    d.cost_integral = {"": 0, "trapezoidal": 1, "left_sum": 2}[integral]
    return d


def min_time_quadratic(d: ProblemDesc, q, r, only_last_n=0) -> ProblemDesc:
    """Turn a time-optimal descriptor (free dt, MinimumTime) into the reference's MinTimeQuadratic(Q, R, integral=False, lsq=True)
    (hybrid_cost.h:189-303): the quadratic form's state and control terms next to the minimum-time term; only_last_n > 0: on the
    last intervals only (the class's option of that name)."""
    assert d.grid in (capi.GRID_FD_VARIABLE, capi.GRID_MS_VARIABLE)
    d.stage_cost = capi.COST_MIN_TIME_QUADRATIC_LSQ
    d.quad_first_interval = max(d.N - only_last_n, 0) if only_last_n > 0 else 0   # _quad_k_min of hybrid_cost.h:224-237
    for i, v in enumerate(q):
        d.q_diag[i] = v
    for i, v in enumerate(r):
        d.r_diag[i] = v
    return d


# ---- SerialIntegratorSystem of order 3 (reference built-in, linear_benchmark_systems.h:50-118): fixed grid + quadratic cost, or time-optimal
def int3_desc(N=30, dt=0.1, defect=capi.DEFECT_CRANK_NICOLSON, time_optimal=False, shooting=False) -> ProblemDesc:
    if time_optimal:
        grid, defect = (capi.GRID_MS_VARIABLE, capi.DEFECT_RK4_SHOOTING) if shooting else (capi.GRID_FD_VARIABLE, capi.DEFECT_CRANK_NICOLSON)
        return make_desc(grid=grid, defect=defect, dynamics=capi.DYN_SERIAL_INTEGRATOR, nx=3, nu=1, N=N,
                         dt=dt, stage_cost=capi.COST_MIN_TIME_LSQ, final_cost=0, u_lb=(-1.0,), u_ub=(1.0,), xf_fixed_mask=0b111,
                         dt_lb=0.01, dt_ub=10.0, dyn_params=(1.0,))
    q = (1.0, 0.5, 0.1)
    return make_desc(grid=capi.GRID_FD, defect=defect, dynamics=capi.DYN_SERIAL_INTEGRATOR, nx=3, nu=1, N=N, dt=dt,
                     q=q, r=(0.1,), qf=tuple(10.0 * v for v in q), u_lb=(-1.0,), u_ub=(1.0,), dyn_params=(1.0,))


INT3_WEIGHTS = (10.0, 10.0, 10.0)


# ---- the reference's other benchmark systems (nonlinear_benchmark_systems.h), their default parameters; the cost / bound
#      set-up of oracle/ref_driver.cpp's scenarios of the same names
BENCHMARK_SYSTEMS = {   # name: (dynamics id, nx, default parameters)
    "duffing": (capi.DYN_DUFFING, 2, (1.0, 1.0, 1.0)),              # damping, spring_alpha, spring_beta
    "rocket": (capi.DYN_FREE_SPACE_ROCKET, 3, ()),
    "pendulum": (capi.DYN_SIMPLE_PENDULUM, 2, (0.205, 0.34, 9.81, 0.0)),   # mass, length, gravitation, friction
    "mpendulum": (capi.DYN_MASSLESS_PENDULUM, 2, (1.0,)),           # omega0
    "toy": (capi.DYN_TOY_EXAMPLE, 2, (0.5,)),                       # mu
    "artstein": (capi.DYN_ARTSTEINS_CIRCLE, 2, ()),
    "cartpole": (capi.DYN_CART_POLE, 4, ()),                        # state [x phi xdot phidot], fixed parameters
}
BENCHMARK_WEIGHTS = (5.0, 5.0, 5.0)


def linear_desc(A, B, N=24, dt=0.1, defect=capi.DEFECT_CRANK_NICOLSON) -> ProblemDesc:
    """LinearStateSpaceModel f = A x + B u (linear_benchmark_systems.h:186-262); cost / bound set-up of oracle/ref_driver.cpp's `lin`."""
    A, B = np.atleast_2d(np.asarray(A, float)), np.atleast_2d(np.asarray(B, float))
    nx, nu = B.shape
    assert A.shape == (nx, nx)
    q = (1.0, 0.5, 0.2, 0.1)[:nx]
    r = (0.1, 0.2, 0.05)[:nu]
    d = make_desc(grid=capi.GRID_FD, defect=defect, dynamics=capi.DYN_LINEAR_STATE_SPACE, nx=nx, nu=nu, N=N, dt=dt, q=q, r=r,
                  qf=tuple(10.0 * v for v in q), u_lb=(-1.5,) * nu, u_ub=(1.5,) * nu)
    for i, v in enumerate(A.reshape(-1)):
        d.lin_a[i] = v
    for i, v in enumerate(B.reshape(-1)):
        d.lin_b[i] = v
    return d


def parallel_integrator_desc(p=2, N=24, dt=0.1, defect=capi.DEFECT_CRANK_NICOLSON) -> ProblemDesc:
    """ParallelIntegratorSystem of dimension p (linear_benchmark_systems.h:120-183), time constant 1; set-up of oracle/ref_driver.cpp's par2 / par3."""
    q = (1.0, 0.5, 0.2)[:p]
    r = (0.1, 0.2, 0.05)[:p]
    return make_desc(grid=capi.GRID_FD, defect=defect, dynamics=capi.DYN_PARALLEL_INTEGRATOR, nx=p, nu=p, N=N, dt=dt, q=q, r=r,
                     qf=tuple(10.0 * v for v in q), u_lb=(-1.5,) * p, u_ub=(1.5,) * p, dyn_params=(1.0,))


def benchmark_desc(name, N=24, dt=0.1, defect=capi.DEFECT_CRANK_NICOLSON) -> ProblemDesc:
    dyn, nx, prm = BENCHMARK_SYSTEMS[name]
    q = (1.0, 0.5, 0.2, 0.1)[:nx]
    return make_desc(grid=capi.GRID_FD, defect=defect, dynamics=dyn, nx=nx, nu=1, N=N, dt=dt, q=q, r=(0.1,), qf=tuple(10.0 * v for v in q),
                     u_lb=(-1.5,), u_ub=(1.5,), dyn_params=prm)


# ---- cfg 5: quadrotor (nx=12, nu=4), MultipleShootingGrid + RK4, u bounds, one nonlinear stage inequality (keep-out ball) ----
QUAD_Q = (1, 1, 1, 0.1, 0.1, 0.1, 0.5, 0.5, 0.5, 0.05, 0.05, 0.05)
QUAD_R = (0.01, 0.1, 0.1, 0.1)


def quad_desc(N=200, dt=0.05, time_optimal=False) -> ProblemDesc:
    if time_optimal:   # vargrid=1 of oracle/ref_driver.cpp: MultipleShootingVariableGrid, free dt, x_f fixed, MinimumTime
        return make_desc(grid=capi.GRID_MS_VARIABLE, defect=capi.DEFECT_RK4_SHOOTING, dynamics=capi.DYN_QUADROTOR, nx=12, nu=4, N=N, dt=dt,
                         stage_cost=capi.COST_MIN_TIME_LSQ, final_cost=0, u_lb=(0.0, -1.0, -1.0, -1.0), u_ub=(20.0, 1.0, 1.0, 1.0),
                         xf_fixed_mask=0xFFF, dt_lb=0.01, dt_ub=10.0, stage_ineq=capi.INEQ_BALL, ineq_params=(1.0, 0.5, 0.6, 0.4),
                         dyn_params=(9.81, 1.0, 0.01, 0.01, 0.02))
    return make_desc(grid=capi.GRID_MS, defect=capi.DEFECT_RK4_SHOOTING, dynamics=capi.DYN_QUADROTOR, nx=12, nu=4, N=N, dt=dt,
                     q=QUAD_Q, r=QUAD_R, qf=tuple(10.0 * v for v in QUAD_Q),
                     u_lb=(0.0, -1.0, -1.0, -1.0), u_ub=(20.0, 1.0, 1.0, 1.0),
                     stage_ineq=capi.INEQ_BALL, ineq_params=(1.0, 0.5, 0.6, 0.4),
                     dyn_params=(9.81, 1.0, 0.01, 0.01, 0.02))


QUAD_WEIGHTS = (10.0, 10.0, 10.0)


def planar_quadrotor_desc(N=50, dt=0.05, time_optimal=False, shooting=True) -> ProblemDesc:
    """The big-block user-model example csrc/models/planar_quadrotor.hpp (public dynamics id DYN_USER + 1; nx = 6, nu = 2): multiple shooting with
    RK4, thrust bounds, keep-out ball on (x, z, theta) -- scenario pquad of oracle/ref_driver.cpp.  time_optimal (vargrid=1 there): free dt on the
    MultipleShootingVariableGrid (shooting) / FiniteDifferencesVariableGrid, x_f fixed, MinimumTime cost."""
    q = (1.0, 1.0, 0.5, 0.1, 0.1, 0.05)
    if time_optimal:
        grid, defect = (capi.GRID_MS_VARIABLE, capi.DEFECT_RK4_SHOOTING) if shooting else (capi.GRID_FD_VARIABLE, capi.DEFECT_CRANK_NICOLSON)
        return make_desc(grid=grid, defect=defect, dynamics=capi.DYN_USER + 1, nx=6, nu=2, N=N, dt=dt, stage_cost=capi.COST_MIN_TIME_LSQ, final_cost=0,
                         u_lb=(0.0, 0.0), u_ub=(12.0, 12.0), xf_fixed_mask=0b111111, dt_lb=0.01, dt_ub=10.0,
                         stage_ineq=capi.INEQ_BALL, ineq_params=(1.0, 0.5, 0.0, 0.3), dyn_params=(1.0, 0.05, 0.25, 9.81))
    return make_desc(grid=capi.GRID_MS, defect=capi.DEFECT_RK4_SHOOTING, dynamics=capi.DYN_USER + 1, nx=6, nu=2, N=N, dt=dt,
                     q=q, r=(0.02, 0.02), qf=tuple(10.0 * v for v in q),
                     u_lb=(0.0, 0.0), u_ub=(12.0, 12.0),
                     stage_ineq=capi.INEQ_BALL, ineq_params=(1.0, 0.5, 0.0, 0.3),
                     dyn_params=(1.0, 0.05, 0.25, 9.81))


def quad_instances(batch: int, seed: int = 20260928, first: int = 0):
    """x0 = hover state near the origin, xf = (2,1,1)+U(-0.3,0.3)^3 position goal, default_rng(seed + i)."""
    x0 = np.zeros((batch, 12))
    xf = np.zeros((batch, 12))
    for b in range(batch):
        rng = np.random.default_rng(seed + first + b)
        x0[b, :3] = rng.uniform(-0.2, 0.2, 3)
        xf[b, :3] = np.array([2.0, 1.0, 1.0]) + rng.uniform(-0.3, 0.3, 3)
    return x0, xf


SCENARIOS = {
    "unicycle": (unicycle_desc, UNICYCLE_WEIGHTS),
    "vdp": (vdp_desc, VDP_WEIGHTS),
    "dint": (dint_desc, DINT_WEIGHTS),
    "int3": (int3_desc, INT3_WEIGHTS),
    "quad": (quad_desc, QUAD_WEIGHTS),
    "pquad": (planar_quadrotor_desc, QUAD_WEIGHTS),
}
SCENARIOS["par2"] = (lambda **kw: parallel_integrator_desc(2, **kw), BENCHMARK_WEIGHTS)
SCENARIOS["par3"] = (lambda **kw: parallel_integrator_desc(3, **kw), BENCHMARK_WEIGHTS)
for _name in BENCHMARK_SYSTEMS:
    SCENARIOS[_name] = ((lambda n: (lambda **kw: benchmark_desc(n, **kw)))(_name), BENCHMARK_WEIGHTS)
