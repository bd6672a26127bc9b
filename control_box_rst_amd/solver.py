"""Host-side mirror of the reference's solver plug-in for a BATCH of OCP instances, over the C-ABI (include/corbo_hip.h).

Method names and semantics follow ``corbo::LevenbergMarquardtSparse`` / ``corbo::NlpSolverInterface``
(reference: src/optimization/include/corbo-optimization/solver/levenberg_marquardt_sparse.h:68-157,
nlp_solver_interface.h:67-115): ``setIterations``, ``setPenaltyWeights``, ``setWeightAdapation``, ``initialize``,
``solve(new_run)``, ``clear``.  The reference's C++ adapter (control_box_rst_amd/adapter/) does the same from C++.
This module never falls back to a CPU implementation: every call goes to libcorbo_hip.so.
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import capi
from .capi import Dims, LmOpts, ProblemDesc, Stats


class CorboHipError(RuntimeError):
    pass


def _dp(a):
    return None if a is None else a.ctypes.data_as(C.POINTER(C.c_double))


def _ip(a):
    return None if a is None else a.ctypes.data_as(C.POINTER(C.c_int32))


def get_dims(desc: ProblemDesc) -> Dims:
    lib = capi.load()
    d = Dims()
    rc = lib.corbo_hip_get_dims(C.byref(desc), C.byref(d))
    if rc != 0:
        raise CorboHipError(f"corbo_hip_get_dims: {lib.corbo_hip_last_error().decode()}")
    return d


def get_structure(desc: ProblemDesc):
    lib = capi.load()
    d = get_dims(desc)
    rows = np.zeros(d.nnz, np.int32)
    cols = np.zeros(d.nnz, np.int32)
    rc = lib.corbo_hip_get_structure(C.byref(desc), _ip(rows), _ip(cols))
    if rc != 0:
        raise CorboHipError(f"corbo_hip_get_structure: {lib.corbo_hip_last_error().decode()}")
    return rows, cols


def init_trajectory(desc: ProblemDesc, x0, xf) -> np.ndarray:
    """FullDiscretizationGridBase::initializeSequences for a batch: x0, xf [B][nx] -> [B][nv]."""
    lib = capi.load()
    x0 = np.ascontiguousarray(np.atleast_2d(x0), np.float64)
    xf = np.ascontiguousarray(np.atleast_2d(xf), np.float64)
    d = get_dims(desc)
    out = np.zeros((x0.shape[0], d.nv))
    rc = lib.corbo_hip_init_trajectory(C.byref(desc), x0.shape[0], _dp(x0), _dp(xf), _dp(out))
    if rc != 0:
        raise CorboHipError(f"corbo_hip_init_trajectory: {lib.corbo_hip_last_error().decode()}")
    return out


class BatchedLevenbergMarquardt:
    """LevenbergMarquardtSparse for `batch` independent instances of one hypergraph structure, on one MI355X."""

    def __init__(self, desc: ProblemDesc, batch: int, device: int = 0, route: int = 0):
        self.lib = capi.load()
        self.desc = desc
        self.batch = int(batch)
        self.device = int(device)
        self.opts: LmOpts = capi.default_lm_opts()
        self.dims = get_dims(desc)
        self._h = C.c_void_p()
        # route: capi.ROUTE_* flags of corbo_hip_create_routed (A/B of two factorisation routes of one descriptor); 0 = the library's choice
        if route:
            rc = self.lib.corbo_hip_create_routed(C.byref(desc), self.batch, self.device, int(route), C.byref(self._h))
        else:
            rc = self.lib.corbo_hip_create(C.byref(desc), self.batch, self.device, C.byref(self._h))
        self._check(rc, "corbo_hip_create")

    # -- reference setters ------------------------------------------------------------------------------------------
    def setIterations(self, iterations: int):
        self.opts.iterations = int(iterations)

    def setPenaltyWeights(self, weight_eq: float, weight_ineq: float, weight_bounds: float):
        self.opts.weight_eq, self.opts.weight_ineq, self.opts.weight_bounds = weight_eq, weight_ineq, weight_bounds

    def setWeightAdapation(self, factor_eq, factor_ineq, factor_bounds, max_eq, max_ineq, max_bounds):  # (sic) reference spelling
        o = self.opts
        o.adapt_factor_eq, o.adapt_factor_ineq, o.adapt_factor_bounds = factor_eq, factor_ineq, factor_bounds
        o.adapt_max_eq, o.adapt_max_ineq, o.adapt_max_bounds = max_eq, max_ineq, max_bounds

    def isLsqSolver(self) -> bool:
        return True

    def initialize(self) -> bool:
        return bool(self._h)

    def clear(self):
        pass

    # -- data ----------------------------------------------------------------------------------------------------------
    def init_trajectory(self, x0, xf) -> np.ndarray:
        return init_trajectory(self.desc, x0, xf)

    def set_instance_data(self, x, lb=None, ub=None, xref=None):
        arrs = [None if a is None else np.ascontiguousarray(a, np.float64) for a in (x, lb, ub, xref)]
        assert arrs[0].shape == (self.batch, self.dims.nv), (arrs[0].shape, (self.batch, self.dims.nv))
        for a in arrs[1:3]:
            assert a is None or a.shape == (self.batch, self.dims.nv), (a.shape, (self.batch, self.dims.nv))
        if arrs[3] is not None:
            assert arrs[3].shape == (self.batch, self.desc.nx)
        rc = self.lib.corbo_hip_set_instance_data(self._h, *[_dp(a) for a in arrs])
        self._check(rc, "corbo_hip_set_instance_data")

    def warm_start(self, x0_new, shift: bool = True):
        """New moving-horizon run on the resident trajectories: the grid update of FullDiscretizationGridBase::update (shifting
        warm start when `shift`, then x_0 = x0_new) on the device.  x0_new: [batch][nx].  Follow with solve(new_run=True)."""
        x0_new = np.ascontiguousarray(x0_new, dtype=np.float64).reshape(self.batch, self.desc.nx)
        self._check(self.lib.corbo_hip_warm_start(self._h, x0_new.ctypes.data_as(C.POINTER(C.c_double)), 1 if shift else 0),
                    "corbo_hip_warm_start")

    def get_first_control(self) -> np.ndarray:
        """u_0 of every instance (getFirstControlInput): [batch][nu]."""
        u0 = np.empty((self.batch, self.desc.nu))
        self._check(self.lib.corbo_hip_get_first_control(self._h, u0.ctypes.data_as(C.POINTER(C.c_double))), "corbo_hip_get_first_control")
        return u0

    # ---- the plant side of a closed loop, on the device (SimulatedPlant with the descriptor's dynamics, full-state output)
    def plant_set_state(self, x):
        """SimulatedPlant::setInitialState: x [batch][nx]."""
        x = np.ascontiguousarray(x, dtype=np.float64).reshape(self.batch, self.desc.nx)
        self._check(self.lib.corbo_hip_plant_set_state(self._h, x.ctypes.data_as(C.POINTER(C.c_double))), "corbo_hip_plant_set_state")

    def plant_step(self, dt=None, integrator=capi.INTEGRATOR_RK4, disturbance=None):
        """SimulatedPlant::control: hold u_0 of every resident trajectory over dt, integrate, add the state disturbance [batch][nx]."""
        dt = float(self.desc.dt_ref if dt is None else dt)
        d = None if disturbance is None else np.ascontiguousarray(disturbance, dtype=np.float64).reshape(self.batch, self.desc.nx)
        self._check(self.lib.corbo_hip_plant_step(self._h, int(integrator), dt, None if d is None else d.ctypes.data_as(C.POINTER(C.c_double))),
                    "corbo_hip_plant_step")

    def plant_set_params(self, params=None):
        """The plants' own model parameters [B][8] (a plant that differs from the controller's model); None = the controller's again."""
        if params is None:
            self._check(self.lib.corbo_hip_plant_set_params(self._h, None), "corbo_hip_plant_set_params")
            return
        pr = np.ascontiguousarray(np.broadcast_to(np.asarray(params, np.float64), (self.batch, 8)))
        self._check(self.lib.corbo_hip_plant_set_params(self._h, _dp(pr)), "corbo_hip_plant_set_params")

    def set_instance_params(self, params=None):
        """Per-instance parameters [B][8] of the controller's dynamics (order of the descriptor's dyn_params); None = the descriptor's again."""
        if params is None:
            self._check(self.lib.corbo_hip_set_instance_params(self._h, None), "corbo_hip_set_instance_params")
            return
        pr = np.ascontiguousarray(np.asarray(params, np.float64))
        assert pr.shape == (self.batch, 8), pr.shape
        self._check(self.lib.corbo_hip_set_instance_params(self._h, _dp(pr)), "corbo_hip_set_instance_params")

    def plant_get_state(self) -> np.ndarray:
        x = np.empty((self.batch, self.desc.nx))
        self._check(self.lib.corbo_hip_plant_get_state(self._h, x.ctypes.data_as(C.POINTER(C.c_double))), "corbo_hip_plant_get_state")
        return x

    def warm_start_from_plant(self, shift: bool = True):
        """warm_start with the device-resident plant states as the measured states."""
        self._check(self.lib.corbo_hip_warm_start_from_plant(self._h, 1 if shift else 0), "corbo_hip_warm_start_from_plant")

    def closed_loop(self, steps, dt=None, integrator=capi.INTEGRATOR_RK4, shift=True, disturbance=None, ocp_iterations=1, log=True):
        """`steps` x [plant_step -> warm_start_from_plant -> solve(new_run) (+ ocp_iterations - 1 solves without new_run)] in one
        call (one synchronisation).  disturbance: [steps][batch][nx] or None.  Returns (states [steps][batch][nx], controls
        [steps][batch][nu]) when `log`, else None."""
        dt = float(self.desc.dt_ref if dt is None else dt)
        d = None if disturbance is None else np.ascontiguousarray(disturbance, dtype=np.float64).reshape(steps, self.batch, self.desc.nx)
        xs = np.empty((steps, self.batch, self.desc.nx)) if log else None
        us = np.empty((steps, self.batch, self.desc.nu)) if log else None
        ptr = lambda a: None if a is None else a.ctypes.data_as(C.POINTER(C.c_double))  # noqa: E731
        self._check(self.lib.corbo_hip_closed_loop(self._h, C.byref(self.opts), int(steps), int(ocp_iterations), int(integrator), dt,
                                                   1 if shift else 0, ptr(d), ptr(xs), ptr(us)), "corbo_hip_closed_loop")
        return (xs, us) if log else None

    def restore_instance_data(self):
        """Device-side re-arm of the batch with the last uploaded x (no PCIe traffic)."""
        self._check(self.lib.corbo_hip_restore_instance_data(self._h), "corbo_hip_restore_instance_data")

    # -- adaptive time-optimal grids: one handle per N, instances move between them (see adaptive_grid.py) ------------------------
    def set_previous_control(self, u_prev=None, dt_prev=None):
        """The previously applied control [B][nu] and its age [B] (StructuredOptimalControlProblem::setPreviousControlInput): what the
        control-deviation edge of interval 0 sees (descriptor field ctrl_dev).  None = zeros resp. the grid's dt."""
        up = None if u_prev is None else np.ascontiguousarray(np.broadcast_to(np.asarray(u_prev, np.float64), (self.batch, self.desc.nu)))
        dp_ = None if dt_prev is None else np.ascontiguousarray(np.broadcast_to(np.asarray(dt_prev, np.float64), (self.batch,)))
        self._check(self.lib.corbo_hip_set_previous_control(self._h, _dp(up), _dp(dp_)), "corbo_hip_set_previous_control")

    def prepare_slots(self, active: int):
        """Use the first `active` slots (and give a never-uploaded handle the descriptor's bound pattern)."""
        self._check(self.lib.corbo_hip_prepare_slots(self._h, int(active)), "corbo_hip_prepare_slots")
        self._active = int(active)

    def get_dt(self, active: int) -> np.ndarray:
        out = np.zeros(self.batch)   # the library writes one value per ACTIVE slot of the handle (<= batch), whatever the caller expects
        self._check(self.lib.corbo_hip_get_dt(self._h, _dp(out)), "corbo_hip_get_dt")
        return out[:active]

    def resample_into(self, dst: "BatchedLevenbergMarquardt", src_index, dst_index):
        """resampleTrajectory of this handle's instances src_index into slots dst_index of `dst` (its N may differ; same N = move)."""
        si = np.ascontiguousarray(src_index, np.int32)
        di = np.ascontiguousarray(dst_index, np.int32)
        assert si.shape == di.shape and si.ndim == 1
        self._check(self.lib.corbo_hip_resample_into(self._h, dst._h, len(si), _ip(si), _ip(di)), "corbo_hip_resample_into")

    def set_option(self, name: str, value: int):
        """Diagnostics / test hooks of the handle (corbo_hip_set_option): pass_limit, run_to_completion, pass_timeline, sweep_timeline."""
        self._check(self.lib.corbo_hip_set_option(self._h, name.encode(), int(value)), "corbo_hip_set_option")

    def set_profiling(self, enable: bool):
        self._check(self.lib.corbo_hip_set_profiling(self._h, 1 if enable else 0), "corbo_hip_set_profiling")

    # -- the hot path ---------------------------------------------------------------------------------------------------
    def solve(self, new_run: bool = True, rearm: bool = False):
        """rearm: a new run from the iterates of the last set_instance_data (restore_instance_data() + solve(new_run=True) in one call, new_run = 2 of the
        C-ABI: the run-to-completion kernel reads its start from the shadow copy itself)."""
        rc = self.lib.corbo_hip_solve(self._h, C.byref(self.opts), 2 if rearm else (1 if new_run else 0))
        self._check(rc, "corbo_hip_solve")

    def solve_async(self, new_run: bool = True, rearm: bool = False):
        """corbo_hip_solve_async: enqueue the solve and return (run-to-completion handles; others solve synchronously).  Results / timing / errors of
        the enqueued solves with the next synchronize() / solve() / get_*() / fetch_solution().  rearm: see solve()."""
        rc = self.lib.corbo_hip_solve_async(self._h, C.byref(self.opts), 2 if rearm else (1 if new_run else 0))
        self._check(rc, "corbo_hip_solve_async")

    def synchronize(self):
        self._check(self.lib.corbo_hip_synchronize(self._h), "corbo_hip_synchronize")

    def get_solution(self):
        x = np.zeros((self.batch, self.dims.nv))
        chi2 = np.zeros(self.batch)
        status = np.zeros(self.batch, np.int32)
        rc = self.lib.corbo_hip_get_solution(self._h, _dp(x), _dp(chi2), _ip(status))
        self._check(rc, "corbo_hip_get_solution")
        return x, chi2, status

    def fetch_solution(self):
        """Results in host-visible (pinned) memory owned by the handle: zero-copy numpy views (x [batch][nv], chi2, status), valid
        until the next call that stages data through the handle."""
        xp, cp, sp = C.POINTER(C.c_double)(), C.POINTER(C.c_double)(), C.POINTER(C.c_int32)()
        stride = C.c_int32(0)
        self._check(self.lib.corbo_hip_fetch_solution(self._h, C.byref(xp), C.byref(stride), C.byref(cp), C.byref(sp)), "corbo_hip_fetch_solution")
        # the pinned buffers belong to the handle and do not move: the numpy views are built once per address (np.ctypeslib.as_array on a
        # ctypes pointer costs ~40 us per call -- more than the rest of a one-OCP step)
        key = (C.cast(xp, C.c_void_p).value, C.cast(cp, C.c_void_p).value, C.cast(sp, C.c_void_p).value, stride.value)
        views = getattr(self, "_fetch_views", None)
        if views is None or views[0] != key:
            x = np.ctypeslib.as_array(xp, shape=(self.batch, stride.value))[:, : self.dims.nv]
            views = (key, x, np.ctypeslib.as_array(cp, shape=(self.batch,)), np.ctypeslib.as_array(sp, shape=(self.batch,)))
            self._fetch_views = views
        return views[1], views[2], views[3]

    def set_result_sink(self, enable: bool):
        """Let the solve kernel write the results into the handle's pinned host memory itself (see corbo_hip_set_result_sink)."""
        self._check(self.lib.corbo_hip_set_result_sink(self._h, 1 if enable else 0), "corbo_hip_set_result_sink")

    def get_phase_cycles(self) -> np.ndarray:
        """Per-instance phase totals of the last run-to-completion solve (option "phase_cycles"): [batch][8] int64, see corbo_hip_get_phase_cycles."""
        out = np.zeros((self.batch, 8), np.int64)
        self._check(self.lib.corbo_hip_get_phase_cycles(self._h, out.ctypes.data_as(C.POINTER(C.c_int64))), "corbo_hip_get_phase_cycles")
        return out

    def get_timing(self, reset=False):
        """(sum of the HIP-event times [ms] of the solves since the last reset, number of solves)."""
        ms, n = C.c_double(0), C.c_int64(0)
        self._check(self.lib.corbo_hip_get_timing(self._h, C.byref(ms), C.byref(n), 1 if reset else 0), "corbo_hip_get_timing")
        return float(ms.value), int(n.value)

    def factor_route(self) -> int:
        """capi.FACTOR_*: the factorisation this handle's solves run through (corbo_hip_factor_route)."""
        return int(self.lib.corbo_hip_factor_route(self._h))

    def get_stats(self) -> dict:
        s = Stats()
        self._check(self.lib.corbo_hip_get_stats(self._h, C.byref(s)), "corbo_hip_get_stats")
        return s.as_dict()

    def eval(self, w_eq=None, w_ineq=None, w_bounds=None, jacobian=True):
        """Residual vector [B][m] and Jacobian values [B][nnz] at the resident iterate (parity hook)."""
        o = self.opts
        w = (o.weight_eq if w_eq is None else w_eq, o.weight_ineq if w_ineq is None else w_ineq,
             o.weight_bounds if w_bounds is None else w_bounds)
        values = np.zeros((self.batch, self.dims.m))
        jac = np.zeros((self.batch, self.dims.nnz)) if jacobian else None
        rc = self.lib.corbo_hip_eval(self._h, *w, _dp(values), _dp(jac))
        self._check(rc, "corbo_hip_eval")
        return values, jac

    def set_references(self, xref_traj=None):
        """Time-varying state reference: xref_traj [B][N][nx] (grid point k; the last one serves the final-stage terms).  Without an
        argument: back to the static reference of set_instance_data."""
        if xref_traj is None:
            self._check(self.lib.corbo_hip_set_references(self._h, None), "corbo_hip_set_references")
            return
        d, B = self.desc, self.batch
        nx, N, S = d.nx, d.N, d.nx + d.nu
        ref = np.zeros((B, self.dims.nv))
        xr = np.broadcast_to(np.asarray(xref_traj, np.float64), (B, N, nx))
        for k in range(N):
            ref[:, k * S:k * S + nx] = xr[:, k]
        ref = np.ascontiguousarray(ref)
        self._check(self.lib.corbo_hip_set_references(self._h, _dp(ref)), "corbo_hip_set_references")

    def set_reference_trajectory(self, traj=None, step=0):
        """Resident tracking reference: traj [B][T][nx] sampled at the grid's dt; control step `step` sees samples step .. step+N-1 and
        closed_loop() advances the window by one sample per control step.  None: static reference again."""
        if traj is None:
            self._check(self.lib.corbo_hip_set_reference_trajectory(self._h, None, 0, 0), "corbo_hip_set_reference_trajectory")
            return
        tr = np.ascontiguousarray(np.broadcast_to(np.asarray(traj, np.float64), (self.batch,) + np.asarray(traj).shape[-2:]))
        assert tr.shape[2] == self.desc.nx
        self._check(self.lib.corbo_hip_set_reference_trajectory(self._h, _dp(tr), int(tr.shape[1]), int(step)), "corbo_hip_set_reference_trajectory")

    def hessian_structure(self, lower_part_only=True):
        """Three (rows, cols) pairs -- objective, equalities, inequalities -- of computeSparseHessiansStructure, in the reference's order."""
        nnz = np.zeros(3, np.int32)
        self._check(self.lib.corbo_hip_hessian_nnz(C.byref(self.desc), int(lower_part_only), _ip(nnz)), "corbo_hip_hessian_nnz")
        rows = [np.zeros(max(1, n), np.int32) for n in nnz]
        cols = [np.zeros(max(1, n), np.int32) for n in nnz]
        rc = self.lib.corbo_hip_hessian_structure(C.byref(self.desc), int(lower_part_only), _ip(rows[0]), _ip(cols[0]), _ip(rows[1]), _ip(cols[1]), _ip(rows[2]), _ip(cols[2]))
        self._check(rc, "corbo_hip_hessian_structure")
        return [(rows[c][:nnz[c]], cols[c][:nnz[c]]) for c in range(3)]

    def eval_hessians_views(self, lower_part_only=True, mult_obj=1.0, mult_eq=None, mult_ineq=None, device=False):
        """The same three value lists without the copy into caller arrays: numpy views of the handle's pinned buffer ([B][nnz] each, valid until
        the next Hessian-path call), or -- device=True -- the raw device addresses (ints) of the lists in HBM."""
        nnz = np.zeros(3, np.int32)
        self._check(self.lib.corbo_hip_hessian_nnz(C.byref(self.desc), int(lower_part_only), _ip(nnz)), "corbo_hip_hessian_nnz")
        me = None if mult_eq is None else np.ascontiguousarray(np.broadcast_to(mult_eq, (self.batch, self.dims.eq)), np.float64)
        mi = None if mult_ineq is None or self.dims.ineq == 0 else np.ascontiguousarray(np.broadcast_to(mult_ineq, (self.batch, self.dims.ineq)), np.float64)
        ptr = [C.POINTER(C.c_double)() for _ in range(3)]
        self._check(self.lib.corbo_hip_eval_hessians_views(self._h, int(lower_part_only), float(mult_obj), _dp(me), _dp(mi), 1 if device else 0,
                                                           C.byref(ptr[0]), C.byref(ptr[1]), C.byref(ptr[2])), "corbo_hip_eval_hessians_views")
        if device:
            return [C.cast(q, C.c_void_p).value for q in ptr]
        rows = getattr(self, "_active", self.batch)   # the library sizes and fills the lists for the handle's ACTIVE instances only
        return [np.ctypeslib.as_array(ptr[c], shape=(rows, int(nnz[c]))) if nnz[c] else np.zeros((rows, 0)) for c in range(3)]

    def eval_hessians(self, lower_part_only=True, mult_obj=1.0, mult_eq=None, mult_ineq=None):
        """computeSparseHessiansValues at the resident iterates: value arrays [B][nnz] of the objective / equality / inequality lists.
        mult_eq [B][eq], mult_ineq [B][ineq] (None = ones)."""
        nnz = np.zeros(3, np.int32)
        self._check(self.lib.corbo_hip_hessian_nnz(C.byref(self.desc), int(lower_part_only), _ip(nnz)), "corbo_hip_hessian_nnz")
        vals = [np.zeros((self.batch, max(1, int(n)))) for n in nnz]
        me = None if mult_eq is None else np.ascontiguousarray(np.broadcast_to(mult_eq, (self.batch, self.dims.eq)), np.float64)
        mi = None if mult_ineq is None or self.dims.ineq == 0 else np.ascontiguousarray(np.broadcast_to(mult_ineq, (self.batch, self.dims.ineq)), np.float64)
        for c in range(3):   # the library writes [B][nnz] densely
            vals[c] = np.zeros((self.batch, int(nnz[c]))) if nnz[c] else np.zeros((self.batch, 0))
        rc = self.lib.corbo_hip_eval_hessians(self._h, int(lower_part_only), float(mult_obj), _dp(me), _dp(mi),
                                              _dp(vals[0]) if nnz[0] else None, _dp(vals[1]) if nnz[1] else None, _dp(vals[2]) if nnz[2] else None)
        self._check(rc, "corbo_hip_eval_hessians")
        return vals

    def objective_gradient(self):
        """computeGradientObjective [B][n] and computeValueObjective [B] at the resident iterates (eval_grad_f / eval_f of an interior-point solver)."""
        grad, obj = np.zeros((self.batch, self.dims.n)), np.zeros(self.batch)
        self._check(self.lib.corbo_hip_eval_objective_gradient(self._h, _dp(grad), _dp(obj)), "corbo_hip_eval_objective_gradient")
        return grad, obj

    def linear_form(self):
        """lbA <= A dx <= ubA of the QP interface at the resident iterates: rows, cols (structure), vals [B][nnz], lbA, ubA [B][rows]."""
        nnz, nrows = C.c_int32(0), C.c_int32(0)
        self._check(self.lib.corbo_hip_linear_form_structure(C.byref(self.desc), C.byref(nnz), C.byref(nrows), None, None), "corbo_hip_linear_form_structure")
        rows, cols = np.zeros(nnz.value, np.int32), np.zeros(nnz.value, np.int32)
        self._check(self.lib.corbo_hip_linear_form_structure(C.byref(self.desc), C.byref(nnz), C.byref(nrows), _ip(rows), _ip(cols)), "corbo_hip_linear_form_structure")
        vals = np.zeros((self.batch, nnz.value))
        lbA, ubA = np.zeros((self.batch, nrows.value)), np.zeros((self.batch, nrows.value))
        self._check(self.lib.corbo_hip_eval_linear_form(self._h, _dp(vals), _dp(lbA), _dp(ubA)), "corbo_hip_eval_linear_form")
        return rows, cols, vals, lbA, ubA

    def time_sweep(self, with_jacobian=True, repeat=20) -> float:
        """Average duration [ms] of one edge/Jacobian sweep launch over the resident batch (HIP events)."""
        o = self.opts
        ms = C.c_float(0)
        rc = self.lib.corbo_hip_time_sweep(self._h, o.weight_eq, o.weight_ineq, o.weight_bounds, 1 if with_jacobian else 0,
                                           int(repeat), C.byref(ms))
        self._check(rc, "corbo_hip_time_sweep")
        return float(ms.value)

    def time_sweep_each(self, with_jacobian=True, repeat=50) -> np.ndarray:
        """Duration [ms] of each of `repeat` sweep launches, every launch bracketed by its own HIP events."""
        o = self.opts
        ms = (C.c_float * int(repeat))()
        self._check(self.lib.corbo_hip_time_sweep_each(self._h, o.weight_eq, o.weight_ineq, o.weight_bounds, 1 if with_jacobian else 0,
                                                       int(repeat), ms), "corbo_hip_time_sweep_each")
        return np.array(ms[:], dtype=np.float64)

    def time_factor(self, repeat=20, timeline=False):
        """Average duration [ms] of one assemble/factor/solve launch; optionally workgroup 0's phase stamps (shader clocks)."""
        ms = C.c_float(0)
        tl = (C.c_longlong * 8)() if timeline else None
        self._check(self.lib.corbo_hip_time_factor(self._h, int(repeat), C.byref(ms), tl), "corbo_hip_time_factor")
        return (float(ms.value), [int(v) for v in tl]) if timeline else float(ms.value)

    def device_views(self):
        """(x_ptr, chi2_ptr, stream) raw device pointers into the library's HBM buffers."""
        xp, cp, st = C.POINTER(C.c_double)(), C.POINTER(C.c_double)(), C.c_void_p()
        self._check(self.lib.corbo_hip_device_views(self._h, C.byref(xp), C.byref(cp), C.byref(st)), "corbo_hip_device_views")
        return C.cast(xp, C.c_void_p).value, C.cast(cp, C.c_void_p).value, st.value

    def device_tensor(self):
        """The resident iterates as a torch CUDA tensor [batch][row_stride] (float64) that ALIASES the library's HBM buffer (no copy):
        what an RCCL collective or any torch op reads.  Columns [0, dims.nv) are the vertex layout.  Synchronise with the handle's
        stream (self.synchronize()) before reading."""
        import torch
        xp, _, _ = self.device_views()
        stride = C.c_int32(0)
        self._check(self.lib.corbo_hip_device_row_stride(self._h, C.byref(stride)), "corbo_hip_device_row_stride")

        class _View:   # __cuda_array_interface__ holder; keeps the solver alive as long as the tensor
            def __init__(s, owner):
                s.owner = owner
                s.__cuda_array_interface__ = {"shape": (owner.batch, stride.value), "typestr": "<f8", "data": (xp, False), "version": 2}
        return torch.as_tensor(_View(self), device=torch.device("cuda", self.device))

    def close(self):
        if getattr(self, "_h", None) and self._h.value:
            self.lib.corbo_hip_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc, what):
        if rc != 0:
            raise CorboHipError(f"{what} failed ({rc}): {self.lib.corbo_hip_last_error().decode()}")
