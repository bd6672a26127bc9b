"""ctypes wrapper of oracle/liboracle.so -- TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

from control_box_rst_amd.capi import Dims, LmOpts, ProblemDesc

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = os.path.join(_HERE, "liboracle.so")
_lib = None


class TraceEntry(C.Structure):
    _fields_ = [("k", C.c_int32), ("inner_passes", C.c_int32), ("accepted", C.c_int32), ("mu", C.c_double),
                ("rho", C.c_double), ("chi2", C.c_double), ("delta_norm", C.c_double)]


def build():
    subprocess.check_call(["make", "-s", "-C", _HERE, "oracle"])


def load():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(_LIB):
        build()
    lib = C.CDLL(_LIB)
    if not hasattr(lib, "oracle_sizeof_problem_desc") or lib.oracle_sizeof_problem_desc() != C.sizeof(ProblemDesc):
        build()   # compiled from another include/corbo_hip.h: a stale checker would read the descriptor with the wrong layout
        lib = C.CDLL(_LIB)
        assert lib.oracle_sizeof_problem_desc() == C.sizeof(ProblemDesc)
    dp, ip = C.POINTER(C.c_double), C.POINTER(C.c_int32)
    lib.oracle_create.argtypes = [C.POINTER(ProblemDesc)]
    lib.oracle_create.restype = C.c_void_p
    lib.oracle_destroy.argtypes = [C.c_void_p]
    lib.oracle_destroy.restype = None
    lib.oracle_get_dims.argtypes = [C.c_void_p, C.POINTER(Dims)]
    lib.oracle_get_structure.argtypes = [C.c_void_p, ip, ip]
    lib.oracle_init_trajectory.argtypes = [C.POINTER(ProblemDesc), dp, dp, dp]
    lib.oracle_set_data.argtypes = [C.c_void_p, dp, dp, dp, dp]
    lib.oracle_get_x.argtypes = [C.c_void_p, dp]
    lib.oracle_set_previous_control.argtypes = [C.c_void_p, dp, C.c_double]
    lib.oracle_warm_start.argtypes = [C.c_void_p, dp, C.c_int]
    lib.oracle_plant_step.argtypes = [C.c_void_p, C.c_int, C.c_double, dp, dp]
    lib.oracle_eval.argtypes = [C.c_void_p, C.c_double, C.c_double, C.c_double, dp, dp]
    lib.oracle_solve.argtypes = [C.c_void_p, C.POINTER(LmOpts), C.c_int, dp, C.POINTER(TraceEntry)]
    lib.oracle_solve_batch.argtypes = [C.POINTER(ProblemDesc), C.c_int, dp, dp, C.POINTER(LmOpts), dp, ip]
    lib.oracle_resample_trajectory.argtypes = [C.c_int, C.c_int, C.c_int, dp, C.c_int, dp]
    lib.oracle_adapt_grid_n.argtypes = [C.c_int, C.c_int, C.c_double, C.c_double, C.c_double, C.c_int, C.c_int]
    i3 = C.POINTER(C.c_int32)
    lib.oracle_set_references.argtypes = [C.c_void_p, dp]
    lib.oracle_get_param_offsets.argtypes = [C.c_void_p, i3]
    lib.oracle_hessian_nnz.argtypes = [C.c_void_p, C.c_int, i3]
    lib.oracle_hessian_structure.argtypes = [C.c_void_p, C.c_int, ip, ip, ip, ip, ip, ip]
    lib.oracle_hessian_values.argtypes = [C.c_void_p, C.c_int, C.c_double, dp, dp, dp, dp, dp]
    lib.oracle_linear_form.argtypes = [C.c_void_p, i3, ip, ip, dp, dp, dp]
    lib.oracle_objective_gradient.argtypes = [C.c_void_p, dp, dp]
    lib.oracle_create_generic.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, dp, dp, GENERIC_FUN]
    lib.oracle_create_generic.restype = C.c_void_p
    _lib = lib
    return lib


GENERIC_FUN = C.CFUNCTYPE(None, C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_double))


def _dp(a):
    return None if a is None else a.ctypes.data_as(C.POINTER(C.c_double))


class OracleProblem:
    """One OCP instance of the CPU restatement."""

    def __init__(self, desc: ProblemDesc):
        self.lib = load()
        self.desc = desc
        self.h = self.lib.oracle_create(C.byref(desc))
        if not self.h:
            raise ValueError("oracle_create: invalid descriptor")
        d = Dims()
        self.lib.oracle_get_dims(self.h, C.byref(d))
        self.dims = d

    def __del__(self):
        if getattr(self, "h", None):
            self.lib.oracle_destroy(self.h)
            self.h = None

    def structure(self):
        rows = np.zeros(self.dims.nnz, np.int32)
        cols = np.zeros(self.dims.nnz, np.int32)
        self.lib.oracle_get_structure(self.h, rows.ctypes.data_as(C.POINTER(C.c_int32)), cols.ctypes.data_as(C.POINTER(C.c_int32)))
        return rows, cols

    def init_trajectory(self, x0, xf):
        x0 = np.ascontiguousarray(x0, np.float64)
        xf = np.ascontiguousarray(xf, np.float64)
        out = np.zeros(self.dims.nv)
        rc = self.lib.oracle_init_trajectory(C.byref(self.desc), _dp(x0), _dp(xf), _dp(out))
        assert rc == 0
        return out

    def set_data(self, x, lb=None, ub=None, xref=None):
        self._keep = [np.ascontiguousarray(a, np.float64) if a is not None else None for a in (x, lb, ub, xref)]
        rc = self.lib.oracle_set_data(self.h, *[_dp(a) for a in self._keep])
        assert rc == 0

    def set_previous_control(self, u_prev=None, dt_prev=0.0):
        """setPreviousControlInput: the fixed vertices the control-deviation edge of interval 0 sees (None / 0 = zeros, dt_ref)."""
        self._keep_up = None if u_prev is None else np.ascontiguousarray(u_prev, np.float64)
        assert self.lib.oracle_set_previous_control(self.h, _dp(self._keep_up), float(dt_prev)) == 0

    def x(self):
        out = np.zeros(self.dims.nv)
        self.lib.oracle_get_x(self.h, _dp(out))
        return out

    def eval(self, w_eq, w_ineq, w_b, jacobian=True):
        values = np.zeros(self.dims.m)
        jac = np.zeros(self.dims.nnz) if jacobian else None
        rc = self.lib.oracle_eval(self.h, w_eq, w_ineq, w_b, _dp(values), _dp(jac))
        assert rc == 0
        return values, jac

    def set_references(self, ref):
        """ref: one reference per vertex component (nv) or None (static reference again)."""
        r = None if ref is None else np.ascontiguousarray(ref, np.float64)
        assert self.lib.oracle_set_references(self.h, _dp(r)) == 0

    def param_offsets(self):
        out = np.zeros(self.dims.n, np.int32)
        assert self.lib.oracle_get_param_offsets(self.h, out.ctypes.data_as(C.POINTER(C.c_int32))) == 0
        return out

    def hessians(self, lower, mult_obj=1.0, mult_eq=None, mult_ineq=None):
        """computeSparseHessians{Structure,Values}: three (rows, cols, values) triplet lists -- objective, equalities, inequalities."""
        nnz = (C.c_int32 * 3)()
        assert self.lib.oracle_hessian_nnz(self.h, int(lower), nnz) == 0
        rows = [np.zeros(max(1, n), np.int32) for n in nnz]
        cols = [np.zeros(max(1, n), np.int32) for n in nnz]
        vals = [np.zeros(max(1, n)) for n in nnz]
        ipp = lambda a: a.ctypes.data_as(C.POINTER(C.c_int32))
        assert self.lib.oracle_hessian_structure(self.h, int(lower), ipp(rows[0]), ipp(cols[0]), ipp(rows[1]), ipp(cols[1]), ipp(rows[2]), ipp(cols[2])) == 0
        me = None if mult_eq is None else np.ascontiguousarray(mult_eq, np.float64)
        mi = None if mult_ineq is None or len(mult_ineq) == 0 else np.ascontiguousarray(mult_ineq, np.float64)
        assert self.lib.oracle_hessian_values(self.h, int(lower), float(mult_obj), _dp(me), _dp(mi), _dp(vals[0]), _dp(vals[1]), _dp(vals[2])) == 0
        return [(rows[i][:nnz[i]], cols[i][:nnz[i]], vals[i][:nnz[i]]) for i in range(3)]

    def objective_gradient(self):
        """computeGradientObjective (n) and computeValueObjective at the current x."""
        g = np.zeros(self.dims.n)
        obj = C.c_double(0)
        assert self.lib.oracle_objective_gradient(self.h, _dp(g), C.byref(obj)) == 0
        return g, obj.value

    def linear_form(self):
        """computeSparseJacobianTwoSideBoundedLinearForm* (with the finite bounds) and its bounds: rows, cols, values, lbA, ubA."""
        nnz = C.c_int32(0)
        assert self.lib.oracle_linear_form(self.h, C.byref(nnz), None, None, None, None, None) == 0
        rows, cols, vals = np.zeros(nnz.value, np.int32), np.zeros(nnz.value, np.int32), np.zeros(nnz.value)
        ma = self.dims.eq + self.dims.ineq + self.dims.bounds
        lbA, ubA = np.zeros(ma), np.zeros(ma)
        ipp = lambda a: a.ctypes.data_as(C.POINTER(C.c_int32))
        assert self.lib.oracle_linear_form(self.h, C.byref(nnz), ipp(rows), ipp(cols), _dp(vals), _dp(lbA), _dp(ubA)) == 0
        return rows, cols, vals, lbA, ubA

    def warm_start(self, x0, shift=True):
        x0 = np.ascontiguousarray(x0, np.float64)
        assert self.lib.oracle_warm_start(self.h, _dp(x0), 1 if shift else 0) == 0

    def plant_step(self, x_plant, integrator, dt, disturbance=None):
        """SimulatedPlant::control on the stored trajectory's first control; returns the new plant state."""
        x = np.array(x_plant, np.float64)
        dist = None if disturbance is None else np.ascontiguousarray(disturbance, np.float64)
        assert self.lib.oracle_plant_step(self.h, int(integrator), float(dt), _dp(dist), _dp(x)) == 0
        return x

    def solve(self, opts: LmOpts, new_run=True):
        chi2 = C.c_double(0)
        trace = (TraceEntry * max(1, opts.iterations))()
        status = self.lib.oracle_solve(self.h, C.byref(opts), 1 if new_run else 0, C.byref(chi2), trace)
        tr = [{f: getattr(trace[i], f) for f, _ in TraceEntry._fields_} for i in range(opts.iterations)]
        return status, chi2.value, tr


class GenericProblem(OracleProblem):
    """A callback problem (the reference's SimpleOptimizationProblemWithCallbacks) run through the SAME oracle_solve as the OCPs.

    lsq / eq / ineq: Python callables x -> sequence of values (or None); their dimensions are given explicitly like in the
    reference's setObjectiveFunction(fun, dim, lsq_form=true) / setEqualityConstraint(fun, dim) / setInequalityConstraint(fun, dim)."""

    def __init__(self, n, lsq=None, dim_lsq=0, eq=None, dim_eq=0, ineq=None, dim_ineq=0, lb=None, ub=None):
        self.lib = load()
        self.desc = None

        def cb(xp, lp, ep, ip):
            x = np.ctypeslib.as_array(xp, shape=(n,))
            for fun, dim, out in ((lsq, dim_lsq, lp), (eq, dim_eq, ep), (ineq, dim_ineq, ip)):
                if fun is not None and dim > 0:
                    v = fun(x)
                    for j in range(dim):
                        out[j] = v[j]

        self._cb = GENERIC_FUN(cb)  # keep the trampoline alive as long as the problem
        lb = None if lb is None else np.ascontiguousarray(lb, np.float64)
        ub = None if ub is None else np.ascontiguousarray(ub, np.float64)
        self.h = self.lib.oracle_create_generic(n, dim_lsq, dim_eq, dim_ineq, _dp(lb), _dp(ub), self._cb)
        if not self.h:
            raise ValueError("oracle_create_generic: invalid arguments")
        d = Dims()
        self.lib.oracle_get_dims(self.h, C.byref(d))
        self.dims = d

    def init_trajectory(self, x0, xf):
        raise NotImplementedError("not an OCP")

    def set_references(self, ref):
        """ref: one reference per vertex component (nv) or None (static reference again)."""
        r = None if ref is None else np.ascontiguousarray(ref, np.float64)
        assert self.lib.oracle_set_references(self.h, _dp(r)) == 0

    def param_offsets(self):
        out = np.zeros(self.dims.n, np.int32)
        assert self.lib.oracle_get_param_offsets(self.h, out.ctypes.data_as(C.POINTER(C.c_int32))) == 0
        return out

    def hessians(self, lower, mult_obj=1.0, mult_eq=None, mult_ineq=None):
        """computeSparseHessians{Structure,Values}: three (rows, cols, values) triplet lists -- objective, equalities, inequalities."""
        nnz = (C.c_int32 * 3)()
        assert self.lib.oracle_hessian_nnz(self.h, int(lower), nnz) == 0
        rows = [np.zeros(max(1, n), np.int32) for n in nnz]
        cols = [np.zeros(max(1, n), np.int32) for n in nnz]
        vals = [np.zeros(max(1, n)) for n in nnz]
        ipp = lambda a: a.ctypes.data_as(C.POINTER(C.c_int32))
        assert self.lib.oracle_hessian_structure(self.h, int(lower), ipp(rows[0]), ipp(cols[0]), ipp(rows[1]), ipp(cols[1]), ipp(rows[2]), ipp(cols[2])) == 0
        me = None if mult_eq is None else np.ascontiguousarray(mult_eq, np.float64)
        mi = None if mult_ineq is None or len(mult_ineq) == 0 else np.ascontiguousarray(mult_ineq, np.float64)
        assert self.lib.oracle_hessian_values(self.h, int(lower), float(mult_obj), _dp(me), _dp(mi), _dp(vals[0]), _dp(vals[1]), _dp(vals[2])) == 0
        return [(rows[i][:nnz[i]], cols[i][:nnz[i]], vals[i][:nnz[i]]) for i in range(3)]

    def linear_form(self):
        """computeSparseJacobianTwoSideBoundedLinearForm* (with the finite bounds) and its bounds: rows, cols, values, lbA, ubA."""
        nnz = C.c_int32(0)
        assert self.lib.oracle_linear_form(self.h, C.byref(nnz), None, None, None, None, None) == 0
        rows, cols, vals = np.zeros(nnz.value, np.int32), np.zeros(nnz.value, np.int32), np.zeros(nnz.value)
        ma = self.dims.eq + self.dims.ineq + self.dims.bounds
        lbA, ubA = np.zeros(ma), np.zeros(ma)
        ipp = lambda a: a.ctypes.data_as(C.POINTER(C.c_int32))
        assert self.lib.oracle_linear_form(self.h, C.byref(nnz), ipp(rows), ipp(cols), _dp(vals), _dp(lbA), _dp(ubA)) == 0
        return rows, cols, vals, lbA, ubA

    def warm_start(self, x0, shift=True):
        raise NotImplementedError("not an OCP")


def resample_trajectory(nx: int, nu: int, x_old: np.ndarray, n_new: int) -> np.ndarray:
    """resampleTrajectory on the vertex layout of a free-dt grid: [x_0 u_0 | ... | x_f | dt] with n points -> n_new points."""
    lib = load()
    x_old = np.ascontiguousarray(x_old, np.float64)
    s = nx + nu
    n = (len(x_old) - nx - 1) // s + 1
    assert (n - 1) * s + nx + 1 == len(x_old), (len(x_old), n)
    out = np.zeros((n_new - 1) * s + nx + 1)
    assert lib.oracle_resample_trajectory(nx, nu, n, _dp(x_old), n_new, _dp(out)) == 0
    return out


ADAPT_SINGLE_STEP, ADAPT_AGGRESSIVE, ADAPT_SHRINK = 1, 2, 3


def adapt_grid_n(strategy: int, n: int, dt: float, dt_ref: float, hyst: float, n_min: int, n_max: int) -> int:
    return load().oracle_adapt_grid_n(strategy, n, dt, dt_ref, hyst, n_min, n_max)


def solve_batch(desc: ProblemDesc, x: np.ndarray, xref: np.ndarray, opts: LmOpts):
    """Sequential single-thread solve of a batch (CPU baseline leg). x: [B][nv] (copied), xref: [B][nx]."""
    lib = load()
    x = np.array(x, dtype=np.float64, order="C", copy=True)
    xref = np.ascontiguousarray(xref, np.float64)
    B = x.shape[0]
    chi2 = np.zeros(B)
    status = np.zeros(B, np.int32)
    rc = lib.oracle_solve_batch(C.byref(desc), B, _dp(x), _dp(xref), C.byref(opts), _dp(chi2), status.ctypes.data_as(C.POINTER(C.c_int32)))
    assert rc == 0
    return x, chi2, status
