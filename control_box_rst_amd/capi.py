"""ctypes mirror of include/corbo_hip.h (the C-ABI drop-in boundary) and loader of libcorbo_hip.so.

The structures here are byte-for-byte the PODs of ``include/corbo_hip.h``.  ``load()`` fails loudly when the HIP
library has not been built -- there is no CPU fallback in the product.
"""
from __future__ import annotations

import ctypes as C
import os

MAX_NX = 16
MAX_NU = 8
INF = 2e30  # CORBO_INF_DBL (reference: src/core/include/corbo-core/types.h:52)

# enums (names follow include/corbo_hip.h)
GRID_FD, GRID_FD_VARIABLE, GRID_MS, GRID_MS_VARIABLE = 0, 1, 2, 3
DEFECT_FORWARD, DEFECT_BACKWARD, DEFECT_MIDPOINT, DEFECT_CRANK_NICOLSON, DEFECT_RK4_SHOOTING = 0, 1, 2, 3, 4
DYN_VAN_DER_POL, DYN_SERIAL_INTEGRATOR, DYN_UNICYCLE, DYN_QUADROTOR = 0, 1, 2, 3
# the reference's other benchmark systems (nonlinear_benchmark_systems.h)
DYN_DUFFING, DYN_FREE_SPACE_ROCKET, DYN_SIMPLE_PENDULUM, DYN_MASSLESS_PENDULUM, DYN_TOY_EXAMPLE, DYN_ARTSTEINS_CIRCLE = 4, 5, 6, 7, 8, 9
DYN_CART_POLE, DYN_PARALLEL_INTEGRATOR, DYN_LINEAR_STATE_SPACE = 10, 11, 12
DYN_USER = 1000   # + slot: user models dropped into csrc/models/ (slot 0: the kinematic-car example)
COST_NONE, COST_QUADRATIC_LSQ, COST_MIN_TIME_LSQ = 0, 1, 2
COST_MIN_TIME_QUADRATIC_LSQ = 3   # MinTimeQuadratic(Q, R, integral=False, lsq=True): state, control and minimum-time terms
INEQ_NONE, INEQ_BALL = 0, 1
FINAL_INEQ_NONE, FINAL_INEQ_TERMINAL_BALL = 0, 1
RULE_TRAPEZOIDAL, RULE_LEFT_SUM = 1, 2         # constraint_integration (corbo_hip.h)
STAGE_EQ_NONE, STAGE_EQ_LINEAR = 0, 1
CTRL_DEV_NONE, CTRL_DEV_RATE = 0, 1
SOLVER_CONVERGED, SOLVER_EARLY_TERMINATED, SOLVER_INFEASIBLE, SOLVER_ERROR = 0, 1, 2, 3


class ProblemDesc(C.Structure):
    _fields_ = [
        ("grid", C.c_int32), ("defect", C.c_int32), ("dynamics", C.c_int32), ("stage_cost", C.c_int32),
        ("final_cost", C.c_int32), ("stage_ineq", C.c_int32),
        ("nx", C.c_int32), ("nu", C.c_int32), ("N", C.c_int32),
        ("xf_fixed_mask", C.c_uint32),
        ("dt_ref", C.c_double), ("dt_lb", C.c_double), ("dt_ub", C.c_double),
        ("x_lb", C.c_double * MAX_NX), ("x_ub", C.c_double * MAX_NX),
        ("u_lb", C.c_double * MAX_NU), ("u_ub", C.c_double * MAX_NU),
        ("q_diag", C.c_double * MAX_NX), ("r_diag", C.c_double * MAX_NU), ("qf_diag", C.c_double * MAX_NX),
        ("dyn_params", C.c_double * 8), ("ineq_params", C.c_double * 8),
        ("final_ineq", C.c_int32), ("final_eq", C.c_int32), ("final_ineq_params", C.c_double * (MAX_NX + 1)),
        ("lin_a", C.c_double * 16), ("lin_b", C.c_double * 12),
        ("quad_first_interval", C.c_int32), ("cost_nonlsq", C.c_int32),
        ("cost_integral", C.c_int32), ("weights_dense", C.c_int32), ("shooting_integrator", C.c_int32), ("final_eq_mask", C.c_uint32),
        ("q_sqrt", C.c_double * 16), ("r_sqrt", C.c_double * 16), ("qf_sqrt", C.c_double * 16),
        ("constraint_integration", C.c_int32), ("stage_ineq_integral", C.c_int32), ("stage_eq", C.c_int32), ("ctrl_dev", C.c_int32),
        ("stage_eq_params", C.c_double * (MAX_NX + MAX_NU + 1)), ("ctrl_dev_params", C.c_double * MAX_NU),
        ("stage_ineq_control", C.c_int32), ("reserved0", C.c_int32), ("ineq_control_params", C.c_double * 8),
    ]


class Dims(C.Structure):
    _fields_ = [(k, C.c_int32) for k in ("nv", "n", "lsq", "eq", "ineq", "bounds", "m", "nnz")]

    def as_dict(self):
        return {k: getattr(self, k) for k, _ in self._fields_}


class LmOpts(C.Structure):
    _fields_ = [
        ("iterations", C.c_int32),
        ("weight_eq", C.c_double), ("weight_ineq", C.c_double), ("weight_bounds", C.c_double),
        ("adapt_factor_eq", C.c_double), ("adapt_factor_ineq", C.c_double), ("adapt_factor_bounds", C.c_double),
        ("adapt_max_eq", C.c_double), ("adapt_max_ineq", C.c_double), ("adapt_max_bounds", C.c_double),
    ]


class Stats(C.Structure):
    _fields_ = [
        ("lm_iterations", C.c_int64), ("accepted_steps", C.c_int64), ("rejected_steps", C.c_int64),
        ("jacobian_sweeps", C.c_int64), ("residual_sweeps", C.c_int64), ("factorizations", C.c_int64),
        ("passes", C.c_int32), ("solve_ms", C.c_float), ("sweep_ms", C.c_float), ("factor_ms", C.c_float), ("inner_loop_cuts", C.c_int32),
        ("counted_iterations", C.c_int64), ("speculative_takeovers", C.c_int64),
    ]

    def as_dict(self):
        return {k: getattr(self, k) for k, _ in self._fields_}


def default_lm_opts(iterations=10, w_eq=2.0, w_ineq=2.0, w_bounds=2.0) -> LmOpts:
    """Reference defaults (levenberg_marquardt_sparse.h:112-124)."""
    return LmOpts(iterations, w_eq, w_ineq, w_bounds, 1.0, 1.0, 1.0, 500.0, 500.0, 500.0)


INTEGRATOR_EULER, INTEGRATOR_RK4 = 0, 1   # corbo_hip_integrator

ROUTE_FREE_DT_BAND, ROUTE_XE_BAND = 1, 2   # corbo_hip_create_routed
FACTOR_STAGE_CR, FACTOR_STAGE_CHAIN, FACTOR_BAND, FACTOR_BLOCK_TRI = 0, 1, 2, 3   # corbo_hip_factor_route
STAGE_FN_USER = 1000                        # CORBO_HIP_STAGE_FN_USER: + slot of a user stage function (csrc/stage_functions/)

# CORBO_HIP_LIB: A/B measurements of two builds of the same C-ABI in one GPU session (development only)
_LIB_PATH = os.environ.get("CORBO_HIP_LIB") or os.path.join(os.path.dirname(os.path.abspath(__file__)), "csrc", "libcorbo_hip.so")
_lib = None

# every symbol include/corbo_hip.h declares
EXPORTED_SYMBOLS = (
    "corbo_hip_default_lm_opts", "corbo_hip_get_dims", "corbo_hip_get_structure", "corbo_hip_init_trajectory",
    "corbo_hip_create", "corbo_hip_destroy", "corbo_hip_set_instance_data", "corbo_hip_solve", "corbo_hip_solve_async",
    "corbo_hip_synchronize", "corbo_hip_get_solution", "corbo_hip_get_stats", "corbo_hip_eval",
    "corbo_hip_device_views", "corbo_hip_time_sweep", "corbo_hip_last_error",
    "corbo_hip_restore_instance_data", "corbo_hip_set_profiling", "corbo_hip_time_factor", "corbo_hip_warm_start", "corbo_hip_get_first_control",
    "corbo_hip_plant_set_state", "corbo_hip_plant_step", "corbo_hip_plant_get_state", "corbo_hip_plant_set_params", "corbo_hip_set_instance_params", "corbo_hip_warm_start_from_plant",
    "corbo_hip_closed_loop", "corbo_hip_fetch_solution", "corbo_hip_get_timing", "corbo_hip_time_sweep_each", "corbo_hip_set_result_sink", "corbo_hip_eval_dynamics", "corbo_hip_set_option", "corbo_hip_prepare_slots", "corbo_hip_get_dt", "corbo_hip_resample_into",
    "corbo_hip_device_count", "corbo_hip_shard_bounds", "corbo_hip_device_row_stride",
    "corbo_hip_set_references", "corbo_hip_set_reference_trajectory", "corbo_hip_hessian_nnz", "corbo_hip_hessian_structure", "corbo_hip_eval_hessians", "corbo_hip_eval_hessians_views", "corbo_hip_linear_form_structure", "corbo_hip_eval_linear_form", "corbo_hip_eval_objective_gradient",
    "corbo_hip_sizeof", "corbo_hip_set_previous_control", "corbo_hip_get_phase_cycles", "corbo_hip_create_routed", "corbo_hip_factor_route", "corbo_hip_stage_function_kind", "corbo_hip_eval_stage_function",
)


def lib_path() -> str:
    return _LIB_PATH


def load() -> C.CDLL:
    """Load libcorbo_hip.so (built by __graft_entry__.build()).  Raises if it is missing: no fallback."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(_LIB_PATH):
        raise RuntimeError(
            f"{_LIB_PATH} not found: build the HIP extension first (python -c 'import __graft_entry__ as g; g.build()'). "
            "control_box_rst_amd has no CPU fallback.")
    try:
        # One HIP runtime per process: PyTorch-ROCm ships its own libamdhip64; let it load first so that libcorbo_hip.so
        # binds to the same runtime (two runtimes in one process do not both see the GPU).
        import torch  # noqa: F401
    except Exception:  # torch is optional for the C-ABI itself
        pass
    lib = C.CDLL(_LIB_PATH)
    lib.corbo_hip_sizeof.argtypes, lib.corbo_hip_sizeof.restype = [C.c_int], C.c_size_t
    for which, mirror in ((0, ProblemDesc), (1, Dims), (2, LmOpts), (3, Stats)):
        if lib.corbo_hip_sizeof(which) != C.sizeof(mirror):
            raise RuntimeError(f"{_LIB_PATH} was built from another include/corbo_hip.h: sizeof({mirror.__name__}) = {C.sizeof(mirror)} here, "
                               f"{lib.corbo_hip_sizeof(which)} in the library -- rebuild it (python -c 'import __graft_entry__ as g; g.build()')")
    dp, ip = C.POINTER(C.c_double), C.POINTER(C.c_int32)
    H = C.c_void_p
    lib.corbo_hip_default_lm_opts.argtypes = [C.POINTER(LmOpts)]
    lib.corbo_hip_default_lm_opts.restype = None
    lib.corbo_hip_get_dims.argtypes = [C.POINTER(ProblemDesc), C.POINTER(Dims)]
    lib.corbo_hip_get_structure.argtypes = [C.POINTER(ProblemDesc), ip, ip]
    lib.corbo_hip_init_trajectory.argtypes = [C.POINTER(ProblemDesc), C.c_int, dp, dp, dp]
    lib.corbo_hip_create.argtypes = [C.POINTER(ProblemDesc), C.c_int, C.c_int, C.POINTER(H)]
    lib.corbo_hip_stage_function_kind.argtypes = [C.c_int]
    lib.corbo_hip_eval_stage_function.argtypes = [C.c_int, C.c_int, C.c_int, dp, dp, dp]
    lib.corbo_hip_create_routed.argtypes = [C.POINTER(ProblemDesc), C.c_int, C.c_int, C.c_uint32, C.POINTER(H)]
    lib.corbo_hip_factor_route.argtypes = [H]
    lib.corbo_hip_factor_route.restype = C.c_int
    lib.corbo_hip_destroy.argtypes = [H]
    lib.corbo_hip_destroy.restype = None
    lib.corbo_hip_set_instance_data.argtypes = [H, dp, dp, dp, dp]
    lib.corbo_hip_set_previous_control.argtypes = [H, dp, dp]
    lib.corbo_hip_restore_instance_data.argtypes = [H]
    lib.corbo_hip_warm_start.argtypes = [H, C.POINTER(C.c_double), C.c_int]
    lib.corbo_hip_get_first_control.argtypes = [H, C.POINTER(C.c_double)]
    lib.corbo_hip_plant_set_state.argtypes = [H, dp]
    lib.corbo_hip_plant_step.argtypes = [H, C.c_int, C.c_double, dp]
    lib.corbo_hip_plant_get_state.argtypes = [H, dp]
    lib.corbo_hip_plant_set_params.argtypes = [H, dp]
    lib.corbo_hip_set_instance_params.argtypes = [H, dp]
    lib.corbo_hip_warm_start_from_plant.argtypes = [H, C.c_int]
    lib.corbo_hip_closed_loop.argtypes = [H, C.POINTER(LmOpts), C.c_int, C.c_int, C.c_int, C.c_double, C.c_int, dp, dp, dp]
    lib.corbo_hip_set_profiling.argtypes = [H, C.c_int]
    lib.corbo_hip_solve.argtypes = [H, C.POINTER(LmOpts), C.c_int]
    lib.corbo_hip_solve_async.argtypes = [H, C.POINTER(LmOpts), C.c_int]
    lib.corbo_hip_synchronize.argtypes = [H]
    lib.corbo_hip_get_solution.argtypes = [H, dp, dp, ip]
    lib.corbo_hip_get_stats.argtypes = [H, C.POINTER(Stats)]
    lib.corbo_hip_eval.argtypes = [H, C.c_double, C.c_double, C.c_double, dp, dp]
    lib.corbo_hip_device_views.argtypes = [H, C.POINTER(dp), C.POINTER(dp), C.POINTER(C.c_void_p)]
    lib.corbo_hip_time_sweep.argtypes = [H, C.c_double, C.c_double, C.c_double, C.c_int, C.c_int, C.POINTER(C.c_float)]
    if not hasattr(lib, "corbo_hip_set_option"):   # an older build of the C-ABI loaded through CORBO_HIP_LIB (A/B measurements)
        _lib = lib
        return lib
    lib.corbo_hip_time_sweep_each.argtypes = [H, C.c_double, C.c_double, C.c_double, C.c_int, C.c_int, C.POINTER(C.c_float)]
    lib.corbo_hip_fetch_solution.argtypes = [H, C.POINTER(dp), C.POINTER(C.c_int32), C.POINTER(dp), C.POINTER(ip)]
    lib.corbo_hip_set_result_sink.argtypes = [H, C.c_int]
    lib.corbo_hip_set_option.argtypes = [H, C.c_char_p, C.c_int]
    lib.corbo_hip_device_count.argtypes = [C.POINTER(C.c_int)]
    lib.corbo_hip_shard_bounds.argtypes = [C.c_int, C.c_int, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int)]
    lib.corbo_hip_device_row_stride.argtypes = [H, C.POINTER(C.c_int32)]
    lib.corbo_hip_prepare_slots.argtypes = [H, C.c_int]
    lib.corbo_hip_get_dt.argtypes = [H, dp]
    lib.corbo_hip_resample_into.argtypes = [H, H, C.c_int, ip, ip]
    lib.corbo_hip_eval_dynamics.argtypes = [C.POINTER(ProblemDesc), C.c_int, dp, dp, dp]
    lib.corbo_hip_get_timing.argtypes = [H, C.POINTER(C.c_double), C.POINTER(C.c_int64), C.c_int]
    lib.corbo_hip_get_phase_cycles.argtypes = [H, C.POINTER(C.c_int64)]
    lib.corbo_hip_time_factor.argtypes = [H, C.c_int, C.POINTER(C.c_float), C.POINTER(C.c_longlong)]
    lib.corbo_hip_set_references.argtypes = [H, dp]
    lib.corbo_hip_set_reference_trajectory.argtypes = [H, dp, C.c_int, C.c_int]
    lib.corbo_hip_hessian_nnz.argtypes = [C.POINTER(ProblemDesc), C.c_int, ip]
    lib.corbo_hip_hessian_structure.argtypes = [C.POINTER(ProblemDesc), C.c_int, ip, ip, ip, ip, ip, ip]
    lib.corbo_hip_eval_hessians.argtypes = [H, C.c_int, C.c_double, dp, dp, dp, dp, dp]
    lib.corbo_hip_eval_hessians_views.argtypes = [H, C.c_int, C.c_double, dp, dp, C.c_int, C.POINTER(dp), C.POINTER(dp), C.POINTER(dp)]
    lib.corbo_hip_linear_form_structure.argtypes = [C.POINTER(ProblemDesc), ip, ip, ip, ip]
    lib.corbo_hip_eval_linear_form.argtypes = [H, dp, dp, dp]
    lib.corbo_hip_eval_objective_gradient.argtypes = [H, dp, dp]
    lib.corbo_hip_last_error.argtypes = []
    lib.corbo_hip_last_error.restype = C.c_char_p
    _lib = lib
    return lib
