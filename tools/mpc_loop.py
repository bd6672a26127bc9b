"""Closed-loop (moving-horizon) rate of a resident batch of unicycle OCPs: per MPC step
    corbo_hip_warm_start (shifting warm start + new measured state: batch x nx doubles up)  ->  corbo_hip_solve(new_run)
    ->  corbo_hip_get_first_control (batch x nu doubles down)  ->  plant step on the host (RK4 of the unicycle over dt + disturbance).
The trajectories never leave HBM.
With "device" as 4th argument the plants live on the device too (corbo_hip_plant_step -> corbo_hip_warm_start_from_plant -> solve):
per step only the disturbance goes up (or nothing).
"call": the same loop as one corbo_hip_closed_loop call (everything enqueued, one synchronisation at the end).
    python tools/mpc_loop.py [batch] [steps] [iterations] [host|device|call]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from control_box_rst_amd import problems  # noqa: E402
from control_box_rst_amd.solver import BatchedLevenbergMarquardt  # noqa: E402


def plant(x, u, dt):
    def f(x):
        return np.stack([u[:, 0] * np.cos(x[:, 2]), u[:, 0] * np.sin(x[:, 2]), u[:, 1]], axis=1)
    k1 = f(x); k2 = f(x + 0.5 * dt * k1); k3 = f(x + 0.5 * dt * k2); k4 = f(x + dt * k3)
    return x + dt / 6.0 * (k1 + 2 * k2 + 2 * k3 + k4)


B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
STEPS = int(sys.argv[2]) if len(sys.argv) > 2 else 50
ITERS = int(sys.argv[3]) if len(sys.argv) > 3 else 10
PLANT = sys.argv[4] if len(sys.argv) > 4 else "host"
d = problems.unicycle_desc()
x0, xf = problems.unicycle_instances(B)
s = BatchedLevenbergMarquardt(d, B)
s.setIterations(ITERS)
s.setPenaltyWeights(*problems.UNICYCLE_WEIGHTS)
s.set_instance_data(s.init_trajectory(x0, xf), xref=xf)
s.solve(new_run=True)
x = x0.copy()
rng = np.random.default_rng(1)
t_ws = t_solve = t_u = 0.0
dist0 = np.linalg.norm(x[:, :2] - xf[:, :2], axis=1).mean()
if PLANT == "call":
    from control_box_rst_amd import capi  # noqa: E402
    s.plant_set_state(x0)
    noise = 1e-3 * rng.normal(size=(STEPS,) + x.shape)
    s.closed_loop(2, integrator=capi.INTEGRATOR_RK4, disturbance=noise[:2], log=False)   # warm-up (scratch buffers)
    t_all = time.perf_counter()
    xs, us = s.closed_loop(STEPS, integrator=capi.INTEGRATOR_RK4, disturbance=noise)
    wall = time.perf_counter() - t_all
    dist1 = np.linalg.norm(xs[-1][:, :2] - xf[:, :2], axis=1).mean()
    print(f"batch={B} N={d.N} iterations={ITERS}, one corbo_hip_closed_loop call of {STEPS} steps: {STEPS / wall:.1f} closed-loop steps/s of the whole "
          f"batch = {B * STEPS / wall / 1e3:.1f} k plant-steps/s ({wall / STEPS * 1e3:.3f} ms per step, on the device {s.get_stats()['solve_ms'] / STEPS:.3f} ms); "
          f"mean distance to goal {dist0:.3f} -> {dist1:.3f}")
    sys.exit(0)
if PLANT == "device":
    from control_box_rst_amd import capi  # noqa: E402
    s.plant_set_state(x0)
    t_p = t_ws = t_solve = 0.0
    t_all = time.perf_counter()
    for k in range(STEPS):
        t0 = time.perf_counter()
        s.plant_step(integrator=capi.INTEGRATOR_RK4, disturbance=1e-3 * rng.normal(size=x.shape))
        t1 = time.perf_counter()
        s.warm_start_from_plant(shift=True)
        t2 = time.perf_counter()
        s.solve(new_run=True)
        t3 = time.perf_counter()
        t_p += t1 - t0; t_ws += t2 - t1; t_solve += t3 - t2
    wall = time.perf_counter() - t_all
    x = s.plant_get_state()
    dist1 = np.linalg.norm(x[:, :2] - xf[:, :2], axis=1).mean()
    print(f"batch={B} N={d.N} iterations={ITERS}, plants on the device: {STEPS / wall:.1f} closed-loop steps/s of the whole batch = "
          f"{B * STEPS / wall / 1e3:.1f} k plant-steps/s (per step: plant step incl. drawing the disturbance {t_p / STEPS * 1e3:.3f} ms, warm start "
          f"{t_ws / STEPS * 1e3:.3f} ms, solve {t_solve / STEPS * 1e3:.3f} ms (last one on the device: {s.get_stats()['solve_ms']:.3f} ms)); "
          f"mean distance to goal {dist0:.3f} -> {dist1:.3f}")
    sys.exit(0)
t_all = time.perf_counter()
for k in range(STEPS):
    t0 = time.perf_counter()
    u0 = s.get_first_control()
    t1 = time.perf_counter()
    x = plant(x, np.clip(u0, -1, 1), d.dt_ref) + 1e-3 * rng.normal(size=x.shape)
    t2 = time.perf_counter()
    s.warm_start(x, shift=True)
    t3 = time.perf_counter()
    s.solve(new_run=True)
    t4 = time.perf_counter()
    t_u += t1 - t0; t_ws += t3 - t2; t_solve += t4 - t3
wall = time.perf_counter() - t_all
dist1 = np.linalg.norm(x[:, :2] - xf[:, :2], axis=1).mean()
print(f"batch={B} N={d.N} iterations={ITERS}: {STEPS / wall:.1f} closed-loop steps/s of the whole batch = {B * STEPS / wall / 1e3:.1f} k plant-steps/s "
      f"(per step: u_0 read-back {t_u / STEPS * 1e3:.3f} ms, warm start {t_ws / STEPS * 1e3:.3f} ms, solve {t_solve / STEPS * 1e3:.3f} ms (last one on the device: {s.get_stats()['solve_ms']:.3f} ms), "
      f"host plant + loop {(wall - t_u - t_ws - t_solve) / STEPS * 1e3:.3f} ms); mean distance to goal {dist0:.3f} -> {dist1:.3f}")
