"""CPU tests of the plant side of the closed loop: the oracle's restatement of SimulatedPlant::control
(plants/src/simulated_plant.cpp:97-160: hold the first control over dt, integrate with explicit Euler / RK4, apply the state
disturbance) against closed loops of the genuine reference -- its StructuredOptimalControlProblem driven by its own SimulatedPlant
(tests/golden/loop_*.json, generator oracle/gen_golden.py -> oracle/_ref/ref_driver loop).

The interval the reference's plant integrates over is not the caller's dt but (t + dt) - t as its time-stamped control buffer rounds
it (systems/src/time_value_buffer.cpp:68-73; 0.10000000000000003 at t = 0.2): the fixture records it per step as "plant_dt" and
the checker passes that value."""
import numpy as np
import pytest

from conftest import desc_for, load_golden
from control_box_rst_amd import capi

LOOPS = ["loop_unicycle_rk4", "loop_unicycle_euler_noshift", "loop_vdp_euler", "loop_int3_rk4", "loop_quad_rk4", "loop_pquad_rk4", "loop_pendulum_rk4", "loop_duffing_euler", "loop_cartpole_rk4", "loop_lin32_rk4"]
POLYNOMIAL = {"loop_vdp_euler", "loop_int3_rk4", "loop_duffing_euler", "loop_lin32_rk4"}   # dynamics without sin / cos: nothing depends on the host's libm


def integrator_of(g):
    return capi.INTEGRATOR_RK4 if g["integrator"] == "rk4" else capi.INTEGRATOR_EULER


@pytest.mark.parametrize("name", LOOPS)
def test_plant_step_alone_is_exact(oracle_mod, name):
    """From the reference's own solution and measured state of a step, the plant step reproduces the reference's next plant state:
    same operations in the same order (bit for bit where no libm call is involved, else to the last ulps of the host's sin / cos)."""
    g = load_golden(name)
    d = desc_for(g)
    p = oracle_mod.OracleProblem(d)
    for k, st in enumerate(g["steps"]):
        p.set_data(np.array(st["vertex"])[: p.dims.nv], xref=np.array(g["xf"]))
        x = p.plant_step(st["x0"], integrator_of(g), st["plant_dt"], st["disturbance"])
        ref = np.array(st["plant_after"])
        if name in POLYNOMIAL:
            assert np.array_equal(x, ref), (name, k)
        else:
            assert np.abs(x - ref).max() <= 4e-16 * max(1.0, np.abs(ref).max()), (name, k)
        if k + 1 < len(g["steps"]):   # the loop is closed: the next measured state is this plant state
            assert np.array_equal(ref, np.array(g["steps"][k + 1]["x0"]))


@pytest.mark.parametrize("name", LOOPS)
def test_closed_loop_vs_reference(oracle_mod, name):
    """The whole loop on the oracle's own iterates: plant.output -> grid update -> solve -> plant.control, every step compared."""
    g = load_golden(name)
    d = desc_for(g)
    p = oracle_mod.OracleProblem(d)
    nv = p.dims.nv
    xf = np.array(g["xf"])
    x = np.array(g["steps"][0]["x0"])
    tol = 3e-4 if "quad" in name else 5e-6   # quadrotor: nearly flat directions, see tests/test_oracle_golden.py
    for k, st in enumerate(g["steps"]):
        if k == 0:
            p.set_data(p.init_trajectory(x, xf), xref=xf)    # first compute(): the grid initialises its sequences
        else:
            p.warm_start(x, shift=bool(g["shift"]))
        status, chi2, _ = p.solve(capi.default_lm_opts(g["iters"], *g["weights"]), new_run=True)
        assert np.abs(p.x() - np.array(st["vertex"])[:nv]).max() <= tol, (name, k)
        assert abs(chi2 - st["chi2"]) <= 2e-6 * abs(st["chi2"]), (name, k)
        x = p.plant_step(x, integrator_of(g), st["plant_dt"], st["disturbance"])
        assert np.abs(x - np.array(st["plant_after"])).max() <= tol, (name, k)
