"""Full-size parity on the CPU side: the oracle against the genuine reference's result for EVERY instance of the headline batch
(tests/golden/unicycle_seeded1024.npz: 1024 final trajectories of the compiled reference + the reference's OWN reproducibility per
instance under a one-ulp change of x0, oracle/gen_golden.py fullsize), and the cfg 5 tolerance demonstrated on the reference itself.

Why a per-instance bound: central differences with delta = 1e-9 amplify the last bits of every edge evaluation to ~1e-7 in the
Jacobian, and a handful of instances sit where that noise decides how far a damped step goes.  For those the
REFERENCE does not reproduce its own trajectory to 1e-5 when x0 moves by one ulp (ulp_spread up to 5e-5; median 2.4e-7).  A
checker cannot ask more of the build than the reference delivers against itself: the bound is max(5e-6, 4 x ulp_spread).
"""
import json
import os

import numpy as np
import pytest
import scipy.sparse as sp

from conftest import GOLDEN, load_npz, stiff_part
from control_box_rst_amd import capi, problems

X_TOL = 5e-6


def test_oracle_vs_reference_all_1024_headline_instances(oracle_mod):
    z = load_npz("unicycle_seeded1024")
    d = problems.unicycle_desc()
    B = 1024
    x0, xf = problems.unicycle_instances(B, seed=int(z["seed"]))
    p = oracle_mod.OracleProblem(d)
    X0 = np.stack([p.init_trajectory(x0[b], xf[b]) for b in range(B)])
    opts = capi.default_lm_opts(10, *problems.UNICYCLE_WEIGHTS)
    X, chi2, status = oracle_mod.solve_batch(d, X0, xf, opts)
    ref, spread = z["vertex"][:, : p.dims.nv], z["ulp_spread"]
    err = np.abs(X - ref).max(axis=1)
    bound = np.maximum(X_TOL, 4.0 * spread)
    assert (err <= bound).all(), (int((err > bound).sum()), int(err.argmax()), float(err.max()))
    assert np.median(err) <= 5e-7 and (err > X_TOL).sum() <= 16          # 1e-6 target met by the bulk; a handful of noise-limited instances
    rel = np.abs(chi2 / z["chi2"] - 1)
    assert (rel <= np.maximum(2e-6, 0.5 * spread)).all(), float(rel.max())
    assert (status <= capi.SOLVER_EARLY_TERMINATED).all()


def test_reference_reproducibility_of_the_headline_batch_is_recorded():
    """The fixture's own statistics (what the bound above rests on): the bulk of the batch reproduces to < 1e-6, a few instances do not
    reproduce to 1e-5 -- in the reference itself."""
    s = load_npz("unicycle_seeded1024")["ulp_spread"]
    assert np.median(s) < 1e-6 and np.quantile(s, 0.95) < 5e-6
    assert (s > 1e-5).sum() >= 1 and s.max() < 1e-4


def test_cfg5_difference_lies_in_the_soft_directions_reference_against_itself_and_oracle(oracle_mod):
    """cfg 5 (quadrotor): the reference run twice with x0 one ulp apart differs by 3e-5 .. 8e-5 in the trajectory -- all of it in the
    eigen-directions of J^T J with eigenvalue <= 0.2 (thrust / rate / torque components, cost weights 0.01 .. 0.1): the component in
    the stiff directions is < 1e-6, |J dx|^2 < 1e-10 chi2.  The oracle differs from the reference by the same kind of vector."""
    with open(os.path.join(GOLDEN, "quad_n40_seeded_ulp.json")) as f:
        g = json.load(f)
    d = problems.quad_desc(N=g["N"])
    opts = capi.default_lm_opts(g["iters"], *g["weights"])
    p = oracle_mod.OracleProblem(d)
    nv = p.dims.nv
    rows, cols = p.structure()
    for inst in g["instances"]:
        x0, xf, ref = np.array(inst["x0"]), np.array(inst["xf"]), np.array(inst["vertex"])[:nv]
        q = oracle_mod.OracleProblem(d)
        q.set_data(ref, xref=xf)
        _, jac = q.eval(*g["weights"])
        J = sp.coo_matrix((jac, (rows, cols)), shape=(p.dims.m, p.dims.n)).tocsr()
        spread = max(np.abs(np.array(v)[:nv] - ref).max() for v in inst["vertex_ulp"])
        assert 1e-5 < spread < 2e-4          # the reference misses SURVEY 8d's 1e-5 against ITSELF on this family
        Xo, chi2o, _ = oracle_mod.solve_batch(d, p.init_trajectory(x0, xf)[None, :], xf[None, :], opts)
        for other in [Xo[0]] + [np.array(v)[:nv] for v in inst["vertex_ulp"]]:
            dp = (other - ref)[d.nx:]          # parameter layout = vertex layout without the fixed x_0
            stiff, jdx2 = stiff_part(J, dp)
            assert np.abs(stiff).max() <= 1e-6
            assert jdx2 <= 1e-10 * inst["chi2"]
        assert np.abs(Xo[0] - ref).max() <= 3.0 * spread
        assert abs(chi2o[0] / inst["chi2"] - 1) <= 1e-8
