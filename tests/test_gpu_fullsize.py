"""Full-size parity on the GPU (-m gpu): the HIP path through the C-ABI against
  * the genuine reference's result for EVERY instance of the headline batch (tests/golden/unicycle_seeded1024.npz) and the oracle run
    on the same 1024 instances,
  * cfg 5 at its full size (N = 200): 32 seeded instances against the oracle, instance 0 against the reference's own result,
  * cfg 5 at N = 40: 6 seeded instances against the reference.

Tolerances (see tests/test_oracle_fullsize.py for where they come from): headline per instance max(1e-5 [SURVEY 8d hard bound],
4 x the reference's own one-ulp reproducibility of that instance), bulk <= 1e-6; cfg 5: 3 x the reference's own one-ulp
reproducibility (~1e-4) on the whole vector, 1e-6 on its component in the stiff eigen-directions of J^T J, chi2 to 1e-8 relative.
"""
import json
import os

import numpy as np
import pytest
import scipy.sparse as sp

from conftest import GOLDEN, load_npz, stiff_part
from control_box_rst_amd import capi, problems
from control_box_rst_amd.solver import BatchedLevenbergMarquardt, get_structure

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module", autouse=True)
def _built():
    import torch
    assert torch.cuda.is_available(), "GPU tests need a GPU"
    import __graft_entry__ as g
    g.build()


def _solve(d, weights, x0, xf, iters=10):
    s = BatchedLevenbergMarquardt(d, len(x0))
    s.setIterations(iters)
    s.setPenaltyWeights(*weights)
    X0 = s.init_trajectory(x0, xf)
    s.set_instance_data(X0, xref=xf)
    s.solve()
    X, chi2, status = s.get_solution()
    return s, X0, X, chi2, status


def test_headline_batch_every_instance_vs_reference_and_oracle(oracle_mod):
    z = load_npz("unicycle_seeded1024")
    d = problems.unicycle_desc()
    B = 1024
    x0, xf = problems.unicycle_instances(B, seed=int(z["seed"]))
    s, X0, X, chi2, status = _solve(d, problems.UNICYCLE_WEIGHTS, x0, xf)
    ref, spread = z["vertex"][:, : s.dims.nv], z["ulp_spread"]
    bound = np.maximum(1e-5, 4.0 * spread)
    err = np.abs(X - ref).max(axis=1)
    assert (err <= bound).all(), (int((err > bound).sum()), int(err.argmax()), float(err.max()))
    assert np.median(err) <= 1e-6 and np.quantile(err, 0.95) <= 5e-6 and (err > 1e-5).sum() <= 8
    rel = np.abs(chi2 / z["chi2"] - 1)
    assert (rel <= np.maximum(2e-6, 2.0 * spread)).all() and (rel > 2e-6).sum() <= 8, (float(rel.max()), int((rel > 2e-6).sum()))
    # the oracle on the same 1024 instances (runs in about a second on the box's host)
    Xo, chi2o, so = oracle_mod.solve_batch(d, X0, xf, s.opts)
    erro = np.abs(X - Xo).max(axis=1)
    assert (erro <= bound).all(), (int((erro > bound).sum()), int(erro.argmax()), float(erro.max()))
    assert np.median(erro) <= 1e-6
    assert np.array_equal(status, so)
    # the result views in pinned host memory (kernel-written sink and the copy path) hold the same numbers
    s.set_result_sink(True)
    s.restore_instance_data()
    s.solve()
    Xs, cs, ss = s.fetch_solution()
    assert np.array_equal(Xs, X) and np.array_equal(cs, chi2) and np.array_equal(ss, status)
    s.set_result_sink(False)
    s.restore_instance_data()
    s.solve()
    Xc, cc, sc = s.fetch_solution()
    assert np.array_equal(Xc, X) and np.array_equal(cc, chi2) and np.array_equal(sc, status)


def _jac_at(s, d, b):
    _, jac = s.eval()
    rows, cols = get_structure(d)
    return sp.coo_matrix((jac[b], (rows, cols)), shape=(s.dims.m, s.dims.n)).tocsr()


def test_cfg5_n40_vs_reference_soft_directions():
    with open(os.path.join(GOLDEN, "quad_n40_seeded_ulp.json")) as f:
        g = json.load(f)
    d = problems.quad_desc(N=g["N"])
    x0 = np.array([i["x0"] for i in g["instances"]])
    xf = np.array([i["xf"] for i in g["instances"]])
    s, X0, X, chi2, status = _solve(d, g["weights"], x0, xf, iters=g["iters"])
    for b, inst in enumerate(g["instances"]):
        ref = np.array(inst["vertex"])[: s.dims.nv]
        spread = max(np.abs(np.array(v)[: s.dims.nv] - ref).max() for v in inst["vertex_ulp"])
        assert np.abs(X[b] - ref).max() <= 3.0 * spread, (b, np.abs(X[b] - ref).max(), spread)
        assert abs(chi2[b] / inst["chi2"] - 1) <= 1e-8
        stiff, jdx2 = stiff_part(_jac_at(s, d, b), (X[b] - ref)[d.nx:])
        assert np.abs(stiff).max() <= 1e-6, (b, np.abs(stiff).max())
        assert jdx2 <= 1e-10 * inst["chi2"]


def test_cfg5_full_size_vs_oracle_and_reference(oracle_mod):
    z = load_npz("quad_n200_seeded_ulp")
    d = problems.quad_desc()
    B = 32
    x0, xf = problems.quad_instances(B, seed=int(z["seed"]))
    assert np.array_equal(x0[0], z["x0"]) and np.array_equal(xf[0], z["xf"])
    s, X0, X, chi2, status = _solve(d, problems.QUAD_WEIGHTS, x0, xf)
    nv = s.dims.nv
    ref0 = z["vertex"][:nv]
    spread = float(np.abs(z["vertex_ulp"][:nv] - ref0).max())    # the reference against itself, x0 one ulp apart: ~1e-4
    assert 1e-5 < spread < 3e-4
    assert np.abs(X[0] - ref0).max() <= 3.0 * spread
    assert abs(chi2[0] / float(z["chi2"]) - 1) <= 1e-8
    Xo, chi2o, so = oracle_mod.solve_batch(d, X0, xf, s.opts)
    assert np.abs(X - Xo).max() <= 3.0 * spread, np.abs(X - Xo).max()
    assert np.abs(chi2 / chi2o - 1).max() <= 1e-8
    assert np.array_equal(status, so)
    _, jac = s.eval()
    rows, cols = get_structure(d)
    for b in range(B):
        J = sp.coo_matrix((jac[b], (rows, cols)), shape=(s.dims.m, s.dims.n)).tocsr()
        dp = (X[b] - Xo[b])[d.nx:]
        assert np.linalg.norm(J @ dp) ** 2 <= 1e-10 * chi2[b]
    for b in (0, 17):        # the eigen-decomposition of a 3184 x 3184 matrix takes a few seconds: two instances
        J = sp.coo_matrix((jac[b], (rows, cols)), shape=(s.dims.m, s.dims.n)).tocsr()
        for other in (Xo[b],) + ((ref0,) if b == 0 else ()):
            stiff, _ = stiff_part(J, (X[b] - other)[d.nx:])
            assert np.abs(stiff).max() <= 1e-6, (b, np.abs(stiff).max())
