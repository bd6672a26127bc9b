"""Device and oracle against the reference on the fuzz-seed goldens (tests/golden/fuzz_*.json), iterate by iterate.
    python tools/fuzz_seed_vs_reference.py   (GPU box)
"""
import glob
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import desc_for   # noqa: E402
from control_box_rst_amd import capi   # noqa: E402
from control_box_rst_amd.solver import BatchedLevenbergMarquardt   # noqa: E402
from oracle import oracle as O   # noqa: E402

O.build()
O.load()
for path in sorted(glob.glob(os.path.join(ROOT, "tests", "golden", "fuzz_*.json"))):
    g = json.load(open(path))
    d = desc_for(g)
    for a in g["after_iter"]:
        s = BatchedLevenbergMarquardt(d, 1)
        s.setIterations(a["k"])
        s.setPenaltyWeights(*g["weights"])
        start = np.array(g["vertex_init"])[None, : s.dims.nv]
        s.set_instance_data(start, xref=np.array(g["xf"])[None, :])
        s.solve()
        x, chi2, _ = s.get_solution()
        st = s.get_stats()
        p = O.OracleProblem(d)
        p.set_data(start[0], xref=np.array(g["xf"]))
        _, chi2o, tr = p.solve(capi.default_lm_opts(a["k"], *g["weights"]))
        ref = np.array(a["vertex"])[: s.dims.nv]
        print(os.path.basename(path), "k", a["k"], "device-ref %.3e" % np.abs(x[0] - ref).max(), "oracle-ref %.3e" % np.abs(p.x() - ref).max(),
              "chi2 ref %.10g dev %.10g oracle %.10g" % (a["chi2"], chi2[0], chi2o), "device accepted/rejected", st["accepted_steps"], st["rejected_steps"],
              "oracle inner passes", [t["inner_passes"] for t in tr])
