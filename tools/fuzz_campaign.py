"""Extended randomized parity campaign on a GPU box: the bodies of tests/test_gpu_fuzz.py over seeds the suite does not run.
    python tools/fuzz_campaign.py [first_seed] [count]      -> failing seeds with their assertion text (nothing = all passed)
"""
import os
import sys
import traceback

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import test_gpu_fuzz as F   # noqa: E402
import test_gpu_hessian as H   # noqa: E402
import test_gpu_extra_edges as X   # noqa: E402
from oracle import oracle as O   # noqa: E402

first = int(sys.argv[1]) if len(sys.argv) > 1 else 200
count = int(sys.argv[2]) if len(sys.argv) > 2 else 500
O.build()
O.load()
bad = []
ROUTES = []   # corbo_hip_factor_route of the 129 .. 256-grid-point cases (3 = block-tridiagonal route, 2 = band route: more list rounds than the BIG instantiation holds)
for name, fn, seeds in (("descriptor", F.test_random_descriptor_vs_oracle, range(first, first + count)),
                        ("quadrotor", F.test_random_quadrotor_descriptor_vs_oracle, range(first, first + count // 10)),
                        ("quadrotor, free dt", F.test_random_free_dt_quadrotor_descriptor_vs_oracle, range(first, first + count // 10)),
                        ("bounds+weights", F.test_per_instance_bounds_and_weight_adaptation_vs_oracle, range(first, first + count // 5)),
                        ("closed loop", F.test_random_closed_loop_call_vs_stepwise_and_oracle_plant, range(first, first + count // 5)),
                        ("hessian operators", H.test_random_descriptor_hessians_vs_oracle, range(first, first + count)),
                        ("counted iterations on / off", lambda o, seed: F.test_counted_converged_iterations_random_descriptors(seed), range(first, first + count // 3)),
                        ("dense weights (incl. long horizons)", F.test_random_dense_weights_vs_oracle, range(first, first + count // 2)),
                        ("hessian operators, partial terminal equality", H.test_random_descriptor_hessians_partial_terminal_equality, range(first, first + count // 2)),
                        # random combinations of the extra-edge kinds (incl. a user control function from seed 12 on) through the block-tridiagonal route (round 6)
                        ("extra edges, block-tridiagonal route", X.test_random_batches_vs_oracle, range(first, first + count // 2)),
                        ("extra edges + non-diagonal weights (band route)", X.test_random_batches_with_dense_weights_vs_oracle, range(first, first + count // 5)),
                        ("extra edges, 129 .. 256 grid points (BIG instantiation)", lambda o, seed: ROUTES.append(X.random_batch_129_to_256(o, seed)), range(first, first + count // 5))):
    n_bad = 0
    for seed in seeds:
        try:
            fn(O, seed)
        except Exception as e:   # noqa: BLE001
            n_bad += 1
            bad.append((name, seed, type(e).__name__, str(e)[:300]))
    print(f"{name}: {len(seeds)} seeds, {n_bad} failed", flush=True)
print(f"129 .. 256 grid points: {sum(1 for r in ROUTES if r == 3)} of {len(ROUTES)} cases through the block-tridiagonal route, {sum(1 for r in ROUTES if r == 2)} through the band route")
for b in bad:
    print("FAILED", b)
esc = F.ESCALATIONS
print(f"escalations: {sum(1 for e in esc if e['stage'] >= 1)} beyond the base tolerance, {sum(1 for e in esc if e['stage'] >= 2)} needed the 48-trial spread")
for e in esc:
    print("ESCALATED", e)
