"""A batch of time-optimal predictive controllers whose grids adapt their resolution per instance, resident on one MI355X.

Reference behaviour mirrored (per instance): ``PredictiveController::step`` (controllers/src/predictive_controller.cpp:46-80: K
``compute()`` calls per control step, ``new_run`` only for the first) on a ``FiniteDifferencesVariableGrid`` with
``setGridAdaptTimeBasedSingleStep`` / ``...AggressiveEstimate`` / ``...SimpleShrinkingHorizon``
(finite_differences_variable_grid.cpp:44-163), or on a ``MultipleShootingVariableGrid`` with its setters of the same names
(multiple_shooting_variable_grid.cpp:42-152; same rules except the aggressive estimate, which rounds dt / dt_ref before it multiplies;
ShootingGridBase::resampleTrajectory, shooting_grid_base.cpp:473-547, is the full-discretisation resampling on the same vertex layout).  Every ``compute()`` starts with the grid update
(full_discretization_grid_base.cpp:38-131): adaptGrid -- unless it is a new run without ``adapt_first_iter``, or the very first run --
then, on a new run, x_0 = measured state and fixed goal components = reference; then the solver.

N differs from instance to instance, a device handle has one N: the batch is kept as BUCKETS, one handle per N in use, each with
capacity for the whole batch.  After a solve the per-instance dt comes down (batch doubles), the reference's rule picks each instance's new
N on the host (a few comparisons per instance), and instances whose N changed are moved to the bucket of their new N by
``corbo_hip_resample_into`` (resampleTrajectory on the device, bit-identical to the oracle's restatement); holes in a bucket are closed
by moving its last instances down.  No trajectory crosses PCIe.
"""
from __future__ import annotations

import math
from typing import Callable, Dict, List

import numpy as np

from .capi import GRID_MS_VARIABLE, ProblemDesc
from .solver import BatchedLevenbergMarquardt

NO_ADAPT, SINGLE_STEP, AGGRESSIVE, SHRINK = 0, 1, 2, 3
AGGRESSIVE_SHOOTING = 4   # what AGGRESSIVE means on a MultipleShootingVariableGrid (picked by AdaptiveGridBatch from the descriptor's grid kind)


def adapt_grid_n(strategy: int, n: int, dt: float, dt_ref: float, hyst: float, n_min: int, n_max: int) -> int:
    """adaptGridTimeBasedSingleStep / AggressiveEstimate / SimpleShrinkingHorizon (finite_differences_variable_grid.cpp:101-163)."""
    if strategy == SINGLE_STEP:
        if dt > dt_ref * (1.0 + hyst) and n < n_max:
            return n + 1
        if dt < dt_ref * (1.0 - hyst) and n > n_min:
            return n - 1
        return n
    if strategy == AGGRESSIVE:
        if dt_ref * (1.0 - hyst) <= dt <= dt_ref * (1.0 + hyst):
            return n
        v = float(n) * (dt / dt_ref)
        new_n = int(math.floor(v + 0.5)) if v >= 0 else -int(math.floor(-v + 0.5))   # std::round: halves away from zero
        return min(max(new_n, n_min), n_max)
    if strategy == SHRINK:
        return n - 1 if n > n_min else n
    if strategy == AGGRESSIVE_SHOOTING:   # multiple_shooting_variable_grid.cpp:115-141: n * (int)round(dt / dt_ref) -- the ratio is rounded first
        if dt_ref * (1.0 - hyst) <= dt <= dt_ref * (1.0 + hyst):
            return n
        v = dt / dt_ref
        new_n = n * (int(math.floor(v + 0.5)) if v >= 0 else -int(math.floor(-v + 0.5)))
        return min(max(new_n, n_min), n_max)
    return n


class AdaptiveGridBatch:
    """`batch` time-optimal OCP instances of one family; make_desc(N) -> the family's descriptor with N grid points."""

    def __init__(self, make_desc: Callable[[int], ProblemDesc], batch: int, n_ref: int, *, strategy=SINGLE_STEP, n_min=2, n_max=1000,
                 hyst=0.1, adapt_first_iter=False, device=0):
        self.make_desc, self.batch, self.n_ref = make_desc, int(batch), int(n_ref)
        self.strategy, self.n_min, self.n_max, self.hyst, self.adapt_first = strategy, n_min, n_max, hyst, adapt_first_iter
        self.device = device
        d_ref = make_desc(n_ref)
        self.dt_ref = float(d_ref.dt_ref)
        if self.strategy == AGGRESSIVE and d_ref.grid == GRID_MS_VARIABLE:
            self.strategy = AGGRESSIVE_SHOOTING
        self.iterations = 10
        self.weights = (2.0, 2.0, 2.0)
        self.adaptation = (1.0, 1.0, 1.0, 500.0, 500.0, 500.0)
        self.buckets: Dict[int, BatchedLevenbergMarquardt] = {}
        self.ids: Dict[int, List[int]] = {}        # N -> global instance ids in slot order
        self.first_run = True
        self.n_of = np.full(self.batch, self.n_ref, dtype=int)
        self.moves = 0
        self._w = self.weights   # current (adapted) penalty weights: kept HERE, buckets come and go (levenberg_marquardt_sparse.cpp:83-86, 270-287)

    # -- the reference solver's setters ------------------------------------------------------------------------------------------
    def setIterations(self, iterations: int):
        self.iterations = int(iterations)
        for s in self.buckets.values():
            s.setIterations(iterations)

    def setPenaltyWeights(self, w_eq, w_ineq, w_b):
        self.weights = (w_eq, w_ineq, w_b)
        for s in self.buckets.values():
            s.setPenaltyWeights(*self.weights)

    def setWeightAdapation(self, *a):
        self.adaptation = tuple(a)
        for s in self.buckets.values():
            s.setWeightAdapation(*a)

    def _bucket(self, n: int) -> BatchedLevenbergMarquardt:
        if n not in self.buckets:
            s = BatchedLevenbergMarquardt(self.make_desc(n), self.batch, device=self.device)
            s.setIterations(self.iterations)
            s.setPenaltyWeights(*self.weights)
            s.setWeightAdapation(*self.adaptation)
            s.prepare_slots(0)
            self.buckets[n], self.ids[n] = s, []
        return self.buckets[n]

    # -- data ------------------------------------------------------------------------------------------------------------------------
    def initialize(self, x0, xf):
        """initializeSequences for every instance on the reference grid (N = n_ref)."""
        x0, xf = np.atleast_2d(x0), np.atleast_2d(xf)
        assert len(x0) == self.batch
        s = self._bucket(self.n_ref)
        s.set_instance_data(s.init_trajectory(x0, xf), xref=xf)
        s.prepare_slots(self.batch)
        self.ids[self.n_ref] = list(range(self.batch))
        self.n_of[:] = self.n_ref
        self.first_run = True
        self._xf = np.array(xf, dtype=float)

    def _adapt(self):
        """adaptGrid for every instance, then move what changed."""
        if self.strategy == NO_ADAPT:
            return
        plan = {}   # n_src -> list of (slot, n_new)
        for n, ids in list(self.ids.items()):
            if not ids:
                continue
            dts = self.buckets[n].get_dt(len(ids))
            for slot, dt in enumerate(dts):
                n_new = adapt_grid_n(self.strategy, n, float(dt), self.dt_ref, self.hyst, self.n_min, self.n_max)
                if n_new != n:
                    plan.setdefault(n, []).append((slot, n_new))
        # every destination bucket exists (and its descriptor is accepted by the device) BEFORE anything moves: a refused horizon
        # must not leave the bookkeeping half way through a plan
        for moves in plan.values():
            for _, n_new in moves:
                self._bucket(n_new)
        for n, moves in plan.items():
            src, ids = self.buckets[n], self.ids[n]
            by_dst: Dict[int, List[int]] = {}
            for slot, n_new in moves:
                by_dst.setdefault(n_new, []).append(slot)
            for n_new, slots in by_dst.items():
                dst = self._bucket(n_new)
                base = len(self.ids[n_new])
                src.resample_into(dst, slots, list(range(base, base + len(slots))))
                for slot in slots:
                    gid = ids[slot]
                    self.ids[n_new].append(gid)
                    self.n_of[gid] = n_new
                dst.prepare_slots(len(self.ids[n_new]))
                self.moves += len(slots)
            # close the holes: the last surviving instances move down (same N: a plain copy)
            gone = sorted(slot for slot, _ in moves)
            gone_set = set(gone)
            keep_tail = [i for i in range(len(ids) - 1, -1, -1) if i not in gone_set]
            holes = [h for h in gone if h < len(ids) - len(gone)]
            srcs = keep_tail[: len(holes)]
            if holes:
                src.resample_into(src, srcs, holes)
                for hslot, sslot in zip(holes, srcs):
                    ids[hslot] = ids[sslot]
            del ids[len(ids) - len(gone):]
            src.prepare_slots(len(ids))

    def compute(self, x0_new=None, new_run=True):
        """One compute() of every instance: grid update (adaptGrid, then on a new run x_0 / fixed goal components), then the solve."""
        if not self.first_run and (not new_run or self.adapt_first):
            self._adapt()
        if new_run and not self.first_run:
            assert x0_new is not None
            x0_new = np.atleast_2d(x0_new)
            for n, ids in self.ids.items():
                if ids:
                    self.buckets[n].warm_start(np.pad(x0_new[ids], ((0, self.batch - len(ids)), (0, 0))), shift=False)
        if new_run:
            self._w = self.weights
        else:
            f, m = self.adaptation[:3], self.adaptation[3:]
            self._w = tuple(min(w * fi, mi) for w, fi, mi in zip(self._w, f, m))
        for n, ids in self.ids.items():
            if ids:
                s = self.buckets[n]
                s.setPenaltyWeights(*self._w)      # stated explicitly: new_run = True makes the handle take them as they are
                s.solve(new_run=True)
        self.first_run = False

    def step(self, x0_new, ocp_iterations=1):
        """PredictiveController::step: `ocp_iterations` compute() calls, new_run only for the first."""
        for it in range(ocp_iterations):
            self.compute(x0_new, new_run=(it == 0))

    # -- results ---------------------------------------------------------------------------------------------------------------------
    def trajectories(self) -> List[np.ndarray]:
        """Vertex vector [x_0 u_0 | ... | x_f | dt] of every instance (lengths differ with N)."""
        out: List[np.ndarray] = [None] * self.batch   # type: ignore
        for n, ids in self.ids.items():
            if not ids:
                continue
            X, _, _ = self.buckets[n].get_solution()
            for slot, gid in enumerate(ids):
                out[gid] = X[slot].copy()
        return out

    def first_controls(self) -> np.ndarray:
        d = self.make_desc(self.n_ref)
        u = np.zeros((self.batch, d.nu))
        for n, ids in self.ids.items():
            if ids:
                u[ids] = self.buckets[n].get_first_control()[: len(ids)]
        return u

    def grid_sizes(self) -> np.ndarray:
        return self.n_of.copy()

    def close(self):
        for s in self.buckets.values():
            s.close()
        self.buckets.clear()
