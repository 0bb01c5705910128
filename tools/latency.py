#!/usr/bin/env python3
"""Latency of small batches through the host-pointer API (call site 1 of the boundary: one objective(alpha, rho))."""
import sys
import time

import numpy as np

sys.path.insert(0, ".")
import gpcc_amd  # noqa: E402
from gpcc_amd import synthetic  # noqa: E402

for Nb in (512, 2048):
    t, y, s, _ = synthetic.simulate_lightcurves([Nb, Nb], seed=1)
    alpha, rho = synthetic.default_hyperparameters(y)
    for rl in (0, 64):
      with gpcc_amd.Objective(t, y, s, "matern32", slots_per_stream=64) as obj:
        obj.set_option("right_looking_max", rl)
        print("right_looking_max", rl)
        for M in (1, 4, 8, 16, 32, 64):
            d = np.stack([np.zeros(M), np.linspace(0, 20, M)], 1)
            a = np.tile(alpha, (M, 1)); r = np.full(M, rho)
            obj.loglik_batch(d, a, r)
            ts = []
            for _ in range(5):
                t0 = time.perf_counter(); obj.loglik_batch(d, a, r); ts.append(time.perf_counter() - t0)
            print("N=%d M=%2d: %.2f ms per call, %.1f evals/s" % (2 * Nb, M, np.median(ts) * 1e3, M / np.median(ts)))
