#!/usr/bin/env python3
"""Left-looking (fused update/solve) vs right-looking per-call latency for mid-size batches: where should right_looking_max sit?"""
import sys, time
import numpy as np
sys.path.insert(0, ".")
import gpcc_amd
from gpcc_amd import synthetic
for Nb in (512, 2048):
    t, y, s, _ = synthetic.simulate_lightcurves([Nb, Nb], seed=1)
    alpha, rho = synthetic.default_hyperparameters(y)
    with gpcc_amd.Objective(t, y, s, "matern32", slots_per_stream=64) as obj:
        obj.set_option("shared_prefix", 0)
        for M in (8, 12, 16, 20, 24, 32, 48, 64):
            d = np.stack([np.zeros(M), np.linspace(0, 20, M)], 1); a = np.tile(alpha, (M, 1)); r = np.full(M, rho)
            res = []
            for rlm in (0, 64):
                obj.set_option("right_looking_max", rlm)
                obj.loglik_batch(d, a, r)
                ts = []
                for _ in range(5):
                    t0 = time.perf_counter(); obj.loglik_batch(d, a, r); ts.append(time.perf_counter() - t0)
                res.append(np.median(ts) * 1e3)
            print("N=%d M=%2d: left-looking %.2f ms, right-looking %.2f ms" % (2 * Nb, M, res[0], res[1]))
