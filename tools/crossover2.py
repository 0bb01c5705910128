#!/usr/bin/env python3
"""Left-looking groups: fused update/solve (2 launches per step) vs the three-kernel path, by group size."""
import sys, time
import numpy as np
sys.path.insert(0, ".")
import gpcc_amd
from gpcc_amd import synthetic
for Nb in (512, 2048):
    t, y, s, _ = synthetic.simulate_lightcurves([Nb, Nb], seed=1)
    alpha, rho = synthetic.default_hyperparameters(y)
    with gpcc_amd.Objective(t, y, s, "matern32", slots_per_stream=256) as obj:
        obj.set_option("shared_prefix", 0)
        obj.set_option("right_looking_max", 0)
        obj.set_option("fused_solve_min", 1)      # compare the two paths at every size (default: fused from 112 evaluations on)
        for M in (25, 32, 48, 64, 96, 128, 160, 192, 256):
            d = np.stack([np.zeros(M), np.linspace(0, 20, M)], 1); a = np.tile(alpha, (M, 1)); r = np.full(M, rho)
            res = []
            for fs in (0, 1):
                obj.set_option("fused_solve", fs)
                obj.loglik_batch(d, a, r)
                ts = []
                for _ in range(4):
                    t0 = time.perf_counter(); obj.loglik_batch(d, a, r); ts.append(time.perf_counter() - t0)
                res.append(np.median(ts) * 1e3)
            print("N=%d M=%3d: three-kernel %.2f ms, fused %.2f ms" % (2 * Nb, M, res[0], res[1]))
