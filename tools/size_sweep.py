#!/usr/bin/env python3
"""Evaluations per second (fixed hyper-parameters, batches of 4096 and of 128) across the size range where the small-N family
hands over to the tile kernels: N = 64 ... 1024."""
import sys, time
import numpy as np
sys.path.insert(0, ".")
import gpcc_amd
from gpcc_amd import synthetic
sizes = [int(x) for x in sys.argv[1].split(",")] if len(sys.argv) > 1 else [64, 110, 150, 176, 191, 192, 224, 256, 320, 384, 512, 768, 1024]
for N in sizes:
    t, y, s, _ = synthetic.simulate_lightcurves([N - N // 2, N // 2], seed=1)
    alpha, rho = synthetic.default_hyperparameters(y)
    out = []
    for M in (128, 4096):
        d = np.stack([np.zeros(M), np.linspace(0, 10, M)], 1); a = np.tile(alpha, (M, 1)); r = np.full(M, rho)
        with gpcc_amd.Objective(t, y, s, "matern32") as obj:
            obj.set_option("shared_prefix", 0)
            obj.loglik_batch(d, a, r)
            ts = []
            for _ in range(5):
                t0 = time.perf_counter(); obj.loglik_batch(d, a, r); ts.append(time.perf_counter() - t0)
            out.append("M=%d: %.3f ms, %.0f evals/s, %.2f TFLOP/s (N^3/3)" % (M, np.median(ts) * 1e3, M / np.median(ts), M / np.median(ts) * N ** 3 / 3 / 1e12))
    print("N=%4d  small=%d  " % (N, N <= 191) + " | ".join(out), flush=True)
