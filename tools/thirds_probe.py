#!/usr/bin/env python3
"""Would more than two concurrent parts help a mid-size batch?  The same M evaluations as ONE group (default: two halves on two streams)
against S groups of M/S on S streams (streams = S, slots_per_stream = M/S).  python tools/thirds_probe.py [n_per_band]"""
import sys
import time

import numpy as np

sys.path.insert(0, ".")
import gpcc_amd  # noqa: E402
from gpcc_amd import synthetic  # noqa: E402

Nb = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
t, y, s, _ = synthetic.simulate_lightcurves([Nb, Nb], seed=1)
alpha, rho = synthetic.default_hyperparameters(y)
for M, variants in ((48, ((1, 256), (3, 16), (2, 24))), (64, ((1, 256), (4, 16), (2, 32))), (96, ((1, 256), (3, 32), (4, 24), (6, 16))),
                    (128, ((1, 256), (4, 32), (2, 64))), (32, ((1, 256), (2, 16)))):
    d = np.stack([np.zeros(M), np.linspace(0, 20, M)], 1); a = np.tile(alpha, (M, 1)); r = np.full(M, rho)
    line = []
    for S, cs in variants:
        with gpcc_amd.Objective(t, y, s, "matern32", streams=S, slots_per_stream=cs) as obj:
            obj.set_option("shared_prefix", 0)
            obj.loglik_batch(d, a, r)
            ts = []
            for _ in range(5):
                t0 = time.perf_counter(); obj.loglik_batch(d, a, r); ts.append(time.perf_counter() - t0)
            line.append("%d x %d: %.2f ms (%.0f/s)" % (S, cs, np.median(ts) * 1e3, M / np.median(ts)))
    print("N=%d M=%3d: " % (2 * Nb, M) + " | ".join(line), flush=True)
