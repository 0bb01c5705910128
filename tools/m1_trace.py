#!/usr/bin/env python3
"""One objective(alpha, rho) at N = 4096 (M = 1), a few times: run under rocprofv3 --kernel-trace to see kernel durations and gaps."""
import sys
import time

import numpy as np

sys.path.insert(0, ".")
import gpcc_amd
from gpcc_amd import synthetic

Nb = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
t, y, s, _ = synthetic.simulate_lightcurves([Nb, Nb], seed=1)
alpha, rho = synthetic.default_hyperparameters(y)
with gpcc_amd.Objective(t, y, s, "matern32") as obj:
    for i in range(4):
        t0 = time.perf_counter()
        ll, info = obj.loglik_batch([[0.0, 2.0 + i]], [alpha], [rho])
        print("call %d: %.3f ms  ll %.6f" % (i, (time.perf_counter() - t0) * 1e3, ll[0]))
