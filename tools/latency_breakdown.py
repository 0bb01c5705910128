#!/usr/bin/env python3
"""Per-kernel time of small batches (HIP events around every launch, serialised): where one objective(alpha, rho) goes."""
import sys
import time

import numpy as np

sys.path.insert(0, ".")
import torch  # noqa: E402

torch.cuda.init()
import gpcc_amd  # noqa: E402
from gpcc_amd import synthetic  # noqa: E402

for Nb in (512, 2048):
    t, y, s, _ = synthetic.simulate_lightcurves([Nb, Nb], seed=1)
    alpha, rho = synthetic.default_hyperparameters(y)
    with gpcc_amd.Objective(t, y, s, "matern32") as obj:
        for M in (1, 8, 24):
            d = np.stack([np.zeros(M), np.linspace(0, 20, M)], 1)
            a = np.tile(alpha, (M, 1)); r = np.full(M, rho)
            obj.loglik_batch(d, a, r)
            t0 = time.perf_counter(); obj.loglik_batch(d, a, r); wall = time.perf_counter() - t0
            obj.profile(True); obj.profile_reset()
            obj.loglik_batch(d, a, r)
            prof = obj.profile_get(); obj.profile(False)
            tot = sum(v[1] for v in prof.values())
            print("N=%d M=%2d wall %.2f ms | kernels %.2f ms: " % (2 * Nb, M, wall * 1e3, tot) +
                  ", ".join("%s %d x %.1f us" % (k, v[0], v[1] / max(v[0], 1) * 1e3) for k, v in prof.items()))
