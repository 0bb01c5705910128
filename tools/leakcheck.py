#!/usr/bin/env python3
"""Creates / uses / destroys handles in a loop and watches free device memory (hipMemGetInfo via torch)."""
import sys

import numpy as np
import torch

sys.path.insert(0, ".")
torch.cuda.init()
import gpcc_amd  # noqa: E402
from gpcc_amd import synthetic  # noqa: E402

t, y, s, _ = synthetic.simulate_lightcurves([300, 280], seed=1)
alpha, rho = synthetic.default_hyperparameters(y)
M = 30
d = np.stack([np.zeros(M), np.linspace(0, 10, M)], 1)
free0 = None
for it in range(60):
    with gpcc_amd.Objective(t, y, s, "matern32", precision="fp32" if it % 2 else "fp64", slots_per_stream=32) as obj:
        obj.loglik_batch(d, np.tile(alpha, (M, 1)), np.full(M, rho))
        if it % 5 == 0:
            obj.predict([0.0, 2.0], alpha, rho, [np.linspace(0, 50, 40)] * 2)
            obj.posterior_offsets([0.0, 2.0], alpha, rho)
            obj.model_matrix([0.0, 2.0], alpha, rho)
    gpcc_amd.getprobabilities(np.zeros(10))
    gpcc_amd.delayedCovariance("OU", [1.0, 1.0], [0.0, 1.0], 2.0, [t[0][:10], t[1][:10]])
    torch.cuda.synchronize()
    free, total = torch.cuda.mem_get_info()
    if it == 4:
        free0 = free
    if it % 10 == 0 or it == 59:
        print("iteration %2d: free %.1f MiB" % (it, free / 2**20))
assert free0 is not None and abs(free - free0) < 64 * 2**20, (free0, free)
print("no leak: free memory moved by %.1f MiB over 55 handle lifetimes" % ((free - free0) / 2**20))
