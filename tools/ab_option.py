#!/usr/bin/env python3
"""A/B of a handle option in ONE process (interleaved rounds, median): python tools/ab_option.py diag_skip 0 1"""
import sys
import time

import numpy as np
import torch

sys.path.insert(0, ".")
import gpcc_amd  # noqa: E402
from gpcc_amd import synthetic  # noqa: E402

key, vals = sys.argv[1], [int(v) for v in sys.argv[2:]]
rounds = 7
torch.cuda.init()
dev = torch.device("cuda", 0)
t, y, s, _ = synthetic.simulate_lightcurves([2048, 2048], seed=1)
alpha, rho = synthetic.default_hyperparameters(y)
G = 1024
delays = np.stack([np.zeros(G), np.linspace(0, 20, G)], 1)
obj = gpcc_amd.Objective(t, y, s, "matern32")
d_d = torch.as_tensor(delays, device=dev)
d_a = torch.as_tensor(np.tile(alpha, (G, 1)), device=dev)
d_r = torch.full((G,), float(rho), dtype=torch.float64, device=dev)
out = torch.empty(G, dtype=torch.float64, device=dev)
info = torch.empty(G, dtype=torch.int32, device=dev)
times = {v: [] for v in vals}
for r in range(rounds + 1):
    for v in vals:
        obj.set_option(key, v)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        obj.loglik_batch_device(d_d, d_a, d_r, out=out, info=info)
        torch.cuda.synchronize()
        if r:
            times[v].append(time.perf_counter() - t0)
for v in vals:
    a = np.array(times[v]) * 1e3
    print("%s=%d: median %.2f ms  min %.2f ms  (%.0f evals/s)" % (key, v, np.median(a), a.min(), G / np.median(a) * 1e3))
