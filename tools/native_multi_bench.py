#!/usr/bin/env python3
"""The delay-grid sweep through ONE multi-device handle (gpcc_create_multi: worker threads + RCCL all-gather inside
libgpcc_hip.so; no torch, no torchrun) -- the single-process shape a Julia host uses.
    python tools/native_multi_bench.py 0,1,2,3,4,5,6,7 [grid_per_device] [steps]
On a one-GPU box `0` and `0,0` rehearse it (a repeated device gathers through host memory).  Weak scaling: 1024 delays
per listed device.  Prints one JSON line: whole-job evaluations per second, host pointers in and out."""
import json
import sys
import time

import numpy as np

sys.path.insert(0, ".")
import gpcc_amd  # noqa: E402
from gpcc_amd import synthetic  # noqa: E402

devs = [int(x) for x in (sys.argv[1] if len(sys.argv) > 1 else "0").split(",")]
G = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 3
t, y, s, _ = synthetic.simulate_lightcurves([2048, 2048], seed=1)
alpha, rho = synthetic.default_hyperparameters(y)
Gtot = G * len(devs)
delays = np.stack([np.zeros(Gtot), np.linspace(0.0, 20.0, Gtot)], 1)
alphas, rhos = np.tile(alpha, (Gtot, 1)), np.full(Gtot, rho)
with gpcc_amd.Objective(t, y, s, "matern32", devices=devs) as obj:
    obj.set_option("shared_prefix", 0)
    ll, info = obj.loglik_batch(delays, alphas, rhos)          # warm-up (workspaces, communicator)
    t0 = time.perf_counter()
    for _ in range(steps):
        ll, info = obj.loglik_batch(delays, alphas, rhos)
        p = gpcc_amd.getprobabilities(ll, device=devs[0])
    dt = time.perf_counter() - t0
    print(json.dumps({"metric": "delay-grid loglik evals/sec (N=4096, 2-band Matern-3/2), one process, multi-device handle",
                      "value": round(Gtot * steps / dt, 2), "unit": "evals/s", "devices": devs, "grid_total": Gtot, "steps": steps,
                      "ms_per_step": round(dt / steps * 1e3, 3), "gather_mode": {1: "rccl", 2: "host"}[obj.get_option("gather_mode")],
                      "info_nonzero": int((info != 0).sum()), "posterior_sum": float(p.sum())}))
