#!/usr/bin/env python3
"""Mid-size groups (8-128 evaluations) at N = 1024 / 2048 / 4096: evaluations per second of the paths a group can take --
right-looking, left-looking three-kernel without and with the right-looking tail, left-looking fused -- and of the default dispatch.
  python tools/midsize.py [--sizes 2048,1024,512] [--batches 8,16,32,64]"""
import argparse
import sys
import time

import numpy as np

sys.path.insert(0, ".")
import gpcc_amd  # noqa: E402
from gpcc_amd import synthetic  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--sizes", default="2048,1024")
ap.add_argument("--batches", default="8,12,16,24,32,48,64,96,128")
ap.add_argument("--target", type=int, default=1024)
args = ap.parse_args()
CONFIGS = [
    ("default", {}),
    ("right", {"right_looking_max": 4096, "fused_small_max": 0}),
    ("left3", {"right_looking_max": 0, "fused_solve_min": 100000, "hybrid_tail": 0}),
    ("left3+tail", {"right_looking_max": 0, "fused_solve_min": 100000, "hybrid_tail": 1}),
    ("fused", {"right_looking_max": 0, "fused_solve_min": 1}),
]
for Nb in [int(x) for x in args.sizes.split(",")]:
    t, y, s, _ = synthetic.simulate_lightcurves([Nb, Nb], seed=1)
    alpha, rho = synthetic.default_hyperparameters(y)
    for M in [int(x) for x in args.batches.split(",")]:
        d = np.stack([np.zeros(M), np.linspace(0, 20, M)], 1); a = np.tile(alpha, (M, 1)); r = np.full(M, rho)
        line, ref = [], None
        for name, opts in CONFIGS:
            with gpcc_amd.Objective(t, y, s, "matern32", slots_per_stream=256) as obj:
                obj.set_option("shared_prefix", 0)

                for k, v in opts.items():
                    obj.set_option(k, v)
                ll, info = obj.loglik_batch(d, a, r)
                assert (info == 0).all()
                if ref is None:
                    ref = ll
                err = float(np.max(np.abs(ll - ref) / np.abs(ref)))
                ts = []
                for _ in range(5):
                    t0 = time.perf_counter(); obj.loglik_batch(d, a, r); ts.append(time.perf_counter() - t0)
                line.append("%s %.2f ms (%.0f/s, %.0e)" % (name, np.median(ts) * 1e3, M / np.median(ts), err))
        print("N=%d M=%3d: " % (2 * Nb, M) + " | ".join(line), flush=True)
