#!/usr/bin/env python3
"""Latency of small batches through the host-pointer API (call site 1 of the boundary: one objective(alpha, rho),
marginaliseb.jl:133-141): the persistent launch (chain_max = 12, every group size forced onto it: chain_work_max lifted) against the two-launches-per-step path
(chain_max = 0).  The default policy picks the persistent launch for M <= 12 and M x (N/128)^2 <= 4096.
  python tools/latency.py [--sizes 512,2048] [--batches 1,2,4,8,12] [--reps 20]"""
import argparse
import sys
import time

import numpy as np

sys.path.insert(0, ".")
import gpcc_amd  # noqa: E402
from gpcc_amd import synthetic  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--sizes", default="192,512,2048")      # per band; two bands
ap.add_argument("--batches", default="1,2,4,8,12")
ap.add_argument("--reps", type=int, default=20)
ap.add_argument("--bands", type=int, default=2)
ap.add_argument("--precision", default="fp64", help="fp32: chain_max = 12 is then the fp64 twin's persistent launch (option fp32_chain), 0 the fp32 tiles per step")
args = ap.parse_args()
print("build:", gpcc_amd.build_info() if hasattr(gpcc_amd, "build_info") else "?")
for Nb in [int(v) for v in args.sizes.split(",")]:
    t, y, s, _ = synthetic.simulate_lightcurves([Nb] * args.bands, seed=1)
    alpha, rho = synthetic.default_hyperparameters(y)
    N = Nb * args.bands
    for cm in (0, 12):
        with gpcc_amd.Objective(t, y, s, "matern32", slots_per_stream=64, precision=args.precision) as obj:
            obj.set_option("chain_max", cm)
            obj.set_option("chain_work_max", 1 << 30)   # (every group up to chain_max: the default policy takes evaluations x (N/128)^2 <= 4096)
            for M in [int(v) for v in args.batches.split(",")]:
                d = np.concatenate([np.zeros((M, 1)), np.linspace(0, 20, M)[:, None] * np.ones((1, args.bands - 1))], 1)
                a = np.tile(alpha, (M, 1))
                r = np.full(M, rho)
                ll, info = obj.loglik_batch(d, a, r)
                ts = []
                for _ in range(args.reps):
                    t0 = time.perf_counter()
                    obj.loglik_batch(d, a, r)
                    ts.append(time.perf_counter() - t0)
                med = np.median(ts)
                print("N=%5d M=%2d %s %-28s: %7.3f ms per call (min %7.3f), %8.1f evals/s   loglik[0] %.12e" % (
                    N, M, args.precision, "persistent launch" if cm else "two launches per step", med * 1e3, min(ts) * 1e3, M / med, ll[0]), flush=True)
