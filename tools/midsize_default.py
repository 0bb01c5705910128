#!/usr/bin/env python3
"""Mid-size groups through the default dispatch at N = 4096 / 2048 (host-pointer API, median of 7 calls), with one option toggled:
  python tools/midsize_default.py [--option look_ahead] [--sizes 2048,1024] [--batches 13,16,24,32,48,64,96]"""
import argparse
import sys
import time

import numpy as np

sys.path.insert(0, ".")
import gpcc_amd  # noqa: E402
from gpcc_amd import synthetic  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--option", default="look_ahead")
ap.add_argument("--values", default="1,0")
ap.add_argument("--sizes", default="2048,1024")
ap.add_argument("--batches", default="13,16,20,24,32,48,64,96")
ap.add_argument("--set", action="append", default=[], help="other options key=value")
args = ap.parse_args()
for Nb in [int(x) for x in args.sizes.split(",")]:
    t, y, s, _ = synthetic.simulate_lightcurves([Nb, Nb], seed=1)
    alpha, rho = synthetic.default_hyperparameters(y)
    with gpcc_amd.Objective(t, y, s, "matern32") as obj:
        obj.set_option("shared_prefix", 0)
        for kv in args.set:
            k, v = kv.split("=")
            obj.set_option(k, int(v))
        for M in [int(x) for x in args.batches.split(",")]:
            d = np.stack([np.zeros(M), np.linspace(0, 20, M)], 1); a = np.tile(alpha, (M, 1)); r = np.full(M, rho)
            line = []
            for v in [int(x) for x in args.values.split(",")]:
                obj.set_option(args.option, v)
                ll, info = obj.loglik_batch(d, a, r)
                assert (info == 0).all()
                ts = []
                for _ in range(7):
                    t0 = time.perf_counter(); obj.loglik_batch(d, a, r); ts.append(time.perf_counter() - t0)
                line.append("%s=%d %.2f ms (%.0f/s)" % (args.option, v, np.median(ts) * 1e3, M / np.median(ts)))
            print("N=%d M=%3d: " % (2 * Nb, M) + " | ".join(line), flush=True)
