#!/usr/bin/env python3
"""Mid-size batches (16-128 evaluations) as ONE group vs split into S concurrent groups on S streams: does the update of one
sub-group hide the diagonal-step / panel-solve chain of another?   python tools/midsize_split.py [--sizes 2048] [--batches 16,32,64]"""
import argparse
import sys
import time

import numpy as np

sys.path.insert(0, ".")
import gpcc_amd  # noqa: E402
from gpcc_amd import synthetic  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--sizes", default="2048,1024")
ap.add_argument("--batches", default="16,24,32,48,64,96,128")
ap.add_argument("--splits", default="1,2")
ap.add_argument("--option", action="append", default=[], help="name=value, applied to every handle")
args = ap.parse_args()
for Nb in [int(x) for x in args.sizes.split(",")]:
    t, y, s, _ = synthetic.simulate_lightcurves([Nb, Nb], seed=1)
    alpha, rho = synthetic.default_hyperparameters(y)
    for M in [int(x) for x in args.batches.split(",")]:
        d = np.stack([np.zeros(M), np.linspace(0, 20, M)], 1); a = np.tile(alpha, (M, 1)); r = np.full(M, rho)
        line, ref = [], None
        for S in [int(x) for x in args.splits.split(",")]:
            cs = (M + S - 1) // max(S, 1)
            # S = 2: the library's own split (option "split_min": two halves on two streams, same slots); S > 2: S groups on S streams
            with gpcc_amd.Objective(t, y, s, "matern32", streams=(S if S > 2 else 1), slots_per_stream=(cs if S > 2 else 256)) as obj:
                obj.set_option("shared_prefix", 0)
                if S != 0:      # S = 0: the library's defaults
                    obj.set_option("split_min", 2 if S == 2 else 0)
                    obj.set_option("split_nt_min", 1)
                    obj.set_option("split_small", 0)
                for kv in args.option:
                    k, v = kv.split("=")
                    obj.set_option(k, int(v))
                ll, info = obj.loglik_batch(d, a, r)
                assert (info == 0).all()
                if ref is None:
                    ref = ll
                err = float(np.max(np.abs(ll - ref) / np.abs(ref)))
                ts = []
                for _ in range(7):
                    t0 = time.perf_counter(); obj.loglik_batch(d, a, r); ts.append(time.perf_counter() - t0)
                line.append(("default: %.2f ms (%.0f/s, %.0e)" % (np.median(ts) * 1e3, M / np.median(ts), err)) if S == 0 else "%d x %d: %.2f ms (%.0f/s, %.0e)" % (S, cs, np.median(ts) * 1e3, M / np.median(ts), err))
        print("N=%d M=%3d: " % (2 * Nb, M) + " | ".join(line), flush=True)
