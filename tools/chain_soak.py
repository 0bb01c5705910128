#!/usr/bin/env python3
"""Randomised soak of the persistent few-evaluation launch: random sizes (2-3 bands, N = 384 .. ~2300), group sizes 1 .. 32, kernels, hyper-parameters
and options (column blocks forced on / off, helper and quarter-job thresholds) -- every result against the launch-per-step path of the same handle
(chain_max = 0) and repeated once for bits.   python tools/chain_soak.py [--cases 60] [--seed 1] [--nmax 2300]"""
import argparse
import sys
import time

import numpy as np

sys.path.insert(0, ".")
import gpcc_amd  # noqa: E402
from gpcc_amd import synthetic  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--cases", type=int, default=60)
ap.add_argument("--seed", type=int, default=1)
ap.add_argument("--nmax", type=int, default=2300)
args = ap.parse_args()
rng = np.random.default_rng(args.seed)
worst = 0.0
t0 = time.time()
print("build:", gpcc_amd.build_info())
for case in range(args.cases):
    L = int(rng.integers(2, 4))
    N = int(rng.integers(384, args.nmax))
    while True:   # (bands of at least 40 observations)
        cuts = np.sort(rng.choice(np.arange(40, N - 40), L - 1, replace=False))
        Nl = [int(x) for x in np.diff(np.concatenate([[0], cuts, [N]]))]
        if min(Nl) >= 40:
            break
    kname = ["OU", "rbf", "matern32", "matern52"][int(rng.integers(0, 4))]
    mb = bool(rng.integers(0, 2))
    t, y, s, _ = synthetic.simulate_lightcurves(Nl, seed=int(rng.integers(1, 1000)))
    alpha, rho = synthetic.default_hyperparameters(y)
    M = int(rng.integers(1, 33))
    d = np.concatenate([np.zeros((M, 1)), rng.random((M, L - 1)) * 15], 1)
    a = np.tile(alpha, (M, 1)) * (0.6 + 0.8 * rng.random((M, L)))
    r = rho * (0.5 + 1.5 * rng.random(M))
    opts = {"chain_work_max": 1 << 30, "chain_wide_work_max": 1 << 30, "chain_batch_min": int(rng.choice([0, 40000])),
            "chain_batch": int(rng.choice([1, 2, 4, 8])), "chain_helpers_max": int(rng.choice([0, 6, 16])), "chain_quarters_max": int(rng.choice([0, 2, 16])),
            "chain_workers_max": int(rng.choice([0, 0, 3, 40]))}
    with gpcc_amd.Objective(t, y, s, kname, marginalise_b=mb, slots_per_stream=32, streams=1) as obj:
        for k, v in opts.items():
            obj.set_option(k, v)
        before = obj.get_option("chain_count")
        ll, info = obj.loglik_batch(d, a, r)
        took = obj.get_option("chain_count") - before
        ll2, info2 = obj.loglik_batch(d, a, r)
        obj.set_option("chain_max", 0)
        ref, rinfo = obj.loglik_batch(d, a, r)
    ok = rinfo == 0
    rel = float(np.max(np.abs(ll[ok] - ref[ok]) / np.abs(ref[ok]))) if ok.any() else 0.0
    worst = max(worst, rel)
    good = took == M and np.array_equal(info, rinfo) and np.array_equal(ll, ll2, equal_nan=True) and np.array_equal(info, info2) and rel <= 1e-10
    print("case %3d: N=%4d %s bands=%d mb=%d M=%2d %s | took %2d, bad pivots %d, vs launch-per-step %.1e %s" % (
        case, N, kname, L, mb, M, {k: v for k, v in opts.items() if k not in ("chain_work_max", "chain_wide_work_max")}, took, int((rinfo != 0).sum()), rel,
        "ok" if good else "FAILED"), flush=True)
    if not good:
        sys.exit(1)
print("%d cases, worst relative difference to the launch-per-step path %.2e, %.0f s" % (args.cases, worst, time.time() - t0))
