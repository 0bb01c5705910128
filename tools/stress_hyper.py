#!/usr/bin/env python3
"""Accuracy envelope: device objective (fp64 and fp32) vs the CPU oracle over the hyper-parameter ranges a
Nelder-Mead run visits (alpha 1e-2..1e2, rho 0.1..300 as in README.md:172 rhomax = 300)."""
import sys

import numpy as np

sys.path.insert(0, ".")
import gpcc_amd  # noqa: E402
from gpcc_amd import synthetic  # noqa: E402
from oracle import oracle  # noqa: E402

rng = np.random.default_rng(11)
t, y, s, _ = synthetic.simulate_lightcurves([330, 300], seed=3, gap_band=1)
M = 400
delays = np.stack([np.zeros(M), rng.random(M) * 20], 1)
alpha = 10.0 ** rng.uniform(-2, 2, (M, 2))
rho = 10.0 ** rng.uniform(-1, np.log10(300), M)
ref, rinfo = oracle.loglik_batch("matern32", t, y, s, delays, alpha, rho, True, nthreads=32)
for prec in ("fp64", "fp32"):
    with gpcc_amd.Objective(t, y, s, "matern32", precision=prec) as obj:
        ll, info = obj.loglik_batch(delays, alpha, rho)
    ok = (rinfo == 0) & (info == 0)
    rel = np.abs(ll[ok] - ref[ok]) / np.abs(ref[ok])
    amax = alpha.max(axis=1)
    print("%s: oracle ok %d, device ok %d, both %d; rel err median %.2e  p99 %.2e  max %.2e" %
          (prec, (rinfo == 0).sum(), (info == 0).sum(), ok.sum(), np.median(rel), np.quantile(rel, 0.99), rel.max()))
    for lo, hi in ((0, 1), (1, 10), (10, 100.1)):
        m = ok & (amax >= lo) & (amax < hi)
        if m.any():
            r = np.abs(ll[m] - ref[m]) / np.abs(ref[m])
            print("   max alpha in [%g, %g): n=%d  max rel err %.2e" % (lo, hi, m.sum(), r.max()))
    bad = (rinfo == 0) & (info != 0)
    if bad.any():
        print("   device failed where the oracle did not: %d cases, alpha max there %.1f..%.1f" % (bad.sum(), amax[bad].min(), amax[bad].max()))
