#!/usr/bin/env python3
"""Calibration data for the fp32 accuracy guard: raw fp32 results (guard off) against the fp64 device
path on the same problems, with the pivot-ratio statistics the diagonal kernel reports.  One row per evaluation:
  tag N L kernel mb  relerr  S=sum K_ii/d_i  mx=max K_ii/d_i  |loglik|
python tools/calibrate_fp32.py [minutes] [seed] [norefine] > profiles/r02/fp32_guard_calibration.log"""
import sys
import time

import numpy as np

sys.path.insert(0, ".")
import gpcc_amd as gp  # noqa: E402
from gpcc_amd import synthetic  # noqa: E402

minutes = float(sys.argv[1]) if len(sys.argv) > 1 else 3.0
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
knames = ["OU", "rbf", "matern32", "matern52"]
U = 2.0 ** -24
t_end = time.time() + 60 * minutes
rows = []


def run(tag, t, y, s, kname, mb, delays, alpha, rho):
    with gp.Objective(t, y, s, kname, marginalise_b=mb, precision="fp64") as o64:
        ref, i64 = o64.loglik_batch(delays, alpha, rho)
    with gp.Objective(t, y, s, kname, marginalise_b=mb, precision="fp32") as o32:
        o32.set_option("fp32_guard", 0)
        if len(sys.argv) > 3 and sys.argv[3] == "norefine":
            o32.set_option("fp32_refine", 0)
        ll, i32 = o32.loglik_batch(delays, alpha, rho)
        cond = o32.conditioning(len(rho))
    N = sum(len(a) for a in t)
    for i in range(len(rho)):
        if i64[i] == 0 and i32[i] == 0:
            rel = abs(ll[i] - ref[i]) / abs(ref[i])
            rows.append((tag, N, len(t), kname, int(mb), rel, cond[i, 0], cond[i, 1], abs(ref[i])))
            print("%s %d %d %s %d %.3e %.4e %.4e %.4e" % rows[-1], flush=True)
        elif i64[i] == 0:
            print("# fp32 failed where fp64 did not: %s N=%d %s info=%d" % (tag, N, kname, i32[i]), flush=True)


trial = 0
while time.time() < t_end:
    trial += 1
    mode = trial % 4
    kname = knames[int(rng.integers(0, 4))]
    mb = bool(rng.integers(0, 2))
    if mode == 0:      # soak-like small problems
        L = int(rng.integers(1, 7))
        lo = 2 if mb else 1
        Nl = [int(rng.integers(lo, 600)) if rng.random() < 0.8 else int(rng.integers(lo, 6)) for _ in range(L)]
        t = [rng.random(n) * rng.uniform(5, 80) for n in Nl]
        y = [rng.uniform(-5, 30) + rng.uniform(0.2, 3) * np.sin(0.2 * t[l] + l) + rng.standard_normal(Nl[l]) * 0.4 for l in range(L)]
        s = [rng.uniform(0.05, 1.0, n) for n in Nl]
        M = 24
        delays = rng.uniform(-5, 20, (M, L))
        alpha = 10.0 ** rng.uniform(-0.7, 0.7, (M, L))
        rho = 10.0 ** rng.uniform(-0.5, 1.5, M)
        run("soak", t, y, s, kname, mb, delays, alpha, rho)
    elif mode == 1:    # Nelder-Mead envelope on benchmark-like data
        Nl = [int(rng.integers(200, 700)) for _ in range(2)]
        t, y, s, _ = synthetic.simulate_lightcurves(Nl, seed=int(rng.integers(1, 1000)), gap_band=1, sigma=float(rng.choice([0.75, 0.3, 0.1])))
        M = 48
        delays = np.stack([np.zeros(M), rng.random(M) * 20], 1)
        alpha = 10.0 ** rng.uniform(-2, 2, (M, 2))
        rho = 10.0 ** rng.uniform(-1, np.log10(300), M)
        run("envelope", t, y, s, kname, mb, delays, alpha, rho)
    elif mode == 2:    # bigger, 1-3 bands
        L = int(rng.integers(1, 4))
        Nl = [int(rng.integers(300, 1400)) for _ in range(L)]
        t = [rng.random(n) * rng.uniform(20, 400) for n in Nl]
        y = [rng.uniform(-5, 30) + rng.uniform(0.2, 3) * np.sin(0.2 * t[l] + l) + rng.standard_normal(Nl[l]) * 0.4 for l in range(L)]
        s = [rng.uniform(0.05, 1.0, n) for n in Nl]
        M = 16
        delays = rng.uniform(-5, 20, (M, L))
        alpha = 10.0 ** rng.uniform(-1, 1.5, (M, L))
        rho = 10.0 ** rng.uniform(-0.5, 2, M)
        run("big", t, y, s, kname, mb, delays, alpha, rho)
    else:              # the benchmark's own regime at several sizes
        n = int(rng.choice([512, 1024, 2048]))
        t, y, s, _ = synthetic.simulate_lightcurves([n, n], seed=int(rng.integers(1, 1000)))
        a0, r0 = synthetic.default_hyperparameters(y)
        M = 8
        delays = np.stack([np.zeros(M), rng.random(M) * 20], 1)
        alpha = np.tile(a0, (M, 1)) * 10.0 ** rng.uniform(-0.3, 0.3, (M, 2))
        rho = r0 * 10.0 ** rng.uniform(-0.3, 0.3, M)
        run("bench", t, y, s, kname, mb, delays, alpha, rho)

r = np.array([(x[5], x[6], x[7], x[8], x[1]) for x in rows])
rel, S, mx, ll, N = r.T
print("# %d evaluations; raw fp32 relative error: median %.2e, p99 %.2e, max %.2e; above 1e-3: %d, above 2e-4: %d"
      % (len(rel), np.median(rel), np.quantile(rel, 0.99), rel.max(), (rel > 1e-3).sum(), (rel > 2e-4).sum()))
for name, est in (("u*S/|ll|", U * S / ll), ("u*sqrt(N)*mx/|ll|", U * np.sqrt(N) * mx / ll), ("u*sqrt(S*mx)/|ll|", U * np.sqrt(S * mx) / ll)):
    ratio = rel / est
    print("# estimator %-20s: relerr/est  median %.3f  p99 %.3f  max %.3f" % (name, np.median(ratio), np.quantile(ratio, 0.99), ratio.max()))
    for c in (1.0, 2.0, 4.0, 8.0):
        flagged = c * est > 2e-4
        missed = (~flagged) & (rel > 1e-3)
        worst_unflagged = rel[~flagged].max() if (~flagged).any() else 0.0
        print("#    c=%.0f: flagged %.2f %%, unflagged worst %.2e, unflagged above 1e-3: %d"
              % (c, 100.0 * flagged.mean(), worst_unflagged, missed.sum()))
