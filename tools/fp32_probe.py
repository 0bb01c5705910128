#!/usr/bin/env python3
"""fp32 device path (guard off) vs fp64 device path vs a host emulation (LAPACK spotrf on the fp32-rounded K0) on a
few well- and ill-conditioned hyper-parameter sets: where does the fp32 error come from?"""
import sys
sys.path.insert(0, "."); sys.path.insert(0, "tests/golden")
import numpy as np, scipy.linalg as sla
import gpcc_amd as gp
from gpcc_amd import synthetic
from make_golden import delayed_covariance
kn = sys.argv[1] if len(sys.argv) > 1 else "matern52"
t, y, s, _ = synthetic.simulate_lightcurves([400, 426], seed=5, gap_band=1, sigma=0.1)
Y = np.concatenate(y); r = Y - np.concatenate([np.full(len(a), a.mean()) for a in y])
cases = [([0.02, 0.03], 100.0), ([0.02, 0.03], 1.0), ([1.0, 1.5], 3.5), ([10., 20.], 3.5), ([0.05, 0.01], 30.), ([0.3, 0.3], 300.0)]
u = 2.0 ** -24
for mb in (False, True):
    with gp.Objective(t, y, s, kn, marginalise_b=mb, precision="fp64") as o64, gp.Objective(t, y, s, kn, marginalise_b=mb, precision="fp32") as o32:
        o32.set_option("fp32_guard", 0)
        for rl in (0, 1):
            o32.set_option("fp32_refine", rl)
            for alpha, rho in cases:
                a, b = o64.loglik_batch([[0, 2.0]], [alpha], [rho])
                c, d = o32.loglik_batch([[0, 2.0]], [alpha], [rho])
                cond = o32.conditioning(1)[0]
                K0 = delayed_covariance(kn, alpha, [0, 2.0], rho, t) + np.diag(np.concatenate(s) ** 2)
                c64 = sla.cholesky(K0, lower=True); z = sla.solve_triangular(c64, r, lower=True); q64 = z @ z
                c32 = sla.cholesky(K0.astype(np.float32), lower=True)
                z = sla.solve_triangular(c32.astype(np.float64), r, lower=True); q32 = z @ z
                dl = 2 * np.log(np.diag(c32).astype(np.float64)).sum() - 2 * np.log(np.diag(c64)).sum()
                print("mb=%d refine=%d alpha %s rho %g: ll %.6e  device fp32-fp64 = %.3e (%.1f u*quad)   spotrf emulation (mb=0 model): %.3e   S/N %.2f mx %.2f"
                      % (mb, rl, alpha, rho, a[0], c[0] - a[0], (c[0] - a[0]) / (u * q64), -0.5 * (q32 - q64) - 0.5 * dl, cond[0] / 826, cond[1]))
