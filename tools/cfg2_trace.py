#!/usr/bin/env python3
"""cfg2 (N = 2048, 256 delays) per-launch durations by step (run under rocprofv3 --kernel-trace; --parse <dir> prints them with
the TFLOP/s of each gpcc_update_solve launch: 256 (nt-k-1) jobs of k tile products + one panel solve each)."""
import glob, os, sqlite3, sys
import numpy as np
sys.path.insert(0, ".")
if len(sys.argv) > 2 and sys.argv[1] == "--parse":
    nt = int(sys.argv[3]) if len(sys.argv) > 3 else 16
    for f in glob.glob(os.path.join(sys.argv[2], "**", "*.db"), recursive=True):
        con = sqlite3.connect(f)
        rows = con.execute("select name, grid_x, workgroup_x, start, end from kernels order by start").fetchall()
        us = [(g // w, (e - s) / 1e3) for n, g, w, s, e in rows if "gpcc_update_solve" in n][-(nt - 1):]
        sd = [(e - s) / 1e3 for n, g, w, s, e in rows if "gpcc_syrk_diag" in n][-nt:]
        asm = [(e - s) / 1e3 for n, g, w, s, e in rows if "gpcc_assemble" in n][-1:]
        tot = 0.0
        for k, (wg, d) in enumerate(us):
            jobs = 256 * (nt - k - 1)
            fl = jobs * (2.0 * 128 ** 3 * k + 128 ** 3)
            tot += d
            print("k=%2d  %5d jobs  %7.1f us  %5.1f TFLOP/s   syrk_diag(k) %6.1f us" % (k, jobs, d, fl / d / 1e6, sd[k]))
        print("update_solve total %.2f ms, syrk_diag total %.2f ms, assembly %.2f ms" % (tot / 1e3, sum(sd) / 1e3, sum(asm) / 1e3))
    sys.exit(0)
import gpcc_amd
from gpcc_amd import synthetic
Nb = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
t, y, s, _ = synthetic.simulate_lightcurves([Nb, Nb], seed=1)
alpha, rho = synthetic.default_hyperparameters(y)
M = 256
d = np.stack([np.zeros(M), np.linspace(0, 20, M)], 1); a = np.tile(alpha, (M, 1)); r = np.full(M, rho)
with gpcc_amd.Objective(t, y, s, "matern32") as obj:
    obj.set_option("shared_prefix", 0)
    for _ in range(3):
        obj.loglik_batch(d, a, r)
