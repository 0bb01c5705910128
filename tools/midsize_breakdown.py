#!/usr/bin/env python3
"""Per-kernel time (HIP events around every launch) of one mid-size group on the left-looking three-kernel path, right-looking tail off / on."""
import sys
import numpy as np
sys.path.insert(0, ".")
import gpcc_amd
from gpcc_amd import synthetic
Nb = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
t, y, s, _ = synthetic.simulate_lightcurves([Nb, Nb], seed=1)
alpha, rho = synthetic.default_hyperparameters(y)
for M in (16, 32, 64):
    d = np.stack([np.zeros(M), np.linspace(0, 20, M)], 1); a = np.tile(alpha, (M, 1)); r = np.full(M, rho)
    for split in (0, 1):
        with gpcc_amd.Objective(t, y, s, "matern32", slots_per_stream=256) as obj:
            for k, v in (("shared_prefix", 0), ("right_looking_max", 0), ("fused_solve_min", 100000), ("hybrid_tail", split)):
                obj.set_option(k, v)
            obj.loglik_batch(d, a, r)
            obj.profile(True); obj.profile_reset()
            obj.loglik_batch(d, a, r)
            prof = obj.profile_get(); obj.profile(False)
        print("N=%d M=%d hybrid_tail=%d: " % (2 * Nb, M, split) + ", ".join("%s %d launches %.2f ms" % (k, v[0], v[1]) for k, v in prof.items() if v[0]))
