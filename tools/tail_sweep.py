import sys, time
import numpy as np
sys.path.insert(0, ".")
import gpcc_amd
from gpcc_amd import synthetic
for Nb in (2048, 1024):
    t, y, s, _ = synthetic.simulate_lightcurves([Nb, Nb], seed=1)
    alpha, rho = synthetic.default_hyperparameters(y)
    for M in (13, 16, 24, 32, 64):
        d = np.stack([np.zeros(M), np.linspace(0, 20, M)], 1); a = np.tile(alpha, (M, 1)); r = np.full(M, rho)
        out = []
        for mb, occ in ((160, 384), (400, 384), (400, 512), (400, 768), (700, 512), (700, 768), (1200, 768), (1200, 1536)):
            with gpcc_amd.Objective(t, y, s, "matern32", slots_per_stream=256) as obj:
                for k, v in (("shared_prefix", 0), ("right_looking_max", 0), ("fused_solve_min", 100000), ("hybrid_mall_mb", mb), ("hybrid_occ", occ)):
                    obj.set_option(k, v)
                obj.loglik_batch(d, a, r)
                ts = []
                for _ in range(5):
                    t0 = time.perf_counter(); obj.loglik_batch(d, a, r); ts.append(time.perf_counter() - t0)
                out.append("%dMB/%d %.2f" % (mb, occ, np.median(ts) * 1e3))
        print("N=%d M=%d: " % (2 * Nb, M) + " | ".join(out), flush=True)
