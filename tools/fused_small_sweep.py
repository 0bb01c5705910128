import sys, time
import numpy as np
sys.path.insert(0, ".")
import gpcc_amd
from gpcc_amd import synthetic
for Nb in (512, 2048):
    t, y, s, _ = synthetic.simulate_lightcurves([Nb, Nb], seed=1)
    alpha, rho = synthetic.default_hyperparameters(y)
    with gpcc_amd.Objective(t, y, s, "matern32", slots_per_stream=64) as obj:
        for M in (1, 2, 3, 4, 6, 8, 12, 16, 24):
            d = np.stack([np.zeros(M), np.linspace(0, 20, M)], 1); a = np.tile(alpha, (M, 1)); r = np.full(M, rho)
            res = []
            for fsm in (0, 64):
                obj.set_option("fused_small_max", fsm)
                obj.loglik_batch(d, a, r)
                ts = []
                for _ in range(5):
                    t0 = time.perf_counter(); obj.loglik_batch(d, a, r); ts.append(time.perf_counter() - t0)
                res.append(np.median(ts) * 1e3)
            print("N=%d M=%2d: unfused %.2f ms, fused %.2f ms" % (2 * Nb, M, res[0], res[1]))
