import sys, time
import numpy as np
sys.path.insert(0, ".")
import gpcc_amd
from gpcc_amd import synthetic
for N in (64, 96, 110, 128, 150, 176, 191):
    t, y, s, _ = synthetic.simulate_lightcurves([N - N // 2, N // 2], seed=1)
    alpha, rho = synthetic.default_hyperparameters(y)
    M = 16384
    d = np.stack([np.zeros(M), np.linspace(0, 10, M)], 1); a = np.tile(alpha, (M, 1)); r = np.full(M, rho)
    out = []
    for wide in (0, 10 ** 9):
        with gpcc_amd.Objective(t, y, s, "matern32") as obj:
            obj.set_option("small_wide_max", wide)
            obj.loglik_batch(d, a, r)
            ts = []
            for _ in range(5):
                t0 = time.perf_counter(); obj.loglik_batch(d, a, r); ts.append(time.perf_counter() - t0)
            out.append("%s %.3f ms (%.2f M/s)" % ("four-wave" if wide else "one-wave", np.median(ts) * 1e3, M / np.median(ts) / 1e6))
    print("N=%d M=%d: " % (N, M) + " | ".join(out), flush=True)
