#!/usr/bin/env python3
"""Bitwise check that a handle option does not change results: python tools/option_check.py <key> <v1> <v2> ..."""
import sys

import numpy as np

sys.path.insert(0, ".")
import gpcc_amd  # noqa: E402
from gpcc_amd import synthetic  # noqa: E402

key, vals = sys.argv[1], [int(a) for a in sys.argv[2:]]
t, y, s, _ = synthetic.simulate_lightcurves([640, 640], seed=2)
alpha, rho = synthetic.default_hyperparameters(y)
M = 40
d = np.stack([np.zeros(M), np.linspace(0, 20, M)], 1)
with gpcc_amd.Objective(t, y, s, "matern32", slots_per_stream=40) as obj:
    obj.set_option("right_looking_max", 0)
    ref, i0 = obj.loglik_batch(d, np.tile(alpha, (M, 1)), np.full(M, rho))
    for v in vals:
        obj.set_option(key, v)
        ll, i1 = obj.loglik_batch(d, np.tile(alpha, (M, 1)), np.full(M, rho))
        print(key, v, "identical:", np.array_equal(ll, ref), np.array_equal(i0, i1))
        assert np.array_equal(ll, ref)
print("OK")
