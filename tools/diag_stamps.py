"""Cycle stamps (s_memtime) of one gpcc_diag_factor workgroup -- needs a library built with the STAMP patch
(/tmp/stamp_patch3.py style instrumentation; not part of the product build)."""
import sys, ctypes; sys.path.insert(0, ".")
import numpy as np, torch
torch.cuda.init()
import gpcc_amd as gp
from gpcc_amd import synthetic, _capi
t, y, s, _ = synthetic.simulate_lightcurves([2048, 2048], seed=1)
alpha, rho = synthetic.default_hyperparameters(y)
for M in (1, 256):
    d = np.stack([np.zeros(M), np.linspace(0, 20, M)], 1)
    with gp.Objective(t, y, s, "matern32") as obj:
        obj.set_option("shared_prefix", 0)
        for _ in range(2):
            obj.loglik_batch(d, np.tile(alpha, (M, 1)), np.full(M, rho))
    buf = (ctypes.c_ulonglong * 64)()
    lib = _capi.load()
    lib.gpcc_debug_stamps.argtypes = [ctypes.c_void_p, ctypes.c_int]
    assert lib.gpcc_debug_stamps(buf, 64) == 0
    st = np.array(list(buf), dtype=np.int64)
    print("M=%d total %d cycles" % (M, st[23] - st[0]))
    print(" load            %6d" % (st[1] - st[0]))
    for jb in range(8):
        print(" jb%d  A(jb)|C,X(jb-1) %5d   B(jb) %5d" % (jb, st[2 + 2 * jb] - st[1 + 2 * jb], st[3 + 2 * jb] - st[2 + 2 * jb]))
    print(" X row 7         %6d" % (st[18] - st[17]))
    print(" (after loop)    %6d" % (st[19] - st[18]))
    print(" W store         %6d" % (st[20] - st[19]))
    print(" gram+logdet     %6d" % (st[21] - st[20]))
    print(" final           %6d" % (st[22] - st[21]))
    print(" store           %6d" % (st[23] - st[22]))
