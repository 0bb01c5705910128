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
    tot = st[38] - st[0]
    print("M=%d total %d ticks (100 MHz -> %.1f us)" % (M, tot, tot / 100.0))
    print(" load            %6d" % (st[1] - st[0]))
    for jb in range(8):
        b = 1 + jb * 4
        nxt = st[b + 4] if jb < 7 else st[33]
        print(" jb%d potf2+inv %5d store %5d panel %5d trailing %5d" % (jb, st[b+1]-st[b], st[b+2]-st[b+1], st[b+3]-st[b+2], nxt - st[b+3]))
    print(" log+invlevels   %6d" % (st[34] - st[33]))
    print(" W               %6d" % (st[35] - st[34]))
    print(" gram            %6d" % (st[36] - st[35]))
    print(" logdet/final    %6d" % (st[37] - st[36]))
    print(" store           %6d" % (st[38] - st[37]))
