"""Child process of tests/test_gpu_parity.py::test_two_processes_share_one_gpu: owns a handle on device 0, waits for its
sibling at a file barrier so that both are evaluating at the same time, writes its log-likelihoods to an .npz."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    rank, outdir = int(sys.argv[1]), sys.argv[2]
    import gpcc_amd
    from gpcc_amd import synthetic
    res = {}
    for tag, Nl in (("small", [60, 50]), ("tiles", [330, 310])):
        t, y, s, _ = synthetic.simulate_lightcurves(Nl, seed=11 + rank, span=20.0)
        alpha, rho = synthetic.default_hyperparameters(y)
        M = 48
        delays = np.stack([np.zeros(M), np.linspace(0.0, 10.0, M) + 0.1 * rank], 1)
        with gpcc_amd.Objective(t, y, s, "matern32", device=0) as obj:
            obj.loglik_batch(delays[:2], np.tile(alpha, (2, 1)), np.full(2, rho))          # code objects loaded, workspace up
            open(os.path.join(outdir, "ready_%s_%d" % (tag, rank)), "w").close()
            t0 = time.time()
            while not os.path.exists(os.path.join(outdir, "ready_%s_%d" % (tag, 1 - rank))):   # both processes on the GPU now
                if time.time() - t0 > 120:
                    raise SystemExit("sibling never arrived")
                time.sleep(0.005)
            lls = []
            for rep in range(20):                                                          # ~overlapping with the sibling's
                ll, info = obj.loglik_batch(delays, np.tile(alpha, (M, 1)), np.full(M, rho))
                assert (info == 0).all()
                lls.append(ll)
            assert all(np.array_equal(lls[0], x) for x in lls[1:])                          # deterministic under contention
            res[tag] = lls[0]
            res[tag + "_delays"] = delays
    np.savez(os.path.join(outdir, "result_%d.npz" % rank), **res)


if __name__ == "__main__":
    main()
