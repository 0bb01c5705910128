"""Worker of tests/test_distributed_cpu.py: world_size-2 gloo run of the grid sharding
(gpcc_amd.distributed.sharded_loglik) with the CPU oracle injected as the evaluator -- the N>1
plumbing (partition, padded all_gather, reassembly, getprobabilities input) without a GPU."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch.distributed as dist  # noqa: E402

from gpcc_amd import fit, shard_bounds, sharded_grid_fit, sharded_loglik, synthetic  # noqa: E402
from oracle import oracle  # noqa: E402  (tests may use the oracle)


def main():
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    t, y, s, _ = synthetic.simulate_lightcurves([40, 30], seed=11, span=20.0)
    alpha, rho = synthetic.default_hyperparameters(y)
    for G in (37, 2, 1):   # uneven split, fewer points than ranks
        grid = np.linspace(0.0, 10.0, G)
        delays = np.stack([np.zeros(G), grid], 1)
        alphas = np.tile(alpha, (G, 1))
        rhos = np.full(G, rho)
        rhos[0] = -1.0 if G > 2 else rho            # one invalid point: info = -2 must travel too
        calls = []

        def evaluate(d, a, r):
            calls.append(len(r))
            return oracle.loglik_batch("matern32", t, y, s, d, a, r, True)

        ll, info = sharded_loglik(evaluate, delays, alphas, rhos)
        lo, hi = shard_bounds(G, world, rank)
        assert calls == ([hi - lo] if hi > lo else []), (calls, lo, hi)   # only this rank's block was evaluated
        ref, rinfo = oracle.loglik_batch("matern32", t, y, s, delays, alphas, rhos, True)
        assert np.array_equal(info, rinfo), (info, rinfo)
        ok = rinfo == 0
        assert np.array_equal(ll[ok], ref[ok])
        assert np.isnan(ll[~ok]).all()
    # the per-delay fit, dealt round-robin: every rank ends with the fit of ALL delays, equal to the unsharded fit
    class O:
        def loglik_batch(self, d, a, r):
            return oracle.loglik_batch("OU", t, y, s, d, a, r, True)

    for G in (7, 1):
        cand = np.stack([np.zeros(G), np.linspace(0.5, 6.0, G)], 1)
        seen = []

        def fit_block(c):
            seen.append(c.copy())
            r = fit.gpcc_grid(t, y, s, kernel="OU", candidatedelays=c, iterations=6, rhomax=20.0, objective=O())
            return r.loglikel, r.alpha, r.rho, np.zeros(len(c), dtype=np.int32)

        ll, al, rh, info = sharded_grid_fit(fit_block, cand)
        assert (len(seen) == 1 and np.array_equal(seen[0], cand[rank::world])) or (len(seen) == 0 and rank >= G)
        full = fit.gpcc_grid(t, y, s, kernel="OU", candidatedelays=cand, iterations=6, rhomax=20.0, objective=O())
        assert np.array_equal(ll, full.loglikel) and np.array_equal(al, full.alpha) and np.array_equal(rh, full.rho)
        assert ll.shape == (G,) and al.shape == (G, 2) and (info == 0).all()
    dist.barrier()
    with open(os.path.join(sys.argv[1], "ok_%d" % rank), "w") as f:
        f.write("ok")
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
