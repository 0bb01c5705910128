"""Worker of tests/test_distributed_cpu.py: world_size-2 gloo run of the grid sharding
(gpcc_amd.distributed.sharded_loglik) with the CPU oracle injected as the evaluator -- the N>1
plumbing (partition, padded all_gather, reassembly, getprobabilities input) without a GPU."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch.distributed as dist  # noqa: E402

from gpcc_amd import shard_bounds, sharded_loglik, synthetic  # noqa: E402
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
    dist.barrier()
    with open(os.path.join(sys.argv[1], "ok_%d" % rank), "w") as f:
        f.write("ok")
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
