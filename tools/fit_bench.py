"""Grid points per second of the full per-delay fit (gpcc_grid_loglik: random candidates + lock-step Nelder-Mead),
SURVEY section 8(d): "where the optimiser runs, report grid points/s with the evals-per-point stated".
  python tools/fit_bench.py [--n-per-band 2048] [--grid 512] [--iterations 30]"""
import argparse
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n-per-band", type=int, default=2048)
    ap.add_argument("--grid", type=int, default=512)
    ap.add_argument("--iterations", type=int, default=30)
    ap.add_argument("--restarts", type=int, default=1)
    ap.add_argument("--kernel", default="matern32")
    ap.add_argument("--readme", action="store_true",
                    help="the reference's own documented sweeps instead (README.md:161, :195, :227: N = 110 / 150, 101 / 201 / 12321 "
                         "delays, iterations = 1000) -- tools/readme_bench.py")
    args, rest = ap.parse_known_args()
    if args.readme:
        import readme_bench
        sys.argv = [sys.argv[0]] + rest
        return readme_bench.main()
    import torch

    import gpcc_amd
    from gpcc_amd import synthetic
    torch.cuda.init()
    t, y, s, _ = synthetic.simulate_lightcurves([args.n_per_band] * 2, seed=1)
    grid = np.linspace(0.0, 20.0, args.grid)
    cand = np.stack([np.zeros(args.grid), grid], 1)
    with gpcc_amd.Objective(t, y, s, args.kernel) as obj:
        obj.grid_loglik(cand[:8], 2, rhomax=300.0)                       # warm-up (workspace, code objects)
        t0 = time.perf_counter()
        ll, alpha, rho, info, its, (f_calls, rounds) = obj.grid_loglik(cand, args.iterations, numberofrestarts=args.restarts,
                                                                       rhomax=300.0, seed=1)
        dt = time.perf_counter() - t0
    p = gpcc_amd.getprobabilities(ll)
    print(json.dumps({"metric": "fitted grid points/s (gpcc_grid_loglik)", "value": round(args.grid / dt, 2),
                      "N": 2 * args.n_per_band, "grid": args.grid, "iterations": args.iterations, "restarts": args.restarts,
                      "seconds": round(dt, 2), "objective_evaluations": f_calls, "evals_per_point": round(f_calls / args.grid, 1),
                      "evals_per_s": round(f_calls / dt, 1), "batched_rounds": rounds,
                      "mean_evals_per_round": round(f_calls / rounds, 1), "info_nonzero": int((info != 0).sum()),
                      "posterior_mode_delay": float(grid[int(np.argmax(p))]),
                      "median_iterations_done": float(np.median(its))}))


if __name__ == "__main__":
    main()
