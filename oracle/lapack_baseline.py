#!/usr/bin/env python3
"""LAPACK-class CPU baseline of the hot path: TEST / MEASUREMENT INFRASTRUCTURE ONLY.

What the reference does per evaluation on a CPU (src/gpccfixdelay_marginaliseb.jl:133-141): a scalar-loop assembly of
K = delayedCovariance + Sobs + B (single-threaded Julia, delayedCovariance.jl:23-31) followed by OpenBLAS dpotrf +
a triangular solve inside MvNormal/logpdf.  Restated here as: oracle.model_matrix (the C restatement's scalar loops) +
scipy.linalg.lapack.dpotrf / dtrtrs -- scipy ships the same OpenBLAS family Julia links.  kind = "port+LAPACK": the
reference itself cannot run (no Julia).  Two shapes, as the README parallelises (README.md:181-211):
  blas : ONE evaluation at a time, BLAS using P threads            (julia -t 1, BLAS.set_num_threads(P))
  pmap : P worker processes, ONE BLAS thread each, one evaluation per worker at a time   (pmap over the delay grid)
Runs as its own process (python -m oracle.lapack_baseline ...), never inside a process that has touched the GPU.
Prints one JSON object."""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

_W = {}


def _problem(n_per_band, bands, seed):
    from gpcc_amd import synthetic
    t, y, s, _ = synthetic.simulate_lightcurves([n_per_band] * bands, seed=seed)
    alpha, rho = synthetic.default_hyperparameters(y)
    return t, y, s, alpha, rho


def loglik_lapack(kernel, t, y, s, delays, alpha, rho, marginalise_b=True):
    """One evaluation: C-restatement assembly + LAPACK Cholesky / solve -> (loglik, info)."""
    from scipy.linalg import lapack

    from oracle import oracle
    K, resid = oracle.model_matrix(kernel, t, y, s, delays, alpha, rho, marginalise_b)
    c, info = lapack.dpotrf(K, lower=1, overwrite_a=1, clean=0)
    if info != 0:
        return float("nan"), int(info)
    z, info2 = lapack.dtrtrs(c, resid, lower=1, trans=0)
    N = len(resid)
    return float(-0.5 * (N * np.log(2 * np.pi) + 2.0 * np.sum(np.log(np.diag(c)))) - 0.5 * (z @ z)), 0


def _init_worker(n_per_band, bands, seed, kernel):
    from threadpoolctl import threadpool_limits
    _W["limit"] = threadpool_limits(limits=1)
    _W["prob"] = _problem(n_per_band, bands, seed)
    _W["kernel"] = kernel


def _work(delay_row):
    t, y, s, alpha, rho = _W["prob"]
    return loglik_lapack(_W["kernel"], t, y, s, np.asarray(delay_row), alpha, rho)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n-per-band", type=int, default=2048)
    ap.add_argument("--bands", type=int, default=2)
    ap.add_argument("--kernel", default="matern32")
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--delays", required=True, help="JSON list of delay vectors to evaluate (the sample)")
    ap.add_argument("--workers", type=int, default=0, help="P (0: every core of the affinity mask)")
    ap.add_argument("--blas-evals", type=int, default=3)
    ap.add_argument("--evals-per-worker", type=int, default=6)
    args = ap.parse_args()
    from threadpoolctl import threadpool_limits

    from oracle import oracle
    oracle.build()
    delays = [list(map(float, d)) for d in json.loads(args.delays)]
    cores = len(os.sched_getaffinity(0))
    quota = None   # a container's CPU share (cgroup v2): the affinity mask of a GPU box shows every core of the host
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            quota = max(1, int(round(float(q) / float(per))))
    except (OSError, ValueError):
        pass
    t, y, s, alpha, rho = _problem(args.n_per_band, args.bands, args.seed)
    out = {"cores_available": cores, "cgroup_cpu_quota": quota}

    # shape "blas": one evaluation at a time, BLAS on P threads
    P = args.workers or min(cores, quota or 16)
    with threadpool_limits(limits=P):
        loglik_lapack(args.kernel, t, y, s, np.asarray(delays[0]), alpha, rho)   # warm-up (page-in, thread pool)
        n = min(args.blas_evals, len(delays))
        t0 = time.perf_counter()
        first = [loglik_lapack(args.kernel, t, y, s, np.asarray(d), alpha, rho) for d in delays[:n]]
        dt = time.perf_counter() - t0
    out["blas"] = {"threads": P, "evals": n, "seconds": round(dt, 3), "evals_per_s": round(n / dt, 3)}

    # shape "pmap": P single-threaded workers; P = the CPU share (cgroup quota, else 16: a 1-GPU box's share) and, where the
    # mask shows more cores, a second run with up to 64 workers and a third of the evaluations each (bounded time)
    import multiprocessing as mp
    best = None
    for Pw in sorted({P, P if args.workers else min(cores, 64)}):
        want = Pw * (args.evals_per_worker if Pw == P else max(1, args.evals_per_worker // 3))
        sample = (delays * (-(-want // len(delays))))[:want]
        with mp.get_context("fork").Pool(Pw, initializer=_init_worker,
                                         initargs=(args.n_per_band, args.bands, args.seed, args.kernel)) as pool:
            pool.map(_work, sample[:Pw])      # warm-up: every worker builds its problem and pages in
            t0 = time.perf_counter()
            res = pool.map(_work, sample, chunksize=1)
            dt = time.perf_counter() - t0
        rec = {"workers": Pw, "evals": len(sample), "seconds": round(dt, 3), "evals_per_s": round(len(sample) / dt, 3)}
        out.setdefault("pmap_runs", []).append(rec)
        if best is None or rec["evals_per_s"] > best["evals_per_s"]:
            best = rec
            out["loglik"] = [r[0] for r in res[:min(len(delays), len(res))]]
            out["info"] = [r[1] for r in res[:min(len(delays), len(res))]]
    out["pmap"] = best
    out["blas_loglik"] = [r[0] for r in first]
    print(json.dumps(out))


if __name__ == "__main__":
    main()
