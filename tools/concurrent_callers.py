#!/usr/bin/env python3
"""The reference's own parallel shape on ONE GPU: P concurrent callers, each with its OWN handle, each issuing ONE objective(alpha, rho)
at a time (README.md:195-211, :258-287: `pmap` workers each run gpcc(...), whose Optim loop calls the objective sequentially,
marginaliseb.jl:145-153, :209-211).  Callers are host threads of one process (one handle per thread: the C ABI's rule) or separate
processes (at most 4 here: the GPU box admits 6 processes on the card).  Reports aggregate evaluations/s and the per-call latency.
  python tools/concurrent_callers.py [--sizes 55,512,2048] [--threads 1,2,4,8,16] [--procs 1,2,4] [--seconds 2]"""
import argparse
import os
import sys
import threading
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def make_problem(nb, bands=2):
    from gpcc_amd import synthetic
    t, y, s, _ = synthetic.simulate_lightcurves([nb] * bands, seed=1)
    alpha, rho = synthetic.default_hyperparameters(y)
    return t, y, s, alpha, rho


def caller(obj, alpha, rho, seconds, start_evt, out, idx):
    d = np.array([0.0, 1.0 + 0.1 * idx])
    obj(alpha, rho, d)              # (workspace, first launch)
    start_evt.wait()
    lat = []
    t_end = time.perf_counter() + seconds
    while time.perf_counter() < t_end:
        t0 = time.perf_counter()
        obj(alpha, rho, d)
        lat.append(time.perf_counter() - t0)
    out[idx] = lat


OPTS = []   # (key, value) applied to every handle (--opt key=value)


def run_threads(nb, P, seconds):
    import gpcc_amd
    t, y, s, alpha, rho = make_problem(nb)
    objs = [gpcc_amd.Objective(t, y, s, "matern32", slots_per_stream=16, streams=1) for _ in range(P)]
    for o in objs:
        for k, v in OPTS:
            o.set_option(k, v)
    out = [None] * P
    evt = threading.Event()
    th = [threading.Thread(target=caller, args=(objs[i], alpha, rho, seconds, evt, out, i)) for i in range(P)]
    for x in th:
        x.start()
    time.sleep(0.3)
    t0 = time.perf_counter()
    evt.set()
    for x in th:
        x.join()
    wall = time.perf_counter() - t0
    for o in objs:
        o.close()
    lat = np.concatenate([np.asarray(l) for l in out])
    return len(lat) / wall, np.median(lat) * 1e3, np.percentile(lat, 95) * 1e3


def proc_main(nb, seconds, idx, barrier_file, q, opts=()):
    import torch
    torch.cuda.init()
    import gpcc_amd
    t, y, s, alpha, rho = make_problem(nb)
    d = np.array([0.0, 1.0 + 0.1 * idx])
    with gpcc_amd.Objective(t, y, s, "matern32", slots_per_stream=16, streams=1) as obj:
        for k, v in opts:
            obj.set_option(k, v)
        obj(alpha, rho, d)
        open("%s.%d" % (barrier_file, idx), "w").close()
        while not os.path.exists(barrier_file + ".go"):
            time.sleep(0.001)
        lat = []
        t_end = time.perf_counter() + seconds
        while time.perf_counter() < t_end:
            t0 = time.perf_counter()
            obj(alpha, rho, d)
            lat.append(time.perf_counter() - t0)
    q.put(lat)


def run_procs(nb, P, seconds):
    import multiprocessing as mp
    import tempfile
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    bf = os.path.join(tempfile.mkdtemp(prefix="gpcc_cc_"), "b")
    ps = [ctx.Process(target=proc_main, args=(nb, seconds, i, bf, q, tuple(OPTS))) for i in range(P)]
    for p in ps:
        p.start()
    while not all(os.path.exists("%s.%d" % (bf, i)) for i in range(P)):
        time.sleep(0.01)
        if any(p.exitcode not in (None, 0) for p in ps):
            raise RuntimeError("a caller process died")
    t0 = time.perf_counter()
    open(bf + ".go", "w").close()
    lats = [q.get() for _ in range(P)]
    wall = time.perf_counter() - t0
    for p in ps:
        p.join()
    lat = np.concatenate([np.asarray(l) for l in lats])
    return len(lat) / wall, np.median(lat) * 1e3, np.percentile(lat, 95) * 1e3


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--sizes", default="55,512,2048")      # per band (two bands): N = 110, 1024, 4096
    ap.add_argument("--threads", default="1,2,4,8,16")
    ap.add_argument("--procs", default="1,2,4")
    ap.add_argument("--seconds", type=float, default=2.0)
    ap.add_argument("--opt", action="append", default=[], help="key=value set on every handle (e.g. chain_workers_max=58)")
    args = ap.parse_args()
    OPTS.extend((kv.split("=")[0], int(kv.split("=")[1])) for kv in args.opt)
    import torch
    torch.cuda.init()
    import gpcc_amd
    print("build:", gpcc_amd.build_info(), "| options:", dict(OPTS), "| GPU_MAX_HW_QUEUES =", os.environ.get("GPU_MAX_HW_QUEUES", "(default)"), flush=True)
    for nb in [int(v) for v in args.sizes.split(",")]:
        base = None
        for P in [int(v) for v in args.threads.split(",") if v]:
            rate, med, p95 = run_threads(nb, P, args.seconds)
            base = base or rate
            print("N=%5d  %2d threads  (one handle each, M = 1 per call): %9.0f evals/s aggregate = %5.2fx one caller | per call median %7.3f ms, p95 %7.3f ms" % (
                2 * nb, P, rate, rate / base, med, p95), flush=True)
        for P in [int(v) for v in args.procs.split(",") if v]:
            rate, med, p95 = run_procs(nb, P, args.seconds)
            print("N=%5d  %2d processes (one handle each, M = 1 per call): %9.0f evals/s aggregate = %5.2fx one thread | per call median %7.3f ms, p95 %7.3f ms" % (
                2 * nb, P, rate, rate / base, med, p95), flush=True)


if __name__ == "__main__":
    main()
