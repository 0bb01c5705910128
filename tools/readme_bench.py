"""The reference's OWN documented workloads (README.md:156-179, :181-211, :214-287; simulatedata.jl:119) on the device:
  sweep A: 2 bands, N = 60 + 50, candidate delays 0:0.2:20 (101), gpcc(...; iterations = 1000, rhomax = 300)   README.md:161-179
  sweep B: the same light curves, 0:0.1:20 (201 delays)                                                         README.md:195-211
  sweep C: 3 bands, N = 60 + 50 + 40, (0.5:0.05:6)^2 = 111 x 111 = 12 321 delay pairs                           README.md:227-235
For each: the fixed-hyper-parameter evaluation rate of batches of that many delays (evals/s) and the full per-delay fit
(fitted grid points/s, objective evaluations, batched rounds, ms per round) through gpcc_grid_loglik.
  python tools/readme_bench.py [--iterations 1000] [--sweeps A,B,C] [--cpu-seconds 20] [--option key=value ...]
--cpu-seconds > 0 also times the CPU port (oracle.loglik_batch, the checker, in a child process) on a bounded sample."""
import argparse
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def workloads():
    from gpcc_amd import synthetic
    t2, y2, s2, _ = synthetic.simulate_lightcurves([60, 50], seed=1, gap_band=1, span=20.0)
    t3, y3, s3, _ = synthetic.simulate_lightcurves([60, 50, 40], seed=1, gap_band=1, span=20.0)
    gA = np.arange(0.0, 20.0 + 1e-9, 0.2)
    gB = np.arange(0.0, 20.0 + 1e-9, 0.1)
    gC = np.arange(0.5, 6.0 + 1e-9, 0.05)
    d2, d3 = np.meshgrid(gC, gC, indexing="ij")
    return {
        "A": dict(t=t2, y=y2, s=s2, cand=np.stack([np.zeros_like(gA), gA], 1), what="2 bands N=110, 101 delays (README.md:161)"),
        "B": dict(t=t2, y=y2, s=s2, cand=np.stack([np.zeros_like(gB), gB], 1), what="2 bands N=110, 201 delays (README.md:195)"),
        "C": dict(t=t3, y=y3, s=s3, cand=np.stack([np.zeros(d2.size), d2.ravel(), d3.ravel()], 1),
                  what="3 bands N=150, 111x111 = 12321 delay pairs (README.md:227)"),
    }


def cpu_port_rate(w, kernel, seconds):
    """evals/s of the CPU port (C restatement, 1 thread per evaluation, all host cores) on a bounded sample."""
    from oracle import oracle
    from gpcc_amd import synthetic
    oracle.build()
    alpha, rho = synthetic.default_hyperparameters(w["y"])
    cand = w["cand"]
    ncores = len(os.sched_getaffinity(0))
    nthreads = min(ncores, 16)
    M = min(len(cand), 256)
    a = np.tile(alpha, (M, 1)); r = np.full(M, rho)
    oracle.loglik_batch(kernel, w["t"], w["y"], w["s"], cand[:M], a, r, True, nthreads=nthreads)
    n, t0 = 0, time.perf_counter()
    while time.perf_counter() - t0 < seconds:
        oracle.loglik_batch(kernel, w["t"], w["y"], w["s"], cand[:M], a, r, True, nthreads=nthreads)
        n += M
    return n / (time.perf_counter() - t0), nthreads


class OraclePort:
    """Objective-shaped wrapper of the CPU port (oracle.loglik_batch): the SAME lock-step optimiser host logic
    (gpcc_amd.fit / neldermead.py) then runs the README's fit on the host cores -- the CPU number beside the GPU's."""

    def __init__(self, kernel, w, nthreads):
        from oracle import oracle
        oracle.build()
        self.o, self.k, self.w, self.nthreads = oracle, kernel, w, nthreads

    def loglik_batch(self, delays, alpha, rho):
        return self.o.loglik_batch(self.k, self.w["t"], self.w["y"], self.w["s"], delays, alpha, rho, True, nthreads=self.nthreads)


def cpu_port_fit(w, kernel, iterations, max_points):
    """fitted grid points/s of the CPU port on a bounded sample of the sweep's delays (all of them when <= max_points)."""
    from gpcc_amd import api, fit
    cand = w["cand"]
    if len(cand) > max_points:
        cand = cand[np.linspace(0, len(cand) - 1, max_points).astype(int)]
    nthreads = min(len(os.sched_getaffinity(0)), 16)
    obj = OraclePort(kernel, w, nthreads)
    t0 = time.perf_counter()
    res = fit.gpcc_grid(w["t"], w["y"], w["s"], kernel=kernel, candidatedelays=cand, iterations=iterations, rhomax=300.0, seed=1,
                        objective=obj, engine="python", unpack=api.unpack_params)
    dt = time.perf_counter() - t0
    return {"cpu_port_fit_points": len(cand), "cpu_port_fit_seconds": round(dt, 3), "cpu_port_fitted_grid_points_per_s": round(len(cand) / dt, 2),
            "cpu_port_fit_evaluations": int(res.f_calls), "cpu_port_fit_rounds": int(res.rounds), "cpu_port_threads": nthreads,
            "cpu_port_what": "oracle/ C restatement (scalar assembly + own Cholesky), %d threads over the evaluations of a round, "
                             "numpy lock-step Nelder-Mead (gpcc_amd/neldermead.py); NOT the Julia reference (no Julia here)" % nthreads}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iterations", type=int, default=1000)
    ap.add_argument("--sweeps", default="A,B,C")
    ap.add_argument("--kernel", default="matern32")
    ap.add_argument("--cpu-seconds", type=float, default=0.0)
    ap.add_argument("--reps", type=int, default=20)
    ap.add_argument("--option", action="append", default=[])
    ap.add_argument("--no-fit", action="store_true")
    ap.add_argument("--devices", default="", help="comma-separated device ids: a multi-device handle (the fit is then sharded by delay, one gather); "
                                                  "repeated ids rehearse it on one GPU")
    ap.add_argument("--cpu-fit-points", type=int, default=0, help="> 0: also run the fit on the CPU port, on at most this many delays of each sweep")
    args = ap.parse_args()

    import gpcc_amd
    from gpcc_amd import synthetic
    W = workloads()
    for name in args.sweeps.split(","):
        w = W[name]
        cand = w["cand"]
        G = len(cand)
        alpha, rho = synthetic.default_hyperparameters(w["y"])
        out = {"sweep": name, "workload": w["what"], "kernel": args.kernel, "G": G}
        devs = [int(x) for x in args.devices.split(",")] if args.devices else None
        if devs:
            out["devices"] = devs
        with gpcc_amd.Objective(w["t"], w["y"], w["s"], args.kernel, devices=devs) as obj:
            for kv in args.option:
                k, v = kv.split("=")
                obj.set_option(k, int(v))
            a = np.tile(alpha, (G, 1)); r = np.full(G, rho)
            obj.loglik_batch(cand, a, r)
            ts = []
            for _ in range(args.reps):
                t0 = time.perf_counter(); ll, info = obj.loglik_batch(cand, a, r); ts.append(time.perf_counter() - t0)
            out["fixed_hyper_ms_per_batch"] = round(float(np.median(ts)) * 1e3, 4)
            out["fixed_hyper_evals_per_s"] = round(G / float(np.median(ts)), 1)
            t0 = time.perf_counter(); obj.loglik_batch(cand[:1], a[:1], r[:1]); t1 = time.perf_counter()
            ts1 = []
            for _ in range(args.reps):
                t0 = time.perf_counter(); obj.loglik_batch(cand[:1], a[:1], r[:1]); ts1.append(time.perf_counter() - t0)
            out["single_evaluation_ms"] = round(float(np.median(ts1)) * 1e3, 4)
            if not args.no_fit:
                obj.grid_loglik(cand[:8], 2, rhomax=300.0)
                t0 = time.perf_counter()
                ll, al, rh, info, its, (f_calls, rounds) = obj.grid_loglik(cand, args.iterations, rhomax=300.0, seed=1)
                dt = time.perf_counter() - t0
                p = gpcc_amd.getprobabilities(ll)
                out.update({"fit_iterations": args.iterations, "fit_seconds": round(dt, 4),
                            "fitted_grid_points_per_s": round(G / dt, 2), "objective_evaluations": f_calls,
                            "evals_per_point": round(f_calls / G, 1), "fit_evals_per_s": round(f_calls / dt, 1),
                            "batched_rounds": rounds, "ms_per_round": round(dt / rounds * 1e3, 4),
                            "mean_evals_per_round": round(f_calls / rounds, 1), "info_nonzero": int((info != 0).sum()),
                            "median_iterations_done": float(np.median(its)),
                            "posterior_mode": [float(x) for x in cand[int(np.argmax(p))]]})
                if devs:
                    comp, gms, tms = obj.multi_stats()
                    out["fit_per_device_ms"] = [round(float(x), 2) for x in comp]
                    out["fit_gather_ms"] = round(gms, 3)
        if args.cpu_seconds > 0:
            rate, nthreads = cpu_port_rate(w, args.kernel, args.cpu_seconds)
            out["cpu_port_evals_per_s"] = round(rate, 1)
            out["cpu_port_threads"] = nthreads
            if "objective_evaluations" in out:
                out["cpu_port_fit_seconds_estimate"] = round(out["objective_evaluations"] / rate, 1)
        if args.cpu_fit_points > 0:
            out.update(cpu_port_fit(w, args.kernel, args.iterations, args.cpu_fit_points))
            if "fitted_grid_points_per_s" in out:
                out["gpu_over_cpu_port_fit"] = round(out["fitted_grid_points_per_s"] / out["cpu_port_fitted_grid_points_per_s"], 1)
        print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
