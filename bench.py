#!/usr/bin/env python3
"""bench.py -- delay-grid log-marginal-likelihood evaluations per second on MI355X.

Workload (BASELINE.json metric: N = 4096 total observations, 2 bands, Matern-3/2, fp64): synthetic
irregularly sampled light curves 2 x 2048 (gpcc_amd.synthetic, seed 1), fixed hyper-parameters
(alpha_l = var(y_l), rho = 3.5), a 1024-point delay grid tau_2 in linspace(0, 20, 1024) per GPU.
One STEP = every rank evaluates its block of the grid through the C ABI
(gpcc_loglik_batch_device: assemble K, Cholesky, solve, log-det -> one log-likelihood per delay),
then ONE all_gather (RCCL) of the log-likelihoods and getprobabilities on the gathered vector.
Weak scaling (default): the grid grows with the number of GPUs (1024 delays per GPU).  Strong scaling: --grid-total G splits a
FIXED grid of G delays over the GPUs (BASELINE cfg4: 65 536 = 256 x 256 with --bands 3 --n-per-band 1365; cfg5: 4096 with
--n-per-band 8192 --precision fp32 --kernel matern52), "scaling": "strong" in the line.

Launch: python bench.py [--gpus 1]            or, for N > 1,
        python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
               --master-port P bench.py --gpus N --steps K --warmup W          (one process per GPU, RCCL via torch.distributed)
        python bench.py --gpus N [--native]                                      (no torchrun: ONE process, multi-device handle,
                                                                                  RCCL all-gather inside libgpcc_hip.so; --native takes
                                                                                  that entry also for --gpus 1: comparable scaling points)
Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

FP64_MFMA_PEAK_TFLOPS = 78.6   # MI355X fp64 matrix peak (vendor sheet; BASELINE.md "Bounds")
FP32_MFMA_PEAK_TFLOPS = 157.3  # fp32-input MFMA (MI355X_MICROARCH.md: 157.3 spec, 155 measured)
TILE = 128


def update_flops_per_eval(N, fused=True):
    """Algorithmic flops of the dominant kernel for one N x N Cholesky (padded order nt*128, lower triangle only).
    fused (default path): gpcc_update_solve = dgemm update + dtrsm of every tile (I,k), I > k:
        2*128^3*k + 128^3 per tile (the dsyrk of the diagonal tiles runs in gpcc_syrk_diag);
    three-kernel path: gpcc_panel_update = dgemm tiles + the dsyrk diagonal tile (128*(128+1)*K) per step."""
    nt = (N + TILE - 1) // TILE
    total = 0.0
    per_step = []
    if fused:
        for k in range(0, nt - 1):
            f = (nt - k - 1) * (2.0 * TILE ** 3 * k + 1.0 * TILE ** 3)
            per_step.append(f)
            total += f
        return total, per_step
    for k in range(1, nt):
        K = k * TILE
        f = (nt - k - 1) * 2.0 * TILE * TILE * K + TILE * (TILE + 1) * K   # gemm tiles + syrk diag tile
        per_step.append(f)
        total += f
    return total, per_step


def gpu_state(index=0):
    """One sample of the GPU's shader clock (MHz) and socket power (W): from sysfs only -- no child process is ever started (a sample that sysfs
    cannot give stays None: this runs on a sampler thread during the timed region, and under a profiler preload a child's exec hop is
    what the pool forbids).
    Box-to-box spread of the headline is +-2 %; with this in the line a 2 % move can be told from a clock / power difference."""
    import glob
    out = {"sclk_mhz": None, "power_w": None}
    try:
        path = None
        try:    # the HIP device's own PCI function (the box may expose more cards in sysfs than HIP devices, in another order)
            import torch
            pr = torch.cuda.get_device_properties(index)
            bdf = "%04x:%02x:%02x.0" % (getattr(pr, "pci_domain_id", 0), pr.pci_bus_id, pr.pci_device_id)
            cand = "/sys/bus/pci/devices/%s/pp_dpm_sclk" % bdf
            if os.path.exists(cand):
                path = cand
        except Exception:
            path = None
        cards = sorted(glob.glob("/sys/class/drm/card[0-9]*/device/pp_dpm_sclk"))
        if path is None and cards:
            path = cards[min(index, len(cards) - 1)]
        if path:
            for line in open(path):
                if "*" in line:
                    out["sclk_mhz"] = int("".join(ch for ch in line.split(":")[1] if ch.isdigit()))
            hw = glob.glob(os.path.join(os.path.dirname(path), "hwmon", "hwmon*"))
            for h in hw:
                for name in ("power1_average", "power1_input"):
                    f = os.path.join(h, name)
                    if os.path.exists(f):
                        out["power_w"] = round(int(open(f).read().strip()) / 1e6, 1)
                        break
    except Exception:
        pass
    return out


def small_executed_ops(N, kernel):
    """Double-precision pipe work one small-N evaluation EXECUTES (flops; an fma = 2), by source -- the count behind DESIGN.md 4.6's
    cycle budget.  NB = blocks of the matrix bordered by the right-hand side.  MFMA: row J of the row-wise Cholesky applies J
    finished rows to NB - J blocks and multiplies NB - J - 1 blocks by -inv(L_D): 4 v_mfma_f64_16x16x4_f64 = 8192 flops per block
    product.  Elements (round 4, separable form): NB(NB+1)/2 blocks x 256 elements x (2 mul + min + distance + scale + Matern
    polynomial + Sobs / B: ~10 flops) plus 2 exponentials per point in the set-up (~70 flops each); rbf keeps the direct form (exp:
    19 ops, 16 of them fma = 35 flops, + ~9).  Diagonal step: NB x 16 pivots x ~(15 fma + 12 chain ops), issued as 64-lane
    instructions."""
    NB = (N + 1 + 15) // 16
    mfma_blocks = sum(J * (NB - J) + (NB - J - 1) for J in range(NB))
    poly = {"OU": 0, "rbf": 1, "matern32": 2, "matern52": 4}.get(kernel, 2)
    per_elem = (35.0 + 8.0 + poly) if kernel == "rbf" else (8.0 + poly)
    return {"mfma": mfma_blocks * 8192.0,
            "element_code": NB * (NB + 1) / 2 * 256 * per_elem + (0.0 if kernel == "rbf" else 2.0 * N * 70.0),
            "pivot_steps_16x16": NB * 16 * (2.0 * 15 + 12.0) * 64}


class GpuSampler:
    """Samples gpu_state on a host thread every `period` seconds WHILE the timed region runs (a sample taken before or after it sees
    an idle GPU at its lowest clock): -> {"sclk_mhz": [min, median, max], "power_w": [mean, max], "samples": n}."""

    def __init__(self, index=0, period=0.05):
        import threading
        self.index, self.period, self.rows = index, period, []
        self._stop = threading.Event()
        self._th = threading.Thread(target=self._run, daemon=True)

    def _run(self):
        while not self._stop.is_set():
            self.rows.append(gpu_state(self.index))
            self._stop.wait(self.period)

    def __enter__(self):
        self._th.start()
        return self

    def __exit__(self, *exc):
        self._stop.set()
        self._th.join(timeout=5.0)

    def summary(self):
        clk = sorted(r["sclk_mhz"] for r in self.rows if r["sclk_mhz"] is not None)
        pw = [r["power_w"] for r in self.rows if r["power_w"] is not None]
        return {"sclk_mhz": [clk[0], clk[len(clk) // 2], clk[-1]] if clk else None,
                "power_w": [round(sum(pw) / len(pw), 1), max(pw)] if pw else None, "samples": len(self.rows),
                "what": "sysfs pp_dpm_sclk level / hwmon power1_average sampled every %.0f ms during the timed region: [min, median, max] MHz, [mean, max] W" % (self.period * 1e3)}


def pmc_traffic(kernel, slots, N, prec="fp64"):
    """HBM bytes per launch of `kernel` from the committed rocprofv3 --pmc summary (bench.py cannot
    collect PMC counters itself); None if no summary matches this configuration."""
    path = os.path.join(ROOT, "profiles", "pmc_latest.json")
    try:
        with open(path) as f:
            d = json.load(f)
        if d.get("slots") != slots or d.get("N") != N:
            return None, None
        import gpcc_amd
        if d.get("build") != gpcc_amd.build_info():      # counters of another build of the library say nothing about this one
            return None, "profiles/pmc_latest.json was recorded for build %r, this library is %r: re-run tools/profile_round.sh" % (d.get("build"), gpcc_amd.build_info())
        for name, e in d["kernels"].items():   # template instantiations carry their arguments in the name
            if name.startswith(kernel) and "hbm_bytes_per_launch" in e and ("<float" in name) == (prec == "fp32"):
                return e["hbm_bytes_per_launch"], "profiles/pmc_latest.json: %s (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE)" % name
        return None, None
    except Exception:
        return None, None


def delay_grid(L, Gtot):
    """The whole grid (Gtot x L): 2 bands tau_2 in linspace(0, 20, Gtot); 3 bands (tau_2, tau_3) on a square over [0.5, 6]^2
    (README.md:227), flattened row-major."""
    if L == 2:
        return np.stack([np.zeros(Gtot), np.linspace(0.0, 20.0, Gtot)], 1)
    side = int(np.ceil(np.sqrt(Gtot)))
    g1 = np.linspace(0.5, 6.0, side)
    d2, d3 = np.meshgrid(g1, g1, indexing="ij")
    return np.ascontiguousarray(np.stack([np.zeros(side * side), d2.ravel(), d3.ravel()], 1)[:Gtot])


def roofline_leg(obj, args, N, L, G, Gtot, elapsed, run_once):
    """The roofline object of the bench line: one more pass (`run_once`, profiled: HIP events around every launch on its own stream,
    groups serialised) after the timed region.  Shared by both entries (device pointers under torch / one-process multi-device handle)."""
    roofline = None
    # dominant kernel = the fp64-MFMA panel update; its launches are timed live with HIP events
    # on the stream they run on (gpcc_profile_*), groups serialised on one stream meanwhile.
    obj.profile(True)
    obj.profile_reset()
    run_once()
    prof = obj.profile_get()
    obj.profile(False)
    small = obj.get_option("small_n_active") == 1
    slots = obj.get_option("workspace_slots")     # (= the option slots_per_stream unless the memory was short)
    # the path a group takes: the fused step only for groups of >= fused_solve_min evaluations
    # (gpcc_hip.hip: enqueue_factor_t); a ragged last group may take another path -- the dominant group decides
    group = min(G, slots)
    fused = (not small) and obj.get_option("fused_solve") == 1 and group >= obj.get_option("fused_solve_min") \
        and group > obj.get_option("right_looking_max")
    peak = FP64_MFMA_PEAK_TFLOPS if (args.precision == "fp64" or small) else FP32_MFMA_PEAK_TFLOPS
    timing_note = ("separate profiled pass after the timed region: HIP events around every launch on its own stream, "
                   "groups serialised on one stream (gpcc_profile_*); rocprofv3 --kernel-trace --stats of the same command: profiles/")
    # profile slots -> the kernels that really ran in them on this path
    if small:
        wide = G <= obj.get_option("small_wide_max") or N > 191
        names = {"small_eval": "gpcc_smallw_eval" if wide else "gpcc_small_eval"}
    elif fused:
        names = {"assemble": "gpcc_assemble_tiles", "panel_update": "gpcc_update_solve", "diag_factor": "gpcc_syrk_diag",
                 "refine": "gpcc_back_solve+gpcc_refine_partials+gpcc_refine_finish"}
    else:
        names = {"assemble": "gpcc_assemble_tiles", "panel_update": "gpcc_panel_update", "diag_factor": "gpcc_diag_factor",
                 "panel_trsm": "gpcc_panel_trsm(_rows)", "small_step": "gpcc_small_step",
                 "refine": "gpcc_back_solve+gpcc_refine_partials+gpcc_refine_finish"}
    kernels_ms = {names.get(k, k): round(v[1], 3) for k, v in prof.items() if v[0] > 0}
    end_to_end = (N ** 3 / 3.0) * Gtot * args.steps / elapsed / 1e12     # N^3/3 per evaluation, whole step
    n3 = N ** 3 / 3.0
    executed = None
    if small:
        # ONE kernel does the whole evaluation (assembly + Cholesky + forward solve): algorithmic flops N^3/3 + N^2
        launches, total_ms = prof["small_eval"]
        kname = names["small_eval"]
        flops_eval = N ** 3 / 3.0 + float(N) ** 2
        executed = small_executed_ops(N, args.kernel)
    else:
        launches, total_ms = prof["panel_update"]
        kname = names["panel_update"]
        flops_eval, _ = update_flops_per_eval(N, fused)
    if launches > 0 and total_ms > 0:
        avg_ms = total_ms / launches
        flops_per_launch = flops_eval * G / launches      # algorithmic flops / launch (average over steps k)
        achieved = flops_per_launch / (avg_ms * 1e-3) / 1e12
        traffic, tsrc = pmc_traffic(kname, slots, N, args.precision)
        roofline = {"bound": "mfma", "kernel": kname, "achieved": round(achieved, 3),
                    "peak": peak, "unit": "TFLOP/s", "frac": round(achieved / peak, 4),
                    # the same launches credited with SURVEY 8(d)'s plain N^3/3 per evaluation (more than this kernel does when
                    # the diagonal tiles run elsewhere): both accountings, so that neither has to be re-derived
                    "frac_n3_over_3": round(n3 * G / launches / (avg_ms * 1e-3) / 1e12 / peak, 4),
                    "frac_accounting": "frac: the flops THIS kernel performs (%s); frac_n3_over_3: N^3/3 per evaluation over the same launches"
                                       % ("N^3/3 + N^2" if small else "dgemm + dtrsm of the tiles I > k" if fused else "dgemm tiles + dsyrk diagonal tile"),
                    "traffic": traffic, "traffic_source": tsrc, "launches": launches, "avg_launch_ms": round(avg_ms, 4),
                    "algorithmic_flops_per_launch": flops_per_launch, "timing": timing_note,
                    "end_to_end_tflops": round(end_to_end, 3), "end_to_end_frac": round(end_to_end / peak, 4),
                    "kernels_ms": kernels_ms}
        if not small:   # the HBM-bound assembly kernel, reported beside it
            an, ams = prof["assemble"]
            nt = (N + TILE - 1) // TILE
            esz = 8.0 if args.precision == "fp64" else 4.0
            # fold_assembly: the factorisation evaluates the off-diagonal tiles itself; what the assembly launches
            # still write is the diagonal tiles (+ tile column 0 on the three-kernel path) and 4 N doubles of per-point factors
            folded = (obj.get_option("fold_assembly") == 1 and nt > 1 and G > obj.get_option("fused_small_max")
                      and (args.kernel != "rbf" or (args.precision == "fp32" and obj.get_option("fp32_assemble") == 1)))
            tiles_written = (nt + (0 if fused else nt - 1)) if folded else nt * (nt + 1) / 2
            abytes = (esz * TILE * TILE * tiles_written + (32.0 * N if folded else 0.0)) * G / max(an, 1)
            roofline["assemble"] = {"bound": "hbm", "achieved": round(abytes / (ams / max(an, 1) * 1e-3) / 1e9, 1),
                                    "peak": 8000.0, "unit": "GB/s", "avg_launch_ms": round(ams / max(an, 1), 4),
                                    "algorithmic_bytes_per_launch": abytes, "folded_into_factorisation": bool(folded),
                                    "tiles_written_per_evaluation": tiles_written,
                                    "launches_counted": "gpcc_sep_points + gpcc_assemble_tiles" if folded else "gpcc_assemble_tiles"}
        else:
            # what the fp64 pipe EXECUTES per evaluation beside the algorithmic N^3/3 + N^2: padded MFMA blocks,
            # the element code (exp) and the 16 x 16 pivot steps, all on the same double-precision pipe
            ex_total = sum(executed.values())
            roofline["executed_ops"] = {"per_evaluation_dp_pipe_flops": {k: round(v) for k, v in executed.items()},
                                        "total": round(ex_total), "algorithmic": round(flops_eval),
                                        "executed_over_algorithmic": round(ex_total / flops_eval, 2),
                                        "executed_tflops": round(ex_total * G / launches / (avg_ms * 1e-3) / 1e12, 2),
                                        "executed_frac_of_peak": round(ex_total * G / launches / (avg_ms * 1e-3) / 1e12 / peak, 4)}
            roofline["note"] = ("one wave (or four) per evaluation, matrix in registers: bound by VALU/MFMA issue of the fp64 pipe "
                                "(assembly exp + 16x16 pivot chains + MFMAs), no HBM traffic beyond 3N inputs and 12 bytes out")
    return roofline


def cpu_baseline_leg(args, Nb, L, G, delays, ll_host):
    """The cpu_baseline object of the bench line (rank 0, N = 1 only)."""
    cpu_baseline = None
    # Reported baseline, outside the timed region: the CPU shape of the reference's path -- scalar-loop assembly (the C
    # restatement under oracle/) + OpenBLAS dpotrf/dtrtrs -- timed in its OWN process (it forks worker pools; this
    # process holds the GPU) on a bounded sample of the same grid, both ways the README parallelises.
    import subprocess
    nsample = 8
    idx = np.linspace(0, G - 1, nsample).astype(int)
    cmd = [sys.executable, "-m", "oracle.lapack_baseline", "--n-per-band", str(Nb), "--bands", str(L), "--kernel", args.kernel,
           "--seed", str(args.seed), "--delays", json.dumps(delays[idx].tolist()), "--evals-per-worker", str(args.cpu_sample or 24)]
    c0 = time.perf_counter()
    run = subprocess.run(cmd, capture_output=True, text=True, cwd=ROOT)
    cpu_s = time.perf_counter() - c0
    if run.returncode == 0:
        rec = json.loads(run.stdout.strip().splitlines()[-1])
        ref = np.array(rec["loglik"])
        gpu_ll = np.asarray(ll_host)[idx][:len(ref)]
        rel = float(np.max(np.abs(gpu_ll - ref) / np.abs(ref)))
        best = rec["pmap"] if rec["pmap"]["evals_per_s"] >= rec["blas"]["evals_per_s"] else None
        cpu_baseline = {"value": (best or rec["blas"])["evals_per_s"], "unit": "evals/s",
                        "cores": best["workers"] if best else rec["blas"]["threads"], "kind": "port+LAPACK",
                        "sample": "%d evaluations of the same grid (%d distinct delays), scalar-loop C assembly + OpenBLAS dpotrf/dtrtrs "
                                  "(scipy), %s; reference not executable (no Julia)"
                                  % ((best or rec["blas"])["evals"], nsample,
                                     "P single-threaded worker processes (the README's pmap shape)" if best else "one evaluation at a time, BLAS on all threads"),
                        "pmap_shape": rec["pmap_runs"], "blas_shape": rec["blas"], "cores_available": rec["cores_available"],
                        "seconds": round(cpu_s, 2), "max_rel_err_gpu_vs_cpu": rel}
    else:
        cpu_baseline = {"value": None, "error": run.stderr[-400:]}
    return cpu_baseline


def native_multi(args, ndev):
    """python bench.py --gpus N without torchrun: weak scaling over a multi-device handle, host pointers in and out (the 40 KiB
    of parameters per 1024 delays are part of the timed region).  GPCC_BENCH_OVERSUBSCRIBE=1 rehearses it on fewer GPUs
    (repeated device ids gather through host memory instead of RCCL)."""
    import gpcc_amd
    from gpcc_amd import synthetic
    N_ = args.gpus
    if ndev >= N_:
        devs = list(range(N_))
    elif os.environ.get("GPCC_BENCH_OVERSUBSCRIBE") == "1":
        devs = [i % ndev for i in range(N_)]
    else:
        raise SystemExit("--gpus %d without torchrun needs %d visible GPUs (found %d)" % (N_, N_, ndev))
    Nb, L, G = args.n_per_band, args.bands, args.grid
    t, y, s, _ = synthetic.simulate_lightcurves([Nb] * L, seed=args.seed)
    alpha, rho = synthetic.default_hyperparameters(y)
    Gtot = args.grid_total if args.grid_total else G * N_
    G = (Gtot + N_ - 1) // N_          # the library deals contiguous blocks of ceil(G / n) delays
    delays = delay_grid(L, Gtot)
    alphas, rhos = np.tile(alpha, (Gtot, 1)), np.full(Gtot, float(rho))
    with gpcc_amd.Objective(t, y, s, args.kernel, precision=args.precision, devices=devs, streams=args.streams,
                            slots_per_stream=args.slots) as obj:
        obj.set_option("shared_prefix", 0)
        for kv in args.option:
            key, val = kv.split("=")
            obj.set_option(key, int(val))

        def step():
            ll, info = obj.loglik_batch(delays, alphas, rhos)       # sharded, ONE all-gather inside the library
            return ll, info, gpcc_amd.getprobabilities(ll, device=devs[0])
        for _ in range(args.warmup):
            step()
        with GpuSampler(devs[0]) as smp:
            t0 = time.perf_counter()
            for _ in range(args.steps):
                ll, info, p = step()
            elapsed = time.perf_counter() - t0
        gstate = smp.summary()
        mode = {1: "rccl", 2: "host"}.get(obj.get_option("gather_mode"), "?")
        comp_ms, gather_ms, total_ms = obj.multi_stats()     # of the LAST step: where a scaling loss would come from
        roofline = cpu_baseline = None
        if N_ == 1:     # a SCALE point at N = 1 through this entry is comparable with the BENCH line field by field
            if not args.no_roofline:
                roofline = roofline_leg(obj, args, L * Nb, L, G, Gtot, elapsed, lambda: obj.loglik_batch(delays, alphas, rhos))
            if not args.no_cpu_baseline:
                cpu_baseline = cpu_baseline_leg(args, Nb, L, G, delays, ll)
        build = gpcc_amd.build_info()
    print(json.dumps({
        "metric": "delay-grid loglik evals/sec (N=%d, %d-band %s)" % (L * Nb, L, {"matern32": "Matern-3/2", "matern52": "Matern-5/2"}.get(args.kernel, args.kernel)),
        "value": round(Gtot * args.steps / elapsed, 2), "unit": "evals/s", "n_gpus": N_, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(elapsed / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "strong" if args.grid_total else "weak", "vs_baseline": None,
        "dtype": "f64" if args.precision == "fp64" else "f32", "data": "synthetic",
        "config": {"workload": "%d-band synthetic N=%d per band (N=%d), %s %s, %s" % (L, Nb, L * Nb, args.kernel, args.precision,
                                                                                    ("fixed %d-point delay grid split over the GPUs" % Gtot) if args.grid_total else ("%d-point delay grid per GPU" % G)),
                   "grid_total": Gtot, "grid_per_gpu": G, "devices": devs,
                   "parallelism": "ONE process, multi-device handle x%d, 1 all-gather inside libgpcc_hip (%s)" % (N_, mode)},
        "last_step": {"per_device_compute_ms": [round(float(x), 3) for x in comp_ms], "gather_ms": round(gather_ms, 3),
                      "call_ms": round(total_ms, 3)},
        "clock_mhz": gstate["sclk_mhz"], "power_w": gstate["power_w"], "gpu_state": gstate,
        "info_nonzero": int((info != 0).sum()), "posterior_sum": float(p.sum()), "roofline": roofline, "cpu_baseline": cpu_baseline,
        "entry": "native_multi", "build": build}))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--grid", type=int, default=1024, help="delays per GPU per step")
    ap.add_argument("--grid-total", type=int, default=0,
                    help="STRONG scaling: a fixed grid of this many delays split over the GPUs (contiguous blocks of ceil(G / n)); 0 = weak scaling with --grid per GPU")
    ap.add_argument("--native", action="store_true",
                    help="one process, multi-device handle, host pointers (the entry --gpus N takes without torchrun) also for --gpus 1")
    ap.add_argument("--n-per-band", type=int, default=2048)
    ap.add_argument("--bands", type=int, default=2, help="2 (default metric config) or 3 (cfg4: 2-D delay grid)")
    ap.add_argument("--kernel", default="matern32")
    ap.add_argument("--seed", type=int, default=1, help="seed of the synthetic light curves (SURVEY 8(d): seeds 1, 2, 3)")
    ap.add_argument("--precision", default="fp64", choices=["fp64", "fp32"])
    ap.add_argument("--streams", type=int, default=1,
                    help="groups in flight (library default: 2).  The default here is ONE stream, so that the timed region, the per-launch "
                         "timing behind `roofline` and the committed rocprofv3 averages all describe the same non-overlapping launches; "
                         "--streams 2 is +0.9 %% on the headline, +5 %% at N = 2048 with 4096 delays (profiles/r03/two_streams_ab.log)")
    ap.add_argument("--slots", type=int, default=None)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--shared-extra", action="store_true",
                    help="also time the sweep with the shared-prefix mode asserted (reported separately, never as `value`)")
    ap.add_argument("--option", action="append", default=[], help="handle option key=value (gpcc_set_option), repeatable")
    ap.add_argument("--cpu-sample", type=int, default=0, help="CPU baseline: evaluations per worker process (0: 24)")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

    import gpcc_amd
    from gpcc_amd import build as gbuild
    from gpcc_amd import synthetic

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    gbuild.ensure_present(local)     # source-only checkout: compile libgpcc_hip.so once per node (hipcc, gfx950)
    ndev = torch.cuda.device_count()
    if world == 1 and (args.gpus > 1 or args.native):
        # launched WITHOUT torchrun: ONE process drives the N GPUs through a multi-device handle (gpcc_create_multi: worker
        # threads + the RCCL all-gather inside libgpcc_hip.so) -- the shape a Julia host uses
        return native_multi(args, ndev)
    if world != args.gpus and rank == 0:
        print("warning: --gpus %d but WORLD_SIZE %d" % (args.gpus, world), file=sys.stderr)
    if local >= ndev and os.environ.get("GPCC_BENCH_OVERSUBSCRIBE") == "1":
        local = local % ndev          # rehearsal only: several ranks on one GPU (gloo collective below)
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    backend = os.environ.get("GPCC_BENCH_BACKEND", "nccl")   # "nccl" = RCCL; "gloo" only for one-GPU rehearsals
    # GPCC_BENCH_FORCE_DIST=1: run the collective path also with ONE rank (a one-GPU box can then execute the RCCL branch --
    # init_process_group("nccl"), all_gather_into_tensor, all_reduce, barrier -- as a one-rank group; launch under torchrun)
    use_dist = world > 1 or os.environ.get("GPCC_BENCH_FORCE_DIST") == "1"
    if use_dist:
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)

    Nb = args.n_per_band
    L = args.bands
    t, y, s, _ = synthetic.simulate_lightcurves([Nb] * L, seed=args.seed)
    alpha, rho = synthetic.default_hyperparameters(y)
    N = L * Nb
    if args.grid_total:      # strong scaling: the SAME grid whatever the number of GPUs, contiguous blocks (the last one may be short)
        Gtot = args.grid_total
        G = (Gtot + world - 1) // world
    else:
        G = args.grid
        Gtot = G * world
    lo = min(rank * G, Gtot)
    Gown = max(0, min(G, Gtot - lo))          # this rank's delays; the buffers hold G (all_gather_into_tensor needs equal blocks)
    grid_all = delay_grid(L, Gtot)
    delays = np.zeros((G, L))
    delays[:Gown] = grid_all[lo:lo + Gown]
    if Gown < G:
        delays[Gown:] = grid_all[-1]          # (padding of a short last block: evaluated, dropped after the gather)
    obj = gpcc_amd.Objective(t, y, s, args.kernel, marginalise_b=True, precision=args.precision, device=local,
                             streams=args.streams, slots_per_stream=args.slots)
    obj.set_option("shared_prefix", 0)   # the timed region: every evaluation factorises its full matrix
    for kv in args.option:
        key, val = kv.split("=")
        obj.set_option(key, int(val))
    d_delays = torch.as_tensor(delays, device=dev).contiguous()
    d_alpha = torch.as_tensor(np.tile(alpha, (G, 1)), device=dev).contiguous()
    d_rho = torch.full((G,), float(rho), dtype=torch.float64, device=dev)
    d_ll = torch.empty(G, dtype=torch.float64, device=dev)
    d_info = torch.empty(G, dtype=torch.int32, device=dev)
    d_all = torch.empty(G * world, dtype=torch.float64, device=dev)
    d_prob = torch.empty(Gtot, dtype=torch.float64, device=dev)
    from gpcc_amd import _capi
    lib = _capi.load()

    def step():
        obj.loglik_batch_device(d_delays, d_alpha, d_rho, out=d_ll, info=d_info)
        if use_dist:
            if backend == "nccl":
                dist.all_gather_into_tensor(d_all, d_ll)     # the path's single collective (RCCL over xGMI)
            else:
                host = torch.empty(G * world, dtype=torch.float64)
                dist.all_gather_into_tensor(host, d_ll.cpu())
                d_all.copy_(host)
            src = d_all
        else:
            src = d_ll
        _capi.check(lib.gpcc_probabilities_device(Gtot, src.data_ptr(), None, d_prob.data_ptr(),
                                                  torch.cuda.current_stream(dev).cuda_stream))

    def fence():
        torch.cuda.synchronize(dev)
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize(dev)

    for _ in range(args.warmup):
        step()
    fence()
    import contextlib
    smp = GpuSampler(local) if rank == 0 else None
    with (smp if smp is not None else contextlib.nullcontext()):
        t0 = time.perf_counter()
        for _ in range(args.steps):
            step()
        fence()
        elapsed = time.perf_counter() - t0
    gstate = smp.summary() if smp is not None else None
    if use_dist:
        tt = torch.tensor([elapsed], dtype=torch.float64, device=dev if backend == "nccl" else "cpu")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())

    # Extra, NOT the headline: the same sweep with the shared-prefix mode asserted (all delays of the grid have the
    # same band-1 amplitude, delay and rho, so each group factorises the leading band-1 tile rows once; results are
    # bitwise identical).  `value` above is measured with every evaluation doing all of its own work.
    shared = None
    if world == 1 and args.shared_extra:
        obj.set_option("shared_prefix", 2)
        step(); fence()
        ll_shared = d_ll.clone()
        s0 = time.perf_counter()
        for _ in range(args.steps):
            step()
        fence()
        shared_elapsed = time.perf_counter() - s0
        obj.set_option("shared_prefix", 0)
        step(); fence()
        shared = {"evals_per_s": round(Gtot * args.steps / shared_elapsed, 2),
                  "max_rel_diff_to_plain": float(((ll_shared - d_ll).abs() / d_ll.abs()).max().item()),
                  "note": "section 8(f).4 mode, not used for `value`"}

    info_bad = int((d_info != 0).sum().item())
    psum = float(d_prob.sum().item())
    value = Gtot * args.steps / elapsed

    roofline = None
    if rank == 0 and not args.no_roofline:
        def run_once():
            obj.loglik_batch_device(d_delays, d_alpha, d_rho, out=d_ll, info=d_info)
            torch.cuda.synchronize(dev)
        roofline = roofline_leg(obj, args, N, L, G, Gtot, elapsed, run_once)

    cpu_baseline = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        cpu_baseline = cpu_baseline_leg(args, Nb, L, G, delays, d_ll.cpu().numpy())

    if rank == 0:
        out = {
            "metric": "delay-grid loglik evals/sec (N=%d, %d-band %s)" % (N, L, {"matern32": "Matern-3/2", "matern52": "Matern-5/2"}.get(args.kernel, args.kernel)),
            "value": round(value, 2), "unit": "evals/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "strong" if args.grid_total else "weak",
            "vs_baseline": None, "dtype": "f64" if args.precision == "fp64" else "f32", "data": "synthetic",
            "config": {"workload": "%d-band synthetic N=%d per band (N=%d), %s %s, %s"
                                   % (L, Nb, N, args.kernel, args.precision,
                                      ("fixed %d-point delay grid split over the GPUs" % Gtot) if args.grid_total else ("%d-point delay grid per GPU" % G)),
                       "grid_total": Gtot, "grid_per_gpu": G, "streams": obj.get_option("workspace_streams"),
                       "slots_per_stream": obj.get_option("workspace_slots"),
                       "parallelism": "grid-sharded x%d, 1 all_gather" % world},
            "clock_mhz": gstate["sclk_mhz"], "power_w": gstate["power_w"], "gpu_state": gstate,
            "info_nonzero": info_bad, "posterior_sum": psum,
            "roofline": roofline, "cpu_baseline": cpu_baseline, "shared_prefix_mode": shared,
            "entry": "device_pointer", "build": gpcc_amd.build_info(),
        }
        print(json.dumps(out))
    obj.close()
    if use_dist:
        dist.barrier()     # rank 0 is the last to arrive (roofline leg): leave the group together
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
