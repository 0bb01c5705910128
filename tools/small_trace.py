"""Per-dispatch durations of the small-N kernel by batch size (run under rocprofv3 --kernel-trace; then
`python tools/small_trace.py --parse <dir>` prints duration per (kernel, grid size) and the host time per call beside it).
  rocprofv3 --kernel-trace -d gpurun_out/prof_small -o small -- python3 tools/small_trace.py"""
import csv
import glob
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def run(options, sizes):
    import gpcc_amd
    from gpcc_amd import synthetic
    from readme_bench import workloads
    W = workloads()
    if sizes:   # one band of n points per requested size instead of the README workloads
        W = {}
        for n in sizes:
            t, y, s, _ = synthetic.simulate_lightcurves([n - n // 2, n // 2], seed=1, span=20.0)
            W["N%d" % n] = dict(t=t, y=y, s=s, cand=np.stack([np.zeros(64), np.linspace(0, 10, 64)], 1))
    out = []
    for name in (sorted(W, key=lambda k: int(k[1:])) if sizes else ("A", "C")):
        w = W[name]
        alpha, rho = synthetic.default_hyperparameters(w["y"])
        with gpcc_amd.Objective(w["t"], w["y"], w["s"], "matern32") as obj:
            for kv in options:
                k, v = kv.split("=")
                obj.set_option(k, int(v))
            for M in ((1, 2048, 16384) if sizes else (1, 8, 64, 101, 201, 512, 1024, 2048, 4096, 12321)):
                idx = np.arange(M) % len(w["cand"])
                d = w["cand"][idx]
                a = np.tile(alpha, (M, 1)); r = np.full(M, rho)
                obj.loglik_batch(d, a, r)
                ts = []
                for _ in range(10):
                    t0 = time.perf_counter(); obj.loglik_batch(d, a, r); ts.append(time.perf_counter() - t0)
                out.append({"sweep": name, "N": int(sum(len(x) for x in w["t"])), "M": M, "host_us_per_call": round(float(np.median(ts)) * 1e6, 1)})
    print(json.dumps(out))


def parse(d):
    """per-dispatch durations from rocprofv3's rocpd database(s) under d (or its kernel_trace.csv files)"""
    import sqlite3
    agg = {}
    for f in glob.glob(os.path.join(d, "**", "*.db"), recursive=True):
        con = sqlite3.connect(f)
        for name, gx, wx, a, b, vg, ag, lds in con.execute("select name, grid_x, workgroup_x, start, end, vgpr_count, accum_vgpr_count, lds_size from kernels"):
            if "gpcc" in name:
                agg.setdefault((name.split("(")[0][:60] + " [vgpr %d+%d lds %d]" % (vg, ag, lds), gx // max(wx, 1)), []).append((b - a) / 1e3)
    for f in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
        with open(f) as fh:
            for r in csv.DictReader(fh):
                if "gpcc" in r["Kernel_Name"]:
                    key = (r["Kernel_Name"].split("(")[0][:60], int(r["Grid_Size_X"]) // max(int(r["Workgroup_Size_X"]), 1))
                    agg.setdefault(key, []).append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    for (k, g), v in sorted(agg.items()):
        v = np.array(v)
        print("%-90s blocks %6d  n %4d  median %9.2f us  min %9.2f us  -> %.3f M evals/s" % (k, g, len(v), np.median(v), v.min(), g / np.median(v)))


if __name__ == "__main__":
    if len(sys.argv) > 2 and sys.argv[1] == "--parse":
        parse(sys.argv[2])
    else:
        sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
        opts = [a.split("=", 1)[1] for a in sys.argv[1:] if a.startswith("--option=")]
        sizes = [int(x) for a in sys.argv[1:] if a.startswith("--sizes=") for x in a.split("=", 1)[1].split(",")]
        run(opts, sizes)
