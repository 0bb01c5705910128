#!/usr/bin/env python3
"""Timeline of the calls after the last idle gap (> 0.3 ms) under the default dispatch (run under rocprofv3 --kernel-trace;
--parse <dir> [dump]; the run mode makes three back-to-back calls and prints each call's host-visible time): per kernel type the
launch count, summed duration and mean grid, the wall span of the batch and the union of the kernels' intervals (how much of the wall
at least one kernel was running).
    rocprofv3 --kernel-trace -d gpurun_out/tl -- python3 tools/timeline.py 2048 32 [key=value ...]
    python3 tools/timeline.py --parse gpurun_out/tl"""
import glob, os, sqlite3, sys
import numpy as np
sys.path.insert(0, ".")
if len(sys.argv) > 2 and sys.argv[1] == "--parse":
    for f in glob.glob(os.path.join(sys.argv[2], "**", "*.db"), recursive=True):
        con = sqlite3.connect(f)
        tabs = [r[0] for r in con.execute("select name from sqlite_master where type in ('table','view')")]
        kt = [t for t in tabs if t == "kernels"] or [t for t in tabs if "kernel" in t.lower()]
        rows = con.execute("select name, grid_x, workgroup_x, start, end from %s order by start" % kt[0]).fetchall()
        rows = [r for r in rows if "gpcc" in r[0]]
        # the last batch = everything from the last gpcc_assemble_tiles launch(es) on: find the last gap > 200 us before an assemble
        # the last batch = everything after the last idle gap of more than 0.3 ms (the host's work between two calls)
        first, emax = 0, rows[0][4]
        for i in range(1, len(rows)):
            if rows[i][3] - emax > 3e5:
                first = i
            emax = max(emax, rows[i][4])
        rows = rows[first:]
        t0, t1 = rows[0][3], max(r[4] for r in rows)
        if len(sys.argv) > 3 and sys.argv[3] == "dump":   # every launch of the batch: start, end (us from the first), grid
            for n, g, w, s_, e_ in rows:
                print("%9.1f %9.1f  %6d WG  %s" % ((s_ - t0) / 1e3, (e_ - t0) / 1e3, g // max(w, 1), n.split("(")[0].replace("void ", "")[:50]))
        per = {}
        for n, g, w, s, e in rows:
            k = n.split("(")[0].replace("void ", "")[:44]
            per.setdefault(k, []).append(((e - s) / 1e3, g // max(w, 1)))
        iv = sorted((s, e) for _, _, _, s, e in rows)
        busy, cur_s, cur_e = 0, iv[0][0], iv[0][1]
        for s, e in iv[1:]:
            if s > cur_e:
                busy += cur_e - cur_s
                cur_s, cur_e = s, e
            else:
                cur_e = max(cur_e, e)
        busy += cur_e - cur_s
        print("wall %.3f ms, some kernel running %.3f ms, sum of kernel durations %.3f ms, %d launches" % ((t1 - t0) / 1e6, busy / 1e6, sum(e - s for _, _, _, s, e in rows) / 1e6, len(rows)))
        for k, v in sorted(per.items(), key=lambda kv: -sum(d for d, _ in kv[1])):
            print("  %-46s %4d launches  %8.3f ms  mean %7.1f us  mean grid %6.0f WG" % (k, len(v), sum(d for d, _ in v) / 1e3, np.mean([d for d, _ in v]), np.mean([g for _, g in v])))
    sys.exit(0)
import gpcc_amd
from gpcc_amd import synthetic
Nb = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
M = int(sys.argv[2]) if len(sys.argv) > 2 else 32
t, y, s, _ = synthetic.simulate_lightcurves([Nb, Nb], seed=1)
alpha, rho = synthetic.default_hyperparameters(y)
d = np.stack([np.zeros(M), np.linspace(0, 20, M)], 1); a = np.tile(alpha, (M, 1)); r = np.full(M, rho)
with gpcc_amd.Objective(t, y, s, "matern32") as obj:
    obj.set_option("shared_prefix", 0)
    for kv in sys.argv[3:]:
        k, v = kv.split("=")
        obj.set_option(k, int(v))
    import time
    for _ in range(3):
        t0 = time.perf_counter()
        obj.loglik_batch(d, a, r)
        print("host-visible time of the call: %.3f ms" % ((time.perf_counter() - t0) * 1e3), flush=True)
