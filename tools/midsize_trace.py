#!/usr/bin/env python3
"""Per-launch durations of gpcc_panel_update by step, right-looking tail off then on (run under rocprofv3 --kernel-trace; --parse <dir>)."""
import glob, os, sqlite3, sys
import numpy as np
sys.path.insert(0, ".")
if len(sys.argv) > 2 and sys.argv[1] == "--parse":
    for f in glob.glob(os.path.join(sys.argv[2], "**", "*.db"), recursive=True):
        con = sqlite3.connect(f)
        rows = con.execute("select name, grid_x, workgroup_x, start, end from kernels order by start").fetchall()
        upd = [(g // w, (e - s) / 1e3) for n, g, w, s, e in rows if "gpcc_panel_update" in n or "gpcc_update_streamk" in n]
        other = {}
        for n, g, w, s, e in rows:
            if "gpcc" in n and "panel_update" not in n and "streamk" not in n:
                other.setdefault(n.split("(")[0][:40], []).append((e - s) / 1e3)
        per = len(upd) // 4
        print("launches", len(upd))
        for i in range(0, len(upd), 31):
            blk = upd[i:i + 31]
            print("run %d: total %.2f ms" % (i // 31, sum(d for _, d in blk) / 1e3))
            print("  " + " ".join("k%d:%dwg:%.0fus" % (j + 1, g, d) for j, (g, d) in enumerate(blk)))
        for k, v in other.items():
            print(k, len(v), "mean %.1f us" % np.mean(v))
    sys.exit(0)
import gpcc_amd
from gpcc_amd import synthetic
t, y, s, _ = synthetic.simulate_lightcurves([2048, 2048], seed=1)
alpha, rho = synthetic.default_hyperparameters(y)
M = int(sys.argv[1]) if len(sys.argv) > 1 else 32
d = np.stack([np.zeros(M), np.linspace(0, 20, M)], 1); a = np.tile(alpha, (M, 1)); r = np.full(M, rho)
for split in (0, 1):
    with gpcc_amd.Objective(t, y, s, "matern32", slots_per_stream=256) as obj:
        for k, v in (("shared_prefix", 0), ("right_looking_max", 0), ("fused_solve_min", 100000), ("hybrid_tail", split)):
            obj.set_option(k, v)
        obj.loglik_batch(d, a, r)
        obj.loglik_batch(d, a, r)
