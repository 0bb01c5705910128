#!/usr/bin/env python3
"""Per-launch durations of one 256-evaluation group by step, for the one-launch step (gpcc_step) and the two-launch step
(gpcc_syrk_diag + gpcc_update_solve).  Run under rocprofv3 --kernel-trace:
    rocprofv3 --kernel-trace -d gpurun_out/st1 -- python3 tools/step_trace.py 2048 step_fused=1
    rocprofv3 --kernel-trace -d gpurun_out/st0 -- python3 tools/step_trace.py 2048 step_fused=0
    python3 tools/step_trace.py --parse gpurun_out/st1 gpurun_out/st0 [nt]
"""
import glob, os, sqlite3, sys
import numpy as np
sys.path.insert(0, ".")


def load(d, nt):
    for f in glob.glob(os.path.join(d, "**", "*.db"), recursive=True):
        con = sqlite3.connect(f)
        tabs = [r[0] for r in con.execute("select name from sqlite_master where type in ('table','view')")]
        kt = [t for t in tabs if t == "kernels"] or [t for t in tabs if "kernel" in t.lower()]
        rows = con.execute("select name, start, end from %s order by start" % kt[0]).fetchall()
        step = [(s, e) for n, s, e in rows if "gpcc_step" in n][-nt:]
        us = [(s, e) for n, s, e in rows if "gpcc_update_solve" in n][-(nt - 1):]
        sd = [(s, e) for n, s, e in rows if "gpcc_syrk_diag" in n][-nt:]
        asm = [(s, e) for n, s, e in rows if "gpcc_assemble" in n][-1:]
        return step, us, sd, asm
    raise SystemExit("no .db under " + d)


if len(sys.argv) > 3 and sys.argv[1] == "--parse":
    nt = int(sys.argv[4]) if len(sys.argv) > 4 else 32
    step, _, _, asm1 = load(sys.argv[2], nt)
    _, us, sd, asm0 = load(sys.argv[3], nt)
    d = lambda p: (p[1] - p[0]) / 1e3
    print("launch k | gpcc_step(k) us | syrk_diag(k+1) + update_solve(k) us | gap-inclusive span ratio")
    t1 = t0 = 0.0
    for i in range(nt):
        k = i - 1
        new = d(step[i])
        old = d(sd[i]) + (d(us[i - 1]) if i >= 1 else 0.0)   # launch k of the new path = update_solve(k) + syrk_diag(k+1)
        t1 += new; t0 += old
        print("k=%3d  %9.1f   %9.1f = %7.1f + %7.1f   %.3f" % (k, new, old, d(sd[i]), d(us[i - 1]) if i >= 1 else 0.0, new / old))
    print("kernel time: one-launch %.3f ms, two-launch %.3f ms; wall span first start -> last end: %.3f vs %.3f ms; assembly %.3f / %.3f ms"
          % (t1 / 1e3, t0 / 1e3, (step[-1][1] - step[0][0]) / 1e6, (sd[-1][1] - sd[0][0]) / 1e6, d(asm1[0]) / 1e3, d(asm0[0]) / 1e3))
    sys.exit(0)
import gpcc_amd
from gpcc_amd import synthetic
Nb = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
t, y, s, _ = synthetic.simulate_lightcurves([Nb, Nb], seed=1)
alpha, rho = synthetic.default_hyperparameters(y)
M = 256
d = np.stack([np.zeros(M), np.linspace(0, 20, M)], 1); a = np.tile(alpha, (M, 1)); r = np.full(M, rho)
with gpcc_amd.Objective(t, y, s, "matern32", streams=1) as obj:
    obj.set_option("shared_prefix", 0)
    for kv in sys.argv[2:]:
        k, v = kv.split("=")
        obj.set_option(k, int(v))
    for _ in range(3):
        obj.loglik_batch(d, a, r)
