#!/usr/bin/env python3
"""Re-evaluates the survivors of tools/adversarial_fp32.py (a log of its JSON lines) with the guard OFF and prints, per case,
the fp32 error beside BOTH statistics the diagonal kernel reports: mean pivot ratio S/N and the largest single ratio."""
import json, sys
import numpy as np
sys.path.insert(0, ".")
import gpcc_amd as gp
from gpcc_amd import synthetic
rng = np.random.default_rng(0)
for line in open(sys.argv[1]):
    if not line.startswith("{"):
        continue
    r = json.loads(line)
    if not isinstance(r.get("worst"), list) or not r["worst"]:
        continue
    t, y, s, _ = synthetic.simulate_lightcurves(r["Nl"], seed=r["data_seed"], sigma=r["sigma"])
    N, L = sum(r["Nl"]), len(r["Nl"])
    w = r["worst"][0]
    # the worst survivor and a cloud around it
    base = np.concatenate([np.log10(w["alpha"]), [np.log10(w["rho"])]])
    P = 64
    pts = base + rng.standard_normal((P, L + 1)) * 0.15
    pts[0] = base
    alpha, rho = 10.0 ** pts[:, :L], 10.0 ** pts[:, L]
    delays = np.tile(w["delays"], (P, 1))
    with gp.Objective(t, y, s, r["kernel"], marginalise_b=r["marginalise_b"], precision="fp64") as o64, \
            gp.Objective(t, y, s, r["kernel"], marginalise_b=r["marginalise_b"], precision="fp32") as o32:
        o32.set_option("fp32_guard", 0)
        ref, i64 = o64.loglik_batch(delays, alpha, rho)
        ll, i32 = o32.loglik_batch(delays, alpha, rho)
        cond = o32.conditioning(P)
    ok = (i64 == 0) & (i32 == 0)
    err = np.abs(ll - ref) / np.abs(ref)
    print("# %s sigma %g %s" % (r["Nl"], r["sigma"], r["kernel"]))
    for i in np.argsort(-np.where(ok, err, -1))[:12]:
        print("  err %.3e  S/N %9.1f  max ratio %.3e  info32 %d  |ll| %.3e" % (err[i], cond[i, 0] / N, cond[i, 1], i32[i], abs(ref[i])))
