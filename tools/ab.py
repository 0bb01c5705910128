#!/usr/bin/env python3
"""A/B of option sets in ONE process (interleaved rounds, median), any size and batch:
    python tools/ab.py --n-per-band 1024 --grid 256 [--bands 2] [--precision fp64] [--streams 1] [--slots 256] A B ...
where each variant is "name:key=value,key=value" (name: with no options = the defaults)."""
import argparse
import sys
import time

import numpy as np
import torch

sys.path.insert(0, ".")
import gpcc_amd  # noqa: E402
from gpcc_amd import synthetic  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--n-per-band", type=int, default=2048)
ap.add_argument("--bands", type=int, default=2)
ap.add_argument("--grid", default="1024", help="batch sizes, comma separated")
ap.add_argument("--precision", default="fp64")
ap.add_argument("--kernel", default="matern32")
ap.add_argument("--streams", type=int, default=1)
ap.add_argument("--slots", type=int, default=None)
ap.add_argument("--rounds", type=int, default=7)
ap.add_argument("variants", nargs="+")
args = ap.parse_args()
torch.cuda.init()
dev = torch.device("cuda", 0)
L = args.bands
t, y, s, _ = synthetic.simulate_lightcurves([args.n_per_band] * L, seed=1)
alpha, rho = synthetic.default_hyperparameters(y)
variants = []
for v in args.variants:
    name, _, opts = v.partition(":")
    variants.append((name, [kv.split("=") for kv in opts.split(",") if kv]))
objs = []
for name, opts in variants:
    o = gpcc_amd.Objective(t, y, s, args.kernel, precision=args.precision, streams=args.streams, slots_per_stream=args.slots)
    o.set_option("shared_prefix", 0)
    for k, v in opts:
        o.set_option(k, int(v))
    objs.append(o)
for G in [int(x) for x in args.grid.split(",")]:
    rng = np.random.default_rng(1)
    delays = np.concatenate([np.zeros((G, 1)), rng.random((G, L - 1)) * 20], 1)
    d_d = torch.as_tensor(delays, device=dev)
    d_a = torch.as_tensor(np.tile(alpha, (G, 1)), device=dev)
    d_r = torch.full((G,), float(rho), dtype=torch.float64, device=dev)
    out = torch.empty(G, dtype=torch.float64, device=dev)
    info = torch.empty(G, dtype=torch.int32, device=dev)
    times = [[] for _ in variants]
    ref = None
    line = []
    for r in range(args.rounds + 1):
        for i, o in enumerate(objs):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            o.loglik_batch_device(d_d, d_a, d_r, out=out, info=info)
            torch.cuda.synchronize()
            if r:
                times[i].append(time.perf_counter() - t0)
            else:
                ll = out.cpu().numpy().copy()
                ref = ll if ref is None else ref
                line.append(float(np.nanmax(np.abs(ll - ref) / np.abs(ref))))
    print("N=%d M=%d %s: " % (L * args.n_per_band, G, args.precision) +
          " | ".join("%s %.3f ms (%.0f/s, dev %.0e)" % (variants[i][0], np.median(times[i]) * 1e3, G / np.median(times[i]), line[i])
                     for i in range(len(variants))), flush=True)
