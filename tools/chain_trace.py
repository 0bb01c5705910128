#!/usr/bin/env python3
"""What one evaluation's critical chain consists of inside the persistent launch (csrc/gpcc_chain.hip.h): per diagonal step k the
device wall-clock stamps of the chain workgroup that runs it -- began to build tile (k,k) | image complete (first pivot next) | step
published (xrow = 9) -- and what the table derives from them: the time the chain spent BETWEEN the end of step k - 1 and the first pivot
of step k (the hand-off: last row block of inv(L) out -> last column block of L(k,k-1) solved and published -> folded into the tile),
and the diagonal step itself.
  python tools/chain_trace.py [--n-per-band 2048] [--bands 2] [--evals 1]"""
import argparse
import sys

import numpy as np

sys.path.insert(0, ".")
import gpcc_amd  # noqa: E402
from gpcc_amd import synthetic  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--n-per-band", type=int, default=2048)
ap.add_argument("--bands", type=int, default=2)
ap.add_argument("--evals", type=int, default=1)
ap.add_argument("--every", type=int, default=1, help="print every n-th step")
ap.add_argument("--opt", action="append", default=[], help="key=value set on the handle (e.g. chain_batch=1)")
ap.add_argument("--brief", action="store_true", help="only the sums and the workers' statistics")
args = ap.parse_args()
t, y, s, _ = synthetic.simulate_lightcurves([args.n_per_band] * args.bands, seed=1)
alpha, rho = synthetic.default_hyperparameters(y)
M = args.evals
d = np.concatenate([np.zeros((M, 1)), np.linspace(0, 20, M)[:, None] * np.ones((1, args.bands - 1))], 1)
with gpcc_amd.Objective(t, y, s, "matern32", slots_per_stream=16) as obj:
    obj.set_option("chain_trace", 1)
    obj.set_option("chain_work_max", 1 << 30)   # (whatever the group size: the kernel itself is looked at)
    for kv in args.opt:
        obj.set_option(kv.split("=")[0], int(kv.split("=")[1]))
    for _ in range(3):
        ll, info = obj.loglik_batch(d, np.tile(alpha, (M, 1)), np.full(M, rho))
    for m in range(M):
        tr = obj.chain_trace(m)
        nt = len(tr)
        print("== N = %d, evaluation %d of %d in the group: %d diagonal steps, chain %.1f us from the first stamp to the last" % (
            args.n_per_band * args.bands, m, M, nt, tr[:, 2].max()))
        print("%4s %12s %12s %12s | %10s %10s %10s" % ("k", "build from", "image at", "published", "hand-off", "diag step", "wait+build"))
        hand, diag = [], []
        for k in range(nt):
            prev_end = tr[k - 1, 2] if k else 0.0
            h = tr[k, 1] - prev_end          # end of step k-1 -> first pivot of step k can start
            dg = tr[k, 2] - tr[k, 1]
            hand.append(h)
            diag.append(dg)
            if (k % args.every == 0 or k == nt - 1) and not args.brief:
                print("%4d %12.2f %12.2f %12.2f | %10.2f %10.2f %10.2f" % (k, tr[k, 0], tr[k, 1], tr[k, 2], h, dg, tr[k, 1] - tr[k, 0]))
        ks = nt // 2
        if args.brief:
            print("sum of hand-offs %.1f us, sum of diagonal steps %.1f us" % (sum(hand[1:]), sum(diag)))
            continue
        b = tr[ks, 8:80].reshape(9, 8)
        print("block steps of diagonal step %d (us): wave 0 [fold | load block | 16 pivots | store] | wait at barrier 1 | inverse row + panel | total" % ks)
        for jb in range(9):
            nxt = b[jb + 1, 0] if jb < 8 else tr[ks, 2]
            w0 = "%5.2f | %5.2f | %5.2f | %5.2f" % (b[jb, 4] - b[jb, 0], b[jb, 5] - b[jb, 4], b[jb, 6] - b[jb, 5], b[jb, 7] - b[jb, 6]) if jb < 8 else "%5.2f (last row of the inverse, W)     " % (b[jb, 1] - b[jb, 0])
            print("   jb %d: %s | %5.2f | %5.2f | %5.2f" % (jb, w0, b[jb, 2] - b[jb, 1], (b[jb, 3] if jb < 8 else nxt) - b[jb, 2], nxt - b[jb, 0]))
        print("sum of hand-offs %.1f us (mean %.2f), sum of diagonal steps %.1f us (mean %.2f)" % (
            sum(hand[1:]), np.mean(hand[1:]) if nt > 1 else 0.0, sum(diag), np.mean(diag)))
        if nt > 2:
            print("the hand-off in detail (us after the first stamp of step k's last 16 pivots [block step 7 begins]): inv(D_7) flagged | the last of the four quarters of tile (k+1,k): column block 6 out, "
                  "inv(D_7) seen, column block 7 out | chain of step k+1: block 7 seen, image complete | step k published")
            for k in range(nt // 2 - 2, nt // 2 + 3):
                o = tr[k, 8 + 8 * 7]
                print("   k=%2d: %6.2f | %6.2f %6.2f %6.2f | %6.2f %6.2f | %6.2f" % (k, tr[k, 3] - o, tr[k, 6] - o, tr[k, 4] - o, tr[k, 5] - o, tr[k + 1, 7] - o, tr[k + 1, 1] - o, tr[k, 2] - o))
    jt = obj.chain_jobs_trace()
    if len(jt):
        print("== workers: %d jobs stamped (whole group)" % len(jt))
        for kind, name in ((1, "quarter-tile solve"), (2, "tile update"), (3, "quarter-tile update"), (4, "tile update, column block")):
            r = jt[jt[:, 0] == kind]
            if len(r):
                wait, run = r[:, 4] - r[:, 3], r[:, 5] - r[:, 4]
                print("   %-18s: %6d jobs, waiting for dependencies mean %6.2f us (median %6.2f, max %7.2f), running mean %6.2f us (median %6.2f, max %6.2f)" % (
                    name, len(r), wait.mean(), np.median(wait), wait.max(), run.mean(), np.median(run), run.max()))
        span = jt[:, 5].max() - jt[:, 3].min()
        busy = (jt[:, 5] - jt[:, 4]).sum()
        print("   span %.1f us; sum of running time %.1f us = %.1f workgroups busy on average" % (span, busy, busy / span))
        if M == 1 and not args.brief:
            tr = obj.chain_trace(0)
            print("== what the chain of step k waited for (us, same clock): step k-1 published | last quarter solve of tile (k,k-1) claimed / inputs there / done |")
            print("   update of (k,k-1) by column k-2 done | update of (k,k) by column k-2 done | image of tile (k,k) complete")
            for k in range(2, min(nt, 14)):
                sol = jt[(jt[:, 0] == 1) & (jt[:, 1] == k - 1) & (jt[:, 2] // 1024 == k)]
                u1 = jt[(jt[:, 0] == 3) & (jt[:, 1] == k - 2) & (jt[:, 2] // 1024 == k)]   # the four quarter updates of tile (k,k-1)
                u2 = jt[(jt[:, 0] == 2) & (jt[:, 1] == k - 2) & (jt[:, 2] == k * 1024 + k)]
                f = lambda r, c: ("%8.1f" % r[:, c].max()) if len(r) else "       -"
                print("   k=%2d: %8.1f | %s / %s / %s | %s | %s | %8.1f" % (k, tr[k - 1, 2], f(sol, 3), f(sol, 4), f(sol, 5), f(u1, 5), f(u2, 5), tr[k, 1]))
