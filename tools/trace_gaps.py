#!/usr/bin/env python3
"""Summarises a rocprofv3 kernel_trace.csv: per kernel name count / mean duration, and the idle gaps between consecutive kernels
of the last evaluation (python tools/trace_gaps.py <kernel_trace.csv> [n_last_kernels])."""
import csv
import sys
from collections import defaultdict

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
n = int(sys.argv[2]) if len(sys.argv) > 2 else 64
last = rows[-n:]
dur = defaultdict(list)
for r in last:
    dur[r["Kernel_Name"].split("(")[0].replace("void ", "")].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
gaps = [int(b["Start_Timestamp"]) - int(a["End_Timestamp"]) for a, b in zip(last[:-1], last[1:])]
span = int(last[-1]["End_Timestamp"]) - int(last[0]["Start_Timestamp"])
print("last %d kernels: span %.1f us, busy %.1f us, gaps %.1f us (mean gap %.2f us)" %
      (n, span / 1e3, sum(sum(v) for v in dur.values()) / 1e3, sum(gaps) / 1e3, sum(gaps) / len(gaps) / 1e3))
for k, v in sorted(dur.items(), key=lambda kv: -sum(kv[1])):
    print("  %-50s %3d x %7.2f us (min %.2f max %.2f)" % (k[:50], len(v), sum(v) / len(v) / 1e3, min(v) / 1e3, max(v) / 1e3))
