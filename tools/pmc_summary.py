#!/usr/bin/env python3
"""Condenses the rocprofv3 output of tools/profile_round.sh into one JSON (committed under profiles/).

HBM traffic follows /opt/skills/guides/MI355X_MICROARCH.md section HBM: FETCH_SIZE and WRITE_SIZE are in
KiB; on gfx950 FETCH_SIZE reports exactly half of the bytes of wide (16 B/lane) coalesced streaming
reads, which is what the LDS-DMA operand streams are, so the read side is doubled; WRITE_SIZE is
exact for 16-byte-per-lane stores.  Counters come from separate --pmc passes."""
import collections
import csv
import glob
import json
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from gpcc_amd.build import build_info_string  # noqa: E402  (the library the counters were taken on: bench.py quotes them only for the same build)


def short(name):
    return name.split("(")[0].replace("void ", "").strip()


def counters(d, tag):
    files = glob.glob("%s/pmc_%s/*/*_counter_collection.csv" % (d, tag))
    agg = collections.defaultdict(lambda: collections.defaultdict(float))
    disp = collections.defaultdict(set)
    dur = collections.defaultdict(float)
    for f in files:
        for r in csv.DictReader(open(f)):
            k = short(r["Kernel_Name"])
            agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
            if r["Dispatch_Id"] not in disp[k]:
                disp[k].add(r["Dispatch_Id"])
                dur[k] += float(r["End_Timestamp"]) - float(r["Start_Timestamp"])
    return agg, {k: len(v) for k, v in disp.items()}, dur


def main():
    d = sys.argv[1]
    # N / slots: the configuration the passes ran (bench.py's defaults) -- bench.py only quotes `traffic` from a summary that matches
    out = {"note": "PMC passes: one 256-evaluation group (bench.py --grid 256 --steps 1 --warmup 0); "
                   "stats: default bench command", "N": int(os.environ.get("GPCC_PMC_N", "4096")),
           "slots": int(os.environ.get("GPCC_PMC_SLOTS", "256")), "build": build_info_string(), "source": "tools/profile_round.sh -> tools/pmc_summary.py %s" % d,
           "kernels": {}}
    stats = glob.glob("%s/stats/*/*_kernel_stats.csv" % d)
    if stats:
        for r in csv.DictReader(open(stats[0])):
            k = short(r["Name"])
            if k.startswith("gpcc_"):
                out["kernels"].setdefault(k, {})["stats"] = {"calls": int(r["Calls"]), "avg_ms": float(r["AverageNs"]) / 1e6,
                                                             "total_ms": float(r["TotalDurationNs"]) / 1e6,
                                                             "percent": float(r["Percentage"])}
    for tag in ("GRBM_GUI_ACTIVE", "FETCH_SIZE", "WRITE_SIZE", "TCC_HIT_sum"):
        agg, n, dur = counters(d, tag)
        for k, c in agg.items():
            if not k.startswith("gpcc_"):
                continue
            e = out["kernels"].setdefault(k, {})
            e.setdefault("pmc_launches", n[k])
            if tag == "GRBM_GUI_ACTIVE":
                ns = dur[k]
                e["pmc_total_ms"] = ns / 1e6
                e["eff_clock_ghz"] = c["GRBM_GUI_ACTIVE"] / 8.0 / ns
                cycles = ns * e["eff_clock_ghz"]
                e["mfma_busy_frac"] = c["SQ_VALU_MFMA_BUSY_CYCLES"] / (cycles * 1024.0)   # 1024 SIMDs
                e["wave_cycles_wait_frac"] = c["SQ_WAIT_ANY"] / max(c["SQ_WAVE_CYCLES"], 1.0)
            elif tag == "FETCH_SIZE":
                e["hbm_read_bytes_per_launch"] = 2.0 * c["FETCH_SIZE"] * 1024.0 / n[k]
            elif tag == "WRITE_SIZE":
                e["hbm_write_bytes_per_launch"] = c["WRITE_SIZE"] * 1024.0 / n[k]
            else:
                e["l2_hit_rate"] = c["TCC_HIT_sum"] / max(c["TCC_HIT_sum"] + c["TCC_MISS_sum"], 1.0)
    for e in out["kernels"].values():
        if "hbm_read_bytes_per_launch" in e and "hbm_write_bytes_per_launch" in e:
            e["hbm_bytes_per_launch"] = e["hbm_read_bytes_per_launch"] + e["hbm_write_bytes_per_launch"]
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
