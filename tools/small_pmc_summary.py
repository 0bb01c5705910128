#!/usr/bin/env python3
"""Condenses tools/small_pmc.sh (rocprofv3 --pmc passes of the small-N kernels) into one JSON for profiles/: per kernel and
configuration the instruction mix per evaluation, the busy fractions of the fp64 pipe (MFMA) and of VALU issue, LDS wait, scratch
instructions, waves per SIMD.  SQ_* counters are summed over all SEs/XCDs by rocprofv3; *_CYCLES of the SQ count quad-cycles except
SQ_VALU_MFMA_BUSY_CYCLES and SQ_BUSY_CYCLES (MI355X_MICROARCH.md, cycle constants)."""
import collections, csv, glob, json, os, re, sys


def short(name):
    return name.split("(")[0].replace("void ", "").strip()


def main():
    d = sys.argv[1]
    out = {"source": "tools/small_pmc.sh -> tools/small_pmc_summary.py", "configs": {}}
    for cfg in sorted(set(re.match(r".*/(cfg\d+)_pass\d+$", p).group(1) for p in glob.glob(d + "/cfg*_pass*") if os.path.isdir(p))):
        agg = collections.defaultdict(lambda: collections.defaultdict(float))
        ndisp = collections.defaultdict(int)
        dur = collections.defaultdict(float)
        grid = {}
        for pas in sorted(glob.glob("%s/%s_pass*" % (d, cfg))):
            if not os.path.isdir(pas):
                continue
            seen = collections.defaultdict(set)
            for f in glob.glob(pas + "/*/*_counter_collection.csv"):
                for r in csv.DictReader(open(f)):
                    k = short(r["Kernel_Name"])
                    if "gpcc_small" not in k:
                        continue
                    agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
                    if r["Dispatch_Id"] not in seen[k]:
                        seen[k].add(r["Dispatch_Id"])
                        grid[k] = int(r.get("Grid_Size", 0) or 0) // max(int(r.get("Workgroup_Size", 1) or 1), 1)
                        if pas.endswith("pass1"):
                            dur[k] += float(r["End_Timestamp"]) - float(r["Start_Timestamp"])
            for k, v in seen.items():
                ndisp[k] = max(ndisp[k], len(v))
        log = open("%s/%s_pass1.log" % (d, cfg)).read()
        line = [l for l in log.splitlines() if l.startswith('{"metric"')]
        bench = json.loads(line[-1]) if line else {}
        entry = {"bench": {k: bench.get(k) for k in ("value", "ms_per_step", "config")}, "kernels": {}}
        for k, c in agg.items():
            n = max(ndisp[k], 1)
            evals = grid.get(k, 0)                     # one workgroup per evaluation
            ns = dur[k]
            e = {"launches": n, "evaluations_per_launch": evals, "avg_launch_us": ns / n / 1e3 if ns else None}
            if ns and c.get("GRBM_GUI_ACTIVE"):
                clk = c["GRBM_GUI_ACTIVE"] / 8.0 / ns          # GHz
                cycles = ns * clk
                e["eff_clock_ghz"] = round(clk, 3)
                e["mfma_busy_frac"] = round(c["SQ_VALU_MFMA_BUSY_CYCLES"] / (cycles * 1024.0), 4)
                e["waves_per_simd_avg"] = round(4.0 * c["SQ_WAVE_CYCLES"] / (cycles * 1024.0), 2)
                e["wave_wait_any_frac"] = round(c["SQ_WAIT_ANY"] / max(c["SQ_WAVE_CYCLES"], 1.0), 4)
                e["wave_wait_inst_frac"] = round(c["SQ_WAIT_INST_ANY"] / max(c["SQ_WAVE_CYCLES"], 1.0), 4)
                e["wave_active_inst_frac"] = round(c["SQ_ACTIVE_INST_ANY"] / max(c["SQ_WAVE_CYCLES"], 1.0), 4)
                e["valu_issue_busy_frac"] = round(4.0 * c.get("SQ_ACTIVE_INST_VALU", 0.0) / (cycles * 1024.0), 4)
                e["lds_issue_busy_frac"] = round(4.0 * c.get("SQ_ACTIVE_INST_LDS", 0.0) / (cycles * 1024.0), 4)
                e["mfma_valu_coexec_frac"] = round(c.get("SQ_VALU_MFMA_COEXEC_CYCLES", 0.0) / (cycles * 1024.0), 4)
            tot = max(evals * n, 1)
            per = lambda name: round(c.get(name, 0.0) / tot, 1)
            e["wave_instructions_per_evaluation"] = {
                "valu": per("SQ_INSTS_VALU"), "mfma": per("SQ_INSTS_MFMA"), "lds": per("SQ_INSTS_LDS"), "salu": per("SQ_INSTS_SALU"),
                "vmem_rd": per("SQ_INSTS_VMEM_RD"), "vmem_wr": per("SQ_INSTS_VMEM_WR"), "flat_incl_scratch": per("SQ_INSTS_FLAT"),
                "fma_f64": per("SQ_INSTS_VALU_FMA_F64"), "mul_f64": per("SQ_INSTS_VALU_MUL_F64"), "add_f64": per("SQ_INSTS_VALU_ADD_F64"),
                "trans_f64": per("SQ_INSTS_VALU_TRANS_F64"), "int32": per("SQ_INSTS_VALU_INT32"), "int64": per("SQ_INSTS_VALU_INT64"), "cvt": per("SQ_INSTS_VALU_CVT")}
            e["lds_bank_conflict_cycles_per_evaluation"] = per("SQ_LDS_BANK_CONFLICT")
            e["lds_wait_inst_frac"] = round(c.get("SQ_WAIT_INST_LDS", 0.0) / max(c.get("SQ_WAVE_CYCLES", 0.0), 1.0), 4) if c.get("SQ_WAVE_CYCLES") else None
            entry["kernels"][k] = e
        out["configs"][cfg] = entry
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
