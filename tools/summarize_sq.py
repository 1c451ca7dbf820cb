#!/usr/bin/env python3
"""Condense the SQ counter passes of tools/prof_sq.sh into profiles/<name>.{md,json}: vector instructions per wave, the
share of the launch during which the fp64 VALU issues, wait / issue split, memory instructions per wave.

usage: tools/summarize_sq.py gpurun_out/prof_<tag> profiles/<name>.md [kernel-substring] [min-grid]

busy_frac = SQ_ACTIVE_INST_VALU x 4 / (1024 SIMDs x GRBM_GUI_ACTIVE / 8): an fp64 VALU instruction occupies its SIMD for
4 cycles (16 lanes per clock), GRBM_GUI_ACTIVE is summed over the 8 XCDs (MI355X_MICROARCH.md, DVFS note)."""
import collections
import csv
import glob
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
from summarize_profile import profiled_hash  # noqa: E402


def main():
    src, dst = sys.argv[1], sys.argv[2]
    sub = sys.argv[3] if len(sys.argv) > 3 else "k_trace_chain"
    min_grid = int(sys.argv[4]) if len(sys.argv) > 4 else 1_000_000
    acc = collections.defaultdict(list)
    for d in ("pmc_sq", "pmc_sq2"):
        for f in glob.glob(os.path.join(src, d, "*", "*_counter_collection.csv")):
            for r in csv.DictReader(open(f)):
                if sub in r["Kernel_Name"] and int(r["Grid_Size"]) >= min_grid:
                    acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    c = {k: sum(v) / len(v) for k, v in acc.items()}
    if not c:
        print("no counters found for", sub)
        return 1
    waves = c.get("SQ_WAVES", 0.0)
    res = {"kernel": sub, "launches_averaged": len(next(iter(acc.values()))), "source_hash": profiled_hash(src),
           "unit": "fp64 VALU issue (16 lanes per clock per SIMD)", "counters": {k: round(v, 1) for k, v in sorted(c.items())}}
    if waves:
        res["valu_instructions_per_wave"] = round(c.get("SQ_INSTS_VALU", 0.0) / waves, 1)
        res["salu_instructions_per_wave"] = round(c.get("SQ_INSTS_SALU", 0.0) / waves, 1)
        res["vmem_rd_per_wave"] = round(c.get("SQ_INSTS_VMEM_RD", 0.0) / waves, 2)
        res["vmem_wr_per_wave"] = round(c.get("SQ_INSTS_VMEM_WR", 0.0) / waves, 2)
        res["wave_cycles_per_wave"] = round(c.get("SQ_WAVE_CYCLES", 0.0) / waves, 0)
        if c.get("SQ_WAVE_CYCLES"):
            res["wait_any_frac_of_wave_cycles"] = round(c.get("SQ_WAIT_ANY", 0.0) / c["SQ_WAVE_CYCLES"], 3)
            res["wait_inst_any_frac_of_wave_cycles"] = round(c.get("SQ_WAIT_INST_ANY", 0.0) / c["SQ_WAVE_CYCLES"], 3)
    if c.get("GRBM_GUI_ACTIVE") and c.get("SQ_ACTIVE_INST_VALU"):
        res["busy_frac"] = round(c["SQ_ACTIVE_INST_VALU"] * 4.0 / (1024.0 * c["GRBM_GUI_ACTIVE"] / 8.0), 3)
        res["note"] = "busy_frac = SQ_ACTIVE_INST_VALU x 4 / (1024 SIMDs x GRBM_GUI_ACTIVE / 8)"
    json.dump(res, open(os.path.splitext(dst)[0] + ".json", "w"), indent=1)
    lines = ["# SQ counters: " + os.path.basename(src), "", f"kernel `{sub}` (grid >= {min_grid} threads), averages over "
             f"{res['launches_averaged']} launches, sources {res['source_hash']}; two `rocprofv3 --pmc` passes (tools/prof_sq.sh)", "",
             "| quantity | value |", "|---|---:|"]
    for k, v in res.items():
        if k not in ("counters", "note", "unit", "kernel", "source_hash"):
            lines.append(f"| {k} | {v} |")
    lines += ["", "raw counters (average per launch):", "", "| counter | value |", "|---|---:|"]
    lines += [f"| {k} | {v} |" for k, v in res["counters"].items()]
    bs = os.path.join(src, "box_state.txt")
    if os.path.exists(bs):
        lines += ["", "box state before the passes: `" + os.path.relpath(bs, ROOT) + "`"]
    open(dst, "w").write("\n".join(lines) + "\n")
    print("\n".join(lines))
    return 0


if __name__ == "__main__":
    sys.exit(main())
