#!/usr/bin/env python3
"""Average per launch of every counter collected by tools/prof_counters.sh for the kernels whose name contains a substring
(full-size launches only).  usage: tools/summarize_counters.py gpurun_out/cnt_TAG [kernel-substring] [min-grid]"""
import collections
import csv
import glob
import os
import sys


def collect(src, sub="k_trace_chain", min_grid=1_000_000):
    acc = collections.defaultdict(list)
    for f in glob.glob(os.path.join(src, "pass*", "*", "*_counter_collection.csv")):
        for r in csv.DictReader(open(f)):
            if sub in r["Kernel_Name"] and int(r["Grid_Size"]) >= min_grid:
                acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    return {k: sum(v) / len(v) for k, v in acc.items()}


if __name__ == "__main__":
    src = sys.argv[1]
    c = collect(src, sys.argv[2] if len(sys.argv) > 2 else "k_trace_chain", int(sys.argv[3]) if len(sys.argv) > 3 else 1_000_000)
    for k in sorted(c):
        print(f"{k:36s} {c[k]:16.1f}")
