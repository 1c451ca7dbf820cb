#!/bin/bash
REPO=${GRAFT_REPO_ROOT:-/root/repo}
cd $REPO
timeout -k 10 600 bash tools/prof_sq.sh r02_relay4_sq --steps 20 --warmup 5 > gpurun_out/sq.log 2>&1
python3 - <<'PY'
import csv, glob, collections
for d in ("pmc_sq", "pmc_sq2"):
    fs = glob.glob(f"gpurun_out/prof_r02_relay4_sq/{d}/*/*_counter_collection.csv")
    if not fs:
        print(d, "missing"); continue
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(fs[0])):
        k = r["Kernel_Name"]
        if "k_trace_chain" in k and int(r["Grid_Size"]) > 1000000:
            acc["chain"][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for name, cs in acc.items():
        print(d, name, {c: round(sum(v) / len(v), 1) for c, v in cs.items()}, "launches", len(next(iter(cs.values()))))
PY
