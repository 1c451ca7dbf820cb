#!/bin/bash
# Kernel trace of the end-to-end C3 workflow (tools/e2e_time.py): per-kernel device time of one warm pass.
#   tools/prof_e2e.sh TAG [e2e_time args...]
set -e
TAG=$1; shift
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/prof_$TAG
mkdir -p $OUT
python3 $REPO/tools/source_hash.py > $OUT/source_hash.txt
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $REPO/tools/e2e_time.py "$@" > $OUT/e2e.txt 2> $OUT/trace.err
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections, os
out = sys.argv[1]
rows = []
for f in glob.glob(os.path.join(out, "trace", "**", "*kernel_trace.csv"), recursive=True):
    rows += list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
def short(n):
    return n.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][:60]
# the last complete workflow pass: from the last k_make_source to the end
idx = [i for i, r in enumerate(rows) if "k_make_source" in r["Kernel_Name"]]
start = idx[-1]
agg = collections.OrderedDict()
t_first, t_last = int(rows[start]["Start_Timestamp"]), 0
for r in rows[start:]:
    k = short(r["Kernel_Name"])
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    a = agg.setdefault(k, [0, 0.0])
    a[0] += 1; a[1] += d
    t_last = max(t_last, int(r["End_Timestamp"]))
print("## kernels of the last workflow pass (from its k_make_source on), in order of first appearance")
print("| kernel | launches | total us |")
print("|---|---:|---:|")
tot = 0.0
for k, (c, d) in agg.items():
    print(f"| {k} | {c} | {d:.1f} |")
    tot += d
print(f"| **sum of kernel time** | | **{tot:.1f}** |")
print(f"| first launch -> last kernel end | | {(t_last - t_first) / 1e3:.1f} |")
PY
