#!/bin/bash
# Arbitrary PMC passes of a bench line (one rocprofv3 --pmc run per group of <= 8 SQ counters, kernel trace only):
#   tools/prof_counters.sh TAG "CNT_A CNT_B ...|CNT_C ..." [bench args...]
# Environment knobs of the library (ART_CHAIN_RPL=..., ART_CHAIN_SPECIAL=...) are exported by the caller: rocprofv3 must start
# python3 itself (no env / bash hop between the profiler and the program).  Output: gpurun_out/cnt_TAG/pass<k>/...
set -e
TAG=$1; GROUPS_=$2; shift 2
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/cnt_$TAG
mkdir -p $OUT
python3 $REPO/tools/source_hash.py > $OUT/source_hash.txt
cd /tmp && export TMPDIR=/tmp
k=0
IFS='|' read -ra GR <<< "$GROUPS_"
for g in "${GR[@]}"; do
  rocprofv3 --pmc $g --kernel-trace --output-format csv -d $OUT/pass$k -- python3 $REPO/bench.py --cpu-sample 0 "$@" > $OUT/bench_$k.json 2> $OUT/pass$k.err
  k=$((k+1))
done
