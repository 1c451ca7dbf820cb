#!/bin/bash
# Kernel trace (+ optional FETCH_SIZE / WRITE_SIZE passes: --pmc) of tools/reduce_bench.py:
#   tools/prof_reduce.sh TAG [--pmc] [reduce_bench args...]
# Output: gpurun_out/prof_TAG/{trace,pmc_fetch,pmc_write}/..., summarised by tools/summarize_reduce.py
set -e
TAG=$1; shift
PMC=0
if [ "$1" = "--pmc" ]; then PMC=1; shift; fi
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/prof_$TAG
mkdir -p $OUT
$REPO/tools/box_state.sh $OUT/box_state.txt
python3 $REPO/tools/source_hash.py > $OUT/source_hash.txt
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $REPO/tools/reduce_bench.py "$@" > $OUT/bench_trace.txt 2> $OUT/trace.err
if [ $PMC = 1 ]; then
  rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_fetch -- python3 $REPO/tools/reduce_bench.py "$@" > $OUT/bench_fetch.txt 2> $OUT/fetch.err
  rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pmc_write -- python3 $REPO/tools/reduce_bench.py "$@" > $OUT/bench_write.txt 2> $OUT/write.err
fi
python3 $REPO/tools/summarize_reduce.py $OUT > $OUT/summary.md 2>&1 || true
cat $OUT/summary.md
