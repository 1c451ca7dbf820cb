#!/bin/bash
# Profiling recipe used for profiles/ (run on the GPU box through gpurun):
#   tools/prof.sh <tag> [bench args...]
# Pass 1: kernel trace + stats.  Pass 2/3: HBM counters, each in its own --pmc run (FETCH_SIZE and WRITE_SIZE do
# not fit one pass on gfx950; see MI355X_MICROARCH.md "rocprofv3 PMC slots").
set -e
TAG=$1; shift
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/prof_$TAG
mkdir -p $OUT
$REPO/tools/box_state.sh $OUT/box_state.txt
python3 $REPO/tools/source_hash.py > $OUT/source_hash.txt
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $REPO/bench.py --cpu-sample 0 "$@" > $OUT/bench_trace.json 2> $OUT/trace.err
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_fetch -- python3 $REPO/bench.py --cpu-sample 0 "$@" > $OUT/bench_fetch.json 2> $OUT/fetch.err
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pmc_write -- python3 $REPO/bench.py --cpu-sample 0 "$@" > $OUT/bench_write.json 2> $OUT/write.err
find $OUT -name "*.csv" | head -20
