#!/bin/bash
# SQ counter pass (own run, PMC only + kernel trace): instruction counts and wait/issue cycle split per kernel.
#   tools/prof_sq.sh TAG [bench args...]                       the bench line
#   PROF_SCRIPT=tools/e2e_time.py tools/prof_sq.sh TAG [args]  another script of this tree (its kernels, its arguments)
set -e
TAG=$1; shift
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/prof_$TAG
mkdir -p $OUT
[ -f $OUT/box_state.txt ] || $REPO/tools/box_state.sh $OUT/box_state.txt
python3 $REPO/tools/source_hash.py > $OUT/source_hash.txt
cd /tmp && export TMPDIR=/tmp
if [ -n "$PROF_SCRIPT" ]; then
  CMD="$REPO/$PROF_SCRIPT"
else
  CMD="$REPO/bench.py --cpu-sample 0"
fi
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --kernel-trace --output-format csv -d $OUT/pmc_sq -- python3 $CMD "$@" > $OUT/bench_sq.json 2> $OUT/sq.err
rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VMEM GRBM_GUI_ACTIVE SQ_INSTS_SMEM SQ_WAIT_INST_LDS --kernel-trace --output-format csv -d $OUT/pmc_sq2 -- python3 $CMD "$@" > $OUT/bench_sq2.json 2> $OUT/sq2.err || true
