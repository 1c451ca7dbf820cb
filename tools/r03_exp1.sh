#!/bin/bash
# Round-3 batch 1: box state, GPU suite (incl. the RCCL path with one rank and the survivor records), default bench line.
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/r03_exp1
mkdir -p $OUT
cd $REPO
tools/box_state.sh $OUT/box_state.txt
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $OUT/pytest.log 2>&1; rc=$?; tail -25 $OUT/pytest.log
[ $rc -eq 0 ] || exit $rc
timeout -k 10 300 python bench.py --steps 20 --warmup 5 > $OUT/bench_20_5.json 2> $OUT/bench_20_5.err && \
python3 -c "import json; j=json.load(open('$OUT/bench_20_5.json')); print('relay4 20/5 value %.3e ms %.4f kernel_ms %.4f frac %s sustained %.3e' % (j['value'], j['ms_per_step'], j['roofline']['kernel_ms'], j['roofline']['frac'], j['value_sustained']))"
tools/box_state.sh $OUT/box_state_after.txt
