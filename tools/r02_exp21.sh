#!/bin/bash
# Round-2 batch 21: the separate detector read-out with up to 16384 short-lived workgroups and the LDS-transpose
# reduction, against the round-1 persistent grid of 2048 (ART_READOUT_BLOCKS=2048) -- same binary otherwise.
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/exp21
mkdir -p $OUT
cd $REPO
timeout -k 10 600 python -m pytest tests -m gpu -x -q > $OUT/pytest.log 2>&1; tail -2 $OUT/pytest.log
for rep in 1 2; do
  for nb in 2048 4096 8192 0; do
    ART_READOUT_BLOCKS=$nb timeout -k 10 200 python tools/readout_bench.py > $OUT/rb_${nb}_$rep.log 2>&1 || { tail -3 $OUT/rb_${nb}_$rep.log; exit 1; }
    echo "blocks cap $nb:"; cat $OUT/rb_${nb}_$rep.log | grep "us/launch"
  done
done
