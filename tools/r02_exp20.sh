#!/bin/bash
# Round-2 batch 20: fewer resident workgroups per CU for the fused kernel (ART_CHAIN_DYN_LDS), with the box's bare-pattern
# figures beside it.
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/exp20
mkdir -p $OUT
cd $REPO
timeout -k 10 200 ./tools/_build/stream_floor 10000000 > $OUT/floor.log 2>&1 || exit 1
grep "pass 1" -A120 $OUT/floor.log | grep "E=4" | grep "soa  nt stores\|workgroups per CU\|copy"
for rep in 1 2; do
  for lds in 0 20480 33000 60000; do
    ART_CHAIN_DYN_LDS=$lds ART_DIAG_TAG=dynlds_$lds timeout -k 10 200 python tools/diag_bench.py > $OUT/t_${lds}_$rep.log 2>&1 || exit 1
    grep "chain" $OUT/t_${lds}_$rep.log
    ART_CHAIN_DYN_LDS=$lds ART_DIAG_TAG=dynlds_$lds timeout -k 10 300 python tools/fused_time.py > $OUT/f_${lds}_$rep.log 2>&1 || exit 1
    grep "fused" $OUT/f_${lds}_$rep.log | tail -1
  done
done
