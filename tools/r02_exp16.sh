#!/bin/bash
# Round-2 batch 16: 6 waves per SIMD (80 VGPRs, no spills since the lane id is derived from the slot register) against
# 5 (85 VGPRs) for the fused kernel without defects.
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/exp16
mkdir -p $OUT
cd $REPO
step() {  # step <seconds> <logfile> <cmd...>
  local t=$1 log=$2; shift 2
  echo "== $* (log $log)"
  timeout -k 10 $t "$@" > $OUT/$log 2>&1
  local rc=$?
  echo "   rc=$rc"
  if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "TIMEOUT: stopping the batch"; exit 1; fi
  return 0
}
step 600 pytest.log python -m pytest tests/test_gpu_parity.py -m gpu -x -q
tail -2 $OUT/pytest.log
ART_CHAIN_WAVES=6 step 600 pytest6.log python -m pytest tests/test_gpu_parity.py -m gpu -x -q
tail -2 $OUT/pytest6.log
for rep in 1 2; do
  ART_DIAG_TAG=waves5 step 200 t5_$rep.log python tools/diag_bench.py
  ART_CHAIN_WAVES=6 ART_DIAG_TAG=waves6 step 200 t6_$rep.log python tools/diag_bench.py
  ART_DIAG_TAG=waves5 step 300 f5_$rep.log python tools/fused_time.py
  ART_CHAIN_WAVES=6 ART_DIAG_TAG=waves6 step 300 f6_$rep.log python tools/fused_time.py
done
grep -h "ms per" $OUT/t*_*.log $OUT/f*_*.log | grep -v element
for c in C2 C4; do
  for w in 5 6; do
    ART_CHAIN_WAVES=$w step 300 bench_${c}_$w.log python bench.py --config $c --cpu-sample 0
    tail -1 $OUT/bench_${c}_$w.log | python3 -c "import json,sys; j=json.loads(sys.stdin.read()); print('$c waves $w', 'value %.3e' % j['value'], 'ms %.3f' % j['ms_per_step'], 'kernel_ms', j['roofline']['kernel_ms'])"
  done
done
