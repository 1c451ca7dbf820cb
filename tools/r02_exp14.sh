#!/bin/bash
# Round-2 batch 14: slimmer fused read-out tail (assigned statistics, bare v_min, rows in pass order, DPP moves without
# copies, refined-seed division / square root in detector_ray).  A/B against build/variants/libart_r2a.so (round-2 build
# before the instruction-count work), then the GPU suite and the bench line.
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/exp14
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
V=$REPO/build/variants
step 900 pytest.log python -m pytest tests -m gpu -x -q
tail -4 $OUT/pytest.log
ART_DIAG_TAG=new step 300 f_new1.log python tools/fused_time.py
ART_DIAG_TAG=old ART_HIP_LIB=$V/libart_r2a.so step 300 f_old1.log python tools/fused_time.py
ART_DIAG_TAG=new step 300 f_new2.log python tools/fused_time.py
ART_DIAG_TAG=old ART_HIP_LIB=$V/libart_r2a.so step 300 f_old2.log python tools/fused_time.py
grep -h "ms per" $OUT/f_*.log
step 300 bench.log python bench.py
tail -1 $OUT/bench.log | cut -c1-300
for c in C2 C3 C4 C5; do
  step 300 bench_$c.log python bench.py --config $c --cpu-sample 0
  tail -1 $OUT/bench_$c.log | python3 -c "import json,sys; j=json.loads(sys.stdin.read()); print('$c', 'value %.3e' % j['value'], 'ms %.3f' % j['ms_per_step'], 'frac', j['roofline']['frac'], 'kernel_ms', j['roofline']['kernel_ms'])"
done
