#!/bin/bash
# Round-2 batch 13: fewer VALU instructions per intersection (constants through scalar operands, raw sqrt seed,
# coupled sqrt/rsqrt in the torus function, half the Kahan norms, frame offsets prepared on the host).
# A/B against the previous build (build/variants/libart_r2a.so) on one box, then the GPU suite.
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/exp13
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
ART_DIAG_CHECK=1 step 200 t_new1.log python tools/diag_bench.py
ART_HIP_LIB=$V/libart_r2a.so step 200 t_old1.log python tools/diag_bench.py
step 200 t_new2.log python tools/diag_bench.py
ART_HIP_LIB=$V/libart_r2a.so step 200 t_old2.log python tools/diag_bench.py
grep -h "ms per\|==" $OUT/t_*.log
step 300 kind_new.log python tools/kind_time.py
ART_HIP_LIB=$V/libart_r2a.so step 300 kind_old.log python tools/kind_time.py
echo "--- kind new"; tail -12 $OUT/kind_new.log
echo "--- kind old"; tail -12 $OUT/kind_old.log
step 900 pytest.log python -m pytest tests -m gpu -x -q
tail -8 $OUT/pytest.log
step 300 bench.log python bench.py
tail -1 $OUT/bench.log | cut -c1-600
