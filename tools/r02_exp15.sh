#!/bin/bash
# Round-2 batch 15: torus solver with a square-root-free first step (Newton on the expanded quartic) -- A/B against
# build/variants/libart_r2b.so (the build before it), fuzz against the oracle, GPU suite.
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/exp15
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
ART_DIAG_TAG=new step 200 t_new1.log python tools/diag_bench.py
ART_DIAG_TAG=r2b ART_HIP_LIB=$V/libart_r2b.so step 200 t_old1.log python tools/diag_bench.py
ART_DIAG_TAG=new step 200 t_new2.log python tools/diag_bench.py
ART_DIAG_TAG=r2b ART_HIP_LIB=$V/libart_r2b.so step 200 t_old2.log python tools/diag_bench.py
grep -h "ms per" $OUT/t_*.log
step 120 floor.log ./tools/_build/stream_floor 10000000
grep "pass 1" -A40 $OUT/floor.log | grep "E=4"
step 300 bench.log python bench.py
tail -1 $OUT/bench.log | cut -c1-300
step 800 fuzz.log python tools/gpu_fuzz.py 30000000 30000
tail -3 $OUT/fuzz.log
