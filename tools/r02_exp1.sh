#!/bin/bash
# Round-2 experiment batch 1 (GPU box): GPU test suite, then the store-path / XCD / pitch / Zernike variants.
# A step that times out ends the batch (no further GPU work after a hang).
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/exp1
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
step 900 pytest.log python -m pytest tests -m gpu -x -q
tail -15 $OUT/pytest.log
V=$REPO/build/variants
step 200 t_default.log python tools/diag_bench.py
ART_DIAG_CHECK=1 ART_HIP_LIB=$V/libart_lds4.so step 200 t_lds4.log python tools/diag_bench.py
ART_DIAG_CHECK=1 ART_HIP_LIB=$V/libart_xcd.so step 200 t_xcd.log python tools/diag_bench.py
ART_HIP_LIB=$V/libart_nocompute.so step 200 t_nocompute.log python tools/diag_bench.py
ART_HIP_LIB=$V/libart_lds4_nocompute.so step 200 t_lds4_nocompute.log python tools/diag_bench.py
for k in 1 3 8; do
  ART_PITCH_EXTRA=$k ART_DIAG_TAG=pitch_extra_$k step 200 t_pitch$k.log python tools/diag_bench.py
done
step 200 t_default2.log python tools/diag_bench.py
step 300 c5_default.log python tools/c5_time.py
ART_HIP_LIB=$V/libart_zernlds.so step 300 c5_zernlds.log python tools/c5_time.py
grep -h "ms per\|==" $OUT/t_*.log
echo "--- c5 default"; cat $OUT/c5_default.log | tail -22
echo "--- c5 zern lds"; cat $OUT/c5_zernlds.log | tail -22
