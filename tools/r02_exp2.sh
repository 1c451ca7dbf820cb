#!/bin/bash
# Round-2 experiment batch 2 (GPU box): workgroup->tile mapping sweep (ART_XCD_MAP), LDS-staged stores on top of it,
# memory floors, defect-kernel occupancy, then the new bench.py on every configuration.
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/exp2
mkdir -p $OUT
cd $REPO
step() {
  local t=$1 log=$2; shift 2
  timeout -k 10 $t "$@" > $OUT/$log 2>&1
  local rc=$?
  echo "== $log rc=$rc"
  if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "TIMEOUT: stopping the batch"; exit 1; fi
  return 0
}
V=$REPO/build/variants
for m in 0 -1 2 3 5 7 9 11 -1 0; do
  ART_XCD_MAP=$m ART_DIAG_TAG=xcd_map_$m step 200 t_map_$m.log python tools/diag_bench.py
  grep -h "ms per" $OUT/t_map_$m.log
done
ART_XCD_MAP=-1 ART_DIAG_CHECK=1 ART_DIAG_TAG=lds4_map-1 ART_HIP_LIB=$V/libart_lds4.so step 200 t_lds4.log python tools/diag_bench.py
ART_XCD_MAP=5 ART_DIAG_TAG=lds4_map5 ART_HIP_LIB=$V/libart_lds4.so step 200 t_lds4b.log python tools/diag_bench.py
ART_XCD_MAP=-1 ART_DIAG_TAG=nocompute_map-1 ART_HIP_LIB=$V/libart_nocompute.so step 200 t_nocompute.log python tools/diag_bench.py
ART_XCD_MAP=-1 ART_DIAG_TAG=lds4_nocompute_map-1 ART_HIP_LIB=$V/libart_lds4_nocompute.so step 200 t_lds4_nocompute.log python tools/diag_bench.py
grep -h "ms per\|==" $OUT/t_lds4*.log $OUT/t_nocompute.log
ART_DIAG_RAYS=1000000 ART_XCD_MAP=0 ART_DIAG_TAG=1e6_map0 step 200 t_1e6_0.log python tools/diag_bench.py
ART_DIAG_RAYS=1000000 ART_XCD_MAP=-1 ART_DIAG_TAG=1e6_map-1 step 200 t_1e6_1.log python tools/diag_bench.py
ART_DIAG_RAYS=100000000 ART_XCD_MAP=0 ART_DIAG_TAG=1e8_map0 step 300 t_1e8_0.log python tools/diag_bench.py
ART_DIAG_RAYS=100000000 ART_XCD_MAP=-1 ART_DIAG_TAG=1e8_map-1 step 300 t_1e8_1.log python tools/diag_bench.py
grep -h "ms per" $OUT/t_1e*.log
ART_DEFECT_WAVES=4 step 300 c5_w4.log python tools/c5_time.py
ART_DEFECT_WAVES=5 step 300 c5_w5.log python tools/c5_time.py
echo "--- c5 defect waves 4"; grep "order 6 \|order 16\|plain" $OUT/c5_w4.log
echo "--- c5 defect waves 5"; grep "order 6 \|order 16\|plain" $OUT/c5_w5.log
step 400 bench_default.log python bench.py --steps 50
tail -1 $OUT/bench_default.log
for c in C2 C3 C4 C5; do
  step 400 bench_$c.log python bench.py --config $c --steps 20 --warmup 5
  tail -c 2500 $OUT/bench_$c.log
done
ART_FORCE_DIST=1 step 400 bench_dist1.log python bench.py --steps 20 --cpu-sample 0
tail -c 1500 $OUT/bench_dist1.log
