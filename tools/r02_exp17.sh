#!/bin/bash
# Round-2 batch 17: fused kernel WITH defects at 5 waves per SIMD (96 VGPRs, 15 spilled dwords; build/variants/
# libart_def5.so) against the shipped 4 (112 VGPRs, none), on C5 and on the Zernike order sweep.
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/exp17
mkdir -p $OUT
cd $REPO
run() {  # run <tag> [lib]
  if [ -n "$2" ]; then export ART_HIP_LIB=$2; else unset ART_HIP_LIB; fi
  timeout -k 10 200 python bench.py --config C5 --cpu-sample 0 --steps 50 --warmup 10 > $OUT/$1.json 2> $OUT/$1.err
  rc=$?
  if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "TIMEOUT"; exit 1; fi
  tail -1 $OUT/$1.json | python3 -c "import json,sys; j=json.loads(sys.stdin.read()); print('$1', 'value %.3e' % j['value'], 'ms %.4f' % j['ms_per_step'], 'kernel_ms', j['roofline']['kernel_ms'])"
}
run waves4_a
run waves5_a $REPO/build/variants/libart_def5.so
run waves4_b
run waves5_b $REPO/build/variants/libart_def5.so
unset ART_HIP_LIB
timeout -k 10 300 python tools/c5_time.py > $OUT/c5_w4.log 2>&1 || exit 1
ART_HIP_LIB=$REPO/build/variants/libart_def5.so timeout -k 10 300 python tools/c5_time.py > $OUT/c5_w5.log 2>&1 || exit 1
echo "--- 4 waves"; grep "chain" $OUT/c5_w4.log
echo "--- 5 waves"; grep "chain" $OUT/c5_w5.log
