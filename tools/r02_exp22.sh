#!/bin/bash
# Round-2 batch 22: the 16-byte-store build (-DART_STORE_LDS4, staging tile aliased with the tail's reduction tiles: 91
# VGPRs, 20 KB LDS, 5 waves) against the shipped 8-byte stores, now that the kernel is no longer VALU-bound.
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/exp22
mkdir -p $OUT
cd $REPO
L=$REPO/build/variants/libart_lds4.so
for rep in 1 2; do
  ART_DIAG_TAG=default timeout -k 10 200 python tools/diag_bench.py 2>&1 | grep "chain  " || exit 1
  ART_DIAG_CHECK=1 ART_HIP_LIB=$L ART_DIAG_TAG=lds4 timeout -k 10 200 python tools/diag_bench.py 2>&1 | grep "chain" || exit 1
  ART_DIAG_TAG=default timeout -k 10 300 python tools/fused_time.py 2>&1 | grep fused | tail -1
  ART_HIP_LIB=$L ART_DIAG_TAG=lds4 timeout -k 10 300 python tools/fused_time.py 2>&1 | grep fused | tail -1
done
for c in relay4 C2 C3 C4; do
  for lib in "" $L; do
    if [ -n "$lib" ]; then export ART_HIP_LIB=$lib; tag=lds4; else unset ART_HIP_LIB; tag=default; fi
    timeout -k 10 300 python bench.py --config $c --cpu-sample 0 2>/dev/null | python3 -c "import json,sys; j=json.loads(sys.stdin.read()); print('$c $tag value %.3e ms %.4f kernel_ms %.4f sustained %.3e' % (j['value'], j['ms_per_step'], j['roofline']['kernel_ms'], j['value_sustained']))" || exit 1
  done
done
unset ART_HIP_LIB
./tools/_build/stream_floor 10000000 | grep "pass 1" -A60 | grep "soa  nt stores" | grep "E=4"
