#!/bin/bash
# Round-2 batch 23: 16-byte stores through LDS as the default store path of the fused kernels (dead pairs masked),
# against the build before it (build/variants/libart_r2d.so, 8-byte stores): GPU suite, every configuration, one box.
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/exp23
mkdir -p $OUT
cd $REPO
timeout -k 10 800 python -m pytest tests -m gpu -x -q > $OUT/pytest.log 2>&1; tail -3 $OUT/pytest.log
L=$REPO/build/variants/libart_r2d.so
for c in relay4 C2 C3 C4 C5; do
  for lib in "" $L "" $L; do
    if [ -n "$lib" ]; then export ART_HIP_LIB=$lib; tag=8B; else unset ART_HIP_LIB; tag=16B; fi
    timeout -k 10 300 python bench.py --config $c --cpu-sample 0 2>/dev/null | python3 -c "import json,sys; j=json.loads(sys.stdin.read()); print('$c $tag value %.3e ms %.4f kernel_ms %.4f sustained %.3e' % (j['value'], j['ms_per_step'], j['roofline']['kernel_ms'], j['value_sustained']))" || exit 1
  done
done
unset ART_HIP_LIB
for args in "--steps 20 --warmup 5" "--rays 1000000" "--rays 100000" "--mirrors 8" "--rays 100000000 --steps 20 --warmup 3"; do
  for lib in "" $L; do
    if [ -n "$lib" ]; then export ART_HIP_LIB=$lib; tag=8B; else unset ART_HIP_LIB; tag=16B; fi
    timeout -k 10 300 python bench.py $args --cpu-sample 0 2>/dev/null | python3 -c "import json,sys; j=json.loads(sys.stdin.read()); print('relay4 [$args] $tag value %.3e ms %.4f' % (j['value'], j['ms_per_step']))" || exit 1
  done
done
unset ART_HIP_LIB
./tools/_build/stream_floor 10000000 | grep "pass 1" -A60 | grep "soa  nt stores" | grep "E=4"
