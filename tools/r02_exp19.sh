#!/bin/bash
# Round-2 batch 19: the bench lines of the final build on ONE box -- every configuration as the driver calls it
# (--steps 20 --warmup 5) and with the defaults (100 / 10), fused and separate read-out, the access pattern's floor
# (tools/stream_floor) in between.
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/exp19
mkdir -p $OUT
cd $REPO
run() {  # run <name> <bench args...>
  local name=$1; shift
  timeout -k 10 300 python bench.py --cpu-sample 0 "$@" > $OUT/$name.json 2> $OUT/$name.err
  local rc=$?
  if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "TIMEOUT $name"; exit 1; fi
  tail -1 $OUT/$name.json | python3 -c "
import json,sys
j=json.loads(sys.stdin.read()); r=j['roofline']; ro=j.get('roofline_readout') or {}
print('%-18s value %.3e sustained %.3e ms/step %.4f | %s %.4f ms frac %.3f alg %.2f | readout_ms %s' % ('$name', j['value'], j.get('value_sustained') or 0, j['ms_per_step'], r['kernel'], r['kernel_ms'], r['frac'] or 0, r['frac_algorithmic'], ro.get('kernel_ms')))"
}
run relay4_20 --steps 20 --warmup 5
run relay4_100
run relay4_sep_20 --steps 20 --warmup 5 --readout separate
run relay4_sep_100 --readout separate
timeout -k 10 120 ./tools/_build/stream_floor 10000000 > $OUT/floor.log 2>&1
grep "pass 1" -A40 $OUT/floor.log | grep "E=4" | head -3
for c in C2 C3 C4 C5; do
  run ${c}_20 --config $c --steps 20 --warmup 5
  run ${c}_100 --config $c
done
run C3_sep_20 --config C3 --steps 20 --warmup 5 --readout separate
run C4_fused_20 --config C4 --steps 20 --warmup 5 --readout fused
run C2_sep_20 --config C2 --steps 20 --warmup 5 --readout separate
run relay4_1e5 --rays 100000
run relay4_1e6 --rays 1000000
run relay4_1e8 --rays 100000000 --steps 20 --warmup 3
run relay2 --mirrors 2
run relay8 --mirrors 8
run relay4_element --mode element
timeout -k 10 300 python bench.py > $OUT/default_full.json 2> $OUT/default_full.err
tail -1 $OUT/default_full.json | cut -c1-400
