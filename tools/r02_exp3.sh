#!/bin/bash
# Round-2 batch 3 (GPU box): bench queue-depth check, rocprofv3 trace + PMC passes for every configuration, sweep.
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/exp3
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
short() { python3 - "$1" <<'PY'
import json, sys
try:
    j = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    r = j["roofline"]
    print({k: j[k] for k in ("value", "ms_per_step", "host_enqueue_ms_per_step")}, "kernel_ms", r["kernel_ms"], "frac", r["frac"],
          "frac_alg", r["frac_algorithmic"], "readout_ms", j["roofline_readout"]["kernel_ms"])
except Exception as e:
    print("unreadable", sys.argv[1], e)
PY
}
step 400 bench_100.log python bench.py --steps 100 --cpu-sample 0;  short $OUT/bench_100.log
ART_BENCH_EVENT_STEPS=100 step 400 bench_100_allev.log python bench.py --steps 100 --cpu-sample 0; short $OUT/bench_100_allev.log
step 400 bench_20.log python bench.py --steps 20 --warmup 5 --cpu-sample 0; short $OUT/bench_20.log
step 400 bench_C5.log python bench.py --config C5 --steps 20 --warmup 5 --cpu-sample 0; short $OUT/bench_C5.log
for c in relay4 C2 C3 C4 C5; do
  step 900 prof_$c.log bash tools/prof.sh r02_$c --config $c --steps 20 --warmup 5
  n=10000000; [ $c = C2 ] && n=1000000; [ $c = C4 ] && n=12500000
  step 120 sum_$c.log python tools/summarize_profile.py gpurun_out/prof_r02_$c gpurun_out/prof_r02_$c/r02_$c.md $n "--config $c --steps 20 --warmup 5"
  grep -h "k_trace\|k_detector_readout\|calibration" $OUT/sum_$c.log | head -8
done
step 900 sweep.log python tools/sweep.py gpurun_out/exp3/r02_sweep.md
tail -45 $OUT/sweep.log
