#!/bin/bash
# Round-2 final measurement batch (GPU box): full GPU suite, bench on every configuration, rocprofv3 trace + PMC passes
# for every configuration (as bench runs them by default) + relay4 with the fused read-out, sweep.
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/exp8
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
          "frac_alg", r["frac_algorithmic"], "readout", j["roofline_readout"].get("kernel_ms"), j.get("parity", {}).get("delay_max_rel_err"))
except Exception as e:
    print("unreadable", sys.argv[1], e)
PY
}
step 900 pytest.log python -m pytest tests -m gpu -x -q
tail -22 $OUT/pytest.log
step 400 smoke.log python -c "import __graft_entry__ as g; g.smoke()"; tail -2 $OUT/smoke.log
for c in relay4 C2 C3 C4 C5; do
  step 900 prof_$c.log bash tools/prof.sh r02_$c --config $c --steps 20 --warmup 5
  n=10000000; [ $c = C2 ] && n=1000000; [ $c = C4 ] && n=12500000
  step 120 sum_$c.log python tools/summarize_profile.py gpurun_out/prof_r02_$c gpurun_out/prof_r02_$c/r02_$c.md $n "--config $c --steps 20 --warmup 5"
  grep -h "^| k_trace\|^| k_detector_readout\|calibration" $OUT/sum_$c.log | head -8
done
step 900 prof_relay4_separate.log bash tools/prof.sh r02_relay4_separate --readout separate --steps 20 --warmup 5
step 120 sum_relay4_separate.log python tools/summarize_profile.py gpurun_out/prof_r02_relay4_separate gpurun_out/prof_r02_relay4_separate/r02_relay4_separate.md 10000000 "--readout separate --steps 20 --warmup 5"
grep -h "^| k_trace\|^| k_detector_readout" $OUT/sum_relay4_separate.log | head -4
step 900 prof_C3_separate.log bash tools/prof.sh r02_C3_separate --config C3 --readout separate --steps 20 --warmup 5
step 120 sum_C3_separate.log python tools/summarize_profile.py gpurun_out/prof_r02_C3_separate gpurun_out/prof_r02_C3_separate/r02_C3_separate.md 10000000 "--config C3 --readout separate --steps 20 --warmup 5"
grep -h "^| k_trace\|^| k_detector_readout" $OUT/sum_C3_separate.log | head -4
step 400 bench_20.log python bench.py --steps 20 --warmup 5;  short $OUT/bench_20.log
step 400 bench_100.log python bench.py; short $OUT/bench_100.log
for c in C2 C3 C4 C5; do
  step 400 bench_$c.log python bench.py --config $c --steps 20 --warmup 5; short $OUT/bench_$c.log
done
step 900 sweep.log python tools/sweep.py gpurun_out/exp8/r02_sweep.md
tail -48 $OUT/sweep.log
