#!/bin/bash
# Round-2 batch 4 (GPU box): GPU suite with the fused read-out, bench fused vs separate, alive-store diagnostic,
# profiles of every configuration (fused read-out) + relay4 with the separate read-out.
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/exp4
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
tail -25 $OUT/pytest.log
step 400 bench_20.log python bench.py --steps 20 --warmup 5;  short $OUT/bench_20.log
step 400 bench_100.log python bench.py --cpu-sample 0; short $OUT/bench_100.log
step 400 bench_20_sep.log python bench.py --steps 20 --warmup 5 --readout separate --cpu-sample 0; short $OUT/bench_20_sep.log
step 400 bench_100_sep.log python bench.py --readout separate --cpu-sample 0; short $OUT/bench_100_sep.log
for c in C2 C3 C4 C5; do
  step 400 bench_$c.log python bench.py --config $c --steps 20 --warmup 5; short $OUT/bench_$c.log
done
V=$REPO/build/variants
ART_DIAG_TAG=default step 200 t_default.log python tools/diag_bench.py
ART_DIAG_TAG=noalive ART_HIP_LIB=$V/libart_noalive.so step 200 t_noalive.log python tools/diag_bench.py
ART_DIAG_TAG=nocompute ART_HIP_LIB=$V/libart_nocompute.so step 200 t_nocompute.log python tools/diag_bench.py
ART_DIAG_TAG=default2 step 200 t_default2.log python tools/diag_bench.py
grep -h "ms per" $OUT/t_*.log
for c in relay4 C2 C3 C4 C5; do
  step 900 prof_$c.log bash tools/prof.sh r02_$c --config $c --steps 20 --warmup 5
  n=10000000; [ $c = C2 ] && n=1000000; [ $c = C4 ] && n=12500000
  step 120 sum_$c.log python tools/summarize_profile.py gpurun_out/prof_r02_$c gpurun_out/prof_r02_$c/r02_$c.md $n "--config $c --steps 20 --warmup 5"
  grep -h "k_trace\|k_detector_readout\|calibration" $OUT/sum_$c.log | head -8
done
step 900 prof_relay4_separate.log bash tools/prof.sh r02_relay4_separate --readout separate --steps 20 --warmup 5
step 120 sum_relay4_separate.log python tools/summarize_profile.py gpurun_out/prof_r02_relay4_separate gpurun_out/prof_r02_relay4_separate/r02_relay4_separate.md 10000000 "--readout separate --steps 20 --warmup 5"
grep -h "k_trace\|k_detector_readout\|calibration" $OUT/sum_relay4_separate.log | head -8
