#!/bin/bash
# Round-2 batch 5 (GPU box): GPU suite, fused vs separate read-out, alive-store diagnostic.
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/exp5
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
step 900 pytest.log python -m pytest tests -m gpu -x -q -k "fused or scene_program or endtoend"
tail -25 $OUT/pytest.log
for rep in 1 2; do
step 400 bench_20_$rep.log python bench.py --steps 20 --warmup 5 --cpu-sample 0 --readout fused;  short $OUT/bench_20_$rep.log
step 400 bench_20_sep_$rep.log python bench.py --steps 20 --warmup 5 --readout separate --cpu-sample 0; short $OUT/bench_20_sep_$rep.log
done
step 400 bench_100.log python bench.py --cpu-sample 0 --readout fused; short $OUT/bench_100.log
step 400 bench_100_sep.log python bench.py --readout separate --cpu-sample 0; short $OUT/bench_100_sep.log
for c in C2 C3 C4 C5; do
  step 400 bench_$c.log python bench.py --config $c --steps 20 --warmup 5 --cpu-sample 0 --readout fused; short $OUT/bench_$c.log
  step 400 bench_${c}_sep.log python bench.py --config $c --steps 20 --warmup 5 --cpu-sample 0 --readout separate; short $OUT/bench_${c}_sep.log
done
V=$REPO/build/variants
ART_DIAG_TAG=default step 200 t_default.log python tools/diag_bench.py
ART_DIAG_TAG=noalive ART_HIP_LIB=$V/libart_noalive.so step 200 t_noalive.log python tools/diag_bench.py
ART_DIAG_TAG=default2 step 200 t_default2.log python tools/diag_bench.py
grep -h "ms per" $OUT/t_*.log
