#!/bin/bash
# Does the first ~25 steps' lower rate (0.81 vs 0.71 ms per step at 20 / 100 timed steps) come from warm-up?
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/exp11
mkdir -p $OUT
cd $REPO
for w in 5 5 50 200 5; do
  timeout -k 10 300 python bench.py --steps 20 --warmup $w --cpu-sample 0 > $OUT/b_w$w.log 2>&1
  rc=$?; if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo TIMEOUT; exit 1; fi
  python3 - $OUT/b_w$w.log $w <<'PY'
import json, sys
j = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("warmup", sys.argv[2], "steps 20: ms_per_step %.4f value %.4e kernel_ms %.4f" % (j["ms_per_step"], j["value"], j["roofline"]["kernel_ms"]))
PY
done
for k in 20 50 100 400; do
  timeout -k 10 300 python bench.py --steps $k --warmup 5 --cpu-sample 0 > $OUT/b_k$k.log 2>&1
  python3 - $OUT/b_k$k.log $k <<'PY'
import json, sys
j = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("warmup 5 steps", sys.argv[2], ": ms_per_step %.4f value %.4e" % (j["ms_per_step"], j["value"]))
PY
done
