#!/bin/bash
# bench.py option matrix (1 GPU): every line must be a valid JSON result
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/exp12
mkdir -p $OUT
cd $REPO
i=0
run() {
  i=$((i+1))
  timeout -k 10 300 python bench.py --steps 10 --warmup 3 --cpu-sample 0 "$@" > $OUT/m_$i.json 2> $OUT/m_$i.err
  rc=$?
  if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "TIMEOUT $*"; exit 1; fi
  python3 - $OUT/m_$i.json $rc "$*" <<'PY'
import json, sys
try:
    j = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    r = j["roofline"]
    print(f"rc {sys.argv[2]} [{sys.argv[3]}] value {j['value']:.3e} ms {j['ms_per_step']:.3f} kernel {r['kernel']} {r['kernel_ms']:.3f} frac {r['frac']} alg {r['frac_algorithmic']:.2f} readout {j['config']['readout']}")
except Exception as e:
    print(f"rc {sys.argv[2]} [{sys.argv[3]}] NO RESULT LINE: {e}")
PY
}
run --mode element
run --mirrors 2
run --mirrors 8
run --rays 100000
run --rays 100000 --graph on
run --config C2 --readout separate --graph off
run --config C2 --rays 100000
run --config C3 --shard strided --rays 2000000
run --config C4 --readout fused
run --config C5 --readout separate
run --config C5 --mode element
ART_FORCE_DIST=1 run --config C4
ART_FORCE_DIST=1 run --config C2 --shard strided
tail -3 $OUT/m_*.err | grep -i "error\|traceback" | head
