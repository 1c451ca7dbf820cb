#!/bin/bash
# Round-2 batch 18: profiles of the final round-2 kernels (after the instruction-count work): three rocprofv3 passes per
# configuration (tools/prof.sh), the SQ passes of the headline, and the sweep.
REPO=${GRAFT_REPO_ROOT:-/root/repo}
cd $REPO
for c in C2 C3 C4 C5; do
  timeout -k 10 400 bash tools/prof.sh r02b_$c --config $c --steps 20 --warmup 5 > /dev/null || { echo "prof $c failed"; exit 1; }
  echo "profiled $c"
done
timeout -k 10 300 bash tools/prof_sq.sh r02b_C3_sq --config C3 --steps 20 --warmup 5 || exit 1
timeout -k 10 900 python tools/sweep.py gpurun_out/sweep_r02b.md > gpurun_out/sweep_r02b.log 2>&1 || { echo "sweep failed"; tail -5 gpurun_out/sweep_r02b.log; exit 1; }
tail -3 gpurun_out/sweep_r02b.log
