#!/bin/bash
# Long differential fuzz on the GPU with the round-2 kernels (scene launch + fused read-out included in the checks).
REPO=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p $REPO/gpurun_out/fuzz
cd $REPO
timeout -k 10 1100 python tools/gpu_fuzz.py ${1:-20000000} ${2:-60000} > gpurun_out/fuzz/fuzz_$1.log 2>&1
tail -4 gpurun_out/fuzz/fuzz_$1.log
