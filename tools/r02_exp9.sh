#!/bin/bash
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/exp9
mkdir -p $OUT
cd $REPO
for m in fused separate; do
timeout -k 10 300 python tools/host_timing.py C4 $m 60 > $OUT/ht_C4_$m.log 2>&1; grep -v amdgpu.ids $OUT/ht_C4_$m.log | cut -c1-400
done
timeout -k 10 300 python tools/host_timing.py relay4 fused 100 > $OUT/ht_relay4_fused.log 2>&1; grep -v amdgpu.ids $OUT/ht_relay4_fused.log | cut -c1-600
