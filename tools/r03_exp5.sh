#!/bin/bash
# Round-3 batch 5: two-rays-per-lane body with the workgroup's store bursts aligned by a bare s_barrier (ART_CHAIN_SYNC=1);
# the like-for-like floor of the fused pattern (stream_floor with the read-out's streams).
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/r03_exp5
mkdir -p $OUT
cd $REPO
tools/box_state.sh $OUT/box_state.txt
V="ART_CHAIN_RPL=1;ART_CHAIN_RPL=2 ART_CHAIN_WAVES=4;ART_CHAIN_RPL=2 ART_CHAIN_WAVES=4 ART_CHAIN_SYNC=1"
timeout -k 10 300 python tools/ab_kernel.py --config relay4 --variants "$V" 2>&1 | grep -v "Warning\|amdgpu.ids" | tee -a $OUT/ab.txt
timeout -k 10 300 python tools/ab_kernel.py --config relay4 --readout none --variants "$V" 2>&1 | grep -v "Warning\|amdgpu.ids" | tee -a $OUT/ab.txt
timeout -k 10 300 python tools/ab_kernel.py --config C4 --variants "$V" 2>&1 | grep -v "Warning\|amdgpu.ids" | tee -a $OUT/ab.txt
timeout -k 10 300 python tools/ab_kernel.py --config C4 --readout none --variants "$V" 2>&1 | grep -v "Warning\|amdgpu.ids" | tee -a $OUT/ab.txt
timeout -k 10 300 python tools/ab_kernel.py --config C2 --variants "$V" 2>&1 | grep -v "Warning\|amdgpu.ids" | tee -a $OUT/ab.txt
./tools/_build/stream_floor 10000000 > $OUT/floor.log 2>&1; grep "pass 1" -A200 $OUT/floor.log | grep "E=4\|E=8" | grep "soa  nt stores\|read-out\|workgroups per CU\|16 B per lane, 512"
