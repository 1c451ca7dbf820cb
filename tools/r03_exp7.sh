#!/bin/bash
# Round-3 batch 7: the two-rays-per-lane body for chains WITH defects (C5), end-to-end time of the C3 workflow with the lazy history
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/r03_exp7
mkdir -p $OUT
cd $REPO
tools/box_state.sh $OUT/box_state.txt
ART_CHAIN_RPL=2 timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "c5 or zernike or fourrier or fuzz or batched_variants or edge" > $OUT/pytest_rpl2_def.log 2>&1; rc=$?; tail -3 $OUT/pytest_rpl2_def.log
[ $rc -eq 0 ] || exit $rc
V="ART_CHAIN_RPL=1;ART_CHAIN_RPL=2"
timeout -k 10 300 python tools/ab_kernel.py --config C5 --variants "$V" 2>&1 | grep -v "Warning\|amdgpu.ids" | tee -a $OUT/ab.txt
timeout -k 10 300 python tools/ab_kernel.py --config C5 --readout none --variants "$V" 2>&1 | grep -v "Warning\|amdgpu.ids" | tee -a $OUT/ab.txt
timeout -k 10 300 python tools/e2e_time.py 10000000 batched 2>&1 | grep -v "Warning\|amdgpu.ids" | tee $OUT/e2e.txt
timeout -k 10 300 python examples/twisted_toroids_large.py 2>&1 | grep -v "Warning\|amdgpu.ids" | tail -13 | tee $OUT/example.txt
