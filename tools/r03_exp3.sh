#!/bin/bash
# Round-3 batch 3: in-process A/B (tools/ab_kernel.py) of the one-ray and two-rays-per-lane chain bodies.
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/r03_exp3
mkdir -p $OUT
cd $REPO
tools/box_state.sh $OUT/box_state.txt
ART_CHAIN_RPL=2 timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "chain or fused or batched or scene or dropped or full_size or fuzz or edge" > $OUT/pytest_rpl2.log 2>&1; rc=$?; tail -3 $OUT/pytest_rpl2.log
[ $rc -eq 0 ] || exit $rc
V="ART_CHAIN_RPL=1;ART_CHAIN_RPL=2 ART_CHAIN_WAVES=4;ART_CHAIN_RPL=2 ART_CHAIN_WAVES=3;ART_CHAIN_RPL=2 ART_CHAIN_WAVES=5;ART_CHAIN_RPL=1 ART_CHAIN_WAVES=6"
for c in relay4 C4 C2 C3; do
  timeout -k 10 300 python tools/ab_kernel.py --config $c --variants "$V" 2>&1 | grep -v Warning | tee -a $OUT/ab.txt
done
timeout -k 10 300 python tools/ab_kernel.py --config relay4 --readout none --variants "$V" 2>&1 | grep -v Warning | tee -a $OUT/ab.txt
timeout -k 10 300 python tools/ab_kernel.py --config relay4 --mirrors 8 --variants "$V" 2>&1 | grep -v Warning | tee -a $OUT/ab.txt
timeout -k 10 300 python tools/ab_kernel.py --config relay4 --rays 1000000 --variants "$V" 2>&1 | grep -v Warning | tee -a $OUT/ab.txt
./tools/_build/stream_floor 10000000 > $OUT/floor.log 2>&1; grep "E=4" $OUT/floor.log | head -3
