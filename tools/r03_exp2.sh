#!/bin/bash
# Round-3 batch 2: GPU suite of the current tree, then the two-rays-per-lane chain body (ART_CHAIN_RPL=2; 4 and 5 waves)
# against the shipped one-ray-per-lane body, on ONE box, with the box's state and its bare-pattern floor recorded; the
# one-launch fold rides along.
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/r03_exp2
mkdir -p $OUT
cd $REPO
tools/box_state.sh $OUT/box_state.txt
line() { python3 -c "import json,sys; j=json.loads(sys.stdin.read()); r=j['roofline']; print('$1 value %.3e ms %.4f kernel_ms %.4f frac_compulsory %.3f sustained %.3e' % (j['value'], j['ms_per_step'], r['kernel_ms'], r['frac_compulsory'], j['value_sustained']))"; }
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $OUT/pytest.log 2>&1; rc=$?; tail -32 $OUT/pytest.log
[ $rc -eq 0 ] || exit $rc
# correctness of the experiment body: the chain-mode parity tests with it
ART_CHAIN_RPL=2 timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "chain or fused or batched or scene or dropped or full_size or fuzz" > $OUT/pytest_rpl2.log 2>&1; rc=$?; tail -5 $OUT/pytest_rpl2.log
[ $rc -eq 0 ] || exit $rc
for rep in 1 2; do
  for v in "1 5" "2 4" "2 5" "2 3"; do
    set -- $v
    for c in relay4 C4 C2 C3; do
      ART_CHAIN_RPL=$1 ART_CHAIN_WAVES=$2 timeout -k 10 300 python bench.py --config $c --cpu-sample 0 2>/dev/null | line "$c rpl$1 w$2" || exit 1
    done
  done
done
./tools/_build/stream_floor 10000000 > $OUT/floor.log 2>&1; grep "E=4" $OUT/floor.log | head -12
tools/box_state.sh $OUT/box_state_after.txt
