#!/bin/bash
# What does each part of the fused read-out tail cost?  Diagnostic builds drop the per-ray outputs, the wave reduction,
# the partial-statistics stores (results wrong by design); relay4 and the 8-element C4 chain.
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/exp7
mkdir -p $OUT
cd $REPO
V=$REPO/build/variants
for cfg in relay4 C4; do
for v in default ro_noxyo ro_noreduce ro_noscratch ro_none; do
  if [ $v = default ]; then unset ART_HIP_LIB; else export ART_HIP_LIB=$V/libart_$v.so; fi
  ART_DIAG_CFG=$cfg ART_DIAG_TAG=$v timeout -k 10 300 python tools/fused_time.py > $OUT/ft_${cfg}_$v.log 2>&1
  rc=$?; if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo TIMEOUT; exit 1; fi
  grep "ms per step" $OUT/ft_${cfg}_$v.log || tail -5 $OUT/ft_${cfg}_$v.log
done
done
