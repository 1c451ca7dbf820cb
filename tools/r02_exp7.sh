#!/bin/bash
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/exp7
mkdir -p $OUT
cd $REPO
V=$REPO/build/variants
for v in default ro_noxyo ro_noreduce ro_noscratch ro_none; do
  if [ $v = default ]; then unset ART_HIP_LIB; else export ART_HIP_LIB=$V/libart_$v.so; fi
  ART_DIAG_TAG=$v timeout -k 10 300 python tools/fused_time.py > $OUT/ft_$v.log 2>&1
  rc=$?; if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo TIMEOUT; exit 1; fi
  grep "ms per step" $OUT/ft_$v.log || tail -5 $OUT/ft_$v.log
done
