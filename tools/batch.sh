#!/bin/bash
# ONE parameterised batch runner for the GPU box (replaces the per-experiment scripts of rounds 1-2; their index is in
# tools/README.md).  Everything a batch prints also lands in gpurun_out/<name>/, with the box's state beside it.
#
#   tools/batch.sh NAME [options]
#     --pytest ["-k expr"]        run the GPU suite first (stop the batch if it fails)
#     --configs "relay4 C2 ..."   bench.py configurations to run (default: relay4)
#     --variants "A=1 B=2|A=2"    '|'-separated variants, each a list of environment assignments ("-" = none);
#                                 (another BUILD of the library is compared with tools/ab_kernel.py --variants "...;LIB=<path>":
#                                 the product loader takes no library from the environment)
#     --args "..."                extra bench.py arguments for every run (e.g. "--steps 20 --warmup 5")
#     --reps N                    repeat the whole variant x config matrix N times (default 1), variants interleaved
#     --floor                     tools/_build/stream_floor 10000000 (the bare access pattern of this box)
#     --prof "relay4 C2 ..."      tools/prof.sh (kernel trace + FETCH_SIZE + WRITE_SIZE passes) for these configurations
#     --sq                        tools/prof_sq.sh on the headline configuration
# Example (round-3 batch 2): tools/batch.sh r03_rpl --configs "relay4 C4" --variants "ART_CHAIN_RPL=1|ART_CHAIN_RPL=2 ART_CHAIN_WAVES=4" --reps 2 --floor
REPO=${GRAFT_REPO_ROOT:-/root/repo}
NAME=$1; shift
OUT=$REPO/gpurun_out/$NAME
mkdir -p $OUT
cd $REPO
PYTEST=0; PYK=""; CONFIGS="relay4"; VARIANTS="-"; ARGS=""; REPS=1; FLOOR=0; PROF=""; SQ=0
while [ $# -gt 0 ]; do
  case $1 in
    --pytest) PYTEST=1; if [ "${2:0:2}" = "-k" ]; then PYK="$2"; shift; fi;;
    --configs) CONFIGS="$2"; shift;;
    --variants) VARIANTS="$2"; shift;;
    --args) ARGS="$2"; shift;;
    --reps) REPS=$2; shift;;
    --floor) FLOOR=1;;
    --prof) PROF="$2"; shift;;
    --sq) SQ=1;;
    *) echo "unknown option $1"; exit 2;;
  esac
  shift
done
tools/box_state.sh $OUT/box_state.txt
python3 tools/source_hash.py > $OUT/source_hash.txt
echo "# batch $NAME  sources $(cat $OUT/source_hash.txt)  $(date -u +%FT%TZ)" | tee $OUT/lines.txt
if [ $PYTEST = 1 ]; then
  timeout -k 10 900 python -m pytest tests -m gpu -x -q $PYK > $OUT/pytest.log 2>&1; rc=$?; tail -40 $OUT/pytest.log
  [ $rc -eq 0 ] || exit $rc
fi
line() { python3 -c "
import json,sys
j=json.loads(sys.stdin.read()); r=j['roofline']
print('$1 | value %.3e ms %.4f | kernel_ms %.4f frac %.3f (%s) frac_compulsory %.3f | sustained %.3e | lazy %s' % (j['value'], j['ms_per_step'], r['kernel_ms'], r['frac'], r['frac_basis'], r['frac_compulsory'], j['value_sustained'], ('%.3e' % j['value_lazy_history']) if j.get('value_lazy_history') else '-'))"; }
IFS='|' read -ra VARS <<< "$VARIANTS"
for rep in $(seq 1 $REPS); do
  for v in "${VARS[@]}"; do
    for c in $CONFIGS; do
      tag="$c [$v] rep$rep"
      if [ "$v" = "-" ]; then envs=""; else envs="$v"; fi
      env $envs timeout -k 10 400 python bench.py --config $c --cpu-sample 0 $ARGS 2>$OUT/last.err | tee "$OUT/bench_${c}_$(echo $v | tr ' =/' '___')_$rep.json" | line "$tag" | tee -a $OUT/lines.txt || { tail -5 $OUT/last.err; exit 1; }
    done
  done
done
if [ $FLOOR = 1 ]; then
  [ -x tools/_build/stream_floor ] || { mkdir -p tools/_build; hipcc -O3 --offload-arch=gfx950 tools/stream_floor.hip -o tools/_build/stream_floor; }
  ./tools/_build/stream_floor 10000000 > $OUT/floor.log 2>&1; grep "E=4" $OUT/floor.log | head -12 | tee -a $OUT/lines.txt
fi
for c in $PROF; do
  timeout -k 10 900 bash tools/prof.sh ${NAME}_$c --config $c --steps 20 --warmup 5 $ARGS > $OUT/prof_$c.log 2>&1 || { tail -5 $OUT/prof_$c.log; exit 1; }
done
if [ $SQ = 1 ]; then
  timeout -k 10 600 bash tools/prof_sq.sh ${NAME}_relay4_sq --steps 20 --warmup 5 $ARGS > $OUT/prof_sq.log 2>&1 || { tail -5 $OUT/prof_sq.log; exit 1; }
fi
tools/box_state.sh $OUT/box_state_after.txt
