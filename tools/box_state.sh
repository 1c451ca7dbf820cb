#!/bin/bash
# State of the GPU box beside a measurement: clocks, power cap, memory / compute partition mode, driver and firmware.
#   tools/box_state.sh <out-file>
# (VERDICT r2 #5b: the pool's boxes differ by up to 20 % on the write-dominated access pattern of the tracing kernels;
# every batch records this next to its numbers so that the spread can be traced to a box property.)  Read-only queries;
# an ordinary user may not change any of these settings on the pool.
OUT=${1:-/dev/stdout}
{
  echo "== date"; date -u +%FT%TZ
  echo "== rocm-smi"; rocm-smi --showclocks --showperflevel --showpower --showmaxpower --showtemp --showmemuse --showmemorypartition --showcomputepartition --showfwinfo 2>&1 | grep -v "^$"
  echo "== amd-smi static (partition, limits)"; amd-smi static --gpu 0 --partition --limit --vbios --driver 2>&1 | grep -v "^$" | head -80
  echo "== amd-smi metric (clocks, power)"; amd-smi metric --gpu 0 --clock --power --temperature 2>&1 | grep -v "^$" | head -80
  echo "== rocminfo (gfx950 agent)"; rocminfo 2>/dev/null | grep -E "Marketing Name|Compute Unit|Max Clock|Name: +gfx|Wavefront|Memory Properties|Size:" | head -24
} > "$OUT" 2>&1
