#!/bin/bash
# all gather variants + the box's copy ceilings, one JSON line each -> gpurun_out/r05_gather_sweep.jsonl
out=${1:-gpurun_out/r05_gather_sweep.jsonl}
mkdir -p $(dirname $out); : > $out
T4R_SWEEP_CEILINGS=1 python tools/gather_sweep.py >> $out 2>/dev/null
for u in 2 4 8; do for nt in 0 1; do
  T4R_GATHER_U=$u T4R_GATHER_NT=$nt python tools/gather_sweep.py >> $out 2>/dev/null
done; done
cat $out
