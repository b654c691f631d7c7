#!/bin/bash
# kernel-time table of the bench workload only (the first block of profile_round.sh):  gpurun -- 'bash tools/prof_stats.sh TAG [ENV=VAL ...]'
tag=${1:-r03}
shift
out=gpurun_out
mkdir -p $out
cd /tmp 2>/dev/null; export TMPDIR=/tmp; cd - >/dev/null
for kv in "$@"; do export "$kv"; done
B="python bench.py --steps 30 --warmup 10 --no-cpu-baseline --no-recall --no-live-traffic --preheat-seconds 2"
rocprofv3 --kernel-trace --stats -d $out/${tag}_prof -o r -- $B > $out/${tag}_prof.log 2>&1
db=$(find $out/${tag}_prof -name "*_results.db" | head -1)
python tools/rocpd_stats.py $db --csv $out/${tag}_kernel_stats.csv > /dev/null
rm -rf $out/${tag}_prof
head -${LINES_OUT:-32} $out/${tag}_kernel_stats.csv | cut -c1-200
tail -c 400 $out/${tag}_prof.log | head -c 300
