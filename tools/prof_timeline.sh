#!/bin/bash
# one-step timeline of the bench workload:  gpurun -- 'bash tools/prof_timeline.sh TAG [ENV=VAL ...]'
tag=${1:-r03}
shift
out=gpurun_out
mkdir -p $out
cd /tmp 2>/dev/null; export TMPDIR=/tmp; cd - >/dev/null
for kv in "$@"; do export "$kv"; done
B="python bench.py --steps 12 --warmup 4 --no-cpu-baseline --no-recall --no-live-traffic --preheat-seconds 1"
rocprofv3 --kernel-trace -d $out/${tag}_tl -o r -- $B > $out/${tag}_tl.log 2>&1
db=$(find $out/${tag}_tl -name "*_results.db" | head -1)
python tools/rocpd_timeline.py $db > $out/${tag}_timeline.csv
rm -rf $out/${tag}_tl
head -3 $out/${tag}_timeline.csv
