#!/bin/bash
# SQ counters of the kernels a command runs:  gpurun -- 'bash tools/pmc_kernel.sh "<filter>" "<counters...>" -- python tools/ff_fused_bench.py ...'
filt=$1; ctrs=$2; shift 3
cd /tmp 2>/dev/null; export TMPDIR=/tmp; cd - >/dev/null
rm -rf gpurun_out/pmck
rocprofv3 --pmc $ctrs --kernel-trace --output-format csv -d gpurun_out/pmck -o r -- "$@" > /dev/null 2>&1
f=$(find gpurun_out/pmck -name "*counter_collection.csv" | head -1)
python - "$f" "$filt" <<'PY'
import csv, collections, sys
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(sys.argv[1])):
    if sys.argv[2] in r["Kernel_Name"]:
        agg[r["Kernel_Name"][:44]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in agg.items():
    print(k, {c: round(sum(v) / len(v)) for c, v in sorted(d.items())})
PY
rm -rf gpurun_out/pmck
