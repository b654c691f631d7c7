#!/bin/bash
# End-of-round evidence on ONE GPU box, everything under gpurun_out/<tag>_*:  gpurun --timeout 2400 -- 'bash tools/round_final.sh r06_fin'
#   GPU suite, default bench line (configs[1], live PMC traffic, lockstep + seed spreads), configs[2] bench line, rocprofv3
#   kernel table + one-step timeline of the bench workload, matrix-pipe busy per kernel inside the step (separate --pmc pass,
#   --kernel-trace only beside it), gather counters, configs[3] / configs[4] step times.
tag=${1:-fin}
out=gpurun_out
mkdir -p $out
python -m pytest tests -m gpu -q 2>&1 | tail -5 > $out/${tag}_gputest.log
python bench.py > $out/${tag}_bench.json 2> $out/${tag}_bench.err
python bench.py --config c3 --no-cpu-baseline > $out/${tag}_bench_c3.json 2>> $out/${tag}_bench.err
bash tools/prof_stats.sh $tag > /dev/null 2>&1
bash tools/prof_timeline.sh $tag > /dev/null 2>&1
cd /tmp 2>/dev/null; export TMPDIR=/tmp; cd - >/dev/null
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace -d $out/${tag}_pmc_mfma -o r -- python bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-recall --no-live-traffic --preheat-seconds 0.5 > $out/${tag}_pmc_mfma.log 2>&1
db=$(find $out/${tag}_pmc_mfma -name "*_results.db" | head -1)
python tools/pmc_mfma_util.py $db > $out/${tag}_pmc_mfma_busy.csv
rm -rf $out/${tag}_pmc_mfma $out/${tag}_pmc_mfma.log
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d $out/${tag}_pmc_gather_$c -o r -- python tools/gather_pmc.py > /dev/null 2>&1
done
python tools/pmc_table.py $(find $out/${tag}_pmc_gather_FETCH_SIZE $out/${tag}_pmc_gather_WRITE_SIZE -name "*counter_collection.csv") > $out/${tag}_pmc_gather_fetch_write.csv
rm -rf $out/${tag}_pmc_gather_FETCH_SIZE $out/${tag}_pmc_gather_WRITE_SIZE
python tools/c45_bench.py c4 20 2>/dev/null | tail -1 > $out/${tag}_c4.json
timeout 600 python tools/c45_bench.py c5 3 fp16 2>/dev/null | tail -1 > $out/${tag}_c5_fp16.json
tail -3 $out/${tag}_gputest.log
python - <<PY
import json
for f in ("${tag}_bench.json", "${tag}_bench_c3.json"):
    d = json.loads(open("$out/" + f).read().strip().splitlines()[-1])
    print(f, d["value"], d["ms_per_step"], d["roofline"]["bound"], d["roofline"]["frac"])
PY
cat $out/${tag}_c4.json $out/${tag}_c5_fp16.json
head -12 $out/${tag}_pmc_mfma_busy.csv
