cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
cd /tmp; export TMPDIR=/tmp; cd - >/dev/null
rocprofv3 --kernel-trace --stats -d gpurun_out/c3_prof -o r -- python bench.py --config c3 --steps 30 --warmup 10 --no-cpu-baseline --no-recall --preheat-seconds 2 > gpurun_out/r05_k_c3_prof.log 2>&1
db=$(find gpurun_out/c3_prof -name "*_results.db" | head -1)
python tools/rocpd_stats.py $db --csv gpurun_out/r05_k_c3_kernel_stats.csv > /dev/null
rocprofv3 --kernel-trace -d gpurun_out/c3_tl -o r -- python bench.py --config c3 --steps 12 --warmup 4 --no-cpu-baseline --no-recall --preheat-seconds 1 > gpurun_out/r05_k_c3_tl.log 2>&1
db=$(find gpurun_out/c3_tl -name "*_results.db" | head -1)
python tools/rocpd_timeline.py $db > gpurun_out/r05_k_c3_timeline.csv
rm -rf gpurun_out/c3_prof gpurun_out/c3_tl
head -45 gpurun_out/r05_k_c3_kernel_stats.csv | cut -c1-170
head -3 gpurun_out/r05_k_c3_timeline.csv
