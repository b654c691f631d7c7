# configs[2]: SoftEmbedding kernels with K = 10, D = 8 as compile-time facts; the tables' id sorts as one device sort
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_round5_gpu.py tests/test_e2e_gpu.py tests/test_kernels_gpu.py tests/test_torch_ops.py -m gpu -x -q -k "soft or one_sort or full_size or multi or c3 or fake or traced" > gpurun_out/r05_u_tests.log 2>&1; tail -3 gpurun_out/r05_u_tests.log
for m in 1 0; do
  T4R_EMB_SORT_MULTI=$m timeout 300 python bench.py --config c3 --no-cpu-baseline --no-recall --steps 200 --preheat-seconds 3 2>/dev/null | python -c "
import sys,json; j=json.loads(sys.stdin.read().strip().split('\n')[-1]); print('c3 exact soft kernels, sort_multi=$m', j['value'], j['ms_per_step'], j['ms_per_step_windows']['all'])" | tee -a gpurun_out/r05_u_c3_ab.txt
done
cd /tmp; export TMPDIR=/tmp; cd - > /dev/null
B="python bench.py --config c3 --steps 30 --warmup 10 --no-cpu-baseline --no-recall --preheat-seconds 2"
rocprofv3 --kernel-trace --stats -d gpurun_out/r05_u_prof -o r -- $B > gpurun_out/r05_u_prof.log 2>&1
db=$(find gpurun_out/r05_u_prof -name "*_results.db" | head -1)
python tools/rocpd_stats.py $db --csv gpurun_out/r05_u_c3_kernel_stats.csv > /dev/null
python tools/rocpd_timeline.py $db > gpurun_out/r05_u_c3_timeline.csv
rm -rf gpurun_out/r05_u_prof
head -2 gpurun_out/r05_u_c3_timeline.csv; grep "soft_embedding" gpurun_out/r05_u_c3_kernel_stats.csv | cut -c1-60,140-200
