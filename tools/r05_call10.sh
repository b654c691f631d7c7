cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_round5_gpu.py tests/test_e2e_gpu.py -q -k "one_pass or head_split_path or extreme or full_size" > gpurun_out/r05_j_tests.log 2>&1; tail -3 gpurun_out/r05_j_tests.log
B="python bench.py --steps 100 --warmup 20 --no-cpu-baseline --no-recall --preheat-seconds 3"
for i in 1 2; do
  T4R_HIP_LIB=$PWD/tools/bin/libt4r_hip_prev.so $B 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read().strip().split('\n')[-1]); print('prev', j['ms_per_step'], j['ms_per_step_windows']['all'])" | tee -a gpurun_out/r05_j_ab.txt
  $B 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read().strip().split('\n')[-1]); print('new ', j['ms_per_step'], j['ms_per_step_windows']['all'])" | tee -a gpurun_out/r05_j_ab.txt
done
bash tools/prof_stats.sh r05_j > gpurun_out/r05_j_prof.txt 2>&1; head -8 gpurun_out/r05_j_prof.txt | cut -c1-150
