cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_round5_gpu.py tests/test_kernels_gpu.py tests/test_e2e_gpu.py -q -k "soft or multi or full_size or prepost or concat" > gpurun_out/r05_l_tests.log 2>&1; tail -3 gpurun_out/r05_l_tests.log
B="python bench.py --config c3 --steps 100 --warmup 20 --no-cpu-baseline --no-recall --preheat-seconds 3"
$B 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read().strip().split('\n')[-1]); print('c3 new', j['ms_per_step'], j['ms_per_step_windows']['all'])" | tee -a gpurun_out/r05_l_c3.txt
$B 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read().strip().split('\n')[-1]); print('c3 new', j['ms_per_step'], j['ms_per_step_windows']['all'])" | tee -a gpurun_out/r05_l_c3.txt
