set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_round5_gpu.py tests/test_e2e_gpu.py -x -q -k "one_pass or head_split_path or extreme" > gpurun_out/r05_c_tests.log 2>&1; tail -5 gpurun_out/r05_c_tests.log
B="python bench.py --steps 100 --warmup 20 --no-cpu-baseline --no-recall --preheat-seconds 3"
for i in 1 2; do
  T4R_HEAD_FDX=0 $B 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read().strip().split('\n')[-1]); print('FDX=0', j['ms_per_step'], j['ms_per_step_windows']['all'])" | tee -a gpurun_out/r05_c_ab_fdx.txt
  T4R_HEAD_FDX=1 $B 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read().strip().split('\n')[-1]); print('FDX=1', j['ms_per_step'], j['ms_per_step_windows']['all'])" | tee -a gpurun_out/r05_c_ab_fdx.txt
done
T4R_HEAD_FDX=1 T4R_HEAD_FDX_WGS=1024 $B 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read().strip().split('\n')[-1]); print('FDX=1 WGS=1024', j['ms_per_step'], j['ms_per_step_windows']['all'])" | tee -a gpurun_out/r05_c_ab_fdx.txt
T4R_HEAD_FDX=1 T4R_HEAD_FDX_WGS=256 $B 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read().strip().split('\n')[-1]); print('FDX=1 WGS=256', j['ms_per_step'], j['ms_per_step_windows']['all'])" | tee -a gpurun_out/r05_c_ab_fdx.txt
bash tools/prof_stats.sh r05_c_fdx1 T4R_HEAD_FDX=1 > gpurun_out/r05_c_prof_fdx1.txt 2>&1; head -12 gpurun_out/r05_c_prof_fdx1.txt | cut -c1-150
T4R_SWEEP_CEILINGS=0 python tools/gather_sweep.py 2>/dev/null | tee gpurun_out/r05_c_gather_default.json
