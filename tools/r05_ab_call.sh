cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 300 python -m pytest tests/test_round5_gpu.py tests/test_round4_gpu.py -m gpu -x -q -k "head" 2>&1 | tail -2
run() { tag=$1; shift
  env "$@" timeout 300 python bench.py --no-cpu-baseline --no-recall --steps 150 --preheat-seconds 3 2>/dev/null | python -c "
import sys,json; j=json.loads(sys.stdin.read().strip().split('\n')[-1]); print('$tag', j['value'], j['ms_per_step'], j['ms_per_step_windows']['all'])" | tee -a gpurun_out/r05_ab_dw4.txt
}
run dw4 A=1
run base T4R_HIP_LIB=$GRAFT_REPO_ROOT/tools/bin/libt4r_hip_base.so
run dw4 A=1
run base T4R_HIP_LIB=$GRAFT_REPO_ROOT/tools/bin/libt4r_hip_base.so
