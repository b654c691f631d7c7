cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_round5_gpu.py tests/test_round4_gpu.py tests/test_e2e_gpu.py tests/test_kernels_gpu.py -m gpu -x -q -k "head or e2e or one_pass or planes or layer or step" > gpurun_out/r05_ae_tests.log 2>&1; tail -3 gpurun_out/r05_ae_tests.log
run() { tag=$1; shift
  env "$@" timeout 300 python bench.py --no-cpu-baseline --no-recall --steps 150 --preheat-seconds 3 2>/dev/null | python -c "
import sys,json; j=json.loads(sys.stdin.read().strip().split('\n')[-1]); print('$tag', j['value'], j['ms_per_step'], j['ms_per_step_windows']['all'], j['config']['final_loss'])" | tee -a gpurun_out/r05_ae_ab.txt
}
run new A=1
run base T4R_HIP_LIB=$GRAFT_REPO_ROOT/tools/bin/libt4r_hip_base.so
run new A=1
run base T4R_HIP_LIB=$GRAFT_REPO_ROOT/tools/bin/libt4r_hip_base.so
