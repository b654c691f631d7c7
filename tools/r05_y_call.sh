# A/B: head d W kernel with 3 waves per SIMD (161 registers instead of 105 + 64) on top of the 2-wave one-pass forward; split counts of the forward
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_round5_gpu.py tests/test_round4_gpu.py tests/test_kernels_gpu.py -m gpu -x -q -k "head or dw" > gpurun_out/r05_y_tests2.log 2>&1; tail -2 gpurun_out/r05_y_tests2.log
run() { tag=$1; shift
  env "$@" timeout 300 python bench.py --no-cpu-baseline --no-recall --steps 200 --preheat-seconds 3 2>/dev/null | python -c "
import sys,json; j=json.loads(sys.stdin.read().strip().split('\n')[-1]); print('$tag', j['value'], j['ms_per_step'], j['ms_per_step_windows']['all'], j['roofline']['avg_launch_ms'])" | tee -a gpurun_out/r05_y_ab2.txt
}
run fdx2+dw3 A=1
run fdx2 T4R_HIP_LIB=$GRAFT_REPO_ROOT/tools/bin/libt4r_hip_fdx2.so
run fdx2+dw3 A=1
run fdx2 T4R_HIP_LIB=$GRAFT_REPO_ROOT/tools/bin/libt4r_hip_fdx2.so
run fdx2+dw3_wgs256 T4R_HEAD_FDX_WGS=256
run fdx2+dw3_wgs768 T4R_HEAD_FDX_WGS=768
run fdx2+dw3_wgs1024 T4R_HEAD_FDX_WGS=1024
