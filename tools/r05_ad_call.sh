cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_round5_gpu.py tests/test_kernels_gpu.py tests/test_e2e_gpu.py -m gpu -x -q -k "tiled_weight or layer or xlnet or e2e or step" > gpurun_out/r05_ad_tests.log 2>&1; tail -3 gpurun_out/r05_ad_tests.log
run() { tag=$1; shift
  env "$@" timeout 300 python bench.py --no-cpu-baseline --no-recall --steps 150 --preheat-seconds 3 2>/dev/null | python -c "
import sys,json; j=json.loads(sys.stdin.read().strip().split('\n')[-1]); print('$tag', j['value'], j['ms_per_step'], j['ms_per_step_windows']['all'])" | tee -a gpurun_out/r05_ad_planes_ab.txt
}
run tiled T4R_PLANES_TILED=1
run elementwise T4R_PLANES_TILED=0
run tiled T4R_PLANES_TILED=1
run elementwise T4R_PLANES_TILED=0
bash tools/prof_stats.sh r05_ad > /dev/null 2>&1; grep "layer_planes\|weight_scales" gpurun_out/r05_ad_kernel_stats.csv | cut -c1-50,60-130
