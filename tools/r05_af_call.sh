cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
run() { tag=$1; shift
  env "$@" timeout 300 python bench.py --no-cpu-baseline --no-recall --steps 150 --preheat-seconds 3 2>/dev/null | python -c "
import sys,json; j=json.loads(sys.stdin.read().strip().split('\n')[-1]); print('$tag', j['value'], j['ms_per_step'], j['ms_per_step_windows']['all'], j['config']['final_loss'])" | tee -a gpurun_out/r05_af_ab.txt
}
run late T4R_LAYER_FF_WGRAD_LATE=1
run base T4R_LAYER_FF_WGRAD_LATE=0
run late T4R_LAYER_FF_WGRAD_LATE=1
run base T4R_LAYER_FF_WGRAD_LATE=0
