cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
run() { tag=$1; shift
  env "$@" timeout 300 python bench.py --no-cpu-baseline --no-recall --steps 150 --preheat-seconds 3 2>/dev/null | python -c "
import sys,json; j=json.loads(sys.stdin.read().strip().split('\n')[-1]); print('$tag', j['value'], j['ms_per_step'], j['ms_per_step_windows']['all'])" | tee -a gpurun_out/r05_aa_tile_ab.txt
}
run base A=1
run ln1_R3 T4R_XLNET_R_LN1=3
run ln1_R2 T4R_XLNET_R_LN1=2
run dh_R3 T4R_XLNET_R_DH=3
run dh_R2 T4R_XLNET_R_DH=2
run both_R2 T4R_XLNET_R_LN1=2 T4R_XLNET_R_DH=2
run base A=1
