cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
B="python bench.py --steps 100 --warmup 20 --no-cpu-baseline --no-recall --preheat-seconds 3"
run() { tag="$1"; shift; env "$@" $B 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read().strip().split('\n')[-1]); print('$tag', j['ms_per_step'], j['ms_per_step_windows']['median'])" | tee -a gpurun_out/r05_m_splitk_sweep.txt; }
run default A=1
run min_tiles_10 T4R_GEMM_SPLIT_MIN_TILES=10
run min_tiles_40 T4R_GEMM_SPLIT_MIN_TILES=40
run min_tiles_80 T4R_GEMM_SPLIT_MIN_TILES=80
run default2 A=1
run min_tiles_160 T4R_GEMM_SPLIT_MIN_TILES=160
run target_1024 T4R_GEMM_SPLIT_TARGET=1024
