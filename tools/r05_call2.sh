set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_round5_gpu.py -x -q > gpurun_out/r05_b_round5_tests.log 2>&1; tail -15 gpurun_out/r05_b_round5_tests.log
timeout 600 python -m pytest tests -m gpu -q > gpurun_out/r05_b_gputest.log 2>&1; tail -8 gpurun_out/r05_b_gputest.log
B="python bench.py --steps 100 --warmup 20 --no-cpu-baseline --no-recall --preheat-seconds 3"
for i in 1 2; do
  T4R_HEAD_FDX=0 $B 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read().strip().split('\n')[-1]); print('FDX=0', j['ms_per_step'], j['ms_per_step_windows']['all'])" | tee -a gpurun_out/r05_b_ab_fdx.txt
  T4R_HEAD_FDX=1 $B 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read().strip().split('\n')[-1]); print('FDX=1', j['ms_per_step'], j['ms_per_step_windows']['all'])" | tee -a gpurun_out/r05_b_ab_fdx.txt
done
bash tools/prof_stats.sh r05_b_fdx1 T4R_HEAD_FDX=1 > gpurun_out/r05_b_prof_fdx1.txt 2>&1; head -24 gpurun_out/r05_b_prof_fdx1.txt | cut -c1-150
timeout 200 python tools/occupier_curve.py --steps 40 --warmup 10 --out gpurun_out/r05_b_occupier_q4.json > /dev/null 2>&1
GPU_MAX_HW_QUEUES=8 timeout 200 python tools/occupier_curve.py --steps 40 --warmup 10 --out gpurun_out/r05_b_occupier_q8.json > /dev/null 2>&1
python - <<'PY'
import json
for f in ("q4","q8"):
    try:
        j=json.load(open(f"gpurun_out/r05_b_occupier_{f}.json"))
        print(f, j["baseline_ms"], {n:{k:v["slowdown"] for k,v in c.items()} for n,c in j["curves"].items()})
    except Exception as e: print(f, "failed", e)
PY
timeout 400 python tools/recall_sweep.py 2>/dev/null | tee gpurun_out/r05_b_recall_sweep.jsonl
T4R_SWEEP_CEILINGS=0 python tools/gather_sweep.py 2>/dev/null | tee gpurun_out/r05_b_gather_default.json
