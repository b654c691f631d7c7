set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -q > gpurun_out/r05_e_gputest.log 2>&1; tail -6 gpurun_out/r05_e_gputest.log
timeout 200 python tools/occupier_curve.py --steps 40 --warmup 10 --out gpurun_out/r05_e_occupier_plain.json > /dev/null 2>&1
timeout 200 python tools/occupier_curve.py --steps 40 --warmup 10 --budget --out gpurun_out/r05_e_occupier_budget.json > /dev/null 2>&1
python - <<'PY'
import json
for f in ("plain","budget"):
    try:
        j=json.load(open(f"gpurun_out/r05_e_occupier_{f}.json"))
        print(f, j["baseline_ms"], {n:{k:v["slowdown"] for k,v in c.items()} for n,c in j["curves"].items()})
    except Exception as e: print(f, "failed", e)
PY
