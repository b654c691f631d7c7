set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 300 python -m pytest tests/test_round5_gpu.py tests/test_kernels_gpu.py -q -k "budget or attn or co_resident" > gpurun_out/r05_g_tests.log 2>&1; tail -3 gpurun_out/r05_g_tests.log
timeout 200 python tools/occupier_curve.py --steps 40 --warmup 10 --budget --out gpurun_out/r05_g_occupier_budget.json > /dev/null 2>&1
python - <<'PY'
import json
j=json.load(open("gpurun_out/r05_g_occupier_budget.json"))
print(j["baseline_ms"], {n:{k:v["slowdown"] for k,v in c.items()} for n,c in j["curves"].items()})
PY
