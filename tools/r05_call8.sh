set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_torch_ops.py tests/test_round4_gpu.py tests/test_distributed_gpu.py tests/test_round5_gpu.py -q > gpurun_out/r05_h_tests.log 2>&1; tail -6 gpurun_out/r05_h_tests.log
timeout 200 python tools/occupier_curve.py --steps 40 --warmup 10 --out gpurun_out/r05_h_occupier_plain.json > /dev/null 2>&1
timeout 200 python tools/occupier_curve.py --steps 40 --warmup 10 --budget --out gpurun_out/r05_h_occupier_budget.json > /dev/null 2>&1
python - <<'PY'
import json
for f in ("plain","budget"):
    j=json.load(open(f"gpurun_out/r05_h_occupier_{f}.json"))
    print(f, j["baseline_ms"], j["baseline_ms_after"], {n:{k:v["slowdown"] for k,v in c.items()} for n,c in j["curves"].items()})
PY
cd /tmp; export TMPDIR=/tmp; cd - >/dev/null
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d gpurun_out/pmc_gemm_$c -o r -- python tools/gemm_pmc.py > gpurun_out/r05_h_pmc_$c.log 2>&1
done
python tools/pmc_table.py $(find gpurun_out/pmc_gemm_FETCH_SIZE gpurun_out/pmc_gemm_WRITE_SIZE -name "*counter_collection.csv") > gpurun_out/r05_h_pmc_gemm_fetch_write.csv
rm -rf gpurun_out/pmc_gemm_FETCH_SIZE gpurun_out/pmc_gemm_WRITE_SIZE
grep -i "head\|dropout\|split" gpurun_out/r05_h_pmc_gemm_fetch_write.csv | cut -c1-160
