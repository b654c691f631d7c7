cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_round5_gpu.py tests/test_torch_ops.py tests/test_e2e_gpu.py -m gpu -x -q -k "reproducible or multi or c3 or traced or full_size or fake" > gpurun_out/r05_w_tests.log 2>&1; tail -4 gpurun_out/r05_w_tests.log
for s in "A=1" "A=2"; do
  echo "== run $s" | tee -a gpurun_out/r05_w_c3_ab_probe2.txt
  timeout 200 python tools/c3_ab_probe.py 12 2>/dev/null | tee -a gpurun_out/r05_w_c3_ab_probe2.txt
done
timeout 300 python bench.py --config c3 --no-cpu-baseline --steps 100 2>/dev/null | python -c "
import sys,json; j=json.loads(sys.stdin.read().strip().split('\n')[-1]); print('c3', j['value'], j['ms_per_step'], j['recall_at_20'].get('hip_bench_config'))"
timeout 300 python bench.py --config c3 --no-cpu-baseline --steps 100 2>/dev/null | python -c "
import sys,json; j=json.loads(sys.stdin.read().strip().split('\n')[-1]); print('c3', j['value'], j['ms_per_step'], j['recall_at_20'].get('hip_bench_config'))"
