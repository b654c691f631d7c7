set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 600 python -m pytest tests -m gpu -x -q > gpurun_out/r05_a_gputest.log 2>&1; tail -3 gpurun_out/r05_a_gputest.log
timeout 300 python bench.py > gpurun_out/r05_a_bench_start.json 2> gpurun_out/r05_a_bench_start.err; tail -c 600 gpurun_out/r05_a_bench_start.json
timeout 300 python bench.py --config c3 --steps 60 > gpurun_out/r05_a_bench_c3.json 2> gpurun_out/r05_a_bench_c3.err; head -c 400 gpurun_out/r05_a_bench_c3.json
timeout 200 python tools/occupier_curve.py --steps 40 --warmup 10 > gpurun_out/r05_a_occupier.log 2>&1; tail -40 gpurun_out/r05_a_occupier.log
timeout 400 bash tools/gather_sweep.sh gpurun_out/r05_a_gather_sweep.jsonl
