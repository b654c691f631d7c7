# final-tree evidence of round 5 (gpurun -- 'bash tools/r05_final.sh TAG'): GPU suite, default bench line, configs[2] line,
# kernel table + one-step timeline of the bench workload, matrix-pipe busy per kernel, gather / head traffic counters
tag=${1:-r05_final}
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -q > gpurun_out/${tag}_gputest.log 2>&1; tail -3 gpurun_out/${tag}_gputest.log
timeout 400 python bench.py > gpurun_out/${tag}_bench.json 2> gpurun_out/${tag}_bench.err; python -c "
import json; j=json.loads(open('gpurun_out/${tag}_bench.json').read().strip().split('\n')[-1]); print(j['value'], j['ms_per_step'], j['ms_per_step_windows']['all'], j['roofline']['bound'], j['roofline']['frac'], j['roofline']['avg_launch_ms'], j['roofline_gather']['frac'], j['roofline_gather']['at_global_batch_8192_out_of_cache']['frac'], j['roofline_gather']['c3_multi_feature_at_global_batch']['frac'], j['cpu_baseline']['value'], j['recall_at_20'].get('hip_bench_config'))"
timeout 400 python bench.py --config c3 > gpurun_out/${tag}_bench_c3.json 2> gpurun_out/${tag}_bench_c3.err; python -c "
import json; j=json.loads(open('gpurun_out/${tag}_bench_c3.json').read().strip().split('\n')[-1]); print('c3', j['value'], j['ms_per_step'], j['ms_per_step_windows']['all'], j['cpu_baseline']['value'], j['recall_at_20'].get('hip_bench_config'))"
bash tools/prof_stats.sh ${tag} > gpurun_out/${tag}_prof.txt 2>&1
bash tools/prof_timeline.sh ${tag} > /dev/null 2>&1; head -2 gpurun_out/${tag}_timeline.csv
cd /tmp; export TMPDIR=/tmp; cd - >/dev/null
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace -d gpurun_out/${tag}_pmc_mfma -o r -- python bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-recall --preheat-seconds 0.5 > gpurun_out/${tag}_pmc_mfma.log 2>&1
db=$(find gpurun_out/${tag}_pmc_mfma -name "*_results.db" | head -1)
python tools/pmc_mfma_util.py $db > gpurun_out/${tag}_pmc_mfma_busy.csv; rm -rf gpurun_out/${tag}_pmc_mfma
head -16 gpurun_out/${tag}_pmc_mfma_busy.csv | cut -c1-150
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d gpurun_out/${tag}_pmc_gather_$c -o r -- python tools/gather_pmc.py > gpurun_out/${tag}_pmc_gather_$c.log 2>&1
done
python tools/pmc_table.py $(find gpurun_out/${tag}_pmc_gather_FETCH_SIZE gpurun_out/${tag}_pmc_gather_WRITE_SIZE -name "*counter_collection.csv") > gpurun_out/${tag}_pmc_gather_fetch_write.csv
rm -rf gpurun_out/${tag}_pmc_gather_FETCH_SIZE gpurun_out/${tag}_pmc_gather_WRITE_SIZE
cat gpurun_out/${tag}_pmc_gather_fetch_write.csv | cut -c1-140
# the two remaining BASELINE configurations at their named size, for the record (parity-test configurations, not bench lines)
timeout 300 python tools/c45_bench.py c4 20 > gpurun_out/${tag}_c4.json 2> gpurun_out/${tag}_c4.err; tail -1 gpurun_out/${tag}_c4.json | cut -c1-300
timeout 400 python tools/c45_bench.py c5 3 fp16 > gpurun_out/${tag}_c5_fp16.json 2> gpurun_out/${tag}_c5_fp16.err; tail -1 gpurun_out/${tag}_c5_fp16.json | cut -c1-300
