cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
run() { # tag, env...
  tag=$1; shift
  env "$@" timeout 300 python bench.py --no-cpu-baseline --no-recall --steps 150 --preheat-seconds 3 $EXTRA > gpurun_out/r05_q_$tag.json 2> gpurun_out/r05_q_$tag.err
  python -c "
import json; j=json.loads(open('gpurun_out/r05_q_$tag.json').read().strip().split('\n')[-1]); print('$tag', j['value'], j['ms_per_step'], j['ms_per_step_windows']['all'])"
}
EXTRA="" run pf2_nosort T4R_PREFETCH=2 T4R_EMB_SORT_STREAM=0
EXTRA="" run pf3_nosort T4R_PREFETCH=3 T4R_EMB_SORT_STREAM=0
EXTRA="" run pf1_nosort T4R_PREFETCH=1 T4R_EMB_SORT_STREAM=0
EXTRA="" run pf0_nosort2 T4R_PREFETCH=0 T4R_EMB_SORT_STREAM=0
EXTRA="" run pf3 T4R_PREFETCH=3
python tools/host_trace.py > gpurun_out/r05_q_host_trace.txt 2>&1; tail -15 gpurun_out/r05_q_host_trace.txt
