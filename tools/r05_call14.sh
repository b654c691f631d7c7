cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_torch_ops.py -q -x > gpurun_out/r05_o_tests.log 2>&1; tail -12 gpurun_out/r05_o_tests.log
timeout 100 python tools/host_trace.py > gpurun_out/r05_o_host_trace.txt 2>&1; tail -12 gpurun_out/r05_o_host_trace.txt
