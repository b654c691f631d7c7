cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_torch_ops.py tests/test_distributed_gpu.py -q -x > gpurun_out/r05_i_tests.log 2>&1; tail -6 gpurun_out/r05_i_tests.log
