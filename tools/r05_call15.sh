cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_distributed_gpu.py -q -x -k "ddp" > gpurun_out/r05_p_tests.log 2>&1; tail -6 gpurun_out/r05_p_tests.log
