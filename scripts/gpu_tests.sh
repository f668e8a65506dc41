cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 2400 python -m pytest tests -x -q -m gpu --durations=15 2>&1 | tail -45 | tee gpurun_out/gpu_tests.log
