# round 5: the fan-out's RCCL branch on the checking stand-in (child processes), then the loopback fan-out tests as before
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 1700 python -m pytest tests/test_gpu_rccl_shim.py -x -q -m gpu --durations=5 2>&1 | tail -60 | tee gpurun_out/r5_shim_tests.log
timeout 900 python -m pytest tests/test_gpu_multi.py -x -q -m gpu --durations=5 2>&1 | tail -15 | tee gpurun_out/r5_multi_tests.log
