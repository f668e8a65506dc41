cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3
timeout 1200 python -m pytest tests/test_gpu_multi.py tests/test_gpu_chain.py tests/test_gpu_boundary.py tests/test_gpu_slabs.py -x -q 2>&1 | tail -40 > gpurun_out/r3/job2_tests.log
cat gpurun_out/r3/job2_tests.log
