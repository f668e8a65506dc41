# round 5: the run-time-voxel-count kernels against the oracle, then a slice of the second-generation fuzz sweep (any nv)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_boundary.py -x -q -m gpu --durations=10 2>&1 | tail -40 | tee gpurun_out/r5_nv_tests.log
timeout 900 python scripts/fuzz_parity.py 150 1000000 2>&1 | tail -25 | tee gpurun_out/r5_nv_fuzz.log
