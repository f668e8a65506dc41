cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -12 > gpurun_out/r3/job9_tests.log; tail -4 gpurun_out/r3/job9_tests.log | head -3
python scripts/fuzz_parity.py 300 61000 2>&1 | grep -v amdgpu | grep -v ": ok" | tail -3
