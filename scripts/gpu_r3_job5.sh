cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -12 > gpurun_out/r3/job5_tests.log; cat gpurun_out/r3/job5_tests.log | tail -6
echo "== scaling model"; timeout 1500 python scripts/scaling_model.py C3 r8 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r3/scaling_model_C3_r8.txt
