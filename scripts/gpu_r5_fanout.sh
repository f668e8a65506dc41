# round 5: the fan-out after the exchange-stream change: loopback + RCCL-branch (stand-in) tests, the slab tests, then the scaling model
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 1700 python -m pytest tests/test_gpu_multi.py tests/test_gpu_rccl_shim.py tests/test_gpu_slabs.py tests/test_gpu_bench_cli.py tests/test_gpu_unity_plugin.py -x -q -m gpu 2>&1 | grep -E "passed|failed|error|Error" | tail -5 | tee gpurun_out/r5_fanout_tests.log
timeout 1500 python scripts/fuzz_parity.py 90 1000300 2>&1 | tail -2 | tee gpurun_out/r5_fanout_fuzz.log
timeout 1500 python scripts/scaling_model.py C3 r8 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r5_scaling_model_C3_r8.txt
