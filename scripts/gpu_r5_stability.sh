# round 5: the thread- and timing-dependent tests repeated (RCCL stand-in child runs, loopback fan-out, unity plugin thread, chain watchdog)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
: > gpurun_out/r5_stability.log
for i in $(seq 1 ${REPS:-8}); do
  timeout 900 python -m pytest tests/test_gpu_rccl_shim.py tests/test_gpu_multi.py tests/test_gpu_unity_plugin.py tests/test_gpu_chain.py -x -q -m gpu 2>&1 | grep -E "passed|failed|error" | tail -2 | tee -a gpurun_out/r5_stability.log
done
