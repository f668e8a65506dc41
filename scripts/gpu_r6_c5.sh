# round 6: config 5 through the fan-out slab by slab (test), the driver-command hardening tests, the smoothstep-skip A/B at C3 and C5, the C5 scaling model
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
( time timeout 1500 python -m pytest tests/test_gpu_large_configs.py -x -q -m gpu -k "slab_by_slab" -s 2>&1 | tail -15 ) 2>&1 | tee gpurun_out/r6_c5_slab_test.log
( time timeout 1800 python -m pytest tests/test_gpu_bench_cli.py -x -q -m gpu -k "drivers_scaling or dies_mid_run" --durations=5 2>&1 | tail -25 ) 2>&1 | tee gpurun_out/r6_driver_cmd_tests.log
STEPS=100 bash scripts/gpu_ab.sh 2>&1 | tee gpurun_out/r6_ab_smooth_C3.txt
STEPS=10 BENCH_ARGS="--config C5 --no-variants --no-formula-count" bash scripts/gpu_ab.sh 2>&1 | tee gpurun_out/r6_ab_smooth_C5.txt
( time timeout 2400 python scripts/scaling_model.py C5 r8 2>&1 | tail -30 ) 2>&1 | tee gpurun_out/r6_scaling_model_C5_r8.txt
