# round 6, third batch: C5 scaling model, GEN k_fill_lds at 12 vs 16 waves (C3nv24), the fan-out test with typed solids
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
( time timeout 2400 python scripts/scaling_model.py C5 r8 2>&1 | tail -30 ) 2>&1 | tee gpurun_out/r6_scaling_model_C5_r8.txt
cp gpurun_out/r6_scaling/scaling_model_C5_r8.json gpurun_out/r6_scaling_model_C5_r8.json 2>/dev/null
STEPS=100 BENCH_ARGS="--config C3nv24 --no-variants --no-formula-count" bash scripts/gpu_ab.sh 2>&1 | tee gpurun_out/r6_ab_gen12_C3nv24.txt
timeout 900 python -m pytest tests/test_gpu_multi.py -x -q -m gpu -k "extremes" 2>&1 | tail -4
