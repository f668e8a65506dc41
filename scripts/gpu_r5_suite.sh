# round 5: the whole GPU suite as the driver runs it (timed), then the default bench line
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
( time timeout 2400 python -m pytest tests -x -q -m gpu --durations=12 2>&1 | tail -40 ) 2>&1 | tee gpurun_out/r5_gpu_suite.log
( time python bench.py ) > gpurun_out/r5_bench_default.json 2> gpurun_out/r5_bench_default.err
tail -c 3000 gpurun_out/r5_bench_default.json; tail -5 gpurun_out/r5_bench_default.err
