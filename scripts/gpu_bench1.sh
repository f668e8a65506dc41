cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3
timeout 300 python bench.py --config C1 --steps 5 --warmup 2 --no-cpu-baseline 2>&1 | tail -3
timeout 600 python bench.py --steps 10 --warmup 2 2>&1 | tail -3 | tee gpurun_out/bench_c3.json
