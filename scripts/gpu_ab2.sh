cd $GRAFT_REPO_ROOT
for extra in "" "--no-grey"; do echo "== $extra"; bash scripts/gpu_bench.sh --steps 100 --warmup 3 --no-cpu-baseline $extra | head -1; done
STEPS=40 bash scripts/gpu_ab.sh
