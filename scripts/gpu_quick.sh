cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_boundary.py tests/test_gpu_edge_cases.py -x -q -m gpu 2>&1 | tail -4
for extra in "" "--no-lds-cubemap" "--cubemap f32"; do
  echo "== bench $extra"
  bash scripts/gpu_bench.sh --steps ${STEPS:-30} --warmup 3 --no-cpu-baseline $extra 2>&1 | head -2
done
