# round 4: bin pass for small grids (one-workgroup scan, totals into pinned host memory, lists filled ahead of the host's wait): tests + A/B
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r4_bin; mkdir -p $OUT
timeout 1500 python -m pytest tests/test_gpu_edge_cases.py tests/test_gpu_parity.py tests/test_gpu_chain.py tests/test_gpu_slabs.py tests/test_gpu_multi.py tests/test_gpu_boundary.py -m gpu -x -q 2>&1 | tail -8 | tee $OUT/pytest.txt
timeout 900 python scripts/fuzz_parity.py 600 424242 2>&1 | tail -1 | tee -a $OUT/pytest.txt
for c in DEMO C1 C2 C3; do
  echo "== $c"
  BENCH_ARGS="--config $c --no-formula-count" STEPS=400 bash scripts/gpu_ab.sh 2>&1 | grep -v amdgpu
done | tee $OUT/ab.txt
