# round 4: four samples per ray-march iteration (default) against two (_ab/libvpfx_spi2.so): bench A/B per config, then parity
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r4_spi; mkdir -p $OUT
for c in C1 DEMO C2 C3 C5; do
  steps=300; [ $c = C5 ] && steps=4
  echo "== $c (main = 4 samples per iteration at 4 waves/SIMD, spi2 = 2 at 5)"
  BENCH_ARGS="--config $c --no-formula-count" STEPS=$steps bash scripts/gpu_ab.sh 2>&1 | grep -v amdgpu
done | tee $OUT/ab.txt
timeout 600 python scripts/view_sweep.py C3 2>&1 | grep -v amdgpu.ids | tee $OUT/view_sweep.txt
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_brick_format.py tests/test_gpu_edge_cases.py tests/test_gpu_slabs.py -x -q 2>&1 | grep -E "passed|failed" | tee $OUT/pytest.txt
