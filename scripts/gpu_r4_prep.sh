# round 4: tile-cost estimate inside k_rm_prepare, raster order for images that are resident at once: tests + A/B
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r4_prep; mkdir -p $OUT
timeout 2400 python -m pytest tests -m gpu -x -q 2>&1 | tail -8 | tee $OUT/pytest.txt
timeout 900 python scripts/fuzz_parity.py 500 616161 2>&1 | tail -1 | tee -a $OUT/pytest.txt
for c in DEMO C1 C2 C3; do
  echo "== $c"
  BENCH_ARGS="--config $c --no-formula-count" STEPS=400 bash scripts/gpu_ab.sh 2>&1 | grep -v amdgpu
done | tee $OUT/ab.txt
