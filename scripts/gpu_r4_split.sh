# round 4: the fill hands the light on ahead of its propagate pass (default math): tests, light-map distance to the oracle, A/B
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r4_split; mkdir -p $OUT
timeout 2400 python -m pytest tests -m gpu -x -q 2>&1 | tail -8 | tee $OUT/pytest.txt
timeout 900 python scripts/fuzz_parity.py 800 515151 2>&1 | tail -1 | tee -a $OUT/pytest.txt
for c in DEMO C1 C2 C3; do
  echo "== $c"
  BENCH_ARGS="--config $c --no-formula-count" STEPS=400 bash scripts/gpu_ab.sh 2>&1 | grep -v amdgpu
done | tee $OUT/ab.txt
echo "== C5"; BENCH_ARGS="--config C5 --no-formula-count" STEPS=4 bash scripts/gpu_ab.sh 2>&1 | grep -v amdgpu | tee -a $OUT/ab.txt
