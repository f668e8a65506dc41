# round 4: workgroup-level claim of k_fill_lds -- bit-identity (fingerprints vs the one-atomic-per-unit build), chain tests, A/B timing
cd $GRAFT_REPO_ROOT
PKG=volumetric-particles-for-unity_amd
OUT=gpurun_out/r4_fillclaim; mkdir -p $OUT
cp $PKG/libvpfx.so /tmp/libvpfx_main.so
for v in main claim1; do
  [ $v = main ] && cp /tmp/libvpfx_main.so $PKG/libvpfx.so || cp _ab/libvpfx_$v.so $PKG/libvpfx.so
  for cfg in C1 C2 C3; do echo -n "$v "; timeout 300 python scripts/fill_hash.py $cfg 2 r8; done
  echo -n "$v D=1 "; timeout 300 python scripts/fill_hash.py C2 2 r8 1.0
done 2>&1 | tee $OUT/hash.txt
cp /tmp/libvpfx_main.so $PKG/libvpfx.so
timeout 600 python scripts/chain_stress.py C3 30 2>&1 | tail -1 | tee $OUT/chain_stress.txt
timeout 600 python scripts/chain_stress.py C1 100 2>&1 | tail -1 | tee -a $OUT/chain_stress.txt
timeout 900 python -m pytest tests/test_gpu_chain.py tests/test_gpu_lds_cubemap.py tests/test_gpu_parity.py -x -q 2>&1 | tail -3 | tee $OUT/pytest.txt
STEPS=30 bash scripts/gpu_ab.sh 2>&1 | tee $OUT/ab_C3.txt
BENCH_ARGS="--config C1" STEPS=200 bash scripts/gpu_ab.sh 2>&1 | grep -v p7 | tee $OUT/ab_C1.txt
