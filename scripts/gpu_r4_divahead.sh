# round 4: rcp(1 + density) formed N slices ahead of the fill's serial light chain (main = 4; ahead1 = in place, as before): fingerprints + A/B
cd $GRAFT_REPO_ROOT
PKG=volumetric-particles-for-unity_amd
OUT=gpurun_out/r4_divahead; mkdir -p $OUT
cp $PKG/libvpfx.so /tmp/libvpfx_main.so
for v in main ahead1 ahead8; do
  [ $v = main ] && cp /tmp/libvpfx_main.so $PKG/libvpfx.so || cp _ab/libvpfx_$v.so $PKG/libvpfx.so
  for cfg in C1 C2 C3; do echo -n "$v "; timeout 300 python scripts/fill_hash.py $cfg 2 r8 2>&1 | grep -v amdgpu; done
  echo -n "$v f32 "; timeout 300 python scripts/fill_hash.py C2 2 f32 2>&1 | grep -v amdgpu
done | tee $OUT/hash.txt
cp /tmp/libvpfx_main.so $PKG/libvpfx.so
for c in DEMO C1 C2 C3 C5; do
  steps=300; [ $c = C5 ] && steps=4
  echo "== $c"
  BENCH_ARGS="--config $c --no-formula-count" STEPS=$steps bash scripts/gpu_ab.sh 2>&1 | grep -v amdgpu
done | tee $OUT/ab.txt
