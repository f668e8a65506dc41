# round 5: the split ray-march (K lanes per ray on the most expensive super-tiles) on the small frames.  A/B builds from
#   scripts/build_ab.sh w4 "-DVPFX_AB=1" w3 "-DVPFX_AB=1 -DVPFX_RM_WAVES_SPLIT=3"
# VPFX_RM_SPLIT="log2k,share%" overrides the plan (0 = the unsplit kernel)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
PKG=volumetric-particles-for-unity_amd
out=gpurun_out/r5_split_ab.txt
: > $out
cp $PKG/libvpfx.so /tmp/libvpfx_main.so
for lib in w4; do
  cp _ab/libvpfx_$lib.so $PKG/libvpfx.so
  for cfg in DEMO C1 C2; do
    for sp in 0 1,25 2,12 2,25 2,50 3,25 2,100 1,100; do
      VPFX_RM_SPLIT=$sp python bench.py --config $cfg --steps 400 --warmup 5 --no-cpu-baseline --no-variants 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$lib $cfg split=$sp: ms_per_step %.4f  raymarch_stage %.4f  fill %.4f  samples %d' % (d['ms_per_step'], d['stage_ms']['raymarch_kernel'], d['stage_ms']['fill_kernel'], d['config']['samples_executed']))" | tee -a $out
    done
  done
done
cp /tmp/libvpfx_main.so $PKG/libvpfx.so
