cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
PKG=volumetric-particles-for-unity_amd
cp $PKG/libvpfx.so /tmp/libvpfx_main.so
cp _ab/libvpfx_ab.so $PKG/libvpfx.so
out=gpurun_out/r5_xcd_affine_small_frames.txt
: > $out
for rep in 1 2; do
for cfg in DEMO C1 C2; do
  for sw in 0 1; do
    VPFX_RM_XCD_AFFINE=$sw python bench.py --config $cfg --steps 400 --warmup 5 --no-cpu-baseline --no-variants 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$cfg xcd_affine=$sw: ms_per_step %.4f  raymarch_stage %.4f' % (d['ms_per_step'], d['stage_ms']['raymarch_kernel']))" | tee -a $out
  done
done
done
cp /tmp/libvpfx_main.so $PKG/libvpfx.so
