cd $GRAFT_REPO_ROOT
PKG=volumetric-particles-for-unity_amd
cp $PKG/libvpfx.so /tmp/libvpfx_main.so
for f in /tmp/libvpfx_main.so _ab/libvpfx_blk44.so; do
  cp $f $PKG/libvpfx.so
  echo "== $(basename $f .so): camera sweep C5"; timeout 900 python scripts/camera_sweep.py C5 2>&1 | grep -v amdgpu.ids
done
cp /tmp/libvpfx_main.so $PKG/libvpfx.so
