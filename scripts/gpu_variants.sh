# run a command with every _ab/libvpfx_<name>.so swapped in for libvpfx.so:  bash scripts/gpu_variants.sh python scripts/dbg_c3.py 32 32 10000 0
cd $GRAFT_REPO_ROOT
PKG=volumetric-particles-for-unity_amd
cp $PKG/libvpfx.so /tmp/libvpfx_main.so
for f in /tmp/libvpfx_main.so _ab/libvpfx_*.so; do
  cp $f $PKG/libvpfx.so
  echo -n "$(basename $f) : "
  timeout 300 "$@" 2>&1 | grep -v amdgpu.ids | tail -1 | cut -c1-160
done
cp /tmp/libvpfx_main.so $PKG/libvpfx.so
