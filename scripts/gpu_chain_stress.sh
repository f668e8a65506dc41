cd $GRAFT_REPO_ROOT
PKG=volumetric-particles-for-unity_amd
cp $PKG/libvpfx.so /tmp/libvpfx_main.so
for f in /tmp/libvpfx_main.so _ab/libvpfx_nochain.so; do
  cp $f $PKG/libvpfx.so; echo $(basename $f)
  timeout 900 python scripts/chain_stress.py C3 60 2>&1 | tail -1
  timeout 900 python scripts/chain_stress.py C2 60 2>&1 | tail -1
  timeout 900 python scripts/chain_stress.py C1 100 2>&1 | tail -1
done
cp /tmp/libvpfx_main.so $PKG/libvpfx.so
