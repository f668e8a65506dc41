cd $GRAFT_REPO_ROOT
PKG=volumetric-particles-for-unity_amd
cp $PKG/libvpfx.so /tmp/libvpfx_main.so
cp _ab/libvpfx_noorder_ab.so $PKG/libvpfx.so
for rep in 1 2; do for l in 3 4 5; do echo "== C5 raster order, wave block 2^$l px wide"; VPFX_RM_WAVE_LX=$l timeout 900 python scripts/camera_sweep.py C5 2>&1 | grep -v amdgpu.ids | head -5; done; done
cp /tmp/libvpfx_main.so $PKG/libvpfx.so
