cd $GRAFT_REPO_ROOT
PKG=volumetric-particles-for-unity_amd
cp $PKG/libvpfx.so /tmp/libvpfx_main.so
cp _ab/libvpfx_ordermode.so $PKG/libvpfx.so
for cfg in C3 C5; do for m in 0 -1; do echo "== $cfg order mode $m"; VPFX_RM_ORDER_MODE=$m timeout 900 python scripts/camera_sweep.py $cfg 2>&1 | grep -v amdgpu.ids; done; done
cp /tmp/libvpfx_main.so $PKG/libvpfx.so
