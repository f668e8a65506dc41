# dispatch order of k_raymarch at C5 (memory-paced) and C3: super-tiles by cost (0, the product at C3), raster (-1), blocks of 2^n x 2^n
# super-tiles by cost with raster inside (n = 1..4).  A/B build: scripts/build_ab.sh ordermode "-DVPFX_AB=1".   gpurun -- 'bash scripts/gpu_r4_rm_order_c5.sh'
cd $GRAFT_REPO_ROOT
PKG=volumetric-particles-for-unity_amd
OUT=gpurun_out/rm_order; mkdir -p $OUT
cp $PKG/libvpfx.so /tmp/libvpfx_main.so
cp _ab/libvpfx_ordermode.so $PKG/libvpfx.so
for cfg in C5 C3 C2; do
for m in 0 -1 1 2 3 4; do
  echo -n "$cfg order mode $m : " | tee -a $OUT/log.txt
  VPFX_RM_ORDER_MODE=$m timeout 900 python bench.py --config $cfg --steps $([ $cfg = C5 ] && echo 5 || echo 100) --warmup 2 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('ms/step', round(d['ms_per_step'],4), {k:round(v,4) for k,v in d['stage_ms'].items() if v is not None})" | tee -a $OUT/log.txt
done
done
cp /tmp/libvpfx_main.so $PKG/libvpfx.so
