# A/B of the blocked brick layouts (VPFX_RM_BLK, scripts/build_ab.sh blk44 / blk82) against the product's rows along x: view sweep with image
# fingerprints (the images must be the product's bit for bit), C3 step, C5 ray-march.   gpurun -- 'bash scripts/gpu_r4_blocked.sh'
cd $GRAFT_REPO_ROOT
PKG=volumetric-particles-for-unity_amd
OUT=gpurun_out/blocked; mkdir -p $OUT
cp $PKG/libvpfx.so /tmp/libvpfx_main.so
for f in /tmp/libvpfx_main.so _ab/libvpfx_*.so; do
  n=$(basename $f .so)
  cp $f $PKG/libvpfx.so
  echo "== $n: view sweep C3" | tee -a $OUT/log.txt
  timeout 600 python scripts/view_sweep.py C3 2>&1 | tee -a $OUT/log.txt
  echo "== $n: bench C3" | tee -a $OUT/log.txt
  timeout 600 python bench.py --steps 100 --warmup 5 --no-cpu-baseline 2>&1 | tail -1 > $OUT/bench_C3_$n.json
  python -c "
import json,sys
d=json.load(open('$OUT/bench_C3_$n.json'))
print('ms/step', round(d['ms_per_step'],3), {k:round(v,3) for k,v in d['stage_ms'].items() if v is not None})" | tee -a $OUT/log.txt
done
if [ -n "$WITH_C5" ]; then
for f in /tmp/libvpfx_main.so _ab/libvpfx_blk44.so _ab/libvpfx_fix44.so; do
  n=$(basename $f .so)
  cp $f $PKG/libvpfx.so
  echo "== $n: bench C5" | tee -a $OUT/log.txt
  timeout 900 python bench.py --config C5 --steps 5 --warmup 1 --no-cpu-baseline 2>&1 | tail -1 > $OUT/bench_C5_$n.json
  python -c "
import json,sys
d=json.load(open('$OUT/bench_C5_$n.json'))
print('ms/step', round(d['ms_per_step'],3), {k:round(v,3) for k,v in d['stage_ms'].items() if v is not None})" | tee -a $OUT/log.txt
done
fi
cp /tmp/libvpfx_main.so $PKG/libvpfx.so
