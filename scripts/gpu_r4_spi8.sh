# ray-march samples per iteration / waves per SIMD on small screens (DEMO, C1) and the headline config: A/B builds of scripts/build_ab.sh.   gpurun -- 'bash scripts/gpu_r4_spi8.sh'
cd $GRAFT_REPO_ROOT
PKG=volumetric-particles-for-unity_amd
OUT=gpurun_out/spi8; mkdir -p $OUT
cp $PKG/libvpfx.so /tmp/libvpfx_main.so
for rep in 1 2; do
for f in /tmp/libvpfx_main.so _ab/libvpfx_*.so; do
  cp $f $PKG/libvpfx.so
  for cfg in DEMO C1 C2 C3; do
  echo -n "$(basename $f .so) $cfg : " | tee -a $OUT/log.txt
  timeout 600 python bench.py --config $cfg --steps 300 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('ms/step', round(d['ms_per_step'],4), {k:round(v,4) for k,v in d['stage_ms'].items() if v is not None})" | tee -a $OUT/log.txt
  done
done
done
cp /tmp/libvpfx_main.so $PKG/libvpfx.so
