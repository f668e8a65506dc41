cd $GRAFT_REPO_ROOT
PKG=volumetric-particles-for-unity_amd
cp $PKG/libvpfx.so /tmp/libvpfx_main.so
for f in /tmp/libvpfx_main.so _ab/libvpfx_lx*.so; do
  cp $f $PKG/libvpfx.so
  for cfg in C5 C3; do
  echo -n "$(basename $f .so) $cfg : "
  timeout 900 python bench.py --config $cfg --steps $([ $cfg = C5 ] && echo 5 || echo 100) --warmup 2 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('ms/step', round(d['ms_per_step'],4), {k:round(v,4) for k,v in d['stage_ms'].items() if v is not None})"
  done
done
cp /tmp/libvpfx_main.so $PKG/libvpfx.so
