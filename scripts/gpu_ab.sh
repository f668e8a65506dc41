# A/B timing of library variants on one box: every _ab/libvpfx_<name>.so is swapped in for libvpfx.so and benched.
# usage (here):  scripts/build_ab.sh name "EXTRA flags" ...   then   gpurun -- 'bash scripts/gpu_ab.sh'
cd $GRAFT_REPO_ROOT
PKG=volumetric-particles-for-unity_amd
cp $PKG/libvpfx.so /tmp/libvpfx_main.so
for rep in 1 2; do
for f in /tmp/libvpfx_main.so _ab/libvpfx_*.so; do
  cp $f $PKG/libvpfx.so
  echo -n "$(basename $f) : "
  timeout 600 python bench.py --steps ${STEPS:-10} --warmup 2 --no-cpu-baseline ${BENCH_ARGS} 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('ms/step', round(d['ms_per_step'],3), {k:round(v,3) for k,v in d['stage_ms'].items() if v is not None})
"
done
done
cp /tmp/libvpfx_main.so $PKG/libvpfx.so
