# round 5: LLVM AMDGPU scheduler options on the whole library (scripts/build_ab.sh base "" ilp "-mllvm -amdgpu-sched-strategy=max-ilp" ...): C3 stage times, then
# the oracle-parity tests on any variant that is faster (a different schedule must not change a bit: -ffp-contract=off, no fast-math)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
PKG=volumetric-particles-for-unity_amd
cp $PKG/libvpfx.so /tmp/libvpfx_main.so
out=gpurun_out/r5_sched_ab.txt
: > $out
for rep in 1 2; do
for lib in ${LIBS:-base ilp memclause trackers bias0}; do
  cp _ab/libvpfx_$lib.so $PKG/libvpfx.so
  for cfg in ${CFGS:-C3}; do
    python bench.py --config $cfg --steps ${STEPS:-200} --warmup 5 --no-cpu-baseline --no-variants 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$lib $cfg: ms_per_step %.4f  bin %.4f fill %.4f raymarch_stage %.4f' % (d['ms_per_step'], d['stage_ms']['bin'], d['stage_ms']['fill_kernel'], d['stage_ms']['raymarch_kernel']))" | tee -a $out
  done
done
done
cp /tmp/libvpfx_main.so $PKG/libvpfx.so
