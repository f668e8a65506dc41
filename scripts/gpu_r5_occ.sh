# round 5: the occluders' eye depth inside the ray set-up (k_raymarch<.., OCC>) instead of a k_scene_depth launch: tests that use occluders + the
# oracle-parity tests, then stage times against the library built from HEAD (_ab/libvpfx_head.so) on the same box
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
PKG=volumetric-particles-for-unity_amd
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_boundary.py tests/test_gpu_edge_cases.py tests/test_gpu_unity_plugin.py tests/test_gpu_multi.py tests/test_gpu_brick_format.py -x -q -m gpu 2>&1 | grep -E "passed|failed|error|rror" | tail -8 | tee gpurun_out/r5_occ_tests.log
out=gpurun_out/r5_occ_ab.txt
: > $out
cp $PKG/libvpfx.so /tmp/libvpfx_new.so
for rep in 1 2; do
for lib in head new; do
  if [ $lib = head ]; then cp _ab/libvpfx_head.so $PKG/libvpfx.so; else cp /tmp/libvpfx_new.so $PKG/libvpfx.so; fi
  for cfg in ${CFGS:-DEMO C1 C2 C3}; do
    python bench.py --config $cfg --steps ${STEPS:-400} --warmup 5 --no-cpu-baseline --no-variants 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$lib $cfg: ms_per_step %.4f  bin %.4f fill %.4f raymarch_stage %.4f  samples %d' % (d['ms_per_step'], d['stage_ms']['bin'], d['stage_ms']['fill_kernel'], d['stage_ms']['raymarch_kernel'], d['config']['samples_executed']))" | tee -a $out
  done
done
done
cp /tmp/libvpfx_new.so $PKG/libvpfx.so
