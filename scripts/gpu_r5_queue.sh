# round 5: the phase-split ray-march (k_raymarch_q: traversal / sampling decoupled by a visit queue in LDS) against k_raymarch.
# A/B builds (scripts/build_ab.sh q6 "-DVPFX_AB=1" ...); VPFX_RM_QUEUE=1 selects the new kernel.  First parity (the oracle tests with the switch on), then stage times.
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
PKG=volumetric-particles-for-unity_amd
cp $PKG/libvpfx.so /tmp/libvpfx_main.so
out=gpurun_out/r5_queue_ab.txt
: > $out
first=1
for lib in ${LIBS:-q6 q6w3 q10w3}; do
  cp _ab/libvpfx_$lib.so $PKG/libvpfx.so
  if [ $first = 1 ]; then
    VPFX_RM_QUEUE=1 timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_boundary.py tests/test_gpu_edge_cases.py tests/test_gpu_brick_format.py tests/test_gpu_lds_cubemap.py -x -q -m gpu 2>&1 | grep -E "passed|failed|error|Error|assert" | tail -8 | tee -a $out
    VPFX_RM_QUEUE=1 timeout 600 python scripts/fuzz_parity.py 200 1020000 2>&1 | tail -3 | tee -a $out
    first=0
  fi
  for cfg in ${CFGS:-DEMO C1 C2 C3}; do
    for qsw in 0 1; do
      VPFX_RM_QUEUE=$qsw python bench.py --config $cfg --steps ${STEPS:-300} --warmup 5 --no-cpu-baseline --no-variants 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$lib $cfg queue=$qsw: ms_per_step %.4f  raymarch_stage %.4f  samples %d' % (d['ms_per_step'], d['stage_ms']['raymarch_kernel'], d['config']['samples_executed']))" | tee -a $out
    done
  done
done
cp /tmp/libvpfx_main.so $PKG/libvpfx.so
