# round 4: ray-march locality experiments -- wave pixel-block shape (VPFX_RM_WAVE_LX 3/4/5 = 8x8 / 16x4 / 32x2) x dispatch order (round robin
# in cost order / one compact screen region per XCD): parity, kernel time and L2->fabric read volume (FETCH_SIZE pass) at C3 and C5
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
OUT=gpurun_out/r4_rm_locality; mkdir -p $OUT; : > $OUT/results.txt
for aff in 0 1; do for lx in 3 4 5; do
  export VPFX_RM_XCD_AFFINE=$aff VPFX_RM_WAVE_LX=$lx
  echo "== affine=$aff wave_lx=$lx" | tee -a $OUT/results.txt
  timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_edge_cases.py -x -q 2>&1 | tail -1 | tee -a $OUT/results.txt
  for cfg in ${CFGS:-C3 C5}; do
    steps=40; [ $cfg = C5 ] && steps=3
    timeout 600 python bench.py --config $cfg --steps $steps --warmup 2 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$cfg', 'ms/step', round(d['ms_per_step'],3), {k:round(v,3) for k,v in d['stage_ms'].items() if v is not None}, 'Gsamples/s', round(d['raymarch_msamples_per_s']/1e3,1))" | tee -a $OUT/results.txt
    rm -rf /tmp/abf; (cd /tmp && timeout -k 5 400 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d /tmp/abf -o p -- python $GRAFT_REPO_ROOT/bench.py --config $cfg --steps 3 --warmup 2 --no-cpu-baseline > /dev/null 2>&1)
    python scripts/rocprof_summary.py /tmp/abf/p_results.db 2>/dev/null | grep -E "k_raymarch.*FETCH_SIZE" | cut -c1-110 | tee -a $OUT/results.txt
  done
done; done
