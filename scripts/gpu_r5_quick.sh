# round 5: quick check after a ray-march / fill change: the oracle-parity tests, then the bench stages of DEMO / C1 / C2 / C3 (+ C5's ray-march if asked)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_boundary.py tests/test_gpu_edge_cases.py tests/test_gpu_slabs.py tests/test_gpu_multi.py -x -q -m gpu 2>&1 | grep -E "passed|failed|error" | tail -5 | tee gpurun_out/r5_quick_tests.log
out=gpurun_out/r5_quick_bench.txt
: > $out
for cfg in ${CFGS:-DEMO C1 C2 C3}; do
  python bench.py --config $cfg --steps ${STEPS:-300} --warmup 5 --no-cpu-baseline --no-variants 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$cfg: ms_per_step %.4f  bin %.4f fill %.4f raymarch_stage %.4f  samples %d' % (d['ms_per_step'], d['stage_ms']['bin'], d['stage_ms']['fill_kernel'], d['stage_ms']['raymarch_kernel'], d['config']['samples_executed']))" | tee -a $out
done
