# round 6: the whole GPU suite as the driver runs it (timed), then the default bench line + the DEMO / C1 / C2 lines
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
( time timeout 2400 python -m pytest tests -x -q -m gpu --durations=12 2>&1 | tail -40 ) 2>&1 | tee gpurun_out/r6_gpu_suite.log
( time python bench.py ) > gpurun_out/r6_bench_default.json 2> gpurun_out/r6_bench_default.err
tail -c 3000 gpurun_out/r6_bench_default.json; tail -5 gpurun_out/r6_bench_default.err
for cfg in DEMO C1 C2; do
  python bench.py --config $cfg --steps 300 --warmup 5 --no-cpu-baseline --no-variants 2>/dev/null > gpurun_out/r6_bench_$cfg.json
  python -c "
import json,sys
d=json.loads(open('gpurun_out/r6_bench_$cfg.json').read().strip().splitlines()[-1])
print('$cfg: ms_per_step %.4f  bin %.4f fill %.4f raymarch_stage %.4f  samples %d' % (d['ms_per_step'], d['stage_ms']['bin'], d['stage_ms']['fill_kernel'], d['stage_ms']['raymarch_kernel'], d['config']['samples_executed']))"
done
