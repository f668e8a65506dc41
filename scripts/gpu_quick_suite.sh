# the -m gpu suite (one line) + a short bench of the small and the headline configurations.   gpurun -- 'OUT=gpurun_out/x bash scripts/gpu_quick_suite.sh'
cd $GRAFT_REPO_ROOT
OUT=${OUT:-gpurun_out/quick}; mkdir -p $OUT
timeout 2400 python -m pytest tests -m gpu -q ${PYTEST_ARGS} 2>&1 | grep -E "passed|failed|error|Error" | tail -5 | tee $OUT/pytest_gpu.txt
for cfg in ${CFGS:-DEMO C1 C2 C3}; do
  timeout 600 python bench.py --config $cfg --steps ${STEPS:-400} --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 > $OUT/bench_$cfg.json
  python -c "
import json
d=json.load(open('$OUT/bench_$cfg.json'))
print('$cfg', round(d['ms_per_step'],4), {k:(round(v,4) if v else v) for k,v in d['stage_ms'].items()})" | tee -a $OUT/bench_summary.txt
done
