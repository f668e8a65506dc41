# round 4: the whole GPU suite on the current sources, then a default bench line
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4_tests
timeout 2400 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 | tee gpurun_out/r4_tests/pytest_gpu.txt
timeout 600 python bench.py --steps 100 > gpurun_out/r4_tests/bench_C3.json 2> gpurun_out/r4_tests/bench_C3.err; tail -c 400 gpurun_out/r4_tests/bench_C3.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r4_tests/bench_C3.json').read().strip().splitlines()[-1])
print('C3 ms/step', d['ms_per_step'], d['stage_ms'], 'value', d['value'], 'formula', d['value_formula_units'], 'speedup', d.get('speedup_vs_cpu'))
print(d['config']['samples_executed'], d['config']['samples_formula'])
PY
for cfg in DEMO C1 C2; do
  timeout 600 python bench.py --config $cfg --steps 400 --warmup 5 > gpurun_out/r4_tests/bench_$cfg.json 2> gpurun_out/r4_tests/bench_$cfg.err
  python - <<PY
import json
try:
    d=json.loads(open('gpurun_out/r4_tests/bench_$cfg.json').read().strip().splitlines()[-1])
    print('$cfg ms/step', round(d['ms_per_step'],4), {k:(round(v,4) if v else v) for k,v in d['stage_ms'].items()}, 'value', round(d['value'],1), 'speedup', d.get('speedup_vs_cpu'))
except Exception as e:
    print('$cfg failed', e); print(open('gpurun_out/r4_tests/bench_$cfg.err').read()[-600:])
PY
done
