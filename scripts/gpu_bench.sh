cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python bench.py "$@" 2> gpurun_out/bench_stderr.log | tail -1 > gpurun_out/bench_last.json
python - <<'PY'
import json
d=json.load(open('gpurun_out/bench_last.json'))
print('ms/step', round(d['ms_per_step'],3), 'stage_ms', {k:round(v,3) for k,v in d['stage_ms'].items() if v is not None}, 'value', round(d['value'],1))
print('roof', {k:(round(v['achieved'],1), round(v['frac'],4)) for k,v in d['roofline_all'].items()})
if 'cpu_baseline' in d: print('cpu', d['cpu_baseline'])
PY
