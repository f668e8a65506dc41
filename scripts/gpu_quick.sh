cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu 2>&1 | tail -3
timeout 600 python bench.py --steps 10 --warmup 2 --no-cpu-baseline 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('ms/step', round(d['ms_per_step'],3), 'stage_ms', {k:round(v,3) for k,v in d['stage_ms'].items() if v is not None}, 'value', round(d['value'],1))
print('roof', {k:(round(v['achieved'],1), round(v['frac'],4)) for k,v in d['roofline_all'].items()})
"
