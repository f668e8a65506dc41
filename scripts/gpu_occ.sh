cd $GRAFT_REPO_ROOT
for lds in 0 50000 70000 120000; do
echo "LDS $lds"
VPFX_FILL_LDS=$lds timeout 600 python bench.py --steps 6 --warmup 2 --no-cpu-baseline 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('ms/step', round(d['ms_per_step'],3), 'stage_ms', {k:round(v,3) for k,v in d['stage_ms'].items() if v is not None})
"
done
