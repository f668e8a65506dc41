cd $GRAFT_REPO_ROOT
{
STEPS=30 timeout 600 bash scripts/gpu_ab.sh
timeout 700 python -m pytest tests -m gpu -x -q 2>&1 | grep "passed\|failed" | tail -2
for c in C1 C2; do timeout 200 python bench.py --config $c --steps 200 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$c ms/step', round(d['ms_per_step'],3), {k:round(v,3) for k,v in d['stage_ms'].items() if v is not None})"; done
} 2>&1 | grep -v amdgpu
