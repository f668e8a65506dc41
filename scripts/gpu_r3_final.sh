# round 3: the measurement job behind profiles/r03_* (tests, bench lines, rocprofv3 passes, scaling model, view sweep)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3f
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -12 > gpurun_out/r3f/tests.log; grep -E "passed|failed" gpurun_out/r3f/tests.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print(\"smoke ok\")" 2>&1 | tail -1
timeout 900 python bench.py 2> gpurun_out/r3f/bench_default.err | tail -1 > gpurun_out/r3f/bench_C3_r8_default.json
timeout 600 python bench.py --no-cpu-baseline --displacement-scale 1.0 2>/dev/null | tail -1 > gpurun_out/r3f/bench_C3_r8_D1.json
timeout 600 python bench.py --no-cpu-baseline --cubemap f32 2>/dev/null | tail -1 > gpurun_out/r3f/bench_C3_f32.json
timeout 600 python bench.py --no-cpu-baseline --cubemap f32 --displacement-scale 1.0 2>/dev/null | tail -1 > gpurun_out/r3f/bench_C3_f32_D1.json
timeout 600 python bench.py --no-cpu-baseline --config C1 --steps 500 2>/dev/null | tail -1 > gpurun_out/r3f/bench_C1_r8.json
timeout 600 python bench.py --no-cpu-baseline --config C2 --steps 500 2>/dev/null | tail -1 > gpurun_out/r3f/bench_C2_r8.json
timeout 600 python bench.py --gpus 2 --share-gpu --steps 10 2>/dev/null | tail -1 > gpurun_out/r3f/bench_C3_gpus2_share_gpu_functional.json
timeout 600 python bench.py --gpus 8 --share-gpu --steps 10 --exchange all_gather 2>/dev/null | tail -1 > gpurun_out/r3f/bench_C3_gpus8_share_gpu_all_gather_functional.json
VPFX_RM_FLAT=1 timeout 600 python bench.py --no-cpu-baseline --steps 50 2>/dev/null | tail -1 > gpurun_out/r3f/bench_C3_r8_flat_raymarch.json
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r3f/bench_*.json')):
    try:
        d=json.load(open(f)); print(f.split('/')[-1], round(d['ms_per_step'],3), {k:(round(v,3) if v else v) for k,v in d['stage_ms'].items()}, round(d['value']), d['config'].get('max_abs_rgba_diff_vs_1gpu_frame'))
    except Exception as e: print(f, 'ERR', e)
PY
python scripts/view_sweep.py C3 2>&1 | grep -v amdgpu > gpurun_out/r3f/view_sweep.txt; cat gpurun_out/r3f/view_sweep.txt
python scripts/lane_bound.py C3 2>&1 | tail -1 > gpurun_out/r3f/lane_bound.txt
timeout 1500 python scripts/scaling_model.py C3 r8 2>&1 | grep -v amdgpu.ids > gpurun_out/r3f/scaling_model_C3_r8.txt; cp gpurun_out/r3/scaling_model_C3_r8.json gpurun_out/r3f/; tail -9 gpurun_out/r3f/scaling_model_C3_r8.txt | cut -c1-150
PROF_DIR=prof_r3 bash scripts/gpu_prof_r3.sh > gpurun_out/r3f/prof.log 2>&1; tail -3 gpurun_out/r3f/prof.log | cut -c1-160
