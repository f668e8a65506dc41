# everything the round-2 docs quote, in one GPU call: profiles of the default bench (R8 cube map, LDS fill) and of --cubemap f32,
# per-config bench lines (C1, C2, C3 f32, C3 R8 global table), the scaling model, config 5 on one GPU
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2
PROF_DIR=prof_r2 bash scripts/gpu_prof_r2.sh > gpurun_out/r2/prof_r8.log 2>&1
PROF_DIR=prof_r2_f32 BENCH_ARGS="--cubemap f32" bash scripts/gpu_prof_r2.sh > gpurun_out/r2/prof_f32.log 2>&1
# the bench lines below report roofline.traffic from the PMC passes just made (same kernel sources; copy them into profiles/ here as well)
cp gpurun_out/prof_r2/traffic.json profiles/traffic_C3_r8.json; cp gpurun_out/prof_r2_f32/traffic.json profiles/traffic_C3_f32.json
for cfg in C1 C2; do
  timeout 600 python bench.py --config $cfg --steps 200 --warmup 5 2> /dev/null | tail -1 > gpurun_out/r2/bench_$cfg.json
done
timeout 600 python bench.py --cubemap f32 --steps 200 --warmup 5 --no-cpu-baseline 2> /dev/null | tail -1 > gpurun_out/r2/bench_C3_f32.json
timeout 600 python bench.py --no-lds-cubemap --steps 200 --warmup 5 --no-cpu-baseline 2> /dev/null | tail -1 > gpurun_out/r2/bench_C3_r8_global_table.json
timeout 600 python bench.py --no-grey --steps 200 --warmup 5 --no-cpu-baseline 2> /dev/null | tail -1 > gpurun_out/r2/bench_C3_rgba_bricks.json
timeout 600 python bench.py --steps 200 --warmup 5 2> /dev/null | tail -1 > gpurun_out/r2/bench_C3.json
timeout 900 python scripts/scaling_model.py C3 r8 > gpurun_out/r2/scaling_model.log 2>&1
timeout 900 python scripts/run_c5.py > gpurun_out/r2/c5.log 2>&1
tail -3 gpurun_out/r2/c5.log; tail -4 gpurun_out/r2/scaling_model.log
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r2/bench_*.json')):
    try:
        d=json.load(open(f)); print(f.split('/')[-1], round(d['ms_per_step'],3), {k:round(v,3) for k,v in d['stage_ms'].items() if v}, round(d['value'],1), round(d['roofline']['frac'],4))
    except Exception as e: print(f, 'ERR', e)
PY
