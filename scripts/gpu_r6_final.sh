# round 6, final batch on the final sources: PMC + kernel-trace passes (headline, cliff / any-nv paths, C5, DEMO), the bench lines, the C3 scaling model
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r6_bench_lines
bash scripts/gpu_prof_r6.sh 2>&1 | tee gpurun_out/r6_prof_overview.txt
PROF_DIR=prof_r6_C5 BENCH_ARGS="--config C5" KT_ARGS="--config C5 --steps 5 --no-variants" STEPS=3 KT_TIMEOUT=900 PMC_TIMEOUT=600 bash scripts/gpu_prof_r5.sh > gpurun_out/prof_r6_C5.log 2>&1
PROF_DIR=prof_r6_DEMO BENCH_ARGS="--config DEMO" KT_ARGS="--config DEMO --steps 300" STEPS=50 bash scripts/gpu_prof_r5.sh > gpurun_out/prof_r6_DEMO.log 2>&1
python bench.py > gpurun_out/r6_bench_lines/bench_C3_r8.json 2> gpurun_out/r6_bench_lines/bench_C3_r8.err
python bench.py --config C5 --steps 10 --warmup 2 --no-cpu-baseline --no-variants > gpurun_out/r6_bench_lines/bench_C5_r8_one_gpu.json 2>/dev/null
for cfg in C2 C1 DEMO; do python bench.py --config $cfg --steps 300 --warmup 5 --no-cpu-baseline > gpurun_out/r6_bench_lines/bench_${cfg}_r8.json 2>/dev/null; done
for f in gpurun_out/r6_bench_lines/*.json; do python -c "
import json,sys
d=json.loads(open('$f').read().strip().splitlines()[-1])
print('$f'.split('/')[-1], 'ms/step %.4f' % d['ms_per_step'], {k: round(v,4) for k,v in d['stage_ms'].items() if v}, 'value %.1f' % d['value'], 'frac %.4f' % d['roofline']['frac'])"; done
( time timeout 1800 python scripts/scaling_model.py C3 r8 2>&1 | tail -14 ) 2>&1 | tee gpurun_out/r6_scaling_model_C3_r8.txt
cp gpurun_out/r6_scaling/scaling_model_C3_r8.json gpurun_out/r6_scaling_model_C3_r8.json 2>/dev/null
