# (before this: python scripts/isa_mix.py profiles/r04_valu_classes.json profiles/r05_isa_mix.json on the final sources -- limiters_json.py prices the class counters with it)
# round 5, final evidence on the final kernel sources (one GPU call): -m gpu suite, rocprofv3 passes (C3, C5), bench lines, view sweep, fuzz sweep, DEMO trace
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r5_final; mkdir -p $OUT
( time timeout 2400 python -m pytest tests -m gpu -q 2>&1 | grep -E "passed|failed|error" | tail -3 ) 2>&1 | tee $OUT/pytest_gpu.txt
PROF_DIR=prof_r5_C3 STEPS=10 bash scripts/gpu_prof_r5.sh > $OUT/prof_C3_tail.txt 2>&1
PROF_DIR=prof_r5_C5 BENCH_ARGS="--config C5" KT_ARGS="--config C5 --steps 3 --warmup 2 --no-variants --no-formula-count" STEPS=3 bash scripts/gpu_prof_r5.sh > $OUT/prof_C5_tail.txt 2>&1
for c in C3 C5; do cp gpurun_out/prof_r5_$c/traffic.json profiles/traffic_${c}_r8.json; cp gpurun_out/prof_r5_$c/limiters.json profiles/limiters_${c}_r8.json; done   # (so that the bench lines below carry them)
( time timeout 900 python bench.py ) > $OUT/bench_C3.json 2> $OUT/bench_C3.err
timeout 600 python bench.py --config C5 --steps 5 --warmup 2 --no-cpu-baseline --no-variants > $OUT/bench_C5.json 2> $OUT/bench_C5.err
for cfg in DEMO C1 C2; do timeout 600 python bench.py --config $cfg --steps 400 --warmup 5 > $OUT/bench_$cfg.json 2> $OUT/bench_$cfg.err; done
timeout 600 python scripts/view_sweep.py C3 2>&1 | grep -v amdgpu.ids | tee $OUT/view_sweep_C3.txt
timeout 1500 python scripts/fuzz_parity.py ${FUZZ:-3000} 1010000 2>&1 | tail -1 | tee $OUT/fuzz_any_nv.log
timeout 900 python scripts/fuzz_parity.py ${FUZZ1:-1000} 970000 2>&1 | tail -1 | tee $OUT/fuzz_first_generation.log
# where a frame of the reference's own scene goes: per-kernel trace of the DEMO bench
export TMPDIR=/tmp; rm -rf $OUT/demo_kt; (cd /tmp && timeout -k 5 300 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$OUT/demo_kt -o kt -- python $GRAFT_REPO_ROOT/bench.py --config DEMO --steps 200 --warmup 5 --no-cpu-baseline --no-formula-count > /dev/null 2>&1)
python scripts/rocprof_summary.py $OUT/demo_kt/kt_results.db 2>/dev/null | head -30 | cut -c1-110 | tee $OUT/demo_kernel_trace.txt; rm -rf $OUT/demo_kt
