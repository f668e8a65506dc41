# round 5: the C3 rocprofv3 passes again (kernel trace of the driver's command, counter passes of the headline path alone), then the bench line that carries them
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r5_final; mkdir -p $OUT
PROF_DIR=prof_r5_C3 STEPS=10 bash scripts/gpu_prof_r5.sh > $OUT/prof_C3_tail.txt 2>&1
cp gpurun_out/prof_r5_C3/traffic.json profiles/traffic_C3_r8.json; cp gpurun_out/prof_r5_C3/limiters.json profiles/limiters_C3_r8.json
( time timeout 900 python bench.py ) > $OUT/bench_C3.json 2> $OUT/bench_C3.err
tail -8 $OUT/prof_C3_tail.txt
