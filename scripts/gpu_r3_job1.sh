# round 3, job 1: gpu suite on the DONE-templated fill + A/B of the fill micro-variants at D = 0.7 and D = 1
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 > gpurun_out/r3/job1_tests.log
echo "== D=0.7" > gpurun_out/r3/job1_ab.log
STEPS=20 bash scripts/gpu_ab.sh >> gpurun_out/r3/job1_ab.log 2>&1
echo "== D=1" >> gpurun_out/r3/job1_ab.log
STEPS=20 BENCH_ARGS="--displacement-scale 1.0" bash scripts/gpu_ab.sh >> gpurun_out/r3/job1_ab.log 2>&1
tail -5 gpurun_out/r3/job1_tests.log; cat gpurun_out/r3/job1_ab.log
