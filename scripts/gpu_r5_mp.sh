# round 5: the driver's one-process-per-GPU command on ONE GPU through the multi-process RCCL stand-in (tests/tools/fake_rccl_mp.cpp)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_bench_cli.py -x -q -m gpu --durations=8 2>&1 | tail -80 | tee gpurun_out/r5_mp_bench_cli.log
