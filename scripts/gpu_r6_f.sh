# round 6, sixth batch: the whole GPU suite on the current build (timed), then development fuzz sweeps on it
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
( time timeout 2400 python -m pytest tests -x -q -m gpu --durations=8 2>&1 | tail -20 ) 2>&1 | tee gpurun_out/r6_gpu_suite_nt64.log
( time timeout 3000 python scripts/fuzz_parity.py 4000 3000400 2>&1 | tail -2 ) 2>&1 | tee gpurun_out/r6_fuzz_4000_occluders_3000400.log
( time timeout 2400 python scripts/fuzz_parity.py 2000 2100000 2>&1 | tail -2 ) 2>&1 | tee gpurun_out/r6_fuzz_2000_any_nv_2100000.log
( time timeout 2400 python scripts/fuzz_parity.py 1500 70000 2>&1 | tail -2 ) 2>&1 | tee gpurun_out/r6_fuzz_1500_first_generation_70000.log
