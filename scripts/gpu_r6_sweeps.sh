# round 6: long development sweeps on the final sources (parity evidence; ~50 min)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
( time timeout 3300 python scripts/fuzz_parity.py 8000 3100000 2>&1 | tail -2 ) 2>&1 | tee gpurun_out/r6_fuzz_8000_occluders_3100000.log
( time timeout 1800 python scripts/fuzz_parity.py 4000 2200000 2>&1 | tail -2 ) 2>&1 | tee gpurun_out/r6_fuzz_4000_any_nv_2200000.log
