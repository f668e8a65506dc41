# round 6, second batch: C5 slab test again (fixed sanity threshold), C5 scaling model (planner cut from the whole-grid profile), two-stream pipeline probe,
# the PMC passes over the headline + the cliff / any-nv paths, the HBM-side request counters available on this device
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
( time timeout 1500 python -m pytest tests/test_gpu_large_configs.py -x -q -m gpu -k "slab_by_slab" -s 2>&1 | tail -8 ) 2>&1 | tee gpurun_out/r6_c5_slab_test.log
( time timeout 2400 python scripts/scaling_model.py C5 r8 2>&1 | tail -30 ) 2>&1 | tee gpurun_out/r6_scaling_model_C5_r8.txt
timeout 600 python scripts/pipeline2_probe.py C3 300 2>&1 | tail -3 | tee gpurun_out/r6_pipeline2_probe.txt
timeout 600 python scripts/pipeline2_probe.py C2 300 2>&1 | tail -1 | tee -a gpurun_out/r6_pipeline2_probe.txt
timeout 600 python scripts/pipeline2_probe.py C1 500 2>&1 | tail -1 | tee -a gpurun_out/r6_pipeline2_probe.txt
( cd /tmp && export TMPDIR=/tmp && rocprofv3 -L 2>/dev/null | grep -o "TCC_EA[A-Z0-9_]*\|TCC_[A-Z_]*WR[A-Z0-9_]*\|TCC_[A-Z_]*ATOMIC[A-Z0-9_]*" | sort -u ) > gpurun_out/r6_tcc_counters_available.txt 2>&1
wc -l gpurun_out/r6_tcc_counters_available.txt
bash scripts/gpu_prof_r6.sh 2>&1 | tee gpurun_out/r6_prof_overview.txt
