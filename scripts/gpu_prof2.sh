# rocprofv3 passes over the default bench command; outputs under gpurun_out/prof_r1 (copied to profiles/ by hand)
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_r1
rm -rf $OUT; mkdir -p $OUT
cd /tmp
CMD="python $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 2 --no-cpu-baseline"
rocprofv3 --kernel-trace --stats -d $OUT/kt -o kt -- $CMD > $OUT/kt_bench.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU -d $OUT/pmc1 -o pmc1 -- $CMD > $OUT/pmc1.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_THREAD_CYCLES_VALU GRBM_GUI_ACTIVE -d $OUT/pmc2 -o pmc2 -- $CMD > $OUT/pmc2.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $OUT/pmc3 -o pmc3 -- $CMD > $OUT/pmc3.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum -d $OUT/pmc4 -o pmc4 -- $CMD > $OUT/pmc4.log 2>&1
rocprofv3 --kernel-trace --pmc TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum -d $OUT/pmc5 -o pmc5 -- $CMD > $OUT/pmc5.log 2>&1
python $GRAFT_REPO_ROOT/scripts/rocprof_summary.py $OUT/kt/kt_results.db $OUT/pmc1/pmc1_results.db $OUT/pmc2/pmc2_results.db $OUT/pmc3/pmc3_results.db $OUT/pmc4/pmc4_results.db $OUT/pmc5/pmc5_results.db > $OUT/summary.txt 2>&1
grep -h "^{\"metric\"" $OUT/kt_bench.log | tail -1 > $OUT/bench_line.json
python $GRAFT_REPO_ROOT/scripts/traffic_json.py $OUT/pmc3/pmc3_results.db $OUT/pmc4/pmc4_results.db $OUT/traffic_C3.json
rm -rf $OUT/*/*.db
cat $OUT/summary.txt | grep -E '==|k_fill|k_raymarch|kernel' | head -70
