# rocprofv3 passes over a bench command (rounds 4-5: also used for config C5 with BENCH_ARGS="--config C5", STEPS=3); outputs under gpurun_out/$PROF_DIR (copied to profiles/ by hand).
# --kernel-trace --stats first, then PMC counters in separate passes (never combined with other trace domains).
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/${PROF_DIR:-prof_r5}
rm -rf $OUT; mkdir -p $OUT
cd /tmp
# counter passes: the headline path alone (no content variants: they launch the same kernel names on other inputs), few steps
CMD="python $GRAFT_REPO_ROOT/bench.py --steps ${STEPS:-10} --warmup 2 --no-cpu-baseline --no-formula-count --no-variants ${BENCH_ARGS}"
# kernel trace: the command the driver runs (default steps and warm-up, variants included) minus the CPU leg, unless KT_ARGS says otherwise
KT_CMD="python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline ${KT_ARGS:-${BENCH_ARGS}}"
timeout -k 5 ${KT_TIMEOUT:-400} rocprofv3 --kernel-trace --stats -d $OUT/kt -o kt -- $KT_CMD > $OUT/kt_bench.log 2>&1
i=0
for set in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU" \
           "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_THREAD_CYCLES_VALU GRBM_GUI_ACTIVE" \
           "FETCH_SIZE" "WRITE_SIZE TCC_HIT_sum TCC_MISS_sum" "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS" "SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_SALU SQ_INSTS_VALU_TRANS" \
           "SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_CVT"; do
  i=$((i+1))
  timeout -k 5 ${PMC_TIMEOUT:-300} rocprofv3 --kernel-trace --pmc $set -d $OUT/pmc$i -o pmc$i -- $CMD > $OUT/pmc$i.log 2>&1
done
python $GRAFT_REPO_ROOT/scripts/rocprof_summary.py $OUT/kt/kt_results.db $OUT/pmc*/pmc*_results.db > $OUT/summary.txt 2>&1
grep -h "^{\"metric\"" $OUT/kt_bench.log | tail -1 > $OUT/bench_line.json
python $GRAFT_REPO_ROOT/scripts/traffic_json.py $OUT/pmc3/pmc3_results.db $OUT/pmc4/pmc4_results.db $OUT/traffic.json > $OUT/traffic.log 2>&1
python $GRAFT_REPO_ROOT/scripts/limiters_json.py $OUT/limiters.json $OUT/pmc*/pmc*_results.db > $OUT/limiters.log 2>&1
rm -rf $OUT/*/*.db
grep -E "k_fill|k_raymarch" $OUT/summary.txt | grep -v "k_fill_value\|k_fill_finish" | cut -c1-140 | head -80
