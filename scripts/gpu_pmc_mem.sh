# extra PMC passes on the memory pipes (at most two counters of one hardware block per pass -- more makes rocprofv3 abort and then hang --
# and every pass under its own timeout) (TA / TCP / TCC) of the two dominant kernels; summary under gpurun_out/pmc_mem
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_mem
rm -rf $OUT; mkdir -p $OUT
cd /tmp
CMD="python $GRAFT_REPO_ROOT/bench.py --steps 6 --warmup 2 --no-cpu-baseline"
i=0
for set in "TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum" "TA_DATA_STALLED_BY_TC_CYCLES_sum TA_FLAT_READ_WAVEFRONTS_sum" \
           "TCP_PENDING_STALL_CYCLES_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum" "TCP_TCR_TCP_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum" \
           "TCP_TCC_READ_REQ_LATENCY_sum TCP_GATE_EN1_sum" "TCC_BUSY_sum TCC_TAG_STALL_sum" "TCC_REQ_sum TCC_READ_sum" "GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  timeout -k 5 150 rocprofv3 --kernel-trace --pmc $set -d $OUT/p$i -o p$i -- $CMD > $OUT/p$i.log 2>&1
done
python $GRAFT_REPO_ROOT/scripts/rocprof_summary.py $OUT/p*/p*_results.db 2>&1 | grep -E "k_fill|k_raymarch<32" | grep -vE "^\S+\s+\S+\s+[0-9]+\s+[0-9.]+\s+[0-9.]+\s+[0-9.]+$" > $OUT/summary.txt
rm -rf $OUT/*/*.db
cat $OUT/summary.txt | cut -c1-110
