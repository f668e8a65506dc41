# round 6, fourth batch: third-generation fuzz sweep (opaque solids), what the fill's HBM writes consist of (EA request counters) at C3 and C5
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
( time timeout 2400 python scripts/fuzz_parity.py ${FUZZ_N:-400} 3000000 2>&1 | tail -6 ) 2>&1 | tee gpurun_out/r6_fuzz_occluders_3000000.log
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/r6_ea_counters; rm -rf $OUT; mkdir -p $OUT
cd /tmp
for cfg in C3 C5; do
  steps=10; [ $cfg = C5 ] && steps=3
  CMD="python $GRAFT_REPO_ROOT/bench.py --config $cfg --steps $steps --warmup 2 --no-cpu-baseline --no-formula-count --no-variants"
  i=0
  for set in "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_EA0_ATOMIC_sum" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum" "TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum" "TCC_WRITE_REQ_sum TCC_ATOMIC_sum TCC_WRITE_SECTORS_sum" "TCC_EA0_WRREQ_DRAM_sum TCC_EA0_RDREQ_DRAM_sum"; do
    i=$((i+1))
    timeout -k 5 400 rocprofv3 --kernel-trace --pmc $set -d $OUT/${cfg}_pmc$i -o pmc$i -- $CMD > $OUT/${cfg}_pmc$i.log 2>&1
  done
  python $GRAFT_REPO_ROOT/scripts/rocprof_summary.py $OUT/${cfg}_pmc*/pmc*_results.db 2>&1 | grep -E "k_fill|k_raymarch<" | grep -v "k_fill_value\|k_fill_finish" > $OUT/summary_$cfg.txt
  cat $OUT/summary_$cfg.txt | cut -c1-160
done
rm -rf $OUT/*/*.db $OUT/*_pmc*/
