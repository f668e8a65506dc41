cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/kt_tmp
rm -rf $OUT; mkdir -p $OUT
cd /tmp
rocprofv3 --kernel-trace --stats -d $OUT/kt -o kt -- python $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 2 --no-cpu-baseline > $OUT/log.txt 2>&1
python $GRAFT_REPO_ROOT/scripts/rocprof_summary.py $OUT/kt/kt_results.db | head -16
