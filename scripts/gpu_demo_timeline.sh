# kernel-by-kernel timeline of a few steady-state frames of the DEMO (or $CFG) bench.   gpurun -- 'bash scripts/gpu_demo_timeline.sh'
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/timeline; mkdir -p $OUT; rm -rf $OUT/kt
(cd /tmp && timeout -k 5 300 rocprofv3 --kernel-trace -d $OUT/kt -o kt -- python $GRAFT_REPO_ROOT/bench.py --config ${CFG:-DEMO} --steps 200 --warmup 5 --no-cpu-baseline --no-formula-count > /dev/null 2>&1)
python scripts/frame_timeline.py $OUT/kt/kt_results.db ${FRAMES:-4} | tee $OUT/timeline_${CFG:-DEMO}.txt
rm -rf $OUT/kt
