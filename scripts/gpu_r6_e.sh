# round 6, fifth batch: non-temporal brick stores (A/B): time at C3 / C5, HBM traffic (FETCH_SIZE / WRITE_SIZE) at C3 / C5 with the variant swapped in, parity tests on it
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
STEPS=100 bash scripts/gpu_ab.sh 2>&1 | tee gpurun_out/r6_ab_nt_C3.txt
STEPS=10 BENCH_ARGS="--config C5 --no-variants --no-formula-count" bash scripts/gpu_ab.sh 2>&1 | tee gpurun_out/r6_ab_nt_C5.txt
PKG=volumetric-particles-for-unity_amd
cp $PKG/libvpfx.so /tmp/libvpfx_main.so; cp _ab/libvpfx_nt.so $PKG/libvpfx.so
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/r6_nt_traffic; rm -rf $OUT; mkdir -p $OUT
( cd /tmp
for cfg in C3 C5; do
  steps=10; [ $cfg = C5 ] && steps=3
  CMD="python $GRAFT_REPO_ROOT/bench.py --config $cfg --steps $steps --warmup 2 --no-cpu-baseline --no-formula-count --no-variants"
  timeout -k 5 400 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $OUT/${cfg}_f -o f -- $CMD > $OUT/${cfg}_f.log 2>&1
  timeout -k 5 400 rocprofv3 --kernel-trace --pmc WRITE_SIZE TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum -d $OUT/${cfg}_w -o w -- $CMD > $OUT/${cfg}_w.log 2>&1
  python $GRAFT_REPO_ROOT/scripts/rocprof_summary.py $OUT/${cfg}_f/f_results.db $OUT/${cfg}_w/w_results.db 2>&1 | grep -E "k_fill|k_raymarch<" | grep "SIZE\|REQ" > $OUT/summary_$cfg.txt
  cat $OUT/summary_$cfg.txt | cut -c1-130
done )
rm -rf $OUT/*_f $OUT/*_w
cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_chain.py tests/test_gpu_brick_format.py tests/test_gpu_lds_cubemap.py -x -q -m gpu 2>&1 | tail -3 | tee gpurun_out/r6_nt_parity_tests.log
cp /tmp/libvpfx_main.so $PKG/libvpfx.so
