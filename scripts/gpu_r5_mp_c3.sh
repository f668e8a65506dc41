# round 5: the driver's scaling command line (one process per rank under torch.distributed.run) at the BENCHMARK config on ONE GPU, RCCL replaced by the
# multi-process stand-in (VPFX_RCCL_LIBRARY): a functional record -- N processes share one GPU and the exchanges are staged through host memory, so the
# times say nothing about scaling; the line proves the command runs to its end at full size and that the sharded frame equals the 1-GPU frame.
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r5_mp
export VPFX_RCCL_LIBRARY=$PWD/tests/tools/_build/libfake_rccl_mp.so FAKE_RCCL_TIMEOUT_MS=120000 MASTER_ADDR=127.0.0.1
for n in 2 4 8; do
  for ex in tiles all_gather; do
    timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node $n --master-addr 127.0.0.1 --master-port $((29500 + n)) bench.py --gpus $n --share-gpu \
        --steps 10 --warmup 3 --exchange $ex > gpurun_out/r5_mp/bench_C3_N${n}_${ex}.json 2> gpurun_out/r5_mp/bench_C3_N${n}_${ex}.err
    echo "N=$n $ex rc=$?"; tail -c 600 gpurun_out/r5_mp/bench_C3_N${n}_${ex}.json; echo; tail -3 gpurun_out/r5_mp/bench_C3_N${n}_${ex}.err
  done
done
ls /tmp/fake-rccl-mp-* 2>/dev/null | head
