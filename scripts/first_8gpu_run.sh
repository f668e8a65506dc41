#!/bin/bash
# First run of the fan-out on a real 8-GPU node (it has only ever run as N slabs on ONE GPU through the peer-copy hook, plus one rank on a real
# communicator).  Ordered so that the first failure is the most informative one: RCCL chatter on, one log per rank, ONE collective type
# first (--exchange all_gather: ncclAllGather + one send/recv pair), then the default exchange (grouped ncclSend/ncclRecv all-to-all), first
# one process for all GPUs (ncclCommInitAll from the library's worker threads), then one process per GPU (ncclCommInitRank, torchrun).
# Every exchange is bounded by the library's time-out (vp_config.reserved[2], default 120 s): a rank that leaves an exchange makes every rank
# return VP_ERR_RCCL (and the process exit non-zero) instead of hanging the node; each step also runs under `timeout`.
# usage: scripts/first_8gpu_run.sh [N=8] [steps=20]       logs: gpurun_out/first_8gpu/
N=${1:-8}; STEPS=${2:-20}
cd "$(dirname "$0")/.."
OUT=gpurun_out/first_8gpu; mkdir -p $OUT
export HSA_ENABLE_IPC_MODE_LEGACY=0 NCCL_DEBUG=INFO NCCL_DEBUG_SUBSYS=INIT,COLL,P2P
rc_all=0
step() {   # name, command...
  local name=$1; shift
  echo "== $name: $*" | tee -a $OUT/summary.txt
  NCCL_DEBUG_FILE=$OUT/${name}_rccl_%h_%p.log timeout -k 10 600 "$@" > $OUT/$name.out 2> $OUT/$name.err
  local rc=$?
  echo "   rc=$rc $(grep -h '^{"metric"' $OUT/$name.out | tail -1 | python3 -c "import json,sys
try:
    d=json.loads(sys.stdin.read()); print('ms/step', round(d['ms_per_step'],3), 'err_vs_1gpu', d['config']['max_abs_rgba_diff_vs_1gpu_frame'], 'slabs', d['config']['slabs'])
except Exception as e: print('(no bench line)')")" | tee -a $OUT/summary.txt
  [ $rc -ne 0 ] && { rc_all=$rc; tail -5 $OUT/$name.err | tee -a $OUT/summary.txt; }
  return $rc
}
rocm-smi --showtopo > $OUT/topology.txt 2>&1
step one_process_all_gather  python bench.py --gpus $N --steps $STEPS --warmup 3 --no-cpu-baseline --exchange all_gather || exit $rc_all
step one_process_tiles       python bench.py --gpus $N --steps $STEPS --warmup 3 --no-cpu-baseline || exit $rc_all
step torchrun_all_gather     python -m torch.distributed.run --nnodes=1 --nproc-per-node $N --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus $N --steps $STEPS --warmup 3 --exchange all_gather || exit $rc_all
step torchrun_tiles          python -m torch.distributed.run --nnodes=1 --nproc-per-node $N --master-addr 127.0.0.1 --master-port 29518 bench.py --gpus $N --steps $STEPS --warmup 3 || exit $rc_all
for n in 2 4; do step torchrun_tiles_$n python -m torch.distributed.run --nnodes=1 --nproc-per-node $n --master-addr 127.0.0.1 --master-port 2952$n bench.py --gpus $n --steps $STEPS --warmup 3; done
echo "all steps done, rc=$rc_all" | tee -a $OUT/summary.txt
exit $rc_all
