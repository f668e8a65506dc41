# The one-process-per-GPU launch of bench.py (what the driver's scaling bench uses) on a ONE-GPU box: two ranks, both on cuda:0.  RCCL must refuse
# the communicator ("Duplicate GPU detected") -- which it can only do after its bootstrap exchanged the ranks' information, i.e. after the unique id
# from rank 0 reached rank 1 through the gloo group and both ranks met.  Expected: both processes exit non-zero within seconds with VP_ERR_RCCL.
cd $GRAFT_REPO_ROOT
export MASTER_ADDR=127.0.0.1
timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29571 bench.py --gpus 2 --share-gpu --config C1 --steps 2 \
  > gpurun_out/rendezvous_check.log 2>&1
echo "exit code $?"
grep -E "VP_ERR_RCCL|Duplicate GPU|ncclCommInit|Traceback|timed out" gpurun_out/rendezvous_check.log | head -8
