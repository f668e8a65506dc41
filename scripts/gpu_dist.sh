cd $GRAFT_REPO_ROOT
echo "--- torchrun nproc 1"
timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 1 --steps 3 --warmup 1 --config C1 --no-cpu-baseline 2>&1 | tail -2 | cut -c1-400
echo "--- 2 ranks sharing one GPU over gloo (functional)"
timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29512 bench.py --gpus 2 --steps 3 --warmup 1 --config C1 --backend gloo --share-gpu 2>&1 | tail -4 | cut -c1-900
echo "--- 4 ranks"
timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 4 --master-addr 127.0.0.1 --master-port 29513 bench.py --gpus 4 --steps 3 --warmup 1 --config C2 --backend gloo --share-gpu 2>&1 | tail -3 | cut -c1-900
