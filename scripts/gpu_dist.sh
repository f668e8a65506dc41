# functional multi-rank runs of bench.py on ONE GPU (all ranks share cuda:0, gloo moves the tensors through host memory):
# checks the whole N > 1 code path (slab cut, split fill with the fused tau product, partial ray-march, both exchanges, shard check)
cd $GRAFT_REPO_ROOT
export MASTER_ADDR=127.0.0.1
for n in 2 4 8; do
  for ex in tiles all_gather; do
    echo "== $n ranks, exchange $ex"
    timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node $n --master-addr 127.0.0.1 --master-port $((29500 + n)) bench.py --gpus $n --steps 3 --warmup 1 \
      --backend gloo --share-gpu --exchange $ex --config ${CFG:-C3} 2>&1 | grep -E '^\{"metric"|Error|error|Traceback' | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); c = d['config']
        print('ms/step', round(d['ms_per_step'], 2), 'slabs', c['slabs'], 'max|dRGBA| vs 1-GPU frame', c['max_abs_rgba_diff_vs_1gpu_frame'], 'skipped', c['reference_frame_skipped'])
    else:
        print(l.strip()[:200])
"
  done
done
