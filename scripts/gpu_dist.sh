# functional N-slab runs of bench.py on ONE GPU: `python bench.py --gpus N --share-gpu` = the library's fan-out (worker threads, slab cut,
# split fill with the all-gathered tau product, partial ray-march, both image exchanges, shard check against the 1-GPU frame) through its
# peer-copy test hook.  A bare `python bench.py --gpus N` (no --share-gpu) is the real thing and needs N GPUs.
cd $GRAFT_REPO_ROOT
for n in 2 4 8; do
  for ex in tiles all_gather; do
    echo "== $n slabs, exchange $ex"
    timeout 600 python bench.py --gpus $n --share-gpu --steps 3 --warmup 1 --exchange $ex --config ${CFG:-C3} 2>&1 | grep -E '^\{"metric"|Error|error|Traceback' | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); c = d['config']
        print('ms/step', round(d['ms_per_step'], 2), 'slabs', c['slabs'], 'max|dRGBA| vs 1-GPU frame', c['max_abs_rgba_diff_vs_1gpu_frame'], 'launch', c['launch'][:40])
    else:
        print(l.strip()[:200])
"
  done
done
