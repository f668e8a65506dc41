# round 6: the driver's scaling command on ONE GPU through the multi-process RCCL stand-in (functional records, no timing meaning): C3 at N = 2/4/8, C5e at N = 8
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r6_mp; mkdir -p $OUT
export VPFX_BENCH_SHARE_GPU=1 VPFX_RCCL_LIBRARY=$GRAFT_REPO_ROOT/tests/tools/_build/libfake_rccl_mp.so HSA_ENABLE_IPC_MODE_LEGACY=0
for n in 2 4 8; do
  python -m torch.distributed.run --nnodes=1 --nproc-per-node $n --master-addr 127.0.0.1 --master-port $((29600 + n)) bench.py --gpus $n --steps 20 --warmup 5 > $OUT/bench_C3_r8_${n}ranks.json 2> $OUT/bench_C3_r8_${n}ranks.err
  echo "C3 N=$n rc=$?"
done
python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 --master-port 29650 bench.py --gpus 8 --steps 5 --warmup 3 --config C5e > $OUT/bench_C5e_r8_8ranks.json 2> $OUT/bench_C5e_r8_8ranks.err
echo "C5e N=8 rc=$?"
for f in $OUT/*.json; do python -c "
import json
d=json.loads(open('$f').read().strip().splitlines()[-1]); c=d['config']
print('$f'.split('/')[-1], 'ranks', c['rccl_ranks'], 'slabs', c['slabs'], 'err_vs_1gpu', c['max_abs_rgba_diff_vs_1gpu_frame'], 'ms/step (shared GPU: no meaning)', round(d['ms_per_step'],2))"; done
rm -f $OUT/*.err
