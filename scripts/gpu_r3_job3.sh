cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3
timeout 600 python -m pytest tests/test_gpu_multi.py -x -q 2>&1 | tail -5
echo "== bench N=1"; timeout 600 python bench.py --steps 50 2> gpurun_out/r3/job3_b1.err | tail -1 > gpurun_out/r3/job3_bench1.json; python -c "
import json; d=json.load(open('gpurun_out/r3/job3_bench1.json')); print(d['ms_per_step'], d['stage_ms'], d['value'], d['roofline']['frac'], d['cpu_baseline']['value'], d['cpu_baseline']['single_thread']['parallel_speedup_of_the_port'])"
tail -3 gpurun_out/r3/job3_b1.err
echo "== bench N=2 share"; timeout 600 python bench.py --gpus 2 --share-gpu --steps 10 2> gpurun_out/r3/job3_b2.err | tail -1 > gpurun_out/r3/job3_bench2.json; echo rc=$?; python -c "
import json; d=json.load(open('gpurun_out/r3/job3_bench2.json')); print(d['ms_per_step'], d['stage_ms_slowest_rank'], d['config']['launch'], d['config']['slabs'], d['config']['max_abs_rgba_diff_vs_1gpu_frame'], d['exchange_ms'], d['per_rank'])"
tail -3 gpurun_out/r3/job3_b2.err
echo "== bench N=8 share"; timeout 600 python bench.py --gpus 8 --share-gpu --steps 10 2> gpurun_out/r3/job3_b8.err | tail -1 > gpurun_out/r3/job3_bench8.json; echo rc=$?; python -c "
import json; d=json.load(open('gpurun_out/r3/job3_bench8.json')); print(d['ms_per_step'], d['config']['slabs'], d['config']['max_abs_rgba_diff_vs_1gpu_frame'], d['config']['samples_executed_all_ranks'], d['per_rank'])"
echo "== scaling model"; timeout 1500 python scripts/scaling_model.py C3 r8 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r3/scaling_model_C3_r8.txt
