# round 6: the headline path AND the cliff / any-nv paths on the record with counters -- one rocprofv3 kernel-trace + the PMC
# passes of scripts/gpu_prof_r5.sh per path.  Outputs under gpurun_out/prof_r6_<name>/ (summary.txt, traffic.json, limiters.json, bench_line.json).
#   C3        the headline (8-bit cube map in LDS, grey z-pair bricks)
#   C3_f32    float cube-map texels: k_fill on the global footprint table            (variants.cubemap_f32, 4.1 ms)
#   C3_rgba   RGBA16F bricks (what a coloured ambientColor selects): k_raymarch<.., GREY = false>, four loads per sample (variants.coloured_ambient, 1.5 ms)
#   C3nv24    32^3 x 24^3: the run-time voxel-count instantiations (GEN k_fill_lds, k_raymarch<0>)
cd $GRAFT_REPO_ROOT
for spec in "C3|" "C3_f32|--cubemap f32" "C3_rgba|--no-grey" "C3nv24|--config C3nv24"; do
  name=${spec%%|*}; args=${spec#*|}
  kt="$args --no-variants --steps 100"
  [ "$name" = "C3" ] && kt="$args"
  PROF_DIR=prof_r6_$name BENCH_ARGS="$args" KT_ARGS="$kt" STEPS=10 bash scripts/gpu_prof_r5.sh > gpurun_out/prof_r6_$name.log 2>&1
  echo "== $name"; cat gpurun_out/prof_r6_$name/bench_line.json | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline_all']
print(' ms/step %.3f' % d['ms_per_step'], {k: round(v,3) for k,v in d['stage_ms'].items() if v})
for k in ('fill','raymarch'): print(' ', k, r[k]['kernel'], 'avg_ms %.3f' % r[k]['avg_ms'], 'bytes %.3f GB' % (r[k]['bytes_per_launch']/1e9), 'frac %.4f' % r[k]['frac'])
"
  cat gpurun_out/prof_r6_$name/traffic.json 2>/dev/null | python -c "
import json,sys
try:
    d=json.load(sys.stdin)
    for k,v in d.items():
        if isinstance(v,dict): print('  traffic', k, '%.3f GB' % (v['traffic_bytes']/1e9), '(fetch raw %.3f, write %.3f)' % (v['FETCH_SIZE_bytes_raw']/1e9, v['WRITE_SIZE_bytes']/1e9))
except Exception as e: print('  no traffic json', e)
"
done
