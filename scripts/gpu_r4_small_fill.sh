# small launches of k_fill_lds: units per block (= working waves per workgroup) 16 / 8 / 4 / 2 / 1 at DEMO and C1 (A/B build with the
# VPFX_FILL_CLAIM_LOG2 override: scripts/build_ab.sh claimenv "-DVPFX_AB=1"), bricks fingerprinted.   gpurun -- 'bash scripts/gpu_r4_small_fill.sh'
cd $GRAFT_REPO_ROOT
PKG=volumetric-particles-for-unity_amd
OUT=gpurun_out/small_fill; mkdir -p $OUT
cp $PKG/libvpfx.so /tmp/libvpfx_main.so
cp _ab/libvpfx_claimenv.so $PKG/libvpfx.so
for cfg in DEMO C1; do
for l in 4 3 2 1 0; do
  echo -n "$cfg claim_log2=$l : " | tee -a $OUT/log.txt
  VPFX_FILL_CLAIM_LOG2=$l timeout 600 python bench.py --config $cfg --steps 400 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('ms/step', round(d['ms_per_step'],4), {k:round(v,4) for k,v in d['stage_ms'].items() if v is not None})" | tee -a $OUT/log.txt
done
done
for l in 4 2; do echo -n "fill hash C1 claim_log2=$l : " | tee -a $OUT/log.txt; VPFX_FILL_CLAIM_LOG2=$l python scripts/fill_hash.py C1 2 r8 | tee -a $OUT/log.txt; done
cp /tmp/libvpfx_main.so $PKG/libvpfx.so
echo "product policy:" | tee -a $OUT/log.txt
for cfg in DEMO C1 C2 C3; do
  echo -n "$cfg : " | tee -a $OUT/log.txt
  timeout 600 python bench.py --config $cfg --steps 300 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('ms/step', round(d['ms_per_step'],4), {k:round(v,4) for k,v in d['stage_ms'].items() if v is not None})" | tee -a $OUT/log.txt
done
python scripts/fill_hash.py C1 2 r8 | tee -a $OUT/log.txt
