# like gpu_ab.sh, plus the HBM fetch volume of k_raymarch (rocprofv3 --pmc FETCH_SIZE, its own pass) for every _ab/ library variant
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
PKG=volumetric-particles-for-unity_amd
cp $PKG/libvpfx.so /tmp/libvpfx_main.so
for f in /tmp/libvpfx_main.so _ab/libvpfx_*.so; do
  cp $f $PKG/libvpfx.so
  name=$(basename $f)
  python bench.py --steps ${STEPS:-40} --warmup 2 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$name', 'ms/step', round(d['ms_per_step'],3), {k:round(v,3) for k,v in d['stage_ms'].items() if v is not None})"
  rm -rf /tmp/abf; (cd /tmp && timeout -k 5 200 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d /tmp/abf -o p -- python $GRAFT_REPO_ROOT/bench.py --steps 6 --warmup 2 --no-cpu-baseline > /dev/null 2>&1)
  python scripts/rocprof_summary.py /tmp/abf/p_results.db 2>/dev/null | grep -E "k_raymarch.*FETCH_SIZE|k_fill_lds.*FETCH_SIZE" | cut -c1-100
done
cp /tmp/libvpfx_main.so $PKG/libvpfx.so
