"""Where a k_fill_lds wave spends its time: reads the in-kernel phase timer of a PROFILING build of the library
(scripts/build_ab.sh prof "-DVPFX_AB=1 -DVPFX_PROBE=9"; the timer brackets the phases with s_memtime and sums wave-cycles over all waves).
usage (GPU box): python scripts/fill_phase_profile.py _ab/libvpfx_prof.so [C3] [r8]"""
import sys, os, ctypes, shutil
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root)
lib_path = sys.argv[1]
pkg = os.path.join(root, "volumetric-particles-for-unity_amd")
def swap_in(src):          # rename, never overwrite: a loaded library's mapping must keep its inode
    shutil.copy(src, os.path.join(pkg, "libvpfx.so.new"))
    os.replace(os.path.join(pkg, "libvpfx.so.new"), os.path.join(pkg, "libvpfx.so"))
shutil.copy(os.path.join(pkg, "libvpfx.so"), "/tmp/libvpfx_keep.so")
swap_in(lib_path)
try:
    from __graft_entry__ import load_package
    load_package()
    from vpfx_amd import abi, engine as E, scene as S
    name = sys.argv[2] if len(sys.argv) > 2 else "C3"
    cube = sys.argv[3] if len(sys.argv) > 3 else "r8"
    if name == "DEMO":
        sc, _, boxes = S.make_demo_scene()
        sc.cubemap = S.make_cubemap_r8() if cube == "r8" else sc.cubemap
    else:
        sc = S.make_scene(name, cubemap=cube)
    lib = E.lib()
    lib.vpfx_probe_read.argtypes = [ctypes.POINTER(ctypes.c_ulonglong), ctypes.c_int]
    e = E.Engine(sc.config())
    e.set_frame(sc.light_to_world, sc.grid_center)
    e.bin(sc.particles, sc.layout, sc.psys_local_to_world)
    for _ in range(3):
        e.fill(sc.fill_params())
    e.sync()
    buf = (ctypes.c_ulonglong * 12)()
    assert lib.vpfx_probe_read(buf, 1) == 0
    reps = 5
    for _ in range(reps):
        e.fill(sc.fill_params())
    e.sync()
    ms = e.last_kernel_ms(1)
    assert lib.vpfx_probe_read(buf, 0) == 0
    v = [x / reps for x in buf]
    names = ["unit header (claim, occupancy, offsets, MV position)", "pre-cull (64 particles per test)", "per-particle set-up (record, slice interval, wave OR)",
             "covered-slice loop", "chain wait", "propagate + store", "publish + loop tail"]
    tot = v[7]
    print(f"{name} {cube}: fill kernel {ms:.3f} ms with the timer in; wave-time by phase (s_memtime ticks, all waves):")
    for n, x in zip(names, v[:7]):
        print(f"  {n:58s} {100 * x / tot:5.1f} %")
    print(f"  {'(sum of phases / wave lifetime)':58s} {100 * sum(v[:7]) / tot:5.1f} %")
    if v[8]:
        print(f"  units with a producer in their column {v[8]:.0f}; {100 * v[9] / v[8]:.1f} % of them found the hand-off word missing at the first look and "
              f"polled again, {v[10] / max(v[9], 1):.1f} more polls each on average")
    e.close()
finally:
    swap_in("/tmp/libvpfx_keep.so")
