"""Where a k_raymarch wave spends its time: reads the in-kernel phase timer of a PROFILING build of the library
(scripts/build_ab.sh rmprof "-DVPFX_RM_PROBE=9"; s_memtime brackets, wave-cycles summed over all waves).
usage (GPU box): python scripts/raymarch_phase_profile.py _ab/libvpfx_rmprof.so [C3] [r8]"""
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
    import torch
    from __graft_entry__ import load_package
    load_package()
    from vpfx_amd import engine as E, scene as S
    name = sys.argv[2] if len(sys.argv) > 2 else "C3"
    cube = sys.argv[3] if len(sys.argv) > 3 else "r8"
    boxes = None
    if name == "DEMO":
        sc, _, boxes = S.make_demo_scene()
        if cube == "r8":
            sc.cubemap = S.make_cubemap_r8()
    else:
        sc = S.make_scene(name, cubemap=cube)
    lib = E.lib()
    lib.vpfx_rm_probe_read.argtypes = [ctypes.POINTER(ctypes.c_ulonglong), ctypes.c_int]
    e = E.Engine(sc.config())
    if boxes:
        e.set_occluders(boxes)
    e.set_frame(sc.light_to_world, sc.grid_center)
    e.bin(sc.particles, sc.layout, sc.psys_local_to_world)
    e.fill(sc.fill_params())
    cam, rp = sc.camera(), sc.raymarch_params()
    img = torch.empty((sc.height, sc.width, 4), device="cuda")
    for _ in range(3):
        e.raymarch_device(cam, rp, img.data_ptr())
    e.sync()
    buf = (ctypes.c_ulonglong * 8)()
    assert lib.vpfx_rm_probe_read(buf, 1) == 0
    reps = 5
    for _ in range(reps):
        e.raymarch_device(cam, rp, img.data_ptr())
    e.sync()
    ms = e.last_kernel_ms(2)
    assert lib.vpfx_rm_probe_read(buf, 0) == 0
    v = [x / reps for x in buf]
    names = ["ray set-up", "cell walk (next occupied metavoxel along the ray)", "per-metavoxel set-up (box test, lattice range)", "sample loops",
             "inter-metavoxel blend + bookkeeping", "image / hand-off stores"]
    tot = v[7]
    st = e.stats()
    print(f"{name} {cube}: k_raymarch {ms:.3f} ms with the timer in, {st['samples'] / 1e6:.0f} M samples; wave-time by phase (s_memtime ticks, all waves):")
    for n, x in zip(names, v[:6]):
        print(f"  {n:58s} {100 * x / tot:5.1f} %")
    print(f"  {'(sum of phases / wave lifetime)':58s} {100 * sum(v[:6]) / tot:5.1f} %")
    e.close()
finally:
    swap_in("/tmp/libvpfx_keep.so")
