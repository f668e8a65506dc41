"""Step time through the host-buffer entry points (vp_bin with host particles, vp_raymarch to a host image)."""
import sys, time
sys.path.insert(0, '.')
from __graft_entry__ import load_package; load_package()
from vpfx_amd import scene as S, engine as E
sc = S.make_scene(sys.argv[1] if len(sys.argv) > 1 else "C3", cubemap="r8")     # the bench input
e = E.Engine(sc.config()); e.set_frame(sc.light_to_world, sc.grid_center)
fp0, fp = sc.fill_params(), sc.fill_params(); fp.cubemap = None
cam, rp = sc.camera(), sc.raymarch_params()
e.bin(sc.particles, sc.layout, sc.psys_local_to_world); e.fill(fp0); e.raymarch(cam, rp)
n = 50
t0 = time.perf_counter()
for _ in range(n):
    e.bin(sc.particles, sc.layout, sc.psys_local_to_world); e.fill(fp); img = e.raymarch(cam, rp)
dt = (time.perf_counter() - t0) / n
st = e.stats()
print(f"host-buffer step {dt*1e3:.3f} ms  -> {(st['voxels_filled']+st['samples'])/dt/1e6:.0f} M(voxels+samples)/s")
t0 = time.perf_counter()
for _ in range(n):
    e.bin(sc.particles, sc.layout, sc.psys_local_to_world)
tb = (time.perf_counter() - t0) / n
e.fill(fp)
t0 = time.perf_counter()
for _ in range(n):
    img = e.raymarch(cam, rp)
tr = (time.perf_counter() - t0) / n
print(f"vp_bin (H2D {sc.particles.nbytes/1e6:.1f} MB + bin) {tb*1e3:.3f} ms; vp_raymarch (+ D2H {img.nbytes/1e6:.1f} MB) {tr*1e3:.3f} ms")
# the same with the two per-frame host buffers page-locked (vp_pin_host_buffer)
import numpy as np
out = np.empty_like(img)
e.pin(out); e.pin(sc.particles)
e.raymarch(cam, rp, out=out)
t0 = time.perf_counter()
for _ in range(n):
    e.bin(sc.particles, sc.layout, sc.psys_local_to_world); e.fill(fp); e.raymarch(cam, rp, out=out)
dtp = (time.perf_counter() - t0) / n
t0 = time.perf_counter()
for _ in range(n):
    e.raymarch(cam, rp, out=out)
trp = (time.perf_counter() - t0) / n
assert np.array_equal(out, img)
print(f"pinned host buffers: step {dtp*1e3:.3f} ms -> {(st['voxels_filled']+st['samples'])/dtp/1e6:.0f} M(voxels+samples)/s; vp_raymarch (+ D2H) {trp*1e3:.3f} ms")
# pipelined: vp_raymarch_async + vp_wait_image, the image copy of frame n runs beside bin + fill of frame n + 1 (one frame of latency)
outs = [out, np.empty_like(out)]
e.pin(outs[1])
e.bin(sc.particles, sc.layout, sc.psys_local_to_world); e.fill(fp); e.raymarch_async(cam, rp, outs[0])
t0 = time.perf_counter()
for i in range(n):
    e.bin(sc.particles, sc.layout, sc.psys_local_to_world); e.fill(fp)
    e.wait_image()                                        # frame i's image is in outs[i % 2] from here on
    e.raymarch_async(cam, rp, outs[(i + 1) % 2])
e.wait_image()
dta = (time.perf_counter() - t0) / n
assert np.array_equal(outs[0], img) and np.array_equal(outs[1], img)
print(f"pinned + vp_raymarch_async (copy beside the next frame's bin + fill): step {dta*1e3:.3f} ms -> {(st['voxels_filled']+st['samples'])/dta/1e6:.0f} M(voxels+samples)/s")
e.unpin(outs[1])
e.unpin(out); e.unpin(sc.particles)
