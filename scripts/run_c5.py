"""BASELINE config 5 (64^3 MVs x 64^3 voxels, 1M particles, 3840x2160) on ONE GPU: stats, stage times, sanity properties."""
import sys, time
import numpy as np
sys.path.insert(0, '.')
from __graft_entry__ import load_package; load_package()
from vpfx_amd import scene as S, engine as E
sc = S.make_scene("C5", cubemap=(sys.argv[1] if len(sys.argv) > 1 else "r8"))
e = E.Engine(sc.config())
e.set_frame(sc.light_to_world, sc.grid_center)
e.upload_particles(sc.particles, sc.layout, sc.psys_local_to_world)
t0 = time.perf_counter(); e.bin_resident(); e.sync(); t1 = time.perf_counter()
st = e.stats(); print("bin %.2f ms" % ((t1-t0)*1e3), st, flush=True)
fp = sc.fill_params()
e.fill(fp); e.sync()
fp.cubemap = None
t2 = time.perf_counter(); e.fill(fp); e.sync(); t3 = time.perf_counter()
print("fill wall %.2f ms kernel %.2f ms  -> %.1f Gvoxel/s" % ((t3-t2)*1e3, e.last_kernel_ms(1), st['voxels_filled']/e.last_kernel_ms(1)/1e6), flush=True)
img = e.raymarch(sc.camera(), sc.raymarch_params())
t4 = time.perf_counter(); img = e.raymarch(sc.camera(), sc.raymarch_params()); t5 = time.perf_counter()
st = e.stats()
print("raymarch wall %.2f ms kernel %.2f ms samples %d -> %.1f Gsamples/s bricks_sampled %d" % ((t5-t4)*1e3, e.last_kernel_ms(2), st['samples'], st['samples']/e.last_kernel_ms(2)/1e6, st['bricks_sampled']))
lm = e.read_lightmap()
print("lightmap min/mean/max", lm.min(), lm.mean(), lm.max(), "alpha mean", img[...,3].mean(), "finite", np.isfinite(img).all(), "brick GB", st['brick_bytes']/1e9)
