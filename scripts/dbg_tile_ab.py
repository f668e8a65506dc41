"""A/B helper for the LDS-tile accumulator variants: C3 with the f32 cube map (global-table kernel) -- kernel time + a correctness
probe (bricks vs the EXACT-math build of the same library: <= 1 fp16 ulp expected)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from __graft_entry__ import load_package
load_package()
from vpfx_amd import engine as E, scene as S
sc = S.make_scene("C3")
res = {}
for exact in (False, True):
    g = E.Engine(sc.config(), exact=exact)
    g.set_frame(sc.light_to_world, sc.grid_center)
    g.bin(sc.particles, sc.layout, sc.psys_local_to_world)
    ms = []
    for _ in range(12):
        g.fill(sc.fill_params()); g.sync(); ms.append(g.last_kernel_ms(1))
    cnt = g.bin_counts()
    occ = list(zip(*np.nonzero(cnt)))[::401]
    res[exact] = (np.median(ms[2:]), [g.read_brick(x, y, z).view(np.uint16).astype(np.int32) for z, y, x in occ], g.read_lightmap())
    g.close()
worst = max(int(np.abs(a - b).max()) for a, b in zip(res[False][1], res[True][1]))
print(f"fill kernel {res[False][0]:.3f} ms (EXACT build {res[True][0]:.3f}); default vs EXACT bricks: max {worst} fp16 ulp; "
      f"light map rel diff {float(np.abs(res[False][2] - res[True][2]).max()):.2e}")
