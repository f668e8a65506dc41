"""Experiment: k_fill time at C3 as a function of the displacement cubemap's size (same particles, same grid).  The footprint
table is 6 (S+1)^2 x 16 B: S = 8 is L1-resident (8 KB), S = 32 is 100 KB, S = 128 (the workload) is 1.6 MB (L2-resident).
The arithmetic per covered voxel is identical, so the difference is the cost of the per-voxel table gather's cache misses."""
import sys
import numpy as np
sys.path.insert(0, '.')
from __graft_entry__ import load_package; load_package()
from vpfx_amd import scene as S, engine as E
cfg = sys.argv[1] if len(sys.argv) > 1 else "C3"
sc = S.make_scene(cfg)
for size in (128, 64, 32, 16, 8):
    sc.cubemap = S.make_cubemap(size)
    g = E.Engine(sc.config())
    g.set_frame(sc.light_to_world, sc.grid_center)
    g.bin(sc.particles, sc.layout, sc.psys_local_to_world)
    ms = []
    for it in range(6):
        g.fill(sc.fill_params()); g.sync()
        ms.append(g.last_kernel_ms(1))
    print(f"{cfg} cubemap {size:4d}^2  table {6 * (size + 1) ** 2 * 16 / 1024:8.1f} KB   k_fill {np.mean(ms[2:]):.3f} ms")
    g.close()
