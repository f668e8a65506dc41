"""Stress of the chained fill's light hand-off (fill.hip, FillChain): the light map is the end of every column's chain, so one stale or torn
hand-off anywhere changes it.  Fills the config `reps` times and prints the set of distinct (light map, sampled bricks) fingerprints -- it must
have ONE element (and had the same one for the last build whose fill walked whole columns with the light in a register).
usage: chain_stress.py [config] [reps]"""
import sys, os, hashlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from __graft_entry__ import load_package
load_package()
from vpfx_amd import engine as E, scene as S
name = sys.argv[1] if len(sys.argv) > 1 else "C3"
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 30
sc = S.make_scene(name, cubemap="r8")
g = E.Engine(sc.config())
g.set_frame(sc.light_to_world, sc.grid_center)
g.bin(sc.particles, sc.layout, sc.psys_local_to_world)
occ = list(zip(*np.nonzero(g.bin_counts())))
hs = {}
for r in range(reps):
    g.fill(sc.fill_params())
    h = hashlib.sha256(g.read_lightmap().tobytes())
    if r % 10 == 0:
        for zz, yy, xx in occ[:: max(1, len(occ) // 100)]:
            h.update(g.read_brick(xx, yy, zz).tobytes())
        key = "L+B " + h.hexdigest()[:16]
    else:
        key = "L   " + h.hexdigest()[:16]
    hs[key] = hs.get(key, 0) + 1
print(name, "fill", f"{g.last_kernel_ms(1):.3f} ms", hs)
