"""Bit-level fingerprint of a fill (light map + every brick) for a config, default and EXACT math: identical arithmetic must give
identical hashes across library variants (scripts/gpu_variants.sh) and across repeated runs."""
import sys, os, hashlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from __graft_entry__ import load_package
load_package()
from vpfx_amd import engine as E, scene as S
name = sys.argv[1] if len(sys.argv) > 1 else "C3"
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
cube = sys.argv[3] if len(sys.argv) > 3 else "f32"          # r8: the LDS byte-table kernels (default math only)
dscale = float(sys.argv[4]) if len(sys.argv) > 4 else None
sc = S.make_scene(name, cubemap=cube)
if dscale is not None:
    sc.displacement_scale = dscale
out = []
for exact in ((False,) if cube == "r8" else (False, True)):
    g = E.Engine(sc.config(), exact=exact)
    g.set_frame(sc.light_to_world, sc.grid_center)
    g.bin(sc.particles, sc.layout, sc.psys_local_to_world)
    cnt = g.bin_counts()
    occ = list(zip(*np.nonzero(cnt)))
    hs = set()
    for r in range(reps):
        g.fill(sc.fill_params())
        h = hashlib.sha256(g.read_lightmap().tobytes())
        for zz, yy, xx in occ[:: max(1, len(occ) // 400)]:
            h.update(g.read_brick(xx, yy, zz).tobytes())
        hs.add(h.hexdigest()[:16])
    out.append(f"{'exact' if exact else 'fast'} {sorted(hs)} {g.last_kernel_ms(1):.3f}ms")
    g.close()
print(name, cube, " | ".join(out))
