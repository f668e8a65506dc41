import sys, numpy as np
sys.path.insert(0, '.')
from __graft_entry__ import load_package; load_package()
from vpfx_amd import scene as S, engine as E
name = sys.argv[1] if len(sys.argv) > 1 else "C1"
sc = S.make_scene(name)
g = E.Engine(sc.config(), early_out=False)
g.set_frame(sc.light_to_world, sc.grid_center); g.bin(sc.particles, sc.layout, sc.psys_local_to_world); g.fill(sc.fill_params())
img = g.raymarch(sc.camera(), sc.raymarch_params())
np.save(f"gpurun_out/rm_{name}.npy", img)
print(g.stats())
