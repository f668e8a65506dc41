"""Ray-march time at C3 for a handful of camera positions (all looking at the grid centre) -- guards against view-dependent cliffs."""
import sys
import numpy as np
import torch
sys.path.insert(0, '.')
from __graft_entry__ import load_package; load_package()
from vpfx_amd import scene as S, engine as E
sc = S.make_scene(sys.argv[1] if len(sys.argv) > 1 else "C3")
g = E.Engine(sc.config())
g.set_frame(sc.light_to_world, sc.grid_center)
g.bin(sc.particles, sc.layout, sc.psys_local_to_world)
g.fill(sc.fill_params())
out = torch.empty((sc.height, sc.width, 4), device="cuda")
D = 0.8 * sc.N[0] * sc.mv_scale
for name, pos in [("default", None), ("top (light side)", (0.04 * D, 1.0 * D, 0.03 * D)), ("below", (0.05 * D, -1.0 * D, 0.02 * D)),
                  ("behind light axis", (0.1 * D, 0.05 * D, D)), ("side +x", (D, 0.1 * D, 0.05 * D)), ("near the cloud edge", (0.0, 0.0, -0.45 * D)),
                  ("inside the cloud", (0.02 * D, 0.01 * D, -0.1 * D)), ("far", (0.0, 0.3 * D, -3.0 * D))]:
    if pos is not None:
        sc.set_camera(pos)
    cam, rp = sc.camera(), sc.raymarch_params()
    ms = []
    for it in range(4):
        g.raymarch_device(cam, rp, out.data_ptr()); g.sync(); ms.append(g.last_kernel_ms(2))
    st = g.stats()
    print(f"{name:22s} zBoundary {g.z_boundary(cam):3d}  {np.mean(ms[1:]):7.3f} ms  {st['samples'] / 1e6:8.1f} M samples  {st['samples'] / np.mean(ms[1:]) / 1e6:6.1f} Gsamples/s  alpha mean {float(out[..., 3].mean()):.3f}")
