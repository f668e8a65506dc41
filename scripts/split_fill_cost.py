"""Cost of the split (multi-GPU) fill against the fused fill on one GPU, whole grid: k_fill<MODE 0> vs k_fill<MODE 1> + k_fill_finish."""
import sys
import numpy as np
import torch
sys.path.insert(0, '.')
from __graft_entry__ import load_package; load_package()
from vpfx_amd import scene as S, engine as E
sc = S.make_scene(sys.argv[1] if len(sys.argv) > 1 else "C3", cubemap="r8")
g = E.Engine(sc.config())
g.set_frame(sc.light_to_world, sc.grid_center)
g.bin(sc.particles, sc.layout, sc.psys_local_to_world)
tau = torch.empty((sc.N[1] * sc.nv, sc.N[0] * sc.nv), dtype=torch.float32, device="cuda")
fused, local, fin = [], [], []
for it in range(5):
    g.fill(sc.fill_params()); g.sync(); fused.append(g.last_kernel_ms(1))
    g.fill_local(sc.fill_params(), tau.data_ptr()); g.sync(); local.append(g.last_kernel_ms(1))
    g.fill_finish(None); g.sync(); fin.append(g.last_kernel_ms(3))
st = g.stats()
gb = st["voxels_filled"] * 16 / 1e9
print(f"fused {np.mean(fused[1:]):.3f} ms | local {np.mean(local[1:]):.3f} ms + finish {np.mean(fin[1:]):.3f} ms ({gb:.2f} GB moved by finish -> {gb / np.mean(fin[1:]):.2f} TB/s)")
