"""How much could a wave-coherent ray-march traversal gain?  Per-ray executed sample counts of the benchmark frame (VP_RM_SHOW_RAY_SAMPLES)
-> for every 8x8-pixel wave tile: sum of its rays' samples / (64 x the longest ray) = the lane utilisation a traversal would reach
if lanes never idled while ANY sample of their own ray was left (only rays that finish early idle).  The measured utilisation of the current
kernel (rays idle whenever another ray of the wave has more samples in the SAME metavoxel) is 59 % (PMC).
usage: lane_bound.py [C3]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from __graft_entry__ import load_package
load_package()
from vpfx_amd import abi, engine as E, scene as S
sc = S.make_scene(sys.argv[1] if len(sys.argv) > 1 else "C3", cubemap="r8")
e = E.Engine(sc.config())
e.set_frame(sc.light_to_world, sc.grid_center); e.bin(sc.particles, sc.layout, sc.psys_local_to_world); e.fill(sc.fill_params())
rp = sc.raymarch_params(); rp.flags = abi.VP_RM_SHOW_RAY_SAMPLES
n = e.raymarch(sc.camera(), rp)[..., 0].astype(np.int64)
H, W = n.shape
assert n.sum() == e.stats()["samples"]
t = n[: H // 8 * 8, : W // 8 * 8].reshape(H // 8, 8, W // 8, 8)
tot, mx = t.sum(axis=(1, 3)), t.max(axis=(1, 3))
print(f"samples {n.sum() / 1e6:.1f} M; rays with samples {int((n > 0).sum())}; per-wave bound sum / (64 max): {tot.sum() / (64 * mx.sum()):.3f}; "
      f"mean ray {n[n > 0].mean():.0f}, longest {n.max()}")
