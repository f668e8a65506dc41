"""Where the reference scene's ray-march launch (bench.py --config DEMO: 1024 x 768, the emitter's plume) spends its 0.1 ms: the launch with
the plume, with an empty grid (what the waves cost that find nothing), and the samples-per-ray view (VP_RM_SHOW_RAY_SAMPLES): how long the
longest ray of every 8 x 8 pixel wave is.  GPU only."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from __graft_entry__ import load_package; load_package()
from vpfx_amd import abi, engine as E, scene as S

cfg = sys.argv[1] if len(sys.argv) > 1 else "DEMO"
if cfg == "DEMO":
    sc, _, boxes = S.make_demo_scene()
    sc.cubemap = S.make_cubemap_r8()
else:
    sc, boxes = S.make_scene(cfg, cubemap="r8"), None
img = torch.empty((sc.height, sc.width, 4), device="cuda")


def run(parts, label, flags=0):
    e = E.Engine(sc.config())
    if boxes:
        e.set_occluders(boxes)
    e.set_frame(sc.light_to_world, sc.grid_center)
    e.bin(parts, sc.layout, sc.psys_local_to_world)
    e.fill(sc.fill_params())
    rp = sc.raymarch_params(); rp.flags = flags
    ts = []
    for _ in range(12):
        e.raymarch_device(sc.camera(), rp, img.data_ptr()); e.sync()
        ts.append(e.last_kernel_ms(2))
    st = e.stats()
    print(f"{label:34s}: ray-march stage {np.median(ts[2:]) * 1e3:7.1f} us, {st['samples'] / 1e6:7.2f} M samples, occupied {st['occupied_mv']}", flush=True)
    out = img.cpu().numpy().copy()
    e.close()
    return out


run(sc.particles, f"{cfg} as benchmarked")
run(sc.particles[:0].copy(), "empty grid (no particles)")
spr = run(sc.particles, "samples-per-ray view (flag kernel)", abi.VP_RM_SHOW_RAY_SAMPLES)[..., 0]
H, W = spr.shape
t = spr[: H // 8 * 8, : W // 8 * 8].reshape(H // 8, 8, W // 8, 8)
wmax, wsum = t.max(axis=(1, 3)), t.sum(axis=(1, 3))
nz = wmax > 0
print(f"waves: {wmax.size}, with samples: {int(nz.sum())}; longest ray per wave: max {wmax.max():.0f}, mean over working waves {wmax[nz].mean():.0f}, "
      f"p50 {np.percentile(wmax[nz], 50):.0f} p90 {np.percentile(wmax[nz], 90):.0f} p99 {np.percentile(wmax[nz], 99):.0f}")
print(f"lane use in working waves: {wsum[nz].sum() / (64 * wmax[nz].sum()):.2f} (sum of samples / 64 x longest ray)")
print(f"sum over waves of the longest ray: {wmax.sum():.0f} wave-samples; at 1024 SIMDs x 4 waves: {wmax.sum() / 4096:.0f} wave-samples per slot")
