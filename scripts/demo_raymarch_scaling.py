"""Is the reference scene's ray-march launch bounded by its longest ray (latency) or by the work of all rays (throughput)?  The same frame at
other resolutions (fewer / more rays of the same length) and other rayMarchSteps (same rays, shorter / longer chains).  GPU only."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from __graft_entry__ import load_package; load_package()
from vpfx_amd import abi, engine as E, scene as S


def run(w, h, steps):
    sc, _, boxes = S.make_demo_scene(width=w, height=h)
    sc.cubemap = S.make_cubemap_r8()
    sc.steps = steps
    img = torch.empty((h, w, 4), device="cuda")
    e = E.Engine(sc.config())
    e.set_occluders(boxes)
    e.set_frame(sc.light_to_world, sc.grid_center)
    e.bin(sc.particles, sc.layout, sc.psys_local_to_world)
    e.fill(sc.fill_params())
    ts = []
    for _ in range(14):
        e.raymarch_device(sc.camera(), sc.raymarch_params(), img.data_ptr()); e.sync()
        ts.append(e.last_kernel_ms(2))
    st = e.stats()
    print(f"{w:5d} x {h:4d}, {steps:3d} steps/MV: ray-march stage {np.median(ts[3:]) * 1e3:7.1f} us, {st['samples'] / 1e6:7.2f} M samples, "
          f"{w * h // 64:6d} waves", flush=True)
    e.close()


def run_empty(w, h):
    sc, _, boxes = S.make_demo_scene(width=w, height=h)
    sc.cubemap = S.make_cubemap_r8()
    img = torch.empty((h, w, 4), device="cuda")
    e = E.Engine(sc.config())
    e.set_occluders(boxes)
    e.set_frame(sc.light_to_world, sc.grid_center)
    e.bin(sc.particles[:0].copy(), sc.layout, sc.psys_local_to_world)
    e.fill(sc.fill_params())
    ts = []
    for _ in range(14):
        e.raymarch_device(sc.camera(), sc.raymarch_params(), img.data_ptr()); e.sync()
        ts.append(e.last_kernel_ms(2))
    print(f"{w:5d} x {h:4d}, EMPTY grid: ray-march stage {np.median(ts[3:]) * 1e3:7.1f} us", flush=True)
    e.close()


if len(sys.argv) > 1 and sys.argv[1] == "floor":
    run_empty(1024, 768)
    run_empty(256, 192)
    for steps in (1, 2, 4, 8, 16, 64):
        run(1024, 768, steps)
    for steps in (1, 8, 64):
        run(256, 192, steps)
    sys.exit(0)
for w, h in ((256, 192), (512, 384), (1024, 768), (2048, 1536)):
    run(w, h, 64)
for steps in (16, 32, 128):
    run(1024, 768, steps)
