#!/usr/bin/env python3
"""Generate the committed golden fixtures of tests/golden/*.npz.

PROVENANCE (read this): the reference (rajabala/Volumetric-Particles-For-Unity) ships no tests or golden data and
can be neither compiled nor run here (Unity C# + ShaderLab/HLSL), so these vectors are produced by THIS repo's
CPU oracle (oracle/vp_oracle.c) on the deterministic synthetic scenes of vpfx_amd.scene, after the oracle was
cross-checked against the independent float64 twin (oracle/numpy_twin.py; agreement recorded in the .npz).
They pin the oracle (and hence the HIP path) against regressions; they are not reference outputs.

Run from the repo root:  python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from __graft_entry__ import load_package  # noqa: E402

load_package()
from vpfx_amd import scene as S  # noqa: E402
from oracle import oracle as O  # noqa: E402
from oracle.numpy_twin import Twin  # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))


def run(sc):
    o = O.Oracle(sc.config())
    o.set_frame(sc.light_to_world, sc.grid_center)
    o.bin(sc.particles, sc.layout, sc.psys_local_to_world)
    o.fill(sc.fill_params())
    img = o.raymarch(sc.camera(), sc.raymarch_params())
    return o, img


def pick_bricks(counts, k=3):
    zz, yy, xx = np.nonzero(counts)
    order = np.argsort(-counts[zz, yy, xx], kind="stable")
    sel = [order[0], order[len(order) // 2], order[-1]][:k]
    return [(int(xx[i]), int(yy[i]), int(zz[i])) for i in sel]


def main():
    # ---- T0: full outputs + twin agreement ---------------------------------------------------------
    sc = S.make_scene("T0")
    o, img = run(sc)
    counts = o.bin_counts()
    bricks = pick_bricks(counts)
    tw = Twin(sc)
    tw.grid(); tw.bin(); tw.fill()
    timg = tw.raymarch()
    twin_err = float(np.abs(timg - img).max())
    assert twin_err < 1e-4 and tw.samples == o.stats()["samples"], (twin_err, tw.samples)
    lists = {f"list_{x}_{y}_{z}": o.bin_list(x, y, z) for x, y, z in bricks}
    np.savez_compressed(
        os.path.join(OUT, "T0.npz"), mv_positions=o.mv_positions(), bin_counts=counts, brick_ids=np.array(bricks),
        **{f"brick_{i}": o.read_brick(*b).view(np.uint16) for i, b in enumerate(bricks)}, **lists,
        lightmap=o.read_lightmap(), rgba=img, samples=np.int64(o.stats()["samples"]), z_boundary=np.int32(o.z_boundary(sc.camera())),
        twin_rgba_max_err=np.float64(twin_err))
    # ---- T0 with an OVER phase (camera inside the light-z range) ------------------------------------
    sc = S.make_scene("T0")
    sc.set_camera((1.5, 14.0, 1.0))
    o, img = run(sc)
    np.savez_compressed(os.path.join(OUT, "T0_over.npz"), rgba=img, samples=np.int64(o.stats()["samples"]),
                        z_boundary=np.int32(o.z_boundary(sc.camera())))
    # ---- C1 (BASELINE config 1): stats, light map, 3 bricks, 64x64 crop of the image ----------------
    sc = S.make_scene("C1")
    o, img = run(sc)
    counts = o.bin_counts()
    bricks = pick_bricks(counts)
    st = o.stats()
    np.savez_compressed(
        os.path.join(OUT, "C1.npz"), bin_counts=counts, brick_ids=np.array(bricks),
        **{f"brick_{i}": o.read_brick(*b).view(np.uint16) for i, b in enumerate(bricks)},
        lightmap=o.read_lightmap().astype(np.float32), rgba_crop=img[96:160, 96:160].copy(), rgba_mean=img.mean(axis=(0, 1)),
        alpha_covered=np.float64((img[..., 3] > 0).mean()), samples=np.int64(st["samples"]), occupied=np.int64(st["occupied_mv"]),
        pairs=np.int64(st["pairs"]), max_pairs=np.int64(st["max_pairs_per_mv"]))
    # ---- C1 "clean" variant: every particle has e = 1 ---------------------------
    sc = S.make_scene("C1", size_range=(1.1, 1.9))
    o, img = run(sc)
    st = o.stats()
    np.savez_compressed(os.path.join(OUT, "C1_clean.npz"), bin_counts=o.bin_counts(), rgba_crop=img[96:160, 96:160].copy(),
                        samples=np.int64(st["samples"]), occupied=np.int64(st["occupied_mv"]), pairs=np.int64(st["pairs"]))
    for f in sorted(os.listdir(OUT)):
        if f.endswith(".npz"):
            print(f, os.path.getsize(os.path.join(OUT, f)))


if __name__ == "__main__":
    main()
