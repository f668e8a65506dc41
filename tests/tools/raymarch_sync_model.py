#!/usr/bin/env python3
"""Lane use of k_raymarch's sample loops under three synchronisation granularities of the 64 rays of a wave (CPU model, formula samples):
  slab   what the kernel does: within a light-axis slab the wave marches "the r-th occupied metavoxel of every lane" together and all lanes
         meet again at the slab's end (a lane with one metavoxel in the slab idles while another marches its second);
  mv     a traversal that streams across slabs: round r = the r-th occupied metavoxel of every lane's whole ray (no meeting at slab ends);
  lattice  all lanes step the ray lattice index together, front to back, each sampling where its ray is inside an occupied metavoxel;
  ray    lanes only idle once their whole ray is done (scripts/lane_bound.py's bound; what the 2.2x slower state-machine kernel of round 3 aimed at).
Every round costs the longest lane's samples (two per loop iteration).  Rays, lattice and per-metavoxel sample ranges follow RM.shader:188-240 as
the kernels do; occupancy = the oracle's bins (test infrastructure: this tool lives with the tests).  Early-out is ignored (formula samples).
usage: tests/tools/raymarch_sync_model.py [C3] [wave tiles to sample = 400]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from __graft_entry__ import load_package
load_package()
from vpfx_amd import scene as S
from oracle import oracle as O
name = sys.argv[1] if len(sys.argv) > 1 else "C3"
ntiles = int(sys.argv[2]) if len(sys.argv) > 2 else 400
sc = S.make_scene(name, cubemap="r8")
o = O.Oracle(sc.config())
o.set_frame(sc.light_to_world, sc.grid_center)
o.bin(sc.particles, sc.layout, sc.psys_local_to_world)
occ = o.bin_counts() > 0                                   # [z][y][x]
mvpos = np.asarray(o.mv_positions(), dtype=np.float64)
N = sc.N; s = sc.mv_scale; W, H = sc.width, sc.height
L = np.asarray(sc.light_to_world, dtype=np.float64).reshape(4, 4).T
Rl = L[:3, :3]
c2w = np.asarray(sc.cam_to_world, dtype=np.float64)
w2c = np.linalg.inv(c2w)
A = Rl.T @ c2w[:3, :3] / s                                 # camera -> grid space, linear part
tvec = Rl.T @ (c2w[:3, 3] - mvpos[0, 0, 0]) / s + 0.5
steps = 64
maxdim = max(N)
halfZ = 1.73205 * 0.5 * maxdim * s
csO = (w2c @ np.append(np.asarray(sc.grid_center, dtype=np.float64), 1.0))[:3]
zMin = csO[2] + halfZ
step = 1.73205 / steps                                     # lattice step in metavoxel units along the ray
fov = np.radians(60.0)
rng = np.random.default_rng(3)
tiles = [(int(rng.integers(0, W // 8)), int(rng.integers(0, H // 8))) for _ in range(ntiles)]
tot = dict(useful=0, slab=0, mv=0, ray=0, lattice=0)
for tx, ty in tiles:
    seqs = []
    for ly in range(8):
        for lx in range(8):
            col, row = tx * 8 + lx, ty * 8 + ly
            d = np.array([(2 * (col + 0.5) / W - 1) * (W / H), 2 * (row + 0.5) / H - 1, -1 / np.tan(fov / 2)])
            d /= np.linalg.norm(d)
            start = d * (zMin / d[2])
            og = A @ start + tvec
            dg = A @ d; dg /= np.linalg.norm(dg)
            tcam = int(np.linalg.norm(tvec - og) / step)
            seq = []
            inv = 1.0 / np.where(dg == 0, 1e-30, dg)
            for zz in range(N[2]):
                a, b = (zz - og[2]) * inv[2], (zz + 1 - og[2]) * inv[2]
                ta, tb = min(a, b), max(a, b)
                # cells of the slab the ray crosses, in ray order: sample the interval finely (model accuracy is enough here)
                ts = np.linspace(ta, tb, 9)[:-1] + (tb - ta) / 16
                cells = []
                for t in ts:
                    p = og + t * dg
                    cx, cy = int(np.floor(p[0])), int(np.floor(p[1]))
                    if 0 <= cx < N[0] and 0 <= cy < N[1] and (cx, cy) not in cells: cells.append((cx, cy))
                for cx, cy in cells:
                    if not occ[zz, cy, cx]: continue
                    lo = np.array([cx, cy, zz], dtype=np.float64); hi = lo + 1
                    t0 = (lo - og) * inv; t1 = (hi - og) * inv
                    t_in, t_out = np.max(np.minimum(t0, t1)), np.min(np.maximum(t0, t1))
                    if t_in > t_out: continue
                    te, tx_ = max(int(np.ceil(t_in / step)), tcam), int(np.floor(t_out / step))
                    n = max(0, tx_ - te + 1)
                    if n > 0: seq.append((zz, n, te, tx_))
            seqs.append(seq)
    useful = sum(q[1] for sq in seqs for q in sq)
    it = lambda n: 2 * ((n + 1) // 2)                         # samples are taken two per loop iteration
    slab_cost = 0
    for zz in range(N[2]):
        per = [[q[1] for q in sq if q[0] == zz] for sq in seqs]
        for r in range(max(len(p) for p in per)):
            slab_cost += max(it(p[r]) if r < len(p) else 0 for p in per)
    mv_cost = sum(max(it(sq[r][1]) if r < len(sq) else 0 for sq in seqs) for r in range(max((len(sq) for sq in seqs), default=0)))
    ray_cost = max((sum(q[1] for q in sq) for sq in seqs), default=0)
    ks = set()
    for sq in seqs:
        for q in sq: ks.update(range(q[2] // 2, q[3] // 2 + 1))          # lattice indices, two per loop iteration
    lat_cost = 2 * len(ks)
    tot["useful"] += useful; tot["slab"] += 64 * slab_cost; tot["mv"] += 64 * mv_cost; tot["ray"] += 64 * ray_cost; tot["lattice"] += 64 * lat_cost
print(f"{name}: {ntiles} wave tiles, {tot['useful'] / ntiles / 64:.0f} formula samples per ray")
for k in ("slab", "mv", "lattice", "ray"):
    print(f"  sync per {k:7s}: lane use of the sample loops {tot['useful'] / max(tot[k], 1):.3f}")
