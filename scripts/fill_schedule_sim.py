"""CPU-only: list-scheduling simulation of the fill's work units on the resident waves (no GPU needed).

Cost model of one 8x8-column tile under one particle = slices covered at the tile (chord of the particle's sphere) + a constant per
(tile, particle); the units are handed to 4 096 waves (256 CUs x 16) in launch order, each wave taking the next unit when it is free.
Compares the unit shapes the fill has had: whole column walks ordered by metavoxel-column weight (rounds 1 / early 2), the same sorted by
their true cost, and per-metavoxel units (the chained fill).  Efficiency = mean wave busy time / makespan.
usage: fill_schedule_sim.py [config]"""
import heapq
import importlib
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
scene = importlib.import_module("volumetric-particles-for-unity_amd.scene")


def simulate(costs, waves=4096):
    h = [0.0] * waves
    heapq.heapify(h)
    for c in costs:
        heapq.heappush(h, heapq.heappop(h) + c)
    return (sum(h) / waves) / max(h)


def main():
    sc = scene.make_scene(sys.argv[1] if len(sys.argv) > 1 else "C3", cubemap="r8")
    N, nv = sc.N[0], sc.nv
    L = np.asarray(sc.light_to_world, dtype=np.float64).reshape(4, 4).T
    vs = sc.mv_scale / nv
    p = (sc.particles["position"].astype(np.float64) @ L[:3, :3]) / vs + 0.5 * N * nv      # voxel coordinates in the light-aligned grid
    r = 0.5 * sc.particles["size"].astype(np.float64) / vs
    T = N * nv // 8
    cost = np.zeros((T, T))
    for (x, y, _), rr in zip(p, r):
        tx0, tx1 = int(max(0, (x - rr) // 8)), int(min(T - 1, (x + rr) // 8))
        ty0, ty1 = int(max(0, (y - rr) // 8)), int(min(T - 1, (y + rr) // 8))
        for tx in range(tx0, tx1 + 1):
            for ty in range(ty0, ty1 + 1):
                cx, cy = min(max(x, tx * 8), tx * 8 + 8), min(max(y, ty * 8), ty * 8 + 8)
                d2 = (cx - x) ** 2 + (cy - y) ** 2
                if d2 < rr * rr:
                    cost[ty, tx] += 2 * np.sqrt(rr * rr - d2) + 6
    tpm = nv // 8
    colw = cost.reshape(N, tpm, N, tpm).sum(axis=(1, 3))
    walks = []
    for c in np.argsort(-colw.flatten()):
        cy, cx = divmod(int(c), N)
        walks += [cost[cy * tpm + t // tpm, cx * tpm + t % tpm] for t in range(tpm * tpm)]
    print(f"{len(walks)} column walks, {int((np.array(walks) > 0).sum())} with work, heaviest {max(walks) / np.mean(walks):.1f} x the mean")
    print(f"column walks, heaviest metavoxel column first : {simulate(walks):.3f}")
    print(f"column walks, sorted by their own cost        : {simulate(sorted(walks, reverse=True)):.3f}")
    per_mv = [w / N for w in walks for _ in range(N)]
    print(f"per-metavoxel units (chained fill)            : {simulate(per_mv):.3f}")


if __name__ == "__main__":
    main()
