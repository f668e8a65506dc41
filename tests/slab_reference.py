"""Test infrastructure (NOT product code): independent Python statements of the fan-out's HOST logic -- slab cut, per-slice cost model,
blend plan -- that tests/test_host_logic.py and tests/test_gpu_slabs.py hold the library's own planner (csrc/host_logic.cpp: vp_plan_slabs,
vp_blend_plan, vp_exchange_plan) against, plus `HipSlabEngine`, a thin adapter that drives one slab context per "rank" through the library's
multi-GPU building blocks (vp_fill_local / vp_fill_finish_gathered / vp_raymarch_partial_device / vp_blend_partials_device).
Until round 3 this lived in the package as parallel.py next to a complete second implementation of the exchange (SlabPipeline over
torch.distributed); that pipeline is gone -- the exchange exists once, in csrc/multi.cpp, and its message schedule is exported
(vp_exchange_plan) so that tests/test_fanout_gloo.py can execute THE LIBRARY'S schedule over gloo on host buffers."""
from __future__ import annotations

from typing import List, Sequence, Tuple

import torch


def slab_bounds(nz: int, world: int, weights: Sequence[float] | None = None) -> List[Tuple[int, int]]:
    """Contiguous zz ranges, one per rank.  With `weights` (work per zz slice, e.g. pair counts) the cut points
    balance the prefix sum; every rank gets at least one slice (requires world <= nz)."""
    if world > nz:
        raise ValueError(f"{world} ranks for {nz} z-slices: at most one rank per slice")
    if weights is None:
        cuts = [round(i * nz / world) for i in range(world + 1)]
    else:
        if len(weights) != nz:
            raise ValueError(f"{len(weights)} weights for {nz} z-slices")
        w = [max(float(x), 0.0) for x in weights]
        total = sum(w)
        if not total > 0.0:
            return slab_bounds(nz, world, None)
        # Optimal contiguous partition: minimise the heaviest slab (the frame waits for the slowest rank), every rank owning >= 1
        # slice.  dp[k][z] = best achievable maximum over the first z slices cut into k slabs; nz <= a few hundred, world <= 8.
        pre = [0.0]
        for x in w:
            pre.append(pre[-1] + x)
        INF = float("inf")
        dp = [[INF] * (nz + 1) for _ in range(world + 1)]
        arg = [[0] * (nz + 1) for _ in range(world + 1)]
        dp[0][0] = 0.0
        for k in range(1, world + 1):
            for z in range(k, nz - (world - k) + 1):
                best, besty = INF, k - 1
                for y in range(k - 1, z):                   # last slab = [y, z)
                    if dp[k - 1][y] == INF:
                        continue
                    # ties broken towards equal thickness (secondary key: slab length), so a flat histogram gives uniform slabs
                    v = max(dp[k - 1][y], pre[z] - pre[y])
                    if v < best or (v == best and abs((z - y) - nz / world) < abs((z - besty) - nz / world)):
                        best, besty = v, y
                dp[k][z], arg[k][z] = best, besty
        cuts = [nz]
        z = nz
        for k in range(world, 0, -1):
            z = arg[k][z]
            cuts.append(z)
        cuts.reverse()
    return [(cuts[i], cuts[i + 1]) for i in range(world)]


def slice_costs(bin_counts, mv_positions, cam_pos, mv_scale: float, height: int, fov_y_rad: float, steps_per_mv: int,
                ms_per_pair: float = 7.3e-6, samples_per_ms: float = 2.0e8):
    """Estimated milliseconds of work per light-axis slice zz, the weights `slab_bounds` cuts balanced slabs from:
      fill       ~ (particle, MV) pairs of the slice                      (k_fill is linear in them: 3.5 ms / 480 k pairs at C3)
      ray-march  ~ lattice samples the slice's occupied MVs receive = screen footprint of each MV (pixels, ~ (focal * s / distance)^2:
                   MVs near the camera cost up to 8x the far ones at the benchmark camera) x samples on the chord through it
                   (steps / sqrt(3) per MV unit, mean chord of a unit cube ~ 2/3 ... 1).  A slab cannot know that rays are already
                   saturated by slabs in front of it, so every occupied MV counts.
    bin_counts [Nz,Ny,Nx] ints, mv_positions [Nz,Ny,Nx,3]; host-side numpy, no device work."""
    import numpy as np
    cnt = np.asarray(bin_counts)
    occ = cnt != 0
    d = np.linalg.norm(np.asarray(mv_positions, dtype=np.float64) - np.asarray(cam_pos, dtype=np.float64), axis=-1)
    focal_px = 0.5 * height / np.tan(0.5 * fov_y_rad)
    area_px = (focal_px * mv_scale / np.maximum(d, 0.5 * mv_scale)) ** 2 * 1.5          # silhouette of a cube seen off-axis ~ 1.5 faces
    samples = area_px * (steps_per_mv / 1.73205) * 0.75
    rm_ms = (samples * occ).sum(axis=(1, 2)) / samples_per_ms
    fill_ms = cnt.sum(axis=(1, 2)).astype(np.float64) * ms_per_pair
    return [float(x) for x in (fill_ms + rm_ms)], [float(x) for x in fill_ms], [float(x) for x in rm_ms]


def choose_slabs(nz: int, world: int, fill_ms: Sequence[float], rm_ms: Sequence[float]) -> List[Tuple[int, int]]:
    """Slab cut for the two-stage pipeline.  The stages are separated by collectives, so a frame costs
    max_r(fill_r) * 1.3 (local + finish pass) + max_r(raymarch_r): the cut that balances the SUM per slice need not minimise
    that.  Candidates = optimal contiguous partitions of fill + alpha * raymarch for a few alpha (0 = fill only ... inf = ray-march
    only); the one with the smallest stage-maxima sum wins (ties: the earliest candidate, i.e. the more fill-balanced)."""
    best, best_t = None, float("inf")
    for alpha in (0.0, 0.25, 0.5, 1.0, 2.0, 4.0, None):
        w = [r if alpha is None else f + alpha * r for f, r in zip(fill_ms, rm_ms)]
        b = slab_bounds(nz, world, w)
        t = 1.3 * max(sum(fill_ms[z0:z1]) for z0, z1 in b) + max(sum(rm_ms[z0:z1]) for z0, z1 in b)
        if t < best_t - 1e-12:
            best, best_t = b, t
    return best


def blend_plan(bounds: Sequence[Tuple[int, int]], z_boundary: int):
    """Which partial images exist and the order they are blended in (slab granularity of VPR.cs:652-711).
    Returns (plan, straddler): plan = list of (rank, which, kind) with which in {"over","under"}, kind 0 = OVER,
    1 = UNDER; straddler = rank owning both phases or None."""
    plan, straddler = [], None
    for r, (z0, z1) in enumerate(bounds):          # phase A: zz ascending, blend OVER
        if z0 <= z_boundary:
            plan.append((r, "over", 0))
    for r, (z0, z1) in enumerate(bounds):          # phase B: zz ascending, blend UNDER
        if z1 - 1 > z_boundary:
            plan.append((r, "under", 1))
        if z0 <= z_boundary < z1 - 1:
            straddler = r
    return plan, straddler


class HipSlabEngine:
    """libvpfx behind the SlabPipeline interface; all images / maps are torch tensors on this rank's GPU."""

    def __init__(self, engine, device):
        self.e, self.dev = engine, device
        self.lm_shape = (engine.N[1] * engine.nv, engine.N[0] * engine.nv)
        self.img_shape = (engine.H, engine.W, 4)
        self._tau = torch.empty(self.lm_shape, dtype=torch.float32, device=device)
        self._over = torch.empty(self.img_shape, dtype=torch.float32, device=device)
        self._under = torch.empty(self.img_shape, dtype=torch.float32, device=device)
        self._out = torch.empty(self.img_shape, dtype=torch.float32, device=device)
        self._piece_out = None

    def bin_resident(self):
        self.e.bin_resident()

    def fill(self, params):
        self.e.fill(params)

    def fill_local(self, params):
        self.e.fill_local(params, self._tau.data_ptr())
        return self._tau

    def fill_finish(self, t_in):
        self.e.fill_finish(None if t_in is None else t_in.contiguous().data_ptr())

    def fill_finish_gathered(self, tau_all, rank, world):
        self.e.fill_finish_gathered(tau_all.data_ptr(), rank, world)

    def z_boundary(self, cam):
        return self.e.z_boundary(cam)

    def raymarch(self, cam, rp):
        self.e.raymarch_device(cam, rp, self._out.data_ptr())
        return self._out

    def raymarch_partial(self, cam, rp):
        self.e.raymarch_partial_device(cam, rp, self._over.data_ptr(), self._under.data_ptr())
        return self._over, self._under

    def blend(self, images, kinds):
        """Ordered blend of equally shaped [pixels, 4] pieces (or whole [H, W, 4] images)."""
        shape = tuple(images[0].shape)
        if self._piece_out is None or tuple(self._piece_out.shape) != shape:
            self._piece_out = torch.empty(shape, dtype=torch.float32, device=self.dev)
        npix = self._piece_out.numel() // 4
        self.e.blend_partials_device([t.data_ptr() for t in images], kinds, self._piece_out.data_ptr(), num_pixels=npix)
        return self._piece_out
