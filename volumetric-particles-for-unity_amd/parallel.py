"""One process per GPU: light-axis (zz) slabs of the metavoxel grid sharded over torch.distributed ranks.

The reference is single-GPU; this is the MI355X-node extension SURVEY.md section 8(e) describes.  The path has
exactly two cross-slab dependencies, and each becomes ONE collective (RCCL over xGMI when the backend is
"nccl"; gloo in the CPU tests):

  fill      per-column transmitted light (Fill.shader:224,250).  The update is linear in the incoming light,
            so every rank fills its slab with T_in = 1, publishing the slab's transmittance map tau
            (all_gather, (Ny*nv)*(Nx*nv)*4 B per rank), then finishes with T_in = prod_{slabs nearer the light} tau.
  raymarch  inter-metavoxel blending (VPR.cs:652-711).  The draw order is zz-major in both phases, so a slab's
            MVs are contiguous in it: each rank composites its slab into a premultiplied partial image
            (all_gather, W*H*16 B per rank) and every rank applies the ordered OVER/UNDER blend of the partials.
            Only the one slab that straddles zBoundary has two partials; its second image is broadcast.

The compute engine is injected: `HipSlabEngine` (libvpfx, device tensors) in production; the tests drive the same
pipeline with a CPU engine to check the sharding math without a GPU.
"""
from __future__ import annotations

from typing import List, Sequence, Tuple

import torch
import torch.distributed as dist


def slab_bounds(nz: int, world: int, weights: Sequence[float] | None = None) -> List[Tuple[int, int]]:
    """Contiguous zz ranges, one per rank.  With `weights` (work per zz slice, e.g. pair counts) the cut points
    balance the prefix sum; every rank gets at least one slice (requires world <= nz)."""
    if world > nz:
        raise ValueError(f"{world} ranks for {nz} z-slices: at most one rank per slice")
    if weights is None:
        cuts = [round(i * nz / world) for i in range(world + 1)]
    else:
        w = [max(float(x), 0.0) + 1e-9 for x in weights]
        total = sum(w)
        cuts, acc, nxt = [0], 0.0, 1
        for z in range(nz):
            acc += w[z]
            while nxt < world and acc >= total * nxt / world and z + 1 > cuts[-1]:
                cuts.append(z + 1)
                nxt += 1
        while len(cuts) < world:
            cuts.append(cuts[-1] + 1)
        cuts.append(nz)
        # repair: strictly increasing, last = nz
        for i in range(1, world + 1):
            cuts[i] = max(cuts[i], cuts[i - 1] + 1)
        for i in range(world, 0, -1):
            cuts[i - 1] = min(cuts[i - 1], cuts[i] - 1)
        cuts[0], cuts[world] = 0, nz
    return [(cuts[i], cuts[i + 1]) for i in range(world)]


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


class SlabPipeline:
    """bin -> fill -> raymarch of one frame across the ranks of `group`.

    `engine` must provide: bin_resident(); fill_local(params) -> tau tensor; fill_finish(T_in tensor | None);
    raymarch_partial(cam, rp) -> (over, under) tensors; blend(images, kinds) -> tensor; z_boundary(cam).
    """

    def __init__(self, engine, bounds, rank: int, world: int, group=None):
        self.eng, self.bounds, self.rank, self.world, self.group = engine, list(bounds), rank, world, group
        self._tau_all = None       # [world, LH, LW] receive buffer of the transmittance all-gather
        self._img_all = None       # [world, H, W, 4] receive buffer of the partial-image all-gather
        self._second = None
        self._into_tensor = True   # all_gather_into_tensor (one contiguous receive buffer, no per-rank copies) if supported

    def _all_gather(self, buf, src):
        """all-gather `src` into the preallocated [world, ...] buffer `buf`; returns the per-rank views."""
        if self._into_tensor:
            try:
                dist.all_gather_into_tensor(buf, src, group=self.group)
                return list(buf.unbind(0))
            except (RuntimeError, NotImplementedError):
                self._into_tensor = False          # backend without the fused form (older gloo): list form below
        parts = list(buf.unbind(0))
        dist.all_gather(parts, src, group=self.group)
        return parts

    def fill(self, fill_params):
        self.eng.bin_resident()
        if self.world == 1:
            self.eng.fill(fill_params)
            return
        tau = self.eng.fill_local(fill_params)
        if self._tau_all is None:
            self._tau_all = torch.empty((self.world,) + tuple(tau.shape), dtype=tau.dtype, device=tau.device)
        taus = self._all_gather(self._tau_all, tau)
        t_in = None
        for r in range(self.rank):                 # product in slab order, like the sequential light map
            t_in = taus[r].clone() if t_in is None else t_in.mul_(taus[r])
        self.eng.fill_finish(t_in)

    def render(self, cam, rp):
        if self.world == 1:
            return self.eng.raymarch(cam, rp)
        zb = self.eng.z_boundary(cam)
        plan, straddler = blend_plan(self.bounds, zb)
        over, under = self.eng.raymarch_partial(cam, rp)
        z0, z1 = self.bounds[self.rank]
        primary = over if z0 <= zb else under      # the straddler's primary is its OVER image
        if self._img_all is None:
            self._img_all = torch.empty((self.world,) + tuple(primary.shape), dtype=primary.dtype, device=primary.device)
        prim = self._all_gather(self._img_all, primary)
        second = None
        if straddler is not None:
            if self.rank == straddler:
                second = under
            else:
                if self._second is None:
                    self._second = torch.empty_like(under)
                second = self._second
            dist.broadcast(second, src=straddler, group=self.group)
        images, kinds = [], []
        for r, which, kind in plan:
            images.append(second if (r == straddler and which == "under") else prim[r])
            kinds.append(kind)
        return self.eng.blend(images, kinds)


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

    def bin_resident(self):
        self.e.bin_resident()

    def fill(self, params):
        self.e.fill(params)

    def fill_local(self, params):
        self.e.fill_local(params, self._tau.data_ptr())
        return self._tau

    def fill_finish(self, t_in):
        self.e.fill_finish(None if t_in is None else t_in.contiguous().data_ptr())

    def z_boundary(self, cam):
        return self.e.z_boundary(cam)

    def raymarch(self, cam, rp):
        self.e.raymarch_device(cam, rp, self._out.data_ptr())
        return self._out

    def raymarch_partial(self, cam, rp):
        self.e.raymarch_partial_device(cam, rp, self._over.data_ptr(), self._under.data_ptr())
        return self._over, self._under

    def blend(self, images, kinds):
        self.e.blend_partials_device([t.data_ptr() for t in images], kinds, self._out.data_ptr())
        return self._out
