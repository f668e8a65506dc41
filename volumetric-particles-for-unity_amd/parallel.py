"""One process per GPU: light-axis (zz) slabs of the metavoxel grid sharded over torch.distributed ranks.

The reference is single-GPU; this is the MI355X-node extension SURVEY.md section 8(e) describes.  The path has
exactly two cross-slab dependencies, and each becomes ONE collective (RCCL over xGMI when the backend is
"nccl"; gloo in the CPU tests):

  fill      per-column transmitted light (Fill.shader:224,250).  The update is linear in the incoming light,
            so every rank fills its slab with T_in = 1, publishing the slab's transmittance map tau
            (all_gather, (Ny*nv)*(Nx*nv)*4 B per rank), then finishes with T_in = prod_{slabs nearer the light} tau.
  raymarch  inter-metavoxel blending (VPR.cs:652-711).  The draw order is zz-major in both phases, so a slab's
            MVs are contiguous in it: each rank composites its slab into a premultiplied partial image.  The
            ordered OVER/UNDER blend of the partials is per pixel, so it is sharded too: ONE all-to-all hands
            rank r the r-th piece (1/N of the pixels) of every partial image, rank r blends its piece, and the
            finished pieces are gathered on the display rank (rank 0) -- (N-1)/N * W*H*16 B sent and received
            per rank instead of (N-1) * W*H*16 B for an all-gather of whole images (xGMI is point to point:
            the all-to-all uses all seven links of a GPU at once).  Only the one slab that straddles zBoundary
            has two partials; the pieces of its second image are scattered.

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
    only); the one with the smallest stage-maxima sum wins (ties: the earliest candidate, i.e. the more fill-balanced).
    (This pipeline runs the split fill on every rank, so it keeps this model; the library's fan-out -- csrc/multi.cpp, hl_plan_slabs --
    runs the fused fill on rank 0 and minimises  max_r F_r + max_r (0.42 F_r [r > 0] + R_r)  exactly.)"""
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


class SlabPipeline:
    """bin -> fill -> raymarch of one frame across the ranks of `group`.

    `engine` must provide: bin_resident(); fill_local(params) -> tau tensor; fill_finish(T_in tensor | None);
    raymarch_partial(cam, rp) -> (over, under) [H, W, 4] tensors; blend(images, kinds) -> tensor, for equally shaped
    [pixels, 4] pieces; z_boundary(cam).
    """

    def __init__(self, engine, bounds, rank: int, world: int, group=None, exchange: str = "tiles"):
        """exchange = "tiles": all-to-all of image pieces, sharded final blend, gather (default, least xGMI traffic);
        "all_gather": ONE all-gather of whole partial images, every rank blends the full image (SURVEY.md 8(e) first form)."""
        if exchange not in ("tiles", "all_gather"):
            raise ValueError(f"exchange = {exchange!r}")
        self.eng, self.bounds, self.rank, self.world, self.group = engine, list(bounds), rank, world, group
        self.exchange = exchange
        self._whole_all = None     # [world, H, W, 4] receive buffer of the whole-image all-gather
        self._tau_all = None       # [world, LH, LW] receive buffer of the transmittance all-gather
        self._img_all = None       # [world, H, W, 4] receive buffer of the partial-image all-gather
        self._second = None
        self._final = None
        self._pad = {}
        self._tiles_ok = False
        self._stage = None         # gloo + device tensors: stage the image exchange through host memory
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
        if hasattr(self.eng, "fill_finish_gathered") and self._tau_all.is_contiguous():
            # the product of the maps nearer the light is formed inside the finish kernel, straight from the receive buffer
            self.eng.fill_finish_gathered(self._tau_all, self.rank, self.world)
            return
        t_in = None
        for r in range(self.rank):                 # product in slab order, like the sequential light map
            t_in = taus[r].clone() if t_in is None else t_in.mul_(taus[r])
        self.eng.fill_finish(t_in)

    # -- collectives of the image exchange.  RCCL ("nccl") takes device tensors directly; under gloo (functional tests, several
    #    ranks sharing one GPU) device tensors are staged through host memory, since gloo only moves host buffers for these ops.
    def _staged(self, t):
        if self._stage is None:
            self._stage = dist.get_backend(self.group) == "gloo"
        return self._stage and t.is_cuda

    def _all_to_all(self, out, inp):
        if self._staged(inp):
            o = torch.empty(out.shape, dtype=out.dtype)
            dist.all_to_all_single(o.view(-1), inp.reshape(-1).cpu(), group=self.group)
            out.copy_(o)
        else:
            dist.all_to_all_single(out.view(-1), inp.reshape(-1), group=self.group)

    def _scatter(self, out, chunks, src):
        if self._staged(out):
            o = torch.empty(out.shape, dtype=out.dtype)
            dist.scatter(o, None if chunks is None else [c.cpu() for c in chunks], src=src, group=self.group)
            out.copy_(o)
        else:
            dist.scatter(out, None if chunks is None else [c.contiguous() for c in chunks], src=src, group=self.group)

    def _gather_to(self, final, mine, dst, everywhere):
        """final [world, piece, 4] <- every rank's finished piece, on rank dst (or on all ranks)."""
        staged = self._staged(mine)
        src = mine.cpu() if staged else mine
        buf = torch.empty(final.shape, dtype=final.dtype) if staged else final
        if everywhere:
            dist.all_gather_into_tensor(buf.view(-1), src.reshape(-1), group=self.group)
        else:
            dist.gather(src, list(buf.unbind(0)) if self.rank == dst else None, dst=dst, group=self.group)
        if staged and (everywhere or self.rank == dst):
            final.copy_(buf)

    def _pieces(self, img, tag):
        """[world, piece, 4] view of an [H, W, 4] image, zero-padded up to world * piece pixels (copy only if padding is needed)."""
        npix = img.shape[0] * img.shape[1]
        piece = -(-npix // self.world)
        flat = img.reshape(npix, 4)
        if piece * self.world != npix:
            buf = self._pad.get(tag)
            if buf is None:
                buf = self._pad[tag] = torch.zeros((piece * self.world, 4), dtype=img.dtype, device=img.device)
            buf[:npix].copy_(flat)
            flat = buf
        return flat.view(self.world, piece, 4), npix, piece

    def render(self, cam, rp, result="rank0"):
        """One frame's image.  result = "rank0": the finished image is assembled on rank 0 only (the display GPU; other
        ranks return None); "all": on every rank."""
        if self.world == 1:
            return self.eng.raymarch(cam, rp)
        zb = self.eng.z_boundary(cam)
        plan, straddler = blend_plan(self.bounds, zb)
        over, under = self.eng.raymarch_partial(cam, rp)
        z0, z1 = self.bounds[self.rank]
        primary = over if z0 <= zb else under      # the straddler's primary is its OVER image
        if self.exchange == "all_gather":
            if self._whole_all is None:
                self._whole_all = torch.empty((self.world,) + tuple(primary.shape), dtype=primary.dtype, device=primary.device)
                self._whole_second = torch.empty_like(under)
            prim = self._all_gather(self._whole_all, primary)
            second = None
            if straddler is not None:
                second = under if self.rank == straddler else self._whole_second
                dist.broadcast(second, src=straddler, group=self.group)
            images = [second if (r == straddler and which == "under") else prim[r] for r, which, kind in plan]
            return self.eng.blend(images, [kind for _, _, kind in plan])
        try:
            return self._render_tiles(primary, under, plan, straddler, result)
        except (RuntimeError, NotImplementedError) as e:
            if self._tiles_ok:                     # it worked before: a real failure, not a missing collective
                raise
            # a backend without all_to_all / scatter / gather: every rank sees the same error on the first frame and
            # switches to the one-collective form for good
            import warnings
            warnings.warn(f"tiles exchange unavailable ({e}); falling back to exchange='all_gather'")
            self.exchange = "all_gather"
            return self.render(cam, rp, result)

    def _render_tiles(self, primary, under, plan, straddler, result):
        send, npix, piece = self._pieces(primary, "primary")
        if self._img_all is None:
            self._img_all = torch.empty_like(send)                     # [world, piece, 4]: piece `rank` of every rank's primary image
            self._second = torch.empty_like(send[0])
            self._final = torch.empty_like(send)                       # the assembled image (padded)
        self._all_to_all(self._img_all, send)
        if straddler is not None:
            chunks = None
            if self.rank == straddler:
                chunks = list(self._pieces(under, "second")[0].unbind(0))
            self._scatter(self._second, chunks, straddler)
        images, kinds = [], []
        for r, which, kind in plan:
            images.append(self._second if (r == straddler and which == "under") else self._img_all[r])
            kinds.append(kind)
        mine = self.eng.blend(images, kinds)                           # [piece, 4]
        self._gather_to(self._final, mine, 0, result == "all")
        self._tiles_ok = True
        if result != "all" and self.rank != 0:
            return None
        return self._final.view(-1, 4)[:npix].view(primary.shape)


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
