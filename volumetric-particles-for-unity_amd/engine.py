"""Thin ctypes binding of libvpfx.so (include/vpfx.h).  No compute here, and NO fallback: if the HIP
library is missing or no GPU is visible this module raises -- it never routes to a CPU implementation.

`Engine` has the same call surface as `oracle.oracle.Oracle` so the parity tests read symmetrically.
"""
from __future__ import annotations

import ctypes as C
import os
import sys

import numpy as np

from . import abi

_PKG = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_PKG, "libvpfx.so")
_lib = None


class VpfxError(RuntimeError):
    def __init__(self, what, code, msg):
        super().__init__(f"{what} -> {abi.STATUS_NAMES.get(code, code)}: {msg}")
        self.code = code


def _share_hip_runtime_with_torch():
    """PyTorch-ROCm wheels bundle their own libamdhip64 / libhsa-runtime64.  One process must use ONE HIP runtime: if libvpfx
    pulled in /opt/rocm's copy first, a later `import torch` finds no GPU.  So when torch is installed (not necessarily imported
    yet) its runtime is loaded first and libvpfx's NEEDED libamdhip64.so.7 resolves to it by SONAME.  No torch: nothing to do."""
    if "torch" in sys.modules or os.environ.get("VPFX_NO_TORCH_HIP_PRELOAD"):
        return
    try:
        import importlib.util
        spec = importlib.util.find_spec("torch")
        cand = os.path.join(os.path.dirname(spec.origin), "lib", "libamdhip64.so") if spec and spec.origin else None
        if cand and os.path.exists(cand):
            C.CDLL(cand, mode=C.RTLD_GLOBAL)
    except Exception:       # a broken torch install must not break the binding
        pass


def lib():
    """Load libvpfx.so (built in-tree by __graft_entry__.build() / csrc/Makefile)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise FileNotFoundError(
                f"{LIB_PATH} not found: build the HIP extension first (python -c 'import __graft_entry__ as g; g.build()'). "
                "There is no CPU fallback.")
        _share_hip_runtime_with_torch()
        L = C.CDLL(LIB_PATH)
        L.vp_last_error.restype = C.c_char_p
        L.vp_last_error.argtypes = [C.c_void_p]
        for name in abi.EXPORTED_SYMBOLS:
            getattr(L, name)       # AttributeError if the header and the library ever disagree
        L.vp_destroy.restype = None
        L.vp_emitter_destroy.restype = None
        L.vp_emitter_destroy.argtypes = [C.c_void_p]
        L.vp_emitter_default_config.restype = None
        L.vp_emitter_count.argtypes = [C.c_void_p]
        L.vp_emitter_step.argtypes = [C.c_void_p, C.c_float]
        _lib = L
    return _lib


def _vp(a):
    return C.c_void_p(a) if isinstance(a, int) else a.ctypes.data_as(C.c_void_p)


class Engine:
    """One vp_ctx (one GPU, one light-axis slab)."""

    def __init__(self, cfg: abi.vp_config, *, exact: bool = False, early_out: bool = True):
        self.L = lib()
        self.h = C.c_void_p()
        cfg = _copy_cfg(cfg)
        cfg.exact_math = 1 if exact else 0           # IEEE divisions in the fill kernel (bit-parity builds)
        cfg.no_early_out = 0 if early_out else 1     # saturation early-out of the ray-march
        self.cfg = cfg
        rc = self.L.vp_create(C.byref(cfg), C.byref(self.h))
        if rc:
            raise VpfxError("vp_create", rc, self.L.vp_last_error(None).decode())
        self.N = tuple(cfg.num_mv)
        self.nv = cfg.num_voxels
        self.W, self.H = cfg.width, cfg.height

    # -- lifetime ------------------------------------------------------------------------------------
    def close(self):
        if getattr(self, "h", None):
            self.L.vp_destroy(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _ck(self, rc, what):
        if rc:
            raise VpfxError(what, rc, self.L.vp_last_error(self.h).decode())

    def set_stream(self, hip_stream: int):
        self._ck(self.L.vp_set_stream(self.h, C.c_void_p(hip_stream)), "vp_set_stream")

    def sync(self):
        self._ck(self.L.vp_sync(self.h), "vp_sync")

    def pin(self, array: np.ndarray):
        """Page-lock a numpy buffer passed to bin()/raymarch(out=...) every frame (speed hint; keep `array` alive until unpin)."""
        self._ck(self.L.vp_pin_host_buffer(self.h, _vp(array), C.c_uint64(array.nbytes)), "vp_pin_host_buffer")

    def unpin(self, array: np.ndarray):
        self._ck(self.L.vp_unpin_host_buffer(self.h, _vp(array)), "vp_unpin_host_buffer")

    # -- frame / bin ---------------------------------------------------------------------------------
    def set_frame(self, light_to_world, grid_center):
        l = np.ascontiguousarray(light_to_world, dtype=np.float32)
        g = np.ascontiguousarray(grid_center, dtype=np.float32)
        self._ck(self.L.vp_set_frame(self.h, _vp(l), _vp(g)), "vp_set_frame")

    def mv_positions(self):
        out = np.empty((self.N[2], self.N[1], self.N[0], 3), dtype=np.float32)
        self._ck(self.L.vp_get_mv_positions(self.h, _vp(out)), "vp_get_mv_positions")
        return out

    def bin(self, particles, layout, psys_local_to_world):
        p = np.ascontiguousarray(particles)
        m = np.ascontiguousarray(psys_local_to_world, dtype=np.float32)
        self._ck(self.L.vp_bin(self.h, _vp(p), C.c_int32(len(p)), C.byref(layout), _vp(m)), "vp_bin")

    def upload_particles(self, particles, layout, psys_local_to_world):
        p = np.ascontiguousarray(particles)
        m = np.ascontiguousarray(psys_local_to_world, dtype=np.float32)
        self._ck(self.L.vp_upload_particles(self.h, _vp(p), C.c_int32(len(p)), C.byref(layout), _vp(m)), "vp_upload_particles")

    def bin_resident(self):
        self._ck(self.L.vp_bin_resident(self.h), "vp_bin_resident")

    def bin_counts(self):
        out = np.empty((self.N[2], self.N[1], self.N[0]), dtype=np.int32)
        self._ck(self.L.vp_read_bincounts(self.h, _vp(out)), "vp_read_bincounts")
        return out

    def bin_list(self, xx, yy, zz, cap=1 << 16):
        ids = np.empty(cap, dtype=np.int32)
        n = C.c_int32(0)
        self._ck(self.L.vp_read_binlist(self.h, int(xx), int(yy), int(zz), _vp(ids), cap, C.byref(n)), "vp_read_binlist")
        return ids[: n.value].copy()

    # -- fill ----------------------------------------------------------------------------------------
    def fill(self, params):
        self._ck(self.L.vp_fill(self.h, C.byref(params)), "vp_fill")

    def fill_begin(self, params):
        """Head of FillMetavoxels (VPR.cs:497-503): constants + light-propagation map cleared to 1."""
        self._ck(self.L.vp_fill_begin(self.h, C.byref(params)), "vp_fill_begin")

    def fill_metavoxel(self, xx, yy, zz):
        """FillMetavoxel(xx, yy, zz) (VPR.cs:559)."""
        self._ck(self.L.vp_fill_metavoxel(self.h, int(xx), int(yy), int(zz)), "vp_fill_metavoxel")

    def fill_local(self, params, d_tau_out: int):
        self._ck(self.L.vp_fill_local(self.h, C.byref(params), C.c_void_p(d_tau_out)), "vp_fill_local")

    def fill_finish(self, d_light_in: int | None):
        self._ck(self.L.vp_fill_finish(self.h, C.c_void_p(d_light_in or 0)), "vp_fill_finish")

    def fill_finish_gathered(self, d_tau_all: int, rank: int, world: int):
        self._ck(self.L.vp_fill_finish_gathered(self.h, C.c_void_p(d_tau_all), int(rank), int(world)), "vp_fill_finish_gathered")

    def read_brick(self, xx, yy, zz):
        out = np.empty((self.nv, self.nv, self.nv, 4), dtype=np.uint16)
        self._ck(self.L.vp_read_brick(self.h, int(xx), int(yy), int(zz), _vp(out)), "vp_read_brick")
        return out.view(np.float16)

    def read_lightmap(self):
        out = np.empty((self.N[1] * self.nv, self.N[0] * self.nv), dtype=np.float32)
        self._ck(self.L.vp_read_lightmap(self.h, _vp(out)), "vp_read_lightmap")
        return out

    # -- ray-march -----------------------------------------------------------------------------------
    def raymarch(self, cam, rp, out=None):
        img = np.empty((self.H, self.W, 4), dtype=np.float32) if out is None else out
        if img.shape != (self.H, self.W, 4) or img.dtype != np.float32 or not img.flags.c_contiguous:
            raise ValueError("raymarch(out=...): float32 C-contiguous [H, W, 4] expected")
        self._ck(self.L.vp_raymarch(self.h, C.byref(cam), C.byref(rp), _vp(img)), "vp_raymarch")
        return img

    def raymarch_async(self, cam, rp, out):
        """vp_raymarch_async: queue the ray-march and the copy of its image into `out` (float32 [H, W, 4], ideally pinned with pin());
        wait_image() before reading `out`."""
        if out.shape != (self.H, self.W, 4) or out.dtype != np.float32 or not out.flags.c_contiguous:
            raise ValueError("raymarch_async(out): float32 C-contiguous [H, W, 4] expected")
        self._ck(self.L.vp_raymarch_async(self.h, C.byref(cam), C.byref(rp), _vp(out)), "vp_raymarch_async")

    def wait_image(self):
        self._ck(self.L.vp_wait_image(self.h), "vp_wait_image")

    def clear_particles_rt(self):
        self._ck(self.L.vp_clear_particles_rt(self.h), "vp_clear_particles_rt")

    def render_metavoxel(self, cam, rp, xx, yy, zz, blend_over: bool, order_index: int = 0):
        """RenderMetavoxel(xx, yy, zz, orderIndex) (VPR.cs:766) into the context's particlesRT."""
        self._ck(self.L.vp_render_metavoxel(self.h, C.byref(cam), C.byref(rp), int(xx), int(yy), int(zz), 1 if blend_over else 0,
                                            int(order_index)), "vp_render_metavoxel")

    def read_particles_rt(self):
        img = np.empty((self.H, self.W, 4), dtype=np.float32)
        self._ck(self.L.vp_read_particles_rt(self.h, _vp(img)), "vp_read_particles_rt")
        return img

    def raymarch_device(self, cam, rp, d_rgba: int):
        self._ck(self.L.vp_raymarch_device(self.h, C.byref(cam), C.byref(rp), C.c_void_p(d_rgba)), "vp_raymarch_device")

    def raymarch_partial_device(self, cam, rp, d_over: int, d_under: int) -> int:
        mask = C.c_int32(0)
        self._ck(self.L.vp_raymarch_partial_device(self.h, C.byref(cam), C.byref(rp), C.c_void_p(d_over), C.c_void_p(d_under),
                                                   C.byref(mask)), "vp_raymarch_partial_device")
        return mask.value

    def blend_partials_device(self, d_partials, kinds, d_out: int, num_pixels=None):
        """Ordered blend of whole partial images, or (num_pixels given) of that many consecutive pixels."""
        n = len(d_partials)
        ptrs = (C.c_void_p * n)(*d_partials)
        k = (C.c_int32 * n)(*kinds)
        if num_pixels is None:
            self._ck(self.L.vp_blend_partials_device(self.h, ptrs, k, n, C.c_void_p(d_out)), "vp_blend_partials_device")
        else:
            self._ck(self.L.vp_blend_partials_range_device(self.h, ptrs, k, n, C.c_void_p(d_out), C.c_int64(int(num_pixels))),
                     "vp_blend_partials_range_device")

    def composite_device(self, d_particles: int, d_scene: int):
        self._ck(self.L.vp_composite_device(self.h, C.c_void_p(d_particles), C.c_void_p(d_scene)), "vp_composite_device")

    def set_occluders(self, solids):
        """Boxes (abi.vp_obb) go through vp_set_occluders; any typed solid (abi.vp_occluder: cylinder, ellipsoid) sends the list through vp_set_occluders2."""
        if all(isinstance(b, abi.vp_obb) for b in solids):
            arr = (abi.vp_obb * len(solids))(*solids) if len(solids) else None
            self._ck(self.L.vp_set_occluders(self.h, arr, len(solids)), "vp_set_occluders")
        else:
            self._ck(self.L.vp_set_occluders2(self.h, abi.as_occluders(solids), len(solids)), "vp_set_occluders2")

    def set_occluders2(self, solids):
        self._ck(self.L.vp_set_occluders2(self.h, abi.as_occluders(solids) if len(solids) else None, len(solids)), "vp_set_occluders2")

    def render_light_depth(self, near=0.3, far=1000.0, cam_distance=200.0):
        out = np.empty((self.N[1] * self.nv, self.N[0] * self.nv), dtype=np.float32)
        self._ck(self.L.vp_render_light_depth(self.h, C.c_float(near), C.c_float(far), C.c_float(cam_distance), _vp(out)), "vp_render_light_depth")
        return out

    def render_scene_depth(self, cam):
        out = np.empty((self.H, self.W), dtype=np.float32)
        self._ck(self.L.vp_render_scene_depth(self.h, C.byref(cam), _vp(out)), "vp_render_scene_depth")
        return out

    def z_histogram(self):
        out = np.zeros(self.N[2], dtype=np.int64)
        self._ck(self.L.vp_z_histogram(self.h, _vp(out)), "vp_z_histogram")
        return out

    def z_boundary(self, cam):
        zb = C.c_int32(0)
        self._ck(self.L.vp_z_boundary(self.h, C.byref(cam), C.byref(zb)), "vp_z_boundary")
        return zb.value

    # -- multi-GPU ------------------------------------------------------------------------------------
    def raymarch_partial_handoff_device(self, cam, rp, d_over: int, d_under: int, d_t_in: int = 0, n_in: int = 0, d_t_out0: int = 0,
                                        d_t_out1: int = 0) -> int:
        mask = C.c_int32(0)
        self._ck(self.L.vp_raymarch_partial_handoff_device(self.h, C.byref(cam), C.byref(rp), C.c_void_p(d_over), C.c_void_p(d_under), C.byref(mask),
                                                           C.c_void_p(d_t_in), int(n_in), C.c_void_p(d_t_out0), C.c_void_p(d_t_out1)),
                 "vp_raymarch_partial_handoff_device")
        return mask.value

    def zsamples(self):
        out = np.zeros(self.N[2], dtype=np.int64)
        self._ck(self.L.vp_read_zsamples(self.h, _vp(out)), "vp_read_zsamples")
        return out

    def rebalance(self):
        """Fan-out contexts: re-cut the slabs at the next bin from the measured work histograms."""
        self._ck(self.L.vp_rebalance(self.h), "vp_rebalance")

    def multi_info(self):
        mi = abi.vp_multi_info()
        self._ck(self.L.vp_get_multi_info(self.h, C.byref(mi)), "vp_get_multi_info")
        w = mi.world_size
        return dict(world_size=w, num_local=mi.num_local, first_rank=mi.first_rank, rccl_ranks=mi.rccl_ranks,
                    exchange="all_gather" if mi.exchange else "tiles", rm_groups=mi.rm_groups, slab_cuts=list(mi.slab_cuts[: w + 1]),
                    chain=list(mi.chain[:w]), group_of=list(mi.group_of[:w]), samples=list(mi.samples[:w]),
                    stage_ms=[list(mi.stage_ms[r]) for r in range(w)], exchange_ms=list(mi.exchange_ms[:3]))

    # -- stats ---------------------------------------------------------------------------------------
    def stats(self):
        st = abi.vp_stats()
        self._ck(self.L.vp_get_stats(self.h, C.byref(st)), "vp_get_stats")
        return {k: getattr(st, k) for k, _ in st._fields_ if k != "reserved"}

    def last_kernel_ms(self, stage: int) -> float:
        ms = C.c_float(0)
        self._ck(self.L.vp_last_kernel_ms(self.h, stage, C.byref(ms)), "vp_last_kernel_ms")
        return ms.value


def rccl_unique_id() -> bytes:
    """128 bytes naming a new RCCL communicator (rank 0 of a multi-process job calls this and hands them to every process)."""
    buf = (C.c_uint8 * 128)()
    rc = lib().vp_rccl_unique_id(buf)
    if rc:
        raise VpfxError("vp_rccl_unique_id", rc, lib().vp_last_error(None).decode())
    return bytes(buf)


def plan_slabs(nz: int, world: int, fill_ms=None, rm_ms=None, rm_groups: int = 1):
    """The library's slab cut (host only; no GPU needed): [(z0, z1)] per rank."""
    f = None if fill_ms is None else (C.c_double * nz)(*[float(x) for x in fill_ms])
    r = None if rm_ms is None else (C.c_double * nz)(*[float(x) for x in rm_ms])
    cuts = (C.c_int32 * (world + 1))()
    rc = lib().vp_plan_slabs(int(nz), int(world), f, r, int(rm_groups), cuts)
    if rc:
        raise VpfxError("vp_plan_slabs", rc, lib().vp_last_error(None).decode())
    return [(cuts[i], cuts[i + 1]) for i in range(world)]


def blend_plan(bounds, z_boundary: int):
    """The library's compositing order of the slabs: (chain, plan, straddler); plan = [(rank, which, kind)], which 0 = the slab's first
    image, 1 = the straddler's phase-B image; kind 0 = OVER, 1 = UNDER; chain = ranks front to back."""
    world = len(bounds)
    cuts = (C.c_int32 * (world + 1))(*([b[0] for b in bounds] + [bounds[-1][1]]))
    chain, pr, pw, pk = ((C.c_int32 * (world + 1))() for _ in range(4))
    n, strad = C.c_int32(0), C.c_int32(-1)
    rc = lib().vp_blend_plan(world, cuts, int(z_boundary), chain, pr, pw, pk, C.byref(n), C.byref(strad))
    if rc:
        raise VpfxError("vp_blend_plan", rc, lib().vp_last_error(None).decode())
    return list(chain[:world]), [(pr[i], pw[i], pk[i]) for i in range(n.value)], (None if strad.value < 0 else strad.value)


def exchange_plan(world: int, rank: int, straddler, all_gather: bool, phase: int):
    """The library's message schedule of the image exchange for one rank (host only): a list of abi.vp_xop, in execution order."""
    ops = (abi.vp_xop * (4 * world + 8))()
    n = C.c_int32(0)
    rc = lib().vp_exchange_plan(int(world), int(rank), -1 if straddler is None else int(straddler), 1 if all_gather else 0, int(phase), ops, len(ops), C.byref(n))
    if rc:
        raise VpfxError("vp_exchange_plan", rc, lib().vp_last_error(None).decode())
    return [ops[i] for i in range(n.value)]


def _copy_cfg(cfg):
    out = abi.vp_config()
    C.memmove(C.byref(out), C.byref(cfg), C.sizeof(abi.vp_config))
    return out
