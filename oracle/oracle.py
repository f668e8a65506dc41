"""ctypes wrapper of the CPU oracle (oracle/vp_oracle.c).

TEST INFRASTRUCTURE ONLY (parity unpinned -- see the header of vp_oracle.c).  Importers allowed:
tests/, __graft_entry__.smoke(), and bench.py's cpu_baseline leg.  The product never imports this.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "_build", "libvporacle.so")
_LIB_NATIVE = os.path.join(_HERE, "_build", "libvporacle_native.so")
_lib = None
_lib_native = None


def build_native() -> str:
    """The same source compiled -march=native on THIS machine (bench.py's CPU-baseline leg: the portable build targets x86-64-v3
    because the .so travels to the GPU box; the timed leg should use the box's own cores' ISA).  Never used by the parity tests."""
    subprocess.run(["make", "-C", _HERE, "-s", "-B", "native"], check=True)      # always rebuilt: a copy built elsewhere must not be reused
    return _LIB_NATIVE


def build(force: bool = False) -> str:
    """Compile the C restatement with gcc (oracle/Makefile)."""
    src = os.path.join(_HERE, "vp_oracle.c")
    hdr = os.path.join(_HERE, "..", "include", "vpfx.h")
    stale = (not os.path.exists(_LIB_PATH)) or any(
        os.path.getmtime(p) > os.path.getmtime(_LIB_PATH) for p in (src, hdr) if os.path.exists(p))
    if force or stale:
        subprocess.run(["make", "-C", _HERE, "-s"] + (["-B"] if force else []), check=True)
    return _LIB_PATH


def _declare(_lib):
        _lib.vpo_last_error.restype = C.c_char_p
        _lib.vpo_f16_to_f32.restype = C.c_float
        _lib.vpo_f16_to_f32.argtypes = [C.c_uint16]
        _lib.vpo_f32_to_f16.restype = C.c_uint16
        _lib.vpo_f32_to_f16.argtypes = [C.c_float]
        _lib.vpo_sample_cubemap.restype = C.c_float
        _lib.vpo_sample_cubemap.argtypes = [C.c_void_p, C.c_int, C.c_float, C.c_float, C.c_float]
        _lib.vpo_sincos_deg.argtypes = [C.c_float, C.POINTER(C.c_float), C.POINTER(C.c_float)]
        return _lib


def lib(native: bool = False):
    global _lib, _lib_native
    if native:
        if _lib_native is None:
            _lib_native = _declare(C.CDLL(build_native()))
        return _lib_native
    if _lib is None:
        build()
        _lib = _declare(C.CDLL(_LIB_PATH))
    return _lib


def _fp(a):
    return a.ctypes.data_as(C.c_void_p)


class OracleError(RuntimeError):
    pass


class Oracle:
    """Same call surface as the product's low-level engine (vpfx_amd.engine.Engine)."""

    def __init__(self, cfg, threads: int = 0, literal: bool = False, native: bool = False):
        self.L = lib(native)
        self.h = C.c_void_p()
        self.cfg = cfg
        rc = self.L.vpo_create(C.byref(cfg), C.byref(self.h))
        if rc:
            raise OracleError(f"vpo_create -> {rc}: {self.L.vpo_last_error(None)}")
        self.N = tuple(cfg.num_mv)
        self.nv = cfg.num_voxels
        self.W, self.H = cfg.width, cfg.height
        if threads:
            self.L.vpo_set_threads(self.h, threads)
        if literal:
            self.L.vpo_set_mode(self.h, 1)

    def close(self):
        if self.h:
            self.L.vpo_destroy(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _ck(self, rc, what):
        if rc:
            raise OracleError(f"{what} -> {rc}: {self.L.vpo_last_error(self.h).decode()}")

    def set_threads(self, n):
        self.L.vpo_set_threads(self.h, n)

    @staticmethod
    def max_threads():
        return lib().vpo_max_threads()

    def set_frame(self, light_to_world, grid_center):
        l = np.ascontiguousarray(light_to_world, dtype=np.float32)
        g = np.ascontiguousarray(grid_center, dtype=np.float32)
        self._ck(self.L.vpo_set_frame(self.h, _fp(l), _fp(g)), "vpo_set_frame")

    def mv_positions(self):
        out = np.empty((self.N[2], self.N[1], self.N[0], 3), dtype=np.float32)
        self._ck(self.L.vpo_get_mv_positions(self.h, _fp(out)), "vpo_get_mv_positions")
        return out

    def bin(self, particles, layout, psys_local_to_world):
        p = np.ascontiguousarray(particles)
        m = np.ascontiguousarray(psys_local_to_world, dtype=np.float32)
        self._ck(self.L.vpo_bin(self.h, _fp(p), C.c_int(len(p)), C.byref(layout), _fp(m)), "vpo_bin")

    def bin_counts(self):
        out = np.empty((self.N[2], self.N[1], self.N[0]), dtype=np.int32)
        self._ck(self.L.vpo_read_bincounts(self.h, _fp(out)), "vpo_read_bincounts")
        return out

    def bin_list(self, xx, yy, zz, cap=1 << 16):
        ids = np.empty(cap, dtype=np.int32)
        n = C.c_int(0)
        self._ck(self.L.vpo_read_binlist(self.h, int(xx), int(yy), int(zz), _fp(ids), cap, C.byref(n)), "vpo_read_binlist")
        return ids[: n.value].copy()

    def particle_records(self, count):
        out = np.empty((count, 16), dtype=np.float32)
        self._ck(self.L.vpo_read_particle_records(self.h, _fp(out)), "vpo_read_particle_records")
        return out

    def fill(self, params):
        self._ck(self.L.vpo_fill(self.h, C.byref(params)), "vpo_fill")

    def fill_local(self, params):
        tau = np.empty((self.N[1] * self.nv, self.N[0] * self.nv), dtype=np.float32)
        self._ck(self.L.vpo_fill_local(self.h, C.byref(params), _fp(tau)), "vpo_fill_local")
        return tau

    def fill_finish(self, light_in=None):
        if light_in is None:
            self._ck(self.L.vpo_fill_finish(self.h, None), "vpo_fill_finish")
        else:
            a = np.ascontiguousarray(light_in, dtype=np.float32)
            self._ck(self.L.vpo_fill_finish(self.h, _fp(a)), "vpo_fill_finish")

    def read_brick(self, xx, yy, zz):
        out = np.empty((self.nv, self.nv, self.nv, 4), dtype=np.uint16)
        self._ck(self.L.vpo_read_brick(self.h, int(xx), int(yy), int(zz), _fp(out)), "vpo_read_brick")
        return out.view(np.float16)

    def read_lightmap(self):
        out = np.empty((self.N[1] * self.nv, self.N[0] * self.nv), dtype=np.float32)
        self._ck(self.L.vpo_read_lightmap(self.h, _fp(out)), "vpo_read_lightmap")
        return out

    def raymarch(self, cam, rp):
        img = np.empty((self.H, self.W, 4), dtype=np.float32)
        self._ck(self.L.vpo_raymarch(self.h, C.byref(cam), C.byref(rp), _fp(img)), "vpo_raymarch")
        return img

    def raymarch_partial(self, cam, rp):
        over = np.empty((self.H, self.W, 4), dtype=np.float32)
        under = np.empty((self.H, self.W, 4), dtype=np.float32)
        mask = C.c_int(0)
        self._ck(self.L.vpo_raymarch_partial(self.h, C.byref(cam), C.byref(rp), _fp(over), _fp(under), C.byref(mask)),
                 "vpo_raymarch_partial")
        return over, under, mask.value

    def set_occluders(self, solids):
        """vp_obb records (60 B, boxes) or a list holding vp_occluder records (64 B, typed: box / capped cylinder / ellipsoid)."""
        if all(C.sizeof(b) == 60 for b in solids):
            arr = (type(solids[0]) * len(solids))(*solids) if len(solids) else None
            self._ck(self.L.vpo_set_occluders(self.h, arr, len(solids)), "vpo_set_occluders")
            return
        buf = (C.c_uint8 * (64 * len(solids)))()                    # vp_occluder = vp_obb + int32 type (0 = box)
        for i, b in enumerate(solids):
            C.memmove(C.addressof(buf) + 64 * i, C.byref(b), C.sizeof(b))
        self._ck(self.L.vpo_set_occluders2(self.h, buf, len(solids)), "vpo_set_occluders2")

    def render_light_depth(self, near=0.3, far=1000.0, cam_distance=200.0):
        out = np.empty((self.N[1] * self.nv, self.N[0] * self.nv), dtype=np.float32)
        self._ck(self.L.vpo_render_light_depth(self.h, C.c_float(near), C.c_float(far), C.c_float(cam_distance), _fp(out)), "vpo_render_light_depth")
        return out

    def render_scene_depth(self, cam):
        out = np.empty((self.H, self.W), dtype=np.float32)
        self._ck(self.L.vpo_render_scene_depth(self.h, C.byref(cam), _fp(out)), "vpo_render_scene_depth")
        return out

    def z_boundary(self, cam):
        zb = C.c_int(0)
        self._ck(self.L.vpo_z_boundary(self.h, C.byref(cam), C.byref(zb)), "vpo_z_boundary")
        return zb.value

    def stats(self):
        from_struct = _stats_struct()
        st = from_struct()
        self._ck(self.L.vpo_get_stats(self.h, C.byref(st)), "vpo_get_stats")
        return {k: getattr(st, k) for k, _ in st._fields_ if k != "reserved"}


def _stats_struct():
    class vp_stats(C.Structure):
        _fields_ = [("particles", C.c_int64), ("occupied_mv", C.c_int64), ("pairs", C.c_int64),
                    ("voxels_filled", C.c_int64), ("samples", C.c_int64), ("brick_bytes", C.c_int64),
                    ("max_pairs_per_mv", C.c_int64), ("bricks_sampled", C.c_int64), ("brick_bytes_per_voxel", C.c_int64),
                    ("brick_format", C.c_int64), ("reserved", C.c_int64 * 2)]
    return vp_stats


def blend_partials(W, H, partials, kinds):
    L = lib()
    arrs = [np.ascontiguousarray(p, dtype=np.float32) for p in partials]
    ptrs = (C.c_void_p * len(arrs))(*[a.ctypes.data for a in arrs])
    k = (C.c_int * len(arrs))(*kinds)
    out = np.empty((H, W, 4), dtype=np.float32)
    L.vpo_blend_partials(W, H, ptrs, k, len(arrs), _fp(out))
    return out


def composite(particles_rgba, scene_rgba):
    L = lib()
    p = np.ascontiguousarray(particles_rgba, dtype=np.float32)
    s = np.ascontiguousarray(scene_rgba, dtype=np.float32).copy()
    L.vpo_composite(p.shape[1], p.shape[0], _fp(p), _fp(s))
    return s
