"""Deterministic synthetic workload of SURVEY.md section 8(d) / BASELINE.md section 4.

The reference publishes no inputs (its particles come from a live Unity ParticleSystem), so every
benchmark and parity test runs on this generator: a uniform ball of displaced-sphere particles, the demo
scene's light orientation and renderer defaults (Assets/Volumetric_Particle_System.unity:9013-9026),
a look-at camera and a procedural value-noise displacement cubemap.

Everything here is host-side input preparation (numpy); no product compute happens in this file.
"""
from __future__ import annotations

import ctypes as C
import math
from dataclasses import dataclass, field

import numpy as np

from . import abi

# BASELINE.json "configs": name -> (N, nv, particles, width, height)
CONFIGS = {
    "C1": (8, 16, 1_000, 256, 256),
    "C2": (16, 32, 10_000, 1280, 720),
    "C3": (32, 32, 100_000, 1920, 1080),
    "C5": (64, 64, 1_000_000, 3840, 2160),
    # C3's grid, particles and screen with a voxel count that is none of 16 / 32 / 64: the run-time-nv ("generic") kernels on the benchmark's scale
    "C3nv24": (32, 24, 100_000, 1920, 1080),
    # one eighth of config 5's metavoxels at config 5's voxel count and screen: what EIGHT ranks sharing one test GPU can hold (bench.py --gpus 8 on the stand-in)
    "C5e": (32, 64, 125_000, 3840, 2160),
    # small extras used by tests
    "T0": (4, 16, 120, 96, 64),
    "T1": (6, 32, 400, 160, 120),
}

# Unity 5.0 ParticleSystem.Particle managed layout (84 bytes).  The ABI takes explicit
# offsets, so this is only the layout the synthetic generator happens to emit.
PARTICLE_DTYPE = np.dtype([
    ("position", "<f4", 3), ("velocity", "<f4", 3), ("animatedVelocity", "<f4", 3), ("axisOfRotation", "<f4", 3),
    ("rotation", "<f4"), ("angularVelocity", "<f4"), ("size", "<f4"), ("color", "<u4"), ("randomSeed", "<u4"),
    ("lifetime", "<f4"), ("startLifetime", "<f4"), ("emitAccumulator0", "<f4"), ("emitAccumulator1", "<f4"),
])
assert PARTICLE_DTYPE.itemsize == 84


def particle_layout(rotation_in_radians: bool = False) -> abi.vp_particle_layout:
    f = PARTICLE_DTYPE.fields
    lay = abi.vp_particle_layout()
    lay.stride = PARTICLE_DTYPE.itemsize
    lay.off_position = f["position"][1]
    lay.off_size = f["size"][1]
    lay.off_rotation = f["rotation"][1]
    lay.off_lifetime = f["lifetime"][1]
    lay.off_start_lifetime = f["startLifetime"][1]
    lay.rotation_in_radians = 1 if rotation_in_radians else 0
    return lay


def quat_to_matrix(q) -> np.ndarray:
    x, y, z, w = [float(v) for v in q]
    return np.array([
        [1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y)],
        [2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x)],
        [2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)],
    ], dtype=np.float64)


def trs(pos, rot3, scale=1.0) -> np.ndarray:
    m = np.eye(4, dtype=np.float64)
    m[:3, :3] = np.asarray(rot3, dtype=np.float64) * scale
    m[:3, 3] = pos
    return m


def to_colmajor16(m) -> np.ndarray:
    """4x4 (row, col) -> Unity Matrix4x4 memory order (column-major), float32."""
    return np.ascontiguousarray(np.asarray(m, dtype=np.float64).T.reshape(16)).astype(np.float32)


def make_cubemap(size: int = 128, seed: int = 4321, lo: float = 0.1, hi: float = 0.95) -> np.ndarray:
    """6 x S x S smooth value noise in [lo, hi] (faces +X,-X,+Y,-Y,+Z,-Z; row 0 = top)."""
    rng = np.random.default_rng(seed)
    coarse = rng.random((6, 17, 17))
    t = (np.arange(size) + 0.5) / size * 16.0
    i0 = np.floor(t).astype(int)
    f = t - i0
    i1 = np.minimum(i0 + 1, 16)
    rows = coarse[:, i0, :] * (1 - f)[None, :, None] + coarse[:, i1, :] * f[None, :, None]
    full = rows[:, :, i0] * (1 - f)[None, None, :] + rows[:, :, i1] * f[None, None, :]
    return np.ascontiguousarray((lo + (hi - lo) * full).astype(np.float32))


def make_cubemap_r8(size: int = 128, seed: int = 4321, lo: float = 0.1, hi: float = 0.95) -> np.ndarray:
    """The same smooth value noise as an 8-bit texture (uint8 [6,S,S], texel = byte/255): the format of the reference's own
    displacement asset (ARGB32, Assets/Textures/DisplacementTexture.cubemap:10-23)."""
    return np.ascontiguousarray(np.rint(make_cubemap(size, seed, lo, hi).astype(np.float64) * 255.0).astype(np.uint8))


def look_at_camera(pos, target=(0.0, 0.0, 0.0), up=(0.0, 1.0, 0.0)):
    """Unity Transform.LookAt + Camera.cameraToWorldMatrix (view space looks down -Z)."""
    pos = np.asarray(pos, dtype=np.float64)
    f = np.asarray(target, dtype=np.float64) - pos
    f /= np.linalg.norm(f)
    r = np.cross(np.asarray(up, dtype=np.float64), f)
    r /= np.linalg.norm(r)
    u = np.cross(f, r)
    c2w = np.eye(4)
    c2w[:3, 0] = r
    c2w[:3, 1] = u
    c2w[:3, 2] = -f
    c2w[:3, 3] = pos
    return c2w, np.linalg.inv(c2w)


@dataclass
class Scene:
    name: str
    N: tuple
    nv: int
    border: int
    mv_scale: float
    width: int
    height: int
    particles: np.ndarray            # structured PARTICLE_DTYPE
    layout: abi.vp_particle_layout
    psys_local_to_world: np.ndarray  # float32[16]
    light_to_world: np.ndarray       # float32[16]
    grid_center: np.ndarray          # float32[3]
    cubemap: np.ndarray              # float32[6,S,S]
    light_depth_map: np.ndarray | None
    scene_depth: np.ndarray | None
    # renderer defaults (scene:9013-9026)
    steps: int = 64
    soft_distance: int = 20
    opacity_factor: float = 0.04
    displacement_scale: float = 0.7
    ambient: tuple = (0.2, 0.2, 0.2)
    fade: int = 0
    cam_to_world: np.ndarray = field(default=None)
    world_to_cam: np.ndarray = field(default=None)
    cam_pos: np.ndarray = field(default=None)
    fov_y_deg: float = 60.0
    near: float = 0.3
    far: float = 1000.0

    # ---- ctypes views (keep the numpy arrays alive on self) -------------------------------------
    def config(self, device: int = -1, slab=(0, 0), devices=None, world_size: int = 0, first_rank: int = 0, multi_flags: int = 0,
               rm_groups: int = 0, rccl_unique_id: bytes | None = None) -> abi.vp_config:
        """devices = [ordinals]: a fan-out context (the library cuts the grid into len(devices) slabs, one per GPU; world_size / first_rank /
        rccl_unique_id: this process's share of a multi-process job)."""
        cfg = abi.vp_config()
        if devices is not None:
            cfg.num_devices = len(devices)
            for i, d in enumerate(devices):
                cfg.devices[i] = d
        cfg.world_size, cfg.first_rank, cfg.multi_flags, cfg.rm_groups = world_size, first_rank, multi_flags, rm_groups
        if rccl_unique_id is not None:
            assert len(rccl_unique_id) == 128
            for i, b in enumerate(rccl_unique_id):
                cfg.rccl_unique_id[i] = b
        cfg.num_mv[0], cfg.num_mv[1], cfg.num_mv[2] = self.N
        cfg.num_voxels = self.nv
        cfg.num_border = self.border
        cfg.mv_scale = self.mv_scale
        cfg.width, cfg.height = self.width, self.height
        cfg.device = device
        cfg.slab_z0, cfg.slab_z1 = slab
        return cfg

    def fill_params(self) -> abi.vp_fill_params:
        p = abi.vp_fill_params()
        p.opacity_factor = self.opacity_factor
        p.displacement_scale = self.displacement_scale
        p.fade_out_particles = self.fade
        p.ambient[0], p.ambient[1], p.ambient[2] = self.ambient
        p.init_light_intensity = 1.0
        p.light_near, p.light_far = 0.3, 1000.0
        p.light_cam_distance = 200.0
        abi.set_cubemap(p, self.cubemap)              # float32 [6,S,S] or uint8 (R8, the reference's asset format)
        if self.light_depth_map is not None:
            p.light_depth_map = self.light_depth_map.ctypes.data_as(abi.c_float_p)
        return p

    def camera(self) -> abi.vp_camera:
        cam = abi.vp_camera()
        w2c = to_colmajor16(self.world_to_cam)
        c2w = to_colmajor16(self.cam_to_world)
        for i in range(16):
            cam.world_to_camera[i] = w2c[i]
            cam.camera_to_world[i] = c2w[i]
        for i in range(3):
            cam.cam_pos[i] = np.float32(self.cam_pos[i])
        cam.fov_y = np.float32(math.radians(self.fov_y_deg))
        cam.near_clip, cam.far_clip = self.near, self.far
        return cam

    def raymarch_params(self) -> abi.vp_raymarch_params:
        rp = abi.vp_raymarch_params()
        rp.steps_per_mv = self.steps
        rp.soft_distance = self.soft_distance
        if self.scene_depth is not None:
            rp.scene_depth = self.scene_depth.ctypes.data_as(abi.c_float_p)
        return rp

    def set_camera(self, pos, target=(0.0, 0.0, 0.0)):
        self.cam_to_world, self.world_to_cam = look_at_camera(pos, target)
        self.cam_pos = np.asarray(pos, dtype=np.float32)


def make_scene(name: str = "C1", *, seed: int = 1234, size_range=(0.6, 1.4), rotation_in_radians=False,
               dims=None, border: int = 1, fade: int = 0, cubemap: str = "f32") -> Scene:
    """Build the section-8(d) scene for a named config (or dims=(N, nv, P, W, H)).  cubemap = "f32" (float texels) or "r8"
    (the same noise as an 8-bit texture, the reference asset's format)."""
    N, nv, P, W, H = dims if dims is not None else CONFIGS[name]
    s = 3.0
    rng = np.random.default_rng(seed)
    d = rng.normal(size=(P, 3))
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    r = 0.40 * N * s * rng.random(P) ** (1.0 / 3.0)
    position = d * r[:, None]
    size = s * rng.uniform(size_range[0], size_range[1], P)
    rotation = rng.uniform(0.0, 360.0, P)
    lifetime = rng.uniform(0.0, 6.0, P)

    parts = np.zeros(P, dtype=PARTICLE_DTYPE)
    parts["position"] = position.astype(np.float32)
    parts["size"] = size.astype(np.float32)
    parts["rotation"] = (np.radians(rotation) if rotation_in_radians else rotation).astype(np.float32)
    parts["startLifetime"] = 6.0
    parts["lifetime"] = lifetime.astype(np.float32)
    parts["axisOfRotation"] = (0.0, 0.0, 1.0)
    parts["color"] = 0xFFFFFFFF

    light_rot = quat_to_matrix((0.185593992, 0.0, 0.0, 0.982626557))
    light = trs((0.0, 0.0, -44.34), light_rot)
    sc = Scene(
        name=name, N=(N, N, N), nv=nv, border=border, mv_scale=s, width=W, height=H,
        particles=parts, layout=particle_layout(rotation_in_radians),
        psys_local_to_world=to_colmajor16(np.eye(4)),
        light_to_world=to_colmajor16(light),
        grid_center=np.zeros(3, dtype=np.float32),
        cubemap=make_cubemap_r8() if cubemap == "r8" else make_cubemap(), light_depth_map=None, scene_depth=None, fade=fade,
    )
    D = 0.8 * N * s
    sc.set_camera((-0.125 * D, 0.05 * D, -D))
    return sc


def make_box(center, half_extent, rot3=None) -> abi.vp_obb:
    """Oriented occluder box (rows of `rot3` = the box's unit axes in world space; default axis-aligned)."""
    b = abi.vp_obb()
    r = np.eye(3) if rot3 is None else np.asarray(rot3, dtype=np.float64)
    for i in range(3):
        b.center[i] = float(center[i])
        b.half_extent[i] = float(half_extent[i])
    for i in range(9):
        b.axes[i] = float(r.reshape(9)[i])
    return b


def make_solid(kind, center, half_extent, rot3=None) -> abi.vp_occluder:
    """Typed occluder (ABI 6): kind = abi.VP_OCC_BOX / VP_OCC_CYLINDER (axis = row 1 of `rot3`) / VP_OCC_ELLIPSOID."""
    o = abi.vp_occluder()
    C.memmove(C.byref(o), C.byref(make_box(center, half_extent, rot3)), C.sizeof(abi.vp_obb))
    o.type = int(kind)
    return o


def unity_primitive(kind, position, quat=(0.0, 0.0, 0.0, 1.0), scale=(1.0, 1.0, 1.0), parent_position=(0.0, 0.0, 0.0)) -> abi.vp_occluder:
    """A Unity built-in primitive mesh under a Transform (position, rotation quaternion, scale; parent = a pure translation):
    Cube (mesh 10202) spans [-0.5, 0.5]^3, Cylinder (mesh 10206) has radius 0.5 and height 2 about its local y, Sphere (mesh 10207) radius 0.5."""
    sx, sy, sz = (float(v) for v in scale)
    half = (0.5 * sx, sy, 0.5 * sz) if kind == abi.VP_OCC_CYLINDER else (0.5 * sx, 0.5 * sy, 0.5 * sz)
    center = np.asarray(position, dtype=np.float64) + np.asarray(parent_position, dtype=np.float64)
    return make_solid(kind, center, half, quat_to_matrix(quat).T)


# ---------------------------------------------------------------------------------------------------------------
# The demo scene of the reference (Assets/Volumetric_Particle_System.unity) -- SURVEY section 8(f) row 3
# ---------------------------------------------------------------------------------------------------------------
class DemoEmitter:
    """The demo's Unity ParticleSystem ("Particle System Demo", scene:2264-2620) as the library emits it: a thin binding of the C ABI's
    particle source (`vp_emitter_*`, csrc/emitter.cpp: cone angle 10 deg / radius 0.5, 10 particles/s, lifetime 6 s, speed 3, size 4, random
    start rotation, 4 deg/s, at most `max_particles` (scene: 60), simulated in LOCAL space).  No arithmetic here.
    `particles()` returns the live particles in the ParticleSystem.Particle layout the C ABI consumes."""

    def __init__(self, seed=7, **overrides):
        from . import engine
        self.L = engine.lib()
        cfg = abi.vp_emitter_config()
        self.L.vp_emitter_default_config(C.byref(cfg))
        cfg.seed = seed
        for k, v in overrides.items():
            assert hasattr(cfg, k), k
            setattr(cfg, k, v)
        self.cfg = cfg
        self.h = C.c_void_p()
        rc = self.L.vp_emitter_create(C.byref(cfg), C.byref(self.h))
        if rc:
            raise ValueError(f"vp_emitter_create -> {abi.STATUS_NAMES.get(rc, rc)}")
        self.layout = particle_layout(False)

    def step(self, dt):
        n = self.L.vp_emitter_step(self.h, C.c_float(dt))
        if n < 0:
            raise ValueError(f"vp_emitter_step -> {abi.STATUS_NAMES.get(n, n)}")
        return n

    def particles(self, layout=None):
        lay = self.layout if layout is None else layout
        n = self.L.vp_emitter_count(self.h)
        p = np.zeros(n, dtype=PARTICLE_DTYPE)
        got = self.L.vp_emitter_write_particles(self.h, p.ctypes.data_as(C.c_void_p), n, C.byref(lay))
        assert got == n, (got, n)
        return p

    def close(self):
        if self.h:
            self.L.vp_emitter_destroy(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def make_demo_scene(width=1024, height=768, warm_seconds=6.0, seed=7):
    """The reference's scene: 10^3 metavoxels x 32^3 voxels of size 3 (scene:9013-9016), main camera at (-10,0,-20) looking
    +z with fov 60, directional light quaternion (0.1856,0,0,0.9826), grid centre at world (0,-5,0), particle system at world
    (0,0,11.2) rotated 180 deg about Y, and the scene's eight opaque meshes as occluder solids: ground / back (scaled cubes), two cubes, four
    cylinders.  Returns (scene, emitter, solids)."""
    em = DemoEmitter(seed=seed)
    steps = int(round(warm_seconds * 30))
    for _ in range(steps):
        em.step(1.0 / 30.0)
    parts = em.particles()
    light = trs((0.0, 0.0, -44.34), quat_to_matrix((0.185593992, 0.0, 0.0, 0.982626557)))
    psys = trs((0.0, 0.0, 11.2), quat_to_matrix((0.0, 1.0, 0.0, -4.37113883e-08)))
    sc = Scene(name="DEMO", N=(10, 10, 10), nv=32, border=1, mv_scale=3.0, width=width, height=height, particles=parts,
               layout=particle_layout(False), psys_local_to_world=to_colmajor16(psys), light_to_world=to_colmajor16(light),
               grid_center=np.array([0.0, -5.0, 0.0], dtype=np.float32), cubemap=make_cubemap(), light_depth_map=None,
               scene_depth=None)
    c2w = np.eye(4)
    c2w[:3, 2] = (0.0, 0.0, -1.0)                   # Unity camera looks down its +z; view space looks down -z
    c2w[:3, 3] = (-10.0, 0.0, -20.0)
    sc.cam_to_world, sc.world_to_cam = c2w, np.linalg.inv(c2w)
    sc.cam_pos = np.array([-10.0, 0.0, -20.0], dtype=np.float32)
    # every MeshRenderer of the scene on the Default layer (VPR.cs:346 lightCamera.cullingMask; the main camera draws them too), all children of
    # the object "Scene" at (0,-5,0) (scene:6525-6533): ground and back are CUBES (mesh 10202) scaled (50,1,50) -- scene:4703-4760, 6313-6382 --,
    # "Cube" / "Cube 1" unit cubes, "Cylinder" / "Cylinder 1" / "Cylinder 2" / "Cylinder 3" unit cylinders (mesh 10206: radius 0.5, height 2;
    # scene:8623, 1755, 8382, 5462; "Cylinder 1" and "Cylinder 2" coincide in the scene file)
    par = (0.0, -5.0, 0.0)
    boxes = [
        unity_primitive(abi.VP_OCC_BOX, (0.0, -1.52, 0.0), scale=(50.0, 1.0, 50.0), parent_position=par),                               # ground
        unity_primitive(abi.VP_OCC_BOX, (0.0, 24.0, 24.5), quat=(0.707106829, 0.0, 0.0, 0.707106829), scale=(50.0, 1.0, 50.0), parent_position=par),  # back
        unity_primitive(abi.VP_OCC_BOX, (-5.34, 6.18, -11.89), parent_position=par),                                                    # Cube
        unity_primitive(abi.VP_OCC_BOX, (0.0, 4.57, -11.0), parent_position=par),                                                       # Cube 1
        unity_primitive(abi.VP_OCC_CYLINDER, (-8.24, 0.0, 0.0), parent_position=par),                                                   # Cylinder
        unity_primitive(abi.VP_OCC_CYLINDER, (0.0, 0.0, 0.0), parent_position=par),                                                     # Cylinder 1
        unity_primitive(abi.VP_OCC_CYLINDER, (0.0, 0.0, 0.0), parent_position=par),                                                     # Cylinder 2
        unity_primitive(abi.VP_OCC_CYLINDER, (1.45, -1.05, 9.11), parent_position=par),                                                 # Cylinder 3
    ]
    return sc, em, boxes
