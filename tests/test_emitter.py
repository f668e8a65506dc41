"""CPU: the library's particle source (`vp_emitter_*`, csrc/emitter.cpp) -- SURVEY 8(f) row 3: an emitter with the demo ParticleSystem's
parameters (Assets/Volumetric_Particle_System.unity:2264-2620: cone type 4, angle 10 deg, radius 0.5, 10 particles / s, lifetime 6 s, speed 3,
size 4, random start rotation, 0.0698 rad/s, at most 60 particles).  Unity's own emitter is closed source, so what is checked is the documented
parameters (as properties of the emitted cloud), the generator (against a restatement of PCG32 written here) and the record layout."""
import ctypes as C
import math

import numpy as np
import pytest

from vpfx_amd import abi, engine as E, scene as S


def default_cfg():
    cfg = abi.vp_emitter_config()
    E.lib().vp_emitter_default_config(C.byref(cfg))
    return cfg


def test_defaults_are_the_demo_scene_values():
    c = default_cfg()
    assert (c.rate, c.lifetime, c.speed, c.size, c.cone_angle_deg, c.cone_radius, c.max_particles) == (10.0, 6.0, 3.0, 4.0, 10.0, 0.5, 60)
    assert math.isclose(math.radians(c.angular_velocity_deg), 0.0698, abs_tol=2e-4)     # RotationModule, scene:2587-2621
    assert list(c.reserved) == [0] * 6


def test_same_seed_same_cloud_and_other_seed_another():
    a, b, c = S.DemoEmitter(seed=11), S.DemoEmitter(seed=11), S.DemoEmitter(seed=12)
    for _ in range(100):
        a.step(1 / 30); b.step(1 / 30); c.step(1 / 30)
    pa, pb, pc = a.particles(), b.particles(), c.particles()
    assert len(pa) == len(pb) == len(pc) > 20
    assert pa.tobytes() == pb.tobytes() and pa.tobytes() != pc.tobytes()


def test_rate_cap_lifetime_and_cone():
    em = S.DemoEmitter(seed=3)
    counts = []
    for _ in range(30 * 9):                                   # 9 s at 30 Hz: past the first retirements
        counts.append(em.step(1.0 / 30.0))
    assert counts[29] in (9, 10, 11)                          # 10 particles / s
    assert max(counts) <= 60 and counts[-1] >= 55             # capped by max_particles, steady state = rate x lifetime
    p = em.particles()
    assert len(p) == counts[-1] == E.lib().vp_emitter_count(em.h)
    age = p["startLifetime"] - p["lifetime"]
    assert np.all(p["startLifetime"] == 6.0) and np.all(p["lifetime"] > 0) and np.all(age >= 0) and np.all(p["size"] == 4.0)
    assert np.all(np.diff(age) <= 1e-4)                        # the oldest first
    assert np.all((p["rotation"] >= 0) & (p["rotation"] < 360.0))
    old = age > 0.5
    vz = p["position"][old, 2] / age[old]                     # local +z is the cone's axis; speed 3, at most 10 degrees off it
    assert np.all(vz <= 3.0 + 1e-3) and np.all(vz >= 3.0 * math.cos(math.radians(10.0)) - 1e-3)
    lateral = np.hypot(p["position"][:, 0], p["position"][:, 1])
    assert np.all(lateral <= 0.5 + 3.0 * math.sin(math.radians(10.0)) * age + 1e-3)
    assert lateral[old].std() > 0.05                           # a cone, not a line


def test_prewarm_starts_in_steady_state_like_the_reference_scene():
    """scene:2269 `prewarm: 1`: the system is in steady state at the first frame.  vp_emitter_config.reserved[0] = 1 simulates one lifetime in
    1/30 s steps inside vp_emitter_create: the same cloud as stepping an un-prewarmed emitter by hand; any other value is refused."""
    L = E.lib()
    L.vp_emitter_create.argtypes = [C.c_void_p, C.c_void_p]
    cfg = default_cfg()
    cfg.seed = 21
    cfg.reserved[0] = 1
    h = C.c_void_p()
    assert L.vp_emitter_create(C.byref(cfg), C.byref(h)) == 0
    n = L.vp_emitter_count(h)
    assert 55 <= n <= 60                                       # rate x lifetime, capped by max_particles
    by_hand = S.DemoEmitter(seed=21)
    for _ in range(180):
        by_hand.step(1.0 / 30.0)
    ref = by_hand.particles()
    got = np.zeros(n, dtype=S.PARTICLE_DTYPE)
    lay = S.particle_layout(False)
    assert L.vp_emitter_write_particles(h, got.ctypes.data_as(C.c_void_p), n, C.byref(lay)) == n == len(ref)
    assert got.tobytes() == ref.tobytes()
    L.vp_emitter_destroy(h)
    cfg.reserved[0] = 7
    assert L.vp_emitter_create(C.byref(cfg), C.byref(h)) == abi.VP_ERR_BAD_ARG


def _pcg32_stream(seed):
    """PCG32 (XSH-RR 64/32) with the emitter's fixed stream constant: restated here, independent of the library."""
    M, inc, mask = 6364136223846793005, ((0xda3e39cb94b95bdb << 1) | 1) & (2**64 - 1), 2**64 - 1
    state = 0

    def step():
        nonlocal state
        old = state
        state = (old * M + inc) & mask
        x = (((old >> 18) ^ old) >> 27) & 0xFFFFFFFF
        r = old >> 59
        return ((x >> r) | (x << ((32 - r) & 31))) & 0xFFFFFFFF
    step(); state = (state + seed) & mask; step()
    while True:
        yield np.float32(step() >> 8) * np.float32(1.0 / 16777216.0)


def test_first_emission_follows_the_restated_generator():
    em = S.DemoEmitter(seed=42)
    em.step(0.35)                                              # 3 particles owed, all born in this step (age 0)
    p = em.particles()
    assert len(p) == 3
    u = _pcg32_stream(42)
    for i in range(3):
        a, b, c = next(u), next(u), next(u)
        rn = math.sqrt(float(a)); phi = 2 * math.pi * float(b)
        assert np.allclose(p["position"][i], (0.5 * rn * math.cos(phi), 0.5 * rn * math.sin(phi), 0.0), atol=2e-6)
        assert math.isclose(float(p["rotation"][i]), 360.0 * float(c), abs_tol=1e-4)
        assert p["lifetime"][i] == 6.0
    em.step(0.1)                                               # they move along the cone: tilt = 10 deg x r / radius
    q = em.particles()
    u = _pcg32_stream(42)
    for i in range(3):
        a, b, _ = next(u), next(u), next(u)
        rn = math.sqrt(float(a)); phi = 2 * math.pi * float(b); tilt = math.radians(10.0) * rn
        d = np.array([math.sin(tilt) * math.cos(phi), math.sin(tilt) * math.sin(phi), math.cos(tilt)])
        assert np.allclose(q["position"][i] - p["position"][i], 3.0 * 0.1 * d, atol=3e-6)
        assert math.isclose(float(q["lifetime"][i]), 5.9, abs_tol=1e-6)
        assert math.isclose(float(q["rotation"][i]), (float(p["rotation"][i]) + 0.4) % 360.0, abs_tol=1e-3)


def test_records_follow_the_callers_layout_and_rotation_unit():
    em = S.DemoEmitter(seed=5)
    for _ in range(40):
        em.step(0.05)
    ref = em.particles()
    lay = abi.vp_particle_layout()
    lay.stride, lay.off_size, lay.off_position, lay.off_lifetime, lay.off_start_lifetime, lay.off_rotation = 40, 0, 4, 20, 28, 32
    lay.rotation_in_radians = 1
    n = len(ref)
    buf = np.full(n * 40 + 8, 0xAB, dtype=np.uint8)
    L = E.lib()
    assert L.vp_emitter_write_particles(em.h, buf.ctypes.data_as(C.c_void_p), n, C.byref(lay)) == n
    assert np.all(buf[n * 40:] == 0xAB)                        # nothing past the last record
    rec = buf[:n * 40].reshape(n, 40)
    f = lambda off, k=1: rec[:, off:off + 4 * k].copy().view("<f4").reshape(n, k)
    assert np.array_equal(f(4, 3), ref["position"]) and np.array_equal(f(0)[:, 0], ref["size"])
    assert np.array_equal(f(20)[:, 0], ref["lifetime"]) and np.array_equal(f(28)[:, 0], ref["startLifetime"])
    assert np.allclose(f(32)[:, 0], np.radians(ref["rotation"]), atol=1e-6)
    assert np.all(rec[:, 16:20] == 0) and np.all(rec[:, 24:28] == 0) and np.all(rec[:, 36:40] == 0)     # the rest of a record is zeroed
    # a short buffer takes the oldest particles
    assert L.vp_emitter_write_particles(em.h, buf.ctypes.data_as(C.c_void_p), 2, C.byref(lay)) == 2
    assert np.array_equal(buf[4:16].view("<f4"), ref["position"][0])


def test_bad_arguments_are_refused():
    L = E.lib()
    h = C.c_void_p()
    assert L.vp_emitter_create(None, C.byref(h)) == abi.VP_ERR_BAD_ARG
    for field, value in (("lifetime", 0.0), ("cone_radius", 0.0), ("cone_angle_deg", 90.0), ("rate", -1.0), ("max_particles", -1), ("size", 0.0),
                         ("speed", float("nan")), ("lifetime", float("inf")), ("lifetime", 1.0e30), ("lifetime", float("nan")), ("rate", float("inf"))):
        cfg = default_cfg()
        setattr(cfg, field, value)
        assert L.vp_emitter_create(C.byref(cfg), C.byref(h)) == abi.VP_ERR_BAD_ARG and not h.value, field
    #  an unbounded lifetime with prewarm used to mean lifetime x 30 simulation steps inside vp_emitter_create: refused before any step runs
    cfg = default_cfg()
    cfg.lifetime, cfg.reserved[0] = float("inf"), 1
    assert L.vp_emitter_create(C.byref(cfg), C.byref(h)) == abi.VP_ERR_BAD_ARG and not h.value
    em = S.DemoEmitter()
    assert L.vp_emitter_step(em.h, C.c_float(-0.1)) == abi.VP_ERR_BAD_ARG
    assert L.vp_emitter_step(em.h, C.c_float(float("inf"))) == abi.VP_ERR_BAD_ARG
    assert L.vp_emitter_step(None, C.c_float(0.1)) == abi.VP_ERR_BAD_ARG and L.vp_emitter_count(None) == abi.VP_ERR_BAD_ARG
    lay = S.particle_layout()
    lay.off_position = lay.stride - 8                          # the position does not fit the record
    buf = np.zeros(84 * 4, dtype=np.uint8)
    em.step(1.0)
    assert L.vp_emitter_write_particles(em.h, buf.ctypes.data_as(C.c_void_p), 4, C.byref(lay)) == abi.VP_ERR_BAD_ARG
    assert L.vp_emitter_write_particles(em.h, None, 4, C.byref(S.particle_layout())) == abi.VP_ERR_BAD_ARG
    L.vp_emitter_destroy(None)                                 # like free(NULL)


def test_demo_scene_is_fed_by_the_library_emitter():
    sc, em, boxes = S.make_demo_scene(width=64, height=48)
    assert 50 <= len(sc.particles) <= 60 and len(boxes) == 8      # ground, back, two cubes, four cylinders
    assert sc.particles.tobytes() == em.particles().tobytes()
    again, _, _ = S.make_demo_scene(width=64, height=48)
    assert again.particles.tobytes() == sc.particles.tobytes()
