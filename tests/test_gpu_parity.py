"""GPU parity: libvpfx (HIP, through the C ABI) vs the CPU oracle on identical seeded inputs.

Tolerances: bin lists / counts bit-exact (integer work); bricks bit-identical in `exact` mode and within
1 fp16 ulp in the default fast-reciprocal mode; light map 1e-5 rel; final RGBA 1e-3 abs (north_star).
"""
import os
import numpy as np
import pytest

from vpfx_amd import scene as S
from vpfx_amd import engine as E
from oracle import oracle as O

pytestmark = pytest.mark.gpu


def run_pair(sc, exact=False, early_out=True):
    o = O.Oracle(sc.config())
    g = E.Engine(sc.config(), exact=exact, early_out=early_out)
    for eng in (o, g):
        eng.set_frame(sc.light_to_world, sc.grid_center)
        eng.bin(sc.particles, sc.layout, sc.psys_local_to_world)
        eng.fill(sc.fill_params())
    return o, g


def f16_ulp_diff(a, b):
    """|a-b| in units of fp16 ulps (monotone integer mapping of binary16)."""
    def key(x):
        u = x.view(np.uint16).astype(np.int32)
        return np.where(u & 0x8000, 0x8000 - (u & 0x7fff) - 1 + 0, u + 0x8000)
    return np.abs(key(a) - key(b))


@pytest.mark.parametrize("name", ["T0", "C1"])
def test_grid_and_bins_exact(name):
    sc = S.make_scene(name)
    o, g = run_pair(sc)
    np.testing.assert_array_equal(o.mv_positions(), g.mv_positions())
    co, cg = o.bin_counts(), g.bin_counts()
    np.testing.assert_array_equal(co, cg)
    so, sg = o.stats(), g.stats()
    for k in ("occupied_mv", "pairs", "max_pairs_per_mv", "voxels_filled"):
        assert so[k] == sg[k], k
    zz, yy, xx = np.nonzero(co)
    for i in range(0, len(zz), max(1, len(zz) // 40)):
        np.testing.assert_array_equal(o.bin_list(xx[i], yy[i], zz[i]), g.bin_list(xx[i], yy[i], zz[i]))


@pytest.mark.parametrize("name,exact", [("T0", True), ("C1", True), ("T1", True), ("C1", False), ("T1", False)])
def test_fill_bricks_and_lightmap(name, exact):
    sc = S.make_scene(name)
    o, g = run_pair(sc, exact=exact)
    co = o.bin_counts()
    zz, yy, xx = np.nonzero(co)
    worst = 0
    for i in range(len(zz)):
        bo, bg = o.read_brick(xx[i], yy[i], zz[i]), g.read_brick(xx[i], yy[i], zz[i])
        if exact:
            assert np.array_equal(bo.view(np.uint16), bg.view(np.uint16)), (xx[i], yy[i], zz[i])
        else:
            worst = max(worst, int(f16_ulp_diff(bo, bg).max()))
    assert worst <= 1
    lo, lg = o.read_lightmap(), g.read_lightmap()
    if exact:
        np.testing.assert_array_equal(lo, lg)
    else:
        np.testing.assert_allclose(lg, lo, rtol=1e-5, atol=1e-9)


@pytest.mark.parametrize("name", ["T0", "C1", "T1"])
def test_raymarch_rgba(name):
    sc = S.make_scene(name)
    o, g = run_pair(sc, early_out=False)
    io = o.raymarch(sc.camera(), sc.raymarch_params())
    ig = g.raymarch(sc.camera(), sc.raymarch_params())
    err = np.abs(io - ig).max()
    assert err <= 1e-3, err
    so, sg = o.stats()["samples"], g.stats()["samples"]
    assert so == sg, (so, sg)                                     # every lattice sample of the oracle, none more
    # default (saturation early-out) must give the same image, with no more samples
    g2 = E.Engine(sc.config())
    g2.set_frame(sc.light_to_world, sc.grid_center)
    g2.bin(sc.particles, sc.layout, sc.psys_local_to_world)
    g2.fill(sc.fill_params())
    ig2 = g2.raymarch(sc.camera(), sc.raymarch_params())
    assert np.abs(io - ig2).max() <= 1e-3
    assert g2.stats()["samples"] <= sg


def test_survey_anchor_pixels_with_the_reference_displacement_cubemap():
    """HIP path against the surveyor's independent anchors (see the same test in test_oracle_golden.py)."""
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "survey_anchors_C1.npz"))
    sc = S.make_scene("C1")
    sc.cubemap = np.ascontiguousarray(g["cubemap_r"].astype(np.float32) / np.float32(255.0))
    e = E.Engine(sc.config())
    e.set_frame(sc.light_to_world, sc.grid_center)
    e.bin(sc.particles, sc.layout, sc.psys_local_to_world)
    e.fill(sc.fill_params())
    img = e.raymarch(sc.camera(), sc.raymarch_params())
    assert np.abs(img[g["rows"], g["cols"], 0] - g["r"]).max() <= 5e-5 and np.abs(img[g["rows"], g["cols"], 3] - g["a"]).max() <= 5e-5
    assert abs(float(img[..., 3].mean()) - 0.802) < 1e-3 and abs(float(img[..., :3].max()) - 0.543) < 1e-3


# ---- voxel counts other than 16 / 32 / 64 (the run-time-nv kernels: fill_generic.hip, raymarch_generic.hip) -------------------------------
# numVoxelsInMetavoxel is a public inspector int (VPR.cs:84), used as is for the texture extents (VPR.cs:312-314) and handed to the shaders
# as the FLOAT uniform _NumVoxels (VPR.cs:527, 722): odd counts divide by two as floats in get_voxel_world_pos (Fill.shader:103).
def _nv_scene(nv, border, seed=77, N=4, P=260, W=120, H=88, **kw):
    sc = S.make_scene("fuzz", seed=seed + nv, dims=(N, nv, P, W, H), border=border, **kw)
    sc.set_camera((5.5, 4.0, -7.5), target=(0.2, -0.1, 0.3))
    return sc


@pytest.mark.parametrize("nv,border", [(8, 1), (12, 1), (24, 2), (13, 1), (5, 0), (20, 0), (4, 1), (40, 3), (63, 1), (3, 1), (2, 0)])
def test_any_voxel_count_exact_bricks_lightmap_rgba_and_samples(nv, border):
    sc = _nv_scene(nv, border)
    o, g = run_pair(sc, exact=True, early_out=False)
    co = o.bin_counts()
    np.testing.assert_array_equal(co, g.bin_counts())
    zz, yy, xx = np.nonzero(co)
    assert len(zz) > 3
    for i in range(len(zz)):
        bo, bg = o.read_brick(xx[i], yy[i], zz[i]), g.read_brick(xx[i], yy[i], zz[i])
        assert bo.shape == bg.shape == (nv, nv, nv, 4)
        assert np.array_equal(bo.view(np.uint16), bg.view(np.uint16)), (nv, xx[i], yy[i], zz[i])      # EXACT build: bit-identical
    np.testing.assert_allclose(g.read_lightmap(), o.read_lightmap(), rtol=1e-5, atol=1e-9)
    io, ig = o.raymarch(sc.camera(), sc.raymarch_params()), g.raymarch(sc.camera(), sc.raymarch_params())
    assert np.abs(io - ig).max() <= 1e-3
    assert o.stats()["samples"] == g.stats()["samples"] > 0                     # every lattice sample of the oracle, none more
    assert g.stats()["voxels_filled"] == len(zz) * nv ** 3


@pytest.mark.parametrize("nv,border,r8,ambient", [(8, 1, True, None), (12, 1, False, None), (24, 2, True, None), (24, 1, True, (0.1, 0.3, 0.2)),
                                                   (13, 0, True, None), (48, 1, True, None), (27, 1, False, (0.3, 0.2, 0.1))])
def test_any_voxel_count_default_math(nv, border, r8, ambient):
    """Default (fast-reciprocal) fill: <= 1 fp16 ulp; R8 cube maps run the LDS-resident kernel; grey ambient -> z-pair bricks, coloured -> RGBA16F."""
    sc = _nv_scene(nv, border, seed=91)
    if r8:
        sc.cubemap = S.make_cubemap_r8(128 if nv % 2 == 0 else 77)
    if ambient:
        sc.ambient = ambient
    o, g = run_pair(sc)
    co = o.bin_counts()
    worst = 0
    for zz, yy, xx in zip(*np.nonzero(co)):
        worst = max(worst, int(f16_ulp_diff(o.read_brick(xx, yy, zz), g.read_brick(xx, yy, zz)).max()))
    assert worst <= 1
    np.testing.assert_allclose(g.read_lightmap(), o.read_lightmap(), rtol=5e-5, atol=1e-9)
    io, ig = o.raymarch(sc.camera(), sc.raymarch_params()), g.raymarch(sc.camera(), sc.raymarch_params())
    assert np.abs(io - ig).max() <= 1e-3
    from vpfx_amd import abi
    assert g.stats()["brick_format"] == (abi.VP_BRICKS_GREY_ZPAIR if (ambient is None and border >= 1) else abi.VP_BRICKS_RGBA16F)


@pytest.mark.parametrize("nv,border,world", [(12, 1, 2), (24, 2, 3), (7, 0, 2)])
def test_any_voxel_count_fan_out(nv, border, world):
    """The split fill (local pass + finish) and the partial-image ray-march at a run-time voxel count: the fan-out inside the library on one GPU."""
    from vpfx_amd import abi
    sc = _nv_scene(nv, border, seed=13, N=4, P=300)
    o, g = run_pair(sc)
    cam, rp = sc.camera(), sc.raymarch_params()
    single = g.raymarch(cam, rp)
    for flags in (abi.VP_MULTI_PEER_COPY, abi.VP_MULTI_PEER_COPY | abi.VP_MULTI_EXCHANGE_ALL_GATHER):
        mf = E.Engine(sc.config(devices=[0] * world, multi_flags=flags))
        mf.set_frame(sc.light_to_world, sc.grid_center)
        mf.bin(sc.particles, sc.layout, sc.psys_local_to_world)
        mf.fill(sc.fill_params())
        img = mf.raymarch(cam, rp)
        assert np.abs(img - single).max() <= 1e-4 and np.abs(img - o.raymarch(cam, rp)).max() <= 1e-3
        np.testing.assert_allclose(mf.read_lightmap(), o.read_lightmap(), rtol=6e-5, atol=1e-9)
        mf.close()
