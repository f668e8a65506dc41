"""GPU: the drop-in boundary items of SURVEY 8(b) added in round 2, all through the C ABI, all against the oracle:
R8 displacement cube maps, refusal of out-of-range displacement inputs, the reference's per-metavoxel entry points
FillMetavoxel(xx,yy,zz) / RenderMetavoxel(xx,yy,zz,orderIndex), and the _ShowMetavoxelDrawOrder view."""
import numpy as np
import pytest

from vpfx_amd import abi, engine as E, scene as S
from vpfx_amd.manager import MetavoxelManager
from oracle import oracle as O

pytestmark = pytest.mark.gpu


def both(sc, **kw):
    o, g = O.Oracle(sc.config()), E.Engine(sc.config(), **kw)
    for x in (o, g):
        x.set_frame(sc.light_to_world, sc.grid_center)
        x.bin(sc.particles, sc.layout, sc.psys_local_to_world)
        x.fill(sc.fill_params())
    return o, g


def r8_cubemap(size=32, seed=5):
    rng = np.random.default_rng(seed)
    c = rng.integers(0, 256, size=(6, size, size), dtype=np.uint8)
    c[0, :4, :4] = 0            # exercise both ends of the UNORM range
    c[1, :4, :4] = 255
    return np.ascontiguousarray(c)


@pytest.mark.parametrize("D", [0.7, 1.0, 0.0])
def test_r8_cubemap_matches_the_oracle_fed_the_same_bytes(D):
    """The reference's displacement texture is 8-bit (DisplacementTexture.cubemap:10-23): bytes in, texel = byte/255.
    D = 1 with zero texels makes netDisplacement exactly 0 (smoothstep edges collapse): still the oracle's bricks."""
    sc = S.make_scene("T0")
    sc.cubemap = r8_cubemap()
    sc.displacement_scale = D
    o, g = both(sc, exact=True, early_out=False)
    cnt = o.bin_counts()
    np.testing.assert_array_equal(cnt, g.bin_counts())
    for zz, yy, xx in zip(*np.nonzero(cnt)):
        a, b = o.read_brick(xx, yy, zz).view(np.uint16), g.read_brick(xx, yy, zz).view(np.uint16)
        assert np.array_equal(a, b), (xx, yy, zz)
    np.testing.assert_allclose(g.read_lightmap(), o.read_lightmap(), rtol=1e-5, atol=1e-9)
    io, ig = o.raymarch(sc.camera(), sc.raymarch_params()), g.raymarch(sc.camera(), sc.raymarch_params())
    assert np.abs(io - ig).max() <= 1e-3
    # ... and the same bytes handed over as floats are the same frame: the R8 path is only a format
    sc2 = S.make_scene("T0")
    sc2.cubemap = np.ascontiguousarray(sc.cubemap.astype(np.float32) / np.float32(255.0))
    sc2.displacement_scale = D
    g2 = E.Engine(sc2.config(), exact=True, early_out=False)
    g2.set_frame(sc2.light_to_world, sc2.grid_center)
    g2.bin(sc2.particles, sc2.layout, sc2.psys_local_to_world)
    g2.fill(sc2.fill_params())
    np.testing.assert_array_equal(g2.raymarch(sc2.camera(), sc2.raymarch_params()), ig)


def test_displacement_scale_one_on_zero_texels_default_math():
    """D = 1 on texels that are exactly 0 makes netDisplacement 0: smoothstep(0, 0, x) = saturate(x / +0) = 1 in the reference
    (density = opacityFactor).  The default-math kernels must reproduce that discontinuity too (found by scripts/fuzz_parity.py)."""
    for fmt in ("r8", "f32"):
        sc = S.make_scene("T0")
        cube = r8_cubemap(32, 11)
        cube[:, ::2, :] = 0                                   # half of the rows all zero: many all-zero bilinear quads
        cube[2] = 0
        sc.cubemap = cube if fmt == "r8" else np.ascontiguousarray(cube.astype(np.float32) / np.float32(255.0))
        sc.displacement_scale = 1.0
        o, g = both(sc)                                       # default math (LDS path for r8, global table for f32)
        cnt = o.bin_counts()
        hit_zero = 0
        for zz, yy, xx in zip(*np.nonzero(cnt)):
            fa, fb = o.read_brick(xx, yy, zz), g.read_brick(xx, yy, zz)
            du = np.abs(fa.view(np.uint16).astype(np.int32) - fb.view(np.uint16).astype(np.int32))
            bad = du > 1
            assert not bad.any() or float(np.abs(fa.astype(np.float32) - fb.astype(np.float32))[bad].max()) <= 2e-5, (fmt, xx, yy, zz)
            hit_zero += int((fa[..., 3] >= np.float16(sc.opacity_factor * 0.999)).sum())
        assert hit_zero > 100                                  # the scene really has voxels at full opacityFactor


def test_r8_cubemap_default_math_within_one_fp16_ulp():
    sc = S.make_scene("T0")
    sc.cubemap = r8_cubemap(64, 9)
    o, g = both(sc)
    cnt = o.bin_counts()
    for zz, yy, xx in zip(*np.nonzero(cnt)):
        a, b = o.read_brick(xx, yy, zz).view(np.uint16).astype(np.int32), g.read_brick(xx, yy, zz).view(np.uint16).astype(np.int32)
        assert np.abs(a - b).max() <= 1, (xx, yy, zz)


def test_out_of_range_displacement_inputs_are_refused():
    """netDisplacement = D raw + (1 - D) >= 0 is what Fill.shader:119-126 assumes (slider [0,1], scene:8103-8111)."""
    sc = S.make_scene("T0")
    g = E.Engine(sc.config())
    g.set_frame(sc.light_to_world, sc.grid_center)
    g.bin(sc.particles, sc.layout, sc.psys_local_to_world)
    for D in (1.5, -0.1, float("nan")):
        sc.displacement_scale = D
        with pytest.raises(E.VpfxError) as ei:
            g.fill(sc.fill_params())
        assert ei.value.code == abi.VP_ERR_BAD_ARG and "displacement_scale" in str(ei.value)
    sc.displacement_scale = 0.7
    good = sc.cubemap
    for bad_value in (1.25, -0.01, np.nan):
        sc.cubemap = good.copy()
        sc.cubemap[3, 5, 7] = bad_value
        with pytest.raises(E.VpfxError) as ei:
            g.fill(sc.fill_params())
        assert ei.value.code == abi.VP_ERR_BAD_ARG and "texels" in str(ei.value)
        # the refused map is not resident either
        fp = sc.fill_params()
        fp.cubemap = None
        with pytest.raises(E.VpfxError):
            g.fill(fp)
    fp = sc.fill_params()
    fp.cubemap_format = 7
    with pytest.raises(E.VpfxError):
        g.fill(fp)
    sc.cubemap = good
    g.fill(sc.fill_params())                       # and a good map afterwards works
    o = O.Oracle(sc.config())
    o.set_frame(sc.light_to_world, sc.grid_center)
    o.bin(sc.particles, sc.layout, sc.psys_local_to_world)
    o.fill(sc.fill_params())
    assert np.abs(g.raymarch(sc.camera(), sc.raymarch_params()) - o.raymarch(sc.camera(), sc.raymarch_params())).max() <= 1e-3


def _manager(sc):
    m = MetavoxelManager(sc.N[0], sc.N[1], sc.N[2], sc.mv_scale, sc.nv, sc.border, sc.width, sc.height)
    m.Start()
    m.SetLight(sc.light_to_world)
    m.SetGridCenter(sc.grid_center)
    m.psysLocalToWorld = sc.psys_local_to_world
    m.SetDisplacementTexture(sc.cubemap)
    return m


@pytest.mark.parametrize("cam_pos", [None, (-9.0, 2.0, 1.0), (2.0, 6.0, -1.0)])
def test_per_metavoxel_entry_points_replay_the_reference_loops(cam_pos):
    """FillMetavoxel(xx,yy,zz) for every occupied MV in zz-major order (VPR.cs:505-518) == FillMetavoxels;
    RenderMetavoxel(xx,yy,zz,i) in the submission order of RenderMetavoxels (VPR.cs:652-711) == RenderMetavoxels == oracle."""
    sc = S.make_scene("T0")
    if cam_pos is not None:
        sc.set_camera(cam_pos)                     # zBoundary inside the grid: OVER and UNDER draws
    cam = sc.camera()
    m = _manager(sc)
    m.BinParticlesToMetavoxels(sc.particles, sc.layout)
    m.FillMetavoxels()
    e = m._engine
    cnt = e.bin_counts()
    bricks = {k: e.read_brick(k[2], k[1], k[0]).copy() for k in zip(*np.nonzero(cnt))}
    lm = e.read_lightmap()
    frame = m.RenderMetavoxels(cam)
    n_cov = m.numMetavoxelsCovered
    # the literal per-draw fill
    m.FillMetavoxelsPerDraw()
    assert m.numMetavoxelsCovered == n_cov == int((cnt != 0).sum())
    for k, b in bricks.items():
        assert np.array_equal(e.read_brick(k[2], k[1], k[0]).view(np.uint16), b.view(np.uint16)), k
    np.testing.assert_array_equal(e.read_lightmap(), lm)
    # the literal per-draw render
    per_draw = m.RenderMetavoxelsPerDraw(cam)
    assert np.abs(per_draw - frame).max() <= 2e-6          # same blends, associated the other way round
    o = O.Oracle(sc.config())
    o.set_frame(sc.light_to_world, sc.grid_center)
    o.bin(sc.particles, sc.layout, sc.psys_local_to_world)
    o.fill(sc.fill_params())
    io = o.raymarch(cam, sc.raymarch_params())
    assert np.abs(per_draw - io).max() <= 1e-3
    assert e.stats()["samples"] == o.stats()["samples"]    # the per-draw kernel executes the oracle's samples
    zb = e.z_boundary(cam)
    assert (zb >= 0) == (cam_pos is not None)


@pytest.mark.parametrize("nv,border,cam_pos", [(12, 1, (-7.0, 2.0, 1.0)), (24, 0, None), (9, 1, (2.0, 5.0, -1.0)), (40, 2, None)])
def test_per_metavoxel_entry_points_at_a_run_time_voxel_count(nv, border, cam_pos):
    """The same replay for voxel counts other than 16 / 32 / 64 (k_fill<.., GEN> column-range kernels, k_raymarch_one<0, ..>; border 0:
    the wrap of a brick whose edge is not a power of two)."""
    sc = S.make_scene("fuzz", seed=300 + nv, dims=(3, nv, 90, 72, 56), border=border)
    sc.set_camera(cam_pos if cam_pos is not None else (6.0, 4.5, -8.0))
    cam = sc.camera()
    m = _manager(sc)
    m.BinParticlesToMetavoxels(sc.particles, sc.layout)
    m.FillMetavoxels()
    e = m._engine
    cnt = e.bin_counts()
    bricks = {k: e.read_brick(k[2], k[1], k[0]).copy() for k in zip(*np.nonzero(cnt))}
    assert len(bricks) > 2
    lm = e.read_lightmap()
    frame = m.RenderMetavoxels(cam)
    m.FillMetavoxelsPerDraw()
    for k, b in bricks.items():
        assert np.array_equal(e.read_brick(k[2], k[1], k[0]).view(np.uint16), b.view(np.uint16)), k
    np.testing.assert_array_equal(e.read_lightmap(), lm)
    per_draw = m.RenderMetavoxelsPerDraw(cam)
    assert np.abs(per_draw - frame).max() <= 2e-6
    o = O.Oracle(sc.config())
    o.set_frame(sc.light_to_world, sc.grid_center)
    o.bin(sc.particles, sc.layout, sc.psys_local_to_world)
    o.fill(sc.fill_params())
    io = o.raymarch(cam, sc.raymarch_params())
    assert np.abs(per_draw - io).max() <= 1e-3
    assert e.stats()["samples"] == o.stats()["samples"]


def test_per_metavoxel_fill_with_an_r8_cubemap_and_a_storage_format_change():
    """vpfx.h: the per-metavoxel fill replays vp_fill within 1 fp16 ulp for R8 cube maps (vp_fill: LDS byte kernel; per metavoxel: float table);
    and a vp_fill_begin that changes the brick storage format (grey ambient -> coloured) clears the pool, so a PARTIAL refill leaves cleared
    bricks, never bricks of the other format read as garbage."""
    sc = S.make_scene("T0", cubemap="r8")
    e = E.Engine(sc.config())
    e.set_frame(sc.light_to_world, sc.grid_center)
    e.bin(sc.particles, sc.layout, sc.psys_local_to_world)
    e.fill(sc.fill_params())
    assert e.stats()["brick_format"] == 1                      # grey z-pair entries
    cnt = e.bin_counts()
    occ = list(zip(*np.nonzero(cnt)))
    whole = {k: e.read_brick(k[2], k[1], k[0]).copy() for k in occ}
    e.fill_begin(sc.fill_params())
    for zz, yy, xx in occ:                                     # zz-major = the reference's loop
        e.fill_metavoxel(xx, yy, zz)
    for k in occ:
        a, b = whole[k].view(np.uint16).astype(np.int32), e.read_brick(k[2], k[1], k[0]).view(np.uint16).astype(np.int32)
        assert np.abs(a - b).max() <= 1, k
    # now a coloured ambient and ONE metavoxel refilled: everything else must read as cleared RGBA16F, the refilled one as the oracle's brick
    sc.ambient = (0.1, 0.3, 0.2)
    e.fill_begin(sc.fill_params())
    z0, y0, x0 = occ[0]
    e.fill_metavoxel(x0, y0, z0)
    assert e.stats()["brick_format"] == 0
    o = O.Oracle(sc.config())
    o.set_frame(sc.light_to_world, sc.grid_center)
    o.bin(sc.particles, sc.layout, sc.psys_local_to_world)
    o.fill(sc.fill_params())
    a, b = o.read_brick(x0, y0, z0).view(np.uint16).astype(np.int32), e.read_brick(x0, y0, z0).view(np.uint16).astype(np.int32)
    assert np.abs(a - b).max() <= 1
    for zz, yy, xx in occ[1:]:
        assert not e.read_brick(xx, yy, zz).view(np.uint16).any(), (xx, yy, zz)
    img = e.raymarch(sc.camera(), sc.raymarch_params())
    assert np.isfinite(img).all() and img[..., 3].max() <= 1.0 + 1e-6
    e.close()


def test_fill_one_metavoxel_only_touches_that_metavoxel():
    sc = S.make_scene("T0")
    m = _manager(sc)
    m.BinParticlesToMetavoxels(sc.particles, sc.layout)
    m.FillMetavoxels()
    e = m._engine
    cnt = e.bin_counts()
    occ = list(zip(*np.nonzero(cnt)))
    zz, yy, xx = occ[len(occ) // 2]
    before = {k: e.read_brick(k[2], k[1], k[0]).copy() for k in occ}
    lm0 = e.read_lightmap()
    # refill that one MV with another opacity factor: only its brick changes, and only its light-map footprint
    m.SetOpacityFactor(0.08)
    m.FillMetavoxelsBegin()
    m.FillMetavoxel(xx, yy, zz)
    changed = [k for k in occ if not np.array_equal(e.read_brick(k[2], k[1], k[0]).view(np.uint16), before[k].view(np.uint16))]
    assert changed == [(zz, yy, xx)]
    lm = e.read_lightmap()
    nv = sc.nv
    mask = np.zeros_like(lm, dtype=bool)
    mask[yy * nv:(yy + 1) * nv, xx * nv:(xx + 1) * nv] = True
    np.testing.assert_array_equal(lm[~mask], 1.0)          # cleared to 1 by vp_fill_begin, untouched elsewhere
    assert (lm[mask] < 1.0).any() and lm0.shape == lm.shape
    # empty metavoxel: a no-op, not an error (the reference never submits one, VPR.cs:511)
    ez, ey, ex = [int(v[0]) for v in np.nonzero(cnt == 0)]
    m.FillMetavoxel(ex, ey, ez)
    with pytest.raises(E.VpfxError):
        m.FillMetavoxel(sc.N[0], 0, 0)


@pytest.mark.parametrize("cam_pos", [None, (3.0, 30.0, 2.0)])
def test_draw_order_view_matches_the_oracle(cam_pos):
    """_ShowMetavoxelDrawOrder (RM.shader:123-138, 170-173): the regression guard for the global submission order."""
    sc = S.make_scene("C1")
    if cam_pos is not None:
        sc.set_camera(cam_pos)
    o, g = both(sc)
    rp = sc.raymarch_params()
    rp.flags = abi.VP_RM_SHOW_DRAW_ORDER
    io, ig = o.raymarch(sc.camera(), rp), g.raymarch(sc.camera(), rp)
    # opaque colours blended in order: a mismatch of ONE position in the order changes a channel by >= 1/104
    assert np.abs(io - ig).max() <= 1e-6
    assert len(np.unique(ig.reshape(-1, 4), axis=0)) > 8    # several distinct order colours are visible (the nearest MV is opaque)
    # and the per-draw entry point shows the same picture when handed the same order indices
    m = _manager(sc)
    m.SetShowMetavoxelDrawOrder(True)
    m.BinParticlesToMetavoxels(sc.particles, sc.layout)
    m.FillMetavoxels()
    assert np.abs(m.RenderMetavoxelsPerDraw(sc.camera()) - io).max() <= 1e-6
    assert np.abs(m.RenderMetavoxels(sc.camera()) - io).max() <= 1e-6


def test_draw_order_with_exactly_tied_columns():
    """The columns' draw order is a STABLE sort by distance (VPR.cs:613-632; the list is built yy-major): with an unrotated light and the camera on
    the grid's axis the mirrored columns are exactly equidistant in f32 (4-way ties on an 8 x 8 grid), so the order among them is decided
    by the tie rule alone -- ranked on the device (k_rm_prepare: column_rank), compared with the oracle's sort through the draw-order view."""
    sc = S.make_scene("C1")
    sc.light_to_world = S.to_colmajor16(S.trs((0.0, 0.0, -44.34), np.eye(3)))
    D = 0.8 * sc.N[0] * sc.mv_scale
    sc.set_camera((0.0, 0.0, -D))
    o, g = both(sc)
    pos = o.mv_positions()[0].reshape(-1, 3).astype(np.float32)          # the zz = 0 slice: one position per column
    d = pos - np.asarray(sc.cam_pos, dtype=np.float32)
    key = (d[:, 0] * d[:, 0] + d[:, 1] * d[:, 1]) + d[:, 2] * d[:, 2]
    assert len(np.unique(key)) <= len(key) // 4 + 1                      # the scene really has the ties it is about
    rp = sc.raymarch_params()
    rp.flags = abi.VP_RM_SHOW_DRAW_ORDER
    io, ig = o.raymarch(sc.camera(), rp), g.raymarch(sc.camera(), rp)
    assert np.abs(io - ig).max() <= 1e-6
    assert len(np.unique(ig.reshape(-1, 4), axis=0)) > 8
    img_o, img_g = o.raymarch(sc.camera(), sc.raymarch_params()), g.raymarch(sc.camera(), sc.raymarch_params())
    assert np.abs(img_o - img_g).max() <= 1e-3


def test_grid_wider_than_the_on_chip_tables():
    """72 x 64 columns: more than the 4 096 keys the column ranking keeps in LDS (k_rm_prepare: column_rank recomputes them from global memory)
    and wider than the 32-bit occupancy rows of the cell walk -- both fall back to their table-free paths; frame and draw order vs the oracle."""
    sc = S.make_scene("wide", dims=(8, 16, 300, 160, 120))
    sc.N = (72, 64, 2)
    o, g = both(sc)
    assert o.stats()["occupied_mv"] == g.stats()["occupied_mv"] > 20
    img_o, img_g = o.raymarch(sc.camera(), sc.raymarch_params()), g.raymarch(sc.camera(), sc.raymarch_params())
    assert img_o[..., 3].max() > 0.05 and np.abs(img_o - img_g).max() <= 1e-3
    rp = sc.raymarch_params()
    rp.flags = abi.VP_RM_SHOW_DRAW_ORDER
    io, ig = o.raymarch(sc.camera(), rp), g.raymarch(sc.camera(), rp)
    assert np.abs(io - ig).max() <= 1e-6


def test_draw_order_view_needs_the_whole_grid():
    sc = S.make_scene("T0")
    g = E.Engine(sc.config(slab=(0, 2)))
    g.set_frame(sc.light_to_world, sc.grid_center)
    g.bin(sc.particles, sc.layout, sc.psys_local_to_world)
    g.fill(sc.fill_params())
    rp = sc.raymarch_params()
    rp.flags = abi.VP_RM_SHOW_DRAW_ORDER
    with pytest.raises(E.VpfxError) as ei:
        g.raymarch(sc.camera(), rp)
    assert ei.value.code == abi.VP_ERR_UNSUPPORTED


def test_manager_first_frame_always_fills():
    """A host whose first OnPostRender lands on a frame with frameCount % updateInterval != 0 must not ray-march unfilled textures."""
    sc = S.make_scene("T0")
    m = _manager(sc)
    img = m.OnPostRender(1, sc.particles, sc.layout, sc.camera())       # updateInterval = 2, frame 1
    m2 = _manager(sc)
    ref = m2.OnPostRender(0, sc.particles, sc.layout, sc.camera())
    np.testing.assert_array_equal(img, ref)


def test_displacement_scale_one_never_flips_a_voxel_across_the_smoothstep_jump():
    """Fuzz finding (seeds 404209 / 407542 / 409081 of scripts/fuzz_parity.py): with displacement scale exactly 1 the reference's smoothstep
    jumps at net displacement 0, so reciprocal-based texture coordinates flipped a voxel now and then (density 0 <-> opacityFactor).  Round 2
    sent such fills to the EXACT kernels (2x the time); now the default-math kernels -- LDS byte table for r8, global table for f32 -- recompute
    the in-face coordinates with the oracle's IEEE arithmetic whenever a lane's coordinate is within 1e-4 of an integer (cube_address, DONE):
    no voxel may change sides (<= 1 fp16 ulp / 2e-5 absolute like every default-math fill), on the white-noise maps that provoked the flips."""
    # (nv = 16 / 64: the other brick sizes of the same kernels -- a scheduling-dependent hazard once broke exactly those instantiations:
    # the D == 1 smoothstep's inline-asm multiply read v_rcp_f32's result without the wait state gfx950 needs)
    for seed, nv in ((404209, 32), (407542, 32), (409081, 32), (404209, 16), (407542, 64)):
        rng = np.random.default_rng(seed)
        for fmt in ("r8", "f32"):
            sc = S.make_scene("d1", dims=(2, nv, 600 if nv < 64 else 200, 64, 48), fade=1, seed=seed)
            cube = rng.integers(0, 256, size=(6, 128, 128), dtype=np.uint8)
            cube[rng.random(cube.shape) < 0.25] = 0                  # many zero texels: net displacement 0 wherever a weight is exactly 0
            sc.cubemap = cube if fmt == "r8" else np.ascontiguousarray(cube.astype(np.float32) / np.float32(255.0))
            sc.displacement_scale = 1.0
            o, g = both(sc)
            cnt = o.bin_counts()
            for zz, yy, xx in zip(*np.nonzero(cnt)):
                fa, fb = o.read_brick(xx, yy, zz), g.read_brick(xx, yy, zz)
                du = np.abs(fa.view(np.uint16).astype(np.int32) - fb.view(np.uint16).astype(np.int32))
                bad = du > 1
                assert not bad.any() or float(np.abs(fa.astype(np.float32) - fb.astype(np.float32))[bad].max()) <= 2e-5, (seed, nv, fmt, xx, yy, zz)
            np.testing.assert_allclose(g.read_lightmap(), o.read_lightmap(), rtol=3e-4, atol=1e-9)   # (a flipped voxel would show as 4 %)


def test_samples_per_ray_view_and_the_per_call_early_out_switch():
    """VP_RM_SHOW_RAY_SAMPLES (perf view): every pixel = the samples its ray executed, early-out left on; their sum is the frame's sample count.
    VP_RM_NO_EARLY_OUT (one call marches every lattice sample of SURVEY 8(d)'s formula on the resident bricks): the count equals the oracle's,
    which a context created with no_early_out also executes; the image is the default frame's up to rounding; the next default call is
    unaffected.  (k_raymarch_flat, the wave-coherent A/B traversal this test also covered until round 3, is compiled into VPFX_AB builds only.)"""
    from vpfx_amd import abi
    from oracle import oracle as O
    for ambient, cam_pos in (((0.2, 0.2, 0.2), None), ((0.1, 0.3, 0.2), (0.5, 0.3, 1.0)), ((0.2, 0.2, 0.2), (2.0, 9.0, 1.0))):
        sc = S.make_scene("C1", cubemap="r8")
        sc.ambient = ambient
        if cam_pos is not None:
            sc.set_camera(cam_pos)
        cam, rp = sc.camera(), sc.raymarch_params()
        g0 = E.Engine(sc.config())
        gall = E.Engine(sc.config(), early_out=False)
        o = O.Oracle(sc.config())
        for x in (g0, gall, o):
            x.set_frame(sc.light_to_world, sc.grid_center)
            x.bin(sc.particles, sc.layout, sc.psys_local_to_world)
            x.fill(sc.fill_params())
        i0 = g0.raymarch(cam, rp)
        executed = g0.stats()["samples"]
        o.raymarch(cam, rp)
        formula = o.stats()["samples"]
        rq = sc.raymarch_params()
        rq.flags = abi.VP_RM_NO_EARLY_OUT
        i_all = g0.raymarch(cam, rq)
        assert g0.stats()["samples"] == formula >= executed
        gall.raymarch(cam, rp)
        assert gall.stats()["samples"] == formula
        assert np.abs(i_all - i0).max() <= 2e-6
        i1 = g0.raymarch(cam, rp)                                      # the switch is per call
        np.testing.assert_array_equal(i0, i1)
        assert g0.stats()["samples"] == executed
        rq.flags = abi.VP_RM_SHOW_RAY_SAMPLES
        n = g0.raymarch(cam, rq)
        assert np.all(n[..., 3] == 1.0) and np.array_equal(n[..., 0], n[..., 2])
        assert int(n[..., 0].astype(np.int64).sum()) == g0.stats()["samples"]            # the view's own frame
        if cam_pos is None:        # all phase B: the flag kernel's literal order stops rays exactly where the default kernel does
            assert g0.stats()["samples"] == executed
        for x in (g0, gall):
            x.close()
        o.close()


def test_async_raymarch_overlaps_the_image_copy_with_the_next_frame():
    """vp_raymarch_async + vp_wait_image ("_async + vp_wait"): the image equals vp_raymarch's bit for bit, the next frame's bin +
    fill may be queued before the wait, a second async call reuses the context's image only after the first copy, vp_sync also lands it."""
    sc = S.make_scene("C1", cubemap="r8")
    e = E.Engine(sc.config())
    e.set_frame(sc.light_to_world, sc.grid_center)
    e.bin(sc.particles, sc.layout, sc.psys_local_to_world)
    e.fill(sc.fill_params())
    cam, rp = sc.camera(), sc.raymarch_params()
    ref = e.raymarch(cam, rp)
    a, b = np.zeros_like(ref), np.zeros_like(ref)
    e.pin(a); e.pin(b)
    e.raymarch_async(cam, rp, a)
    e.bin_resident(); e.fill(sc.fill_params())            # the next frame's work, queued while the copy runs
    e.wait_image()
    assert np.array_equal(a, ref)
    sc.set_camera((2.0, 1.0, -1.5))
    cam2 = sc.camera()
    ref2 = e.raymarch(cam2, rp)                            # synchronous call between two async ones
    e.raymarch_async(cam, rp, a)
    e.raymarch_async(cam2, rp, b)                          # waits on the device for a's copy before it overwrites the context image
    e.sync()                                               # vp_sync lands the pending image too
    assert np.array_equal(b, ref2)
    e.raymarch_async(cam, rp, a); e.wait_image(); e.wait_image()      # a second wait is a no-op
    assert np.array_equal(a, ref)
    e.unpin(a); e.unpin(b)
    e.close()



def test_manager_async_readback_shows_each_frame_one_frame_late():
    """asyncReadback (vp_raymarch_async behind the component mirror): frame n returns frame n - 1's image, bit for bit what the synchronous
    component returned for that frame."""
    sc = S.make_scene("T0")
    sync, lazy = _manager(sc), _manager(sc)
    lazy.asyncReadback = True
    cams = [sc.camera()]
    sc.set_camera((2.0, 1.0, -19.0))
    cams.append(sc.camera())
    moved = sc.particles.copy()
    moved["position"] += 0.4
    frames = [(0, sc.particles, cams[0]), (1, sc.particles, cams[1]), (2, moved, cams[1]), (3, moved, cams[0])]
    refs = [sync.OnPostRender(f, p, sc.layout, c).copy() for f, p, c in frames]
    got = []
    for f, p, c in frames:
        g = lazy.OnPostRender(f, p, sc.layout, c)                 # (a view of one of the two read-back buffers: valid until the call after next)
        got.append(None if g is None else g.copy())
    assert got[0] is None
    for i in range(1, len(frames)):
        np.testing.assert_array_equal(got[i], refs[i - 1])
    lazy.OnDestroy(); sync.OnDestroy()                     # (unpins the read-back buffers before numpy frees them)
