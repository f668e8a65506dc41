"""GPU: the chained fill (fill.hip, FillChain).  The fill kernels' unit of work is one metavoxel of one column tile; the light a column has
transmitted so far reaches the unit of the column's next occupied metavoxel through a tagged word in memory.  Long columns, columns with
gaps (empty metavoxels are skipped, VPR.cs:511), columns with no occupied metavoxel at all, slab contexts and many refills of one context
(the tags of successive launches must not collide) -- always against the oracle, whose fill is the reference's plain loop."""
import numpy as np
import pytest

from vpfx_amd import abi, engine as E, scene as S
from oracle import oracle as O

pytestmark = pytest.mark.gpu


def engines(sc, **kw):
    o, g = O.Oracle(sc.config()), E.Engine(sc.config(), **kw)
    for x in (o, g):
        x.set_frame(sc.light_to_world, sc.grid_center)
        x.bin(sc.particles, sc.layout, sc.psys_local_to_world)
        x.fill(sc.fill_params())
    return o, g


def check(o, g, exact):
    cnt = o.bin_counts()
    np.testing.assert_array_equal(cnt, g.bin_counts())
    for zz, yy, xx in zip(*np.nonzero(cnt)):
        a = o.read_brick(xx, yy, zz).view(np.uint16).astype(np.int32)
        b = g.read_brick(xx, yy, zz).view(np.uint16).astype(np.int32)
        assert np.abs(a - b).max() <= (0 if exact else 1), (xx, yy, zz)
    if exact:
        np.testing.assert_array_equal(g.read_lightmap(), o.read_lightmap())
    else:
        np.testing.assert_allclose(g.read_lightmap(), o.read_lightmap(), rtol=1e-5, atol=1e-9)


@pytest.mark.parametrize("cubemap,exact", [("f32", True), ("f32", False), ("r8", False)])      # k_fill EXACT / default, k_fill_lds
@pytest.mark.parametrize("N,nv", [((2, 2, 24), 16), ((3, 2, 12), 32), ((1, 1, 6), 64)])
def test_long_columns_with_gaps(N, nv, cubemap, exact):
    sc = S.make_scene("chain", dims=(max(N), nv, 600, 64, 48), cubemap=cubemap, seed=5)
    sc.N = N
    # particles in clumps along the light axis, so that columns have runs of empty metavoxels between occupied ones
    z = sc.particles["position"][:, 2]
    sc.particles["position"][:, 2] = np.where(np.abs(z) % 9.0 < 4.5, z, z + 4.5).astype(np.float32)
    o, g = engines(sc, exact=exact)
    cnt = o.bin_counts()
    occ_per_col = (cnt > 0).sum(axis=0)
    assert occ_per_col.max() >= 3                                  # real chains
    check(o, g, exact)


def test_columns_without_any_occupied_metavoxel_keep_the_cleared_light():
    sc = S.make_scene("sparse", dims=(6, 16, 12, 64, 48), size_range=(0.2, 0.4), cubemap="r8")
    o, g = engines(sc)
    cnt = o.bin_counts()
    assert ((cnt > 0).sum(axis=0) == 0).any()                      # some columns are empty from end to end
    check(o, g, False)
    lm = g.read_lightmap()
    nv = sc.nv
    for yy, xx in zip(*np.nonzero((cnt > 0).sum(axis=0) == 0)):
        assert np.all(lm[yy * nv:(yy + 1) * nv, xx * nv:(xx + 1) * nv] == 1.0)       # GL.Clear(Color.red)  VPR.cs:499


@pytest.mark.parametrize("cubemap", ["f32", "r8"])
def test_many_refills_of_one_context(cubemap):
    sc = S.make_scene("C1", cubemap=cubemap)
    o, g = engines(sc)
    lm0 = g.read_lightmap()
    np.testing.assert_allclose(lm0, o.read_lightmap(), rtol=1e-5, atol=1e-9)
    for i in range(200):
        g.fill(sc.fill_params())
        if i % 20 == 19:
            np.testing.assert_array_equal(g.read_lightmap(), lm0)
    # another particle set through the same context (other occupancy, other chains), then back
    sc2 = S.make_scene("C1", cubemap=cubemap, seed=99)
    g.bin(sc2.particles, sc2.layout, sc2.psys_local_to_world)
    g.fill(sc2.fill_params())
    o2 = O.Oracle(sc2.config())
    o2.set_frame(sc2.light_to_world, sc2.grid_center)
    o2.bin(sc2.particles, sc2.layout, sc2.psys_local_to_world)
    o2.fill(sc2.fill_params())
    np.testing.assert_allclose(g.read_lightmap(), o2.read_lightmap(), rtol=1e-5, atol=1e-9)
    g.bin(sc.particles, sc.layout, sc.psys_local_to_world)
    g.fill(sc.fill_params())
    np.testing.assert_array_equal(g.read_lightmap(), lm0)


def test_tag_range_wraps_after_many_launches_on_a_deep_grid():
    """tag = launch number x (Nz + 1) + ordinal is 32 bits wide: on a grid 2^20 metavoxels deep the range is used up after 4 095 fills and
    the hand-off words are cleared (launch_fill_lds_variant / chain_begin).  The fill must be the same before, at and after the wrap."""
    sc = S.make_scene("deep", dims=(4, 16, 60, 32, 24), cubemap="r8", seed=3)
    sc.N = (1, 1, 1 << 20)
    sc.particles["position"][:, :2] *= 0.1                         # keep the particles inside the one column ...
    sc.particles["position"][:, 2] *= 8.0                          # ... and spread them over a few dozen metavoxels along the light axis
    o, g = engines(sc)
    cnt = o.bin_counts()
    assert (cnt > 0).sum() >= 6
    check(o, g, False)
    lm0 = g.read_lightmap()
    zz0 = int(np.nonzero(cnt[:, 0, 0])[0][-1])                     # the column's last occupied metavoxel
    b0 = g.read_brick(0, 0, zz0).copy()
    for i in range(4200):
        g.fill(sc.fill_params())
        if i in (4090, 4093, 4094, 4095, 4096, 4199):
            np.testing.assert_array_equal(g.read_lightmap(), lm0)
            np.testing.assert_array_equal(g.read_brick(0, 0, zz0), b0)


def test_watchdog_reports_a_hand_off_that_never_arrives(monkeypatch):
    """Test hook VPFX_TEST_CHAIN_TIMEOUT=1 (environment, read by vp_create only when vp_config.multi_flags carries VP_MULTI_TEST_HOOKS): units await tags nobody
    writes and give up after a few polls.  The fill then completes -- no hung GPU --, the next synchronising call returns an error, the fill
    counts as not done (a ray-march must not composite its bricks), and the context stays usable."""
    sc = S.make_scene("C1", cubemap="r8")
    monkeypatch.setenv("VPFX_TEST_CHAIN_TIMEOUT", "1")
    # without the opt-in bit in vp_config the environment is never read (a stray variable must not break a production context)
    plain = E.Engine(sc.config())
    plain.set_frame(sc.light_to_world, sc.grid_center)
    plain.bin(sc.particles, sc.layout, sc.psys_local_to_world)
    plain.fill(sc.fill_params())
    plain.sync()
    plain.close()
    cfg = sc.config()
    cfg.multi_flags = abi.VP_MULTI_TEST_HOOKS
    g = E.Engine(cfg)
    monkeypatch.delenv("VPFX_TEST_CHAIN_TIMEOUT")
    g.set_frame(sc.light_to_world, sc.grid_center)
    g.bin(sc.particles, sc.layout, sc.psys_local_to_world)
    g.fill(sc.fill_params())
    with pytest.raises(Exception, match="hand-off"):
        g.sync()
    g.sync()                                                       # the flag is consumed: the context keeps working
    assert g.stats()["occupied_mv"] > 0
    with pytest.raises(E.VpfxError) as ei:                         # ... but the timed-out fill is invalid: no ray-march on it
        g.raymarch(sc.camera(), sc.raymarch_params())
    assert ei.value.code == abi.VP_ERR_STATE
    # the device-async entry point never syncs: it must see the flag too
    import torch
    g.fill(sc.fill_params())
    torch.cuda.synchronize()
    out = torch.empty((sc.height, sc.width, 4), device="cuda")
    with pytest.raises(Exception, match="hand-off"):
        g.raymarch_device(sc.camera(), sc.raymarch_params(), out.data_ptr())
    # reserved switches outside the documented values are refused (an uninitialised struct must not silently change behaviour)
    cfg = sc.config()
    cfg.reserved[2] = 1
    with pytest.raises(E.VpfxError) as ei:
        E.Engine(cfg)
    assert ei.value.code == abi.VP_ERR_BAD_ARG
