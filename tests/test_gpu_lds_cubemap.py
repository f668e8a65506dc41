"""GPU: the LDS-resident R8 cube-map fill (k_fill_lds, persistent workgroups) against the oracle and against the global-table
kernel fed the same bytes: every brick size, S = 128 (immediate row pitch) and other sizes (generic pitch), the split
(multi-GPU) fill, and the benchmark configs."""
import numpy as np
import pytest
import torch

from vpfx_amd import abi, engine as E, scene as S
import slab_reference as PAR
from oracle import oracle as O

pytestmark = pytest.mark.gpu


def fill_both(sc, **kw):
    o, g = O.Oracle(sc.config()), E.Engine(sc.config(), **kw)
    for x in (o, g):
        x.set_frame(sc.light_to_world, sc.grid_center)
        x.bin(sc.particles, sc.layout, sc.psys_local_to_world)
        x.fill(sc.fill_params())
    return o, g


def brick_ulp_diff(a_eng, b_eng, cnt, step=1):
    worst = 0
    for zz, yy, xx in list(zip(*np.nonzero(cnt)))[::step]:
        a = a_eng.read_brick(xx, yy, zz).view(np.uint16).astype(np.int32)
        b = b_eng.read_brick(xx, yy, zz).view(np.uint16).astype(np.int32)
        worst = max(worst, int(np.abs(a - b).max()))
    return worst


@pytest.mark.parametrize("dims,size", [((8, 16, 1000, 256, 256), 128), ((6, 32, 400, 160, 120), 128), ((3, 64, 40, 96, 64), 128),
                                        ((6, 32, 400, 160, 120), 64), ((4, 16, 150, 96, 64), 37), ((4, 32, 150, 96, 64), 160)])
def test_lds_fill_matches_the_oracle(dims, size):
    sc = S.make_scene("r8", dims=dims)
    sc.cubemap = S.make_cubemap_r8(size)
    o, g = fill_both(sc)                                   # default math -> the LDS kernel
    cnt = o.bin_counts()
    np.testing.assert_array_equal(cnt, g.bin_counts())
    assert brick_ulp_diff(o, g, cnt) <= 1                  # <= 1 fp16 ulp, like the global-table default path
    np.testing.assert_allclose(g.read_lightmap(), o.read_lightmap(), rtol=1e-5, atol=1e-9)
    io, ig = o.raymarch(sc.camera(), sc.raymarch_params()), g.raymarch(sc.camera(), sc.raymarch_params())
    assert np.abs(io - ig).max() <= 1e-3


def test_lds_fill_is_deterministic_and_close_to_the_global_table_kernel():
    sc = S.make_scene("C2", cubemap="r8")
    g = E.Engine(sc.config())
    cfg = sc.config()
    cfg.reserved[0] = 1                                    # VPFX_CFG_NO_LDS_CUBEMAP: same bytes through the f32 footprint table
    h = E.Engine(cfg)
    for x in (g, h):
        x.set_frame(sc.light_to_world, sc.grid_center)
        x.bin(sc.particles, sc.layout, sc.psys_local_to_world)
        x.fill(sc.fill_params())
    cnt = g.bin_counts()
    assert brick_ulp_diff(g, h, cnt, step=7) <= 1
    np.testing.assert_allclose(g.read_lightmap(), h.read_lightmap(), rtol=2e-5, atol=1e-9)   # (D/255) * bytes vs D * (bytes/255): last-bit differences per voxel
    lm = g.read_lightmap()
    b0 = {k: g.read_brick(k[2], k[1], k[0]).copy() for k in list(zip(*np.nonzero(cnt)))[::97]}
    for _ in range(3):                                     # dynamic tile scheduling must not change a bit
        g.fill(sc.fill_params())
        np.testing.assert_array_equal(g.read_lightmap(), lm)
        for k, b in b0.items():
            assert np.array_equal(g.read_brick(k[2], k[1], k[0]).view(np.uint16), b.view(np.uint16))


def test_lds_fill_too_large_a_cubemap_falls_back_to_the_global_table():
    sc = S.make_scene("T0")
    sc.cubemap = S.make_cubemap_r8(200)                    # 6 * 202^2 = 245 KB > LDS
    o, g = fill_both(sc)
    assert brick_ulp_diff(o, g, o.bin_counts()) <= 1


@pytest.mark.parametrize("world", [2, 3])
def test_lds_split_fill_matches_single_engine(world):
    """MODE 1 of the LDS kernel (slab-local pass of the multi-GPU fill) + vp_fill_finish."""
    from test_gpu_slabs import run_slabs
    sc = S.make_scene("C1", cubemap="r8")
    single = E.Engine(sc.config())
    single.set_frame(sc.light_to_world, sc.grid_center)
    single.bin(sc.particles, sc.layout, sc.psys_local_to_world)
    single.fill(sc.fill_params())
    ref = single.raymarch(sc.camera(), sc.raymarch_params())
    out, lm, bounds, zb, straddler, engs = run_slabs(sc, world)
    assert np.abs(out - ref).max() <= 2e-5
    np.testing.assert_allclose(lm, single.read_lightmap(), rtol=2e-5, atol=1e-9)
    o = O.Oracle(sc.config())
    o.set_frame(sc.light_to_world, sc.grid_center)
    o.bin(sc.particles, sc.layout, sc.psys_local_to_world)
    o.fill(sc.fill_params())
    assert np.abs(out - o.raymarch(sc.camera(), sc.raymarch_params())).max() <= 1e-3


def test_config3_r8_benchmark_workload_against_the_oracle():
    """The bench's default input (C3 with the 8-bit cube map): light map and frame against the oracle, sample counts equal."""
    sc = S.make_scene("C3", cubemap="r8")
    g = E.Engine(sc.config(), early_out=False)
    g.set_frame(sc.light_to_world, sc.grid_center)
    g.bin(sc.particles, sc.layout, sc.psys_local_to_world)
    g.fill(sc.fill_params())
    ig = g.raymarch(sc.camera(), sc.raymarch_params())
    o = O.Oracle(sc.config())
    o.set_frame(sc.light_to_world, sc.grid_center)
    o.bin(sc.particles, sc.layout, sc.psys_local_to_world)
    o.fill(sc.fill_params())
    io = o.raymarch(sc.camera(), sc.raymarch_params())
    np.testing.assert_allclose(g.read_lightmap(), o.read_lightmap(), rtol=1e-5, atol=1e-9)
    assert np.abs(io - ig).max() <= 1e-3
    assert o.stats()["samples"] == g.stats()["samples"]
    cnt = o.bin_counts()
    assert brick_ulp_diff(o, g, cnt, step=211) <= 1
    # EVERY brick of one whole light-axis layer (the densest one), not a sample of the grid: <= 1 fp16 ulp
    zmid = int(np.argmax(cnt.reshape(cnt.shape[0], -1).sum(axis=1)))
    layer = np.zeros_like(cnt)
    layer[zmid] = cnt[zmid]
    assert int((layer != 0).sum()) > 300
    assert brick_ulp_diff(o, g, layer, step=1) <= 1
    # ... and the benchmark's own ray-march -- the DEFAULT saturation early-out -- per pixel against the same oracle frame:
    # it skips exact no-ops only, so the image is the oracle's and the samples are a subset
    ge = E.Engine(sc.config())
    ge.set_frame(sc.light_to_world, sc.grid_center)
    ge.bin(sc.particles, sc.layout, sc.psys_local_to_world)
    ge.fill(sc.fill_params())
    ie = ge.raymarch(sc.camera(), sc.raymarch_params())
    assert np.abs(io - ie).max() <= 1e-3 and np.abs(ig - ie).max() <= 2e-6
    se = ge.stats()
    assert se["samples"] < 0.6 * o.stats()["samples"] and se["brick_format"] == 1 and se["bricks_sampled"] < se["occupied_mv"]
    ge.close()
