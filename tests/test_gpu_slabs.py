"""Slab-sharded path on ONE GPU: K engines, each owning a light-axis slab, driven through the same
SlabPipeline collectives (replaced here by in-process exchanges) must reproduce the single-engine result.

Checks the split fill (vp_fill_local / vp_fill_finish), the partial ray-march and the ordered blend kernels."""
import numpy as np
import pytest
import torch

from vpfx_amd import engine as E, scene as S
import slab_reference as PAR
from oracle import oracle as O

pytestmark = pytest.mark.gpu


def run_slabs(sc, world, cam_pos=None, weights=None, gathered=True, **engine_kw):
    dev = torch.device("cuda", 0)
    if cam_pos is not None:
        sc.set_camera(cam_pos)
    bounds = PAR.slab_bounds(sc.N[2], world, weights)
    engs = []
    for r in range(world):
        e = E.Engine(sc.config(device=0, slab=bounds[r]), **engine_kw)
        e.set_frame(sc.light_to_world, sc.grid_center)
        e.upload_particles(sc.particles, sc.layout, sc.psys_local_to_world)
        engs.append(PAR.HipSlabEngine(e, dev))
    # fill: local -> "all_gather" -> finish
    taus = []
    for h in engs:
        h.bin_resident()
        taus.append(h.fill_local(sc.fill_params()).clone())
    tau_all = torch.stack(taus).contiguous()           # what the all-gather delivers: [world, LH, LW]
    for r, h in enumerate(engs):
        if gathered:                                   # product of the nearer slabs' maps formed inside the finish kernel
            h.fill_finish_gathered(tau_all, r, world)
        else:
            t_in = None
            for j in range(r):
                t_in = taus[j].clone() if t_in is None else t_in.mul_(taus[j])
            h.fill_finish(t_in)
    cam, rp = sc.camera(), sc.raymarch_params()
    zb = engs[0].z_boundary(cam)
    plan, straddler = PAR.blend_plan(bounds, zb)
    parts = {}
    for r, h in enumerate(engs):
        over, under = h.raymarch_partial(cam, rp)
        parts[(r, "over")] = over.clone()
        parts[(r, "under")] = under.clone()
    imgs = [parts[(r, which)] for r, which, _ in plan]
    kinds = [k for _, _, k in plan]
    out = engs[0].blend(imgs, kinds).cpu().numpy()
    # the sharded form of the same blend (vp_blend_partials_range_device): pieces of 1/world of the pixels, then reassembled
    npix = sc.width * sc.height
    piece = -(-npix // world)
    flat = [torch.cat([t.reshape(npix, 4), torch.zeros((piece * world - npix, 4), device=dev)]) for t in imgs]
    pieces = [engs[r % len(engs)].blend([f[r * piece:(r + 1) * piece] for f in flat], kinds).clone() for r in range(world)]
    sharded = torch.cat(pieces)[:npix].reshape(sc.height, sc.width, 4).cpu().numpy()
    assert np.array_equal(sharded, out)
    lightmap = engs[-1].e.read_lightmap()
    return out, lightmap, bounds, zb, straddler, engs


@pytest.mark.parametrize("world", [2, 3])
def test_slabs_match_single_under_phase(world):
    sc = S.make_scene("C1")
    single = E.Engine(sc.config())
    single.set_frame(sc.light_to_world, sc.grid_center)
    single.bin(sc.particles, sc.layout, sc.psys_local_to_world)
    single.fill(sc.fill_params())
    ref = single.raymarch(sc.camera(), sc.raymarch_params())
    out, lm, bounds, zb, straddler, engs = run_slabs(sc, world)
    assert zb == -1 and straddler is None
    assert np.abs(out - ref).max() <= 2e-5
    np.testing.assert_allclose(lm, single.read_lightmap(), rtol=2e-5, atol=1e-9)
    # bricks of a far slab: identical to the single-GPU bricks within 1 fp16 ulp (product reassociation of T_in)
    z0, z1 = bounds[-1]
    cnt = single.bin_counts()
    zz, yy, xx = np.nonzero(cnt[z0:z1])
    for i in range(0, len(zz), max(1, len(zz) // 10)):
        a = single.read_brick(xx[i], yy[i], zz[i] + z0).view(np.uint16).astype(np.int32)
        b = engs[-1].e.read_brick(xx[i], yy[i], zz[i] + z0).view(np.uint16).astype(np.int32)
        assert np.abs(a - b).max() <= 1


@pytest.mark.parametrize("world,cam", [(2, (3.0, 30.0, 2.0)), (4, (3.0, 30.0, 2.0)), (3, (-2.0, -4.0, 1.0))])
def test_slabs_match_oracle_with_over_phase(world, cam):
    """Camera placed so that zBoundary falls inside the grid: OVER and UNDER phases, one straddling slab."""
    sc = S.make_scene("C1")
    sc.set_camera(cam)
    o = O.Oracle(sc.config())
    o.set_frame(sc.light_to_world, sc.grid_center)
    o.bin(sc.particles, sc.layout, sc.psys_local_to_world)
    o.fill(sc.fill_params())
    ref = o.raymarch(sc.camera(), sc.raymarch_params())
    out, lm, bounds, zb, straddler, _ = run_slabs(sc, world, cam_pos=cam)
    assert 0 <= zb < sc.N[2] - 1, zb
    assert np.abs(out - ref).max() <= 1e-3
    # and the single-engine HIP path handles the mixed phases too
    single = E.Engine(sc.config())
    single.set_frame(sc.light_to_world, sc.grid_center)
    single.bin(sc.particles, sc.layout, sc.psys_local_to_world)
    single.fill(sc.fill_params())
    assert np.abs(single.raymarch(sc.camera(), sc.raymarch_params()) - ref).max() <= 1e-3


def test_weighted_slab_bounds_cover_grid():
    sc = S.make_scene("T0")
    b = PAR.slab_bounds(sc.N[2], 3, weights=[0, 10, 10, 0])
    assert b[0][0] == 0 and b[-1][1] == sc.N[2] and all(z1 > z0 for z0, z1 in b)
    assert all(b[i][1] == b[i + 1][0] for i in range(len(b) - 1))


def test_z_histogram_matches_bin_counts_and_balances():
    sc = S.make_scene("C1")
    e = E.Engine(sc.config())
    e.set_frame(sc.light_to_world, sc.grid_center)
    e.upload_particles(sc.particles, sc.layout, sc.psys_local_to_world)
    h = e.z_histogram()
    e.bin_resident()
    np.testing.assert_array_equal(h, e.bin_counts().sum(axis=(1, 2)))
    b = PAR.slab_bounds(sc.N[2], 4, [float(x) for x in h])
    loads = [h[z0:z1].sum() for z0, z1 in b]
    assert max(loads) <= 0.45 * h.sum()
    out, lm, bounds, zb, straddler, _ = run_slabs(sc, 4, weights=[float(x) for x in h])
    single = E.Engine(sc.config())
    single.set_frame(sc.light_to_world, sc.grid_center)
    single.bin(sc.particles, sc.layout, sc.psys_local_to_world)
    single.fill(sc.fill_params())
    assert np.abs(out - single.raymarch(sc.camera(), sc.raymarch_params())).max() <= 2e-5


# ---- BASELINE config 4: the C3 workload (32^3 x 32^3, 100k particles, 1920x1080) in 2 / 4 / 8 light-axis slabs ------------------
@pytest.fixture(scope="module")
def c3_reference():
    """Single-engine C3 frame (no early-out: every lattice sample) + the oracle's frame, computed once for the three shard counts."""
    sc = S.make_scene("C3")
    g = E.Engine(sc.config(), early_out=False)
    g.set_frame(sc.light_to_world, sc.grid_center)
    g.upload_particles(sc.particles, sc.layout, sc.psys_local_to_world)
    hist = [float(x) for x in g.z_histogram()]
    g.bin_resident()
    g.fill(sc.fill_params())
    frame = g.raymarch(sc.camera(), sc.raymarch_params())
    st = g.stats()
    lm = g.read_lightmap()
    g.close()
    o = O.Oracle(sc.config())
    o.set_frame(sc.light_to_world, sc.grid_center)
    o.bin(sc.particles, sc.layout, sc.psys_local_to_world)
    o.fill(sc.fill_params())
    oracle_frame = o.raymarch(sc.camera(), sc.raymarch_params())
    oracle_samples = o.stats()["samples"]
    o.close()
    torch.cuda.empty_cache()
    return dict(frame=frame, lightmap=lm, stats=st, hist=hist, oracle_frame=oracle_frame, oracle_samples=oracle_samples)


@pytest.mark.parametrize("world", [2, 4, 8])
def test_config4_c3_sharded_matches_single_engine_and_oracle(world, c3_reference):
    """BASELINE.json configs[3]: pair-balanced slabs as bench.py --gpus N cuts them, K engines on one GPU."""
    ref = c3_reference
    sc = S.make_scene("C3")
    out, lm, bounds, zb, straddler, engs = run_slabs(sc, world, weights=ref["hist"], early_out=False)
    assert bounds[0][0] == 0 and bounds[-1][1] == 32 and all(z1 > z0 for z0, z1 in bounds)
    loads = [sum(ref["hist"][z0:z1]) for z0, z1 in bounds]
    assert max(loads) <= 1.25 * sum(loads) / world                      # the fill work is balanced (optimal contiguous cut)
    assert np.abs(out - ref["frame"]).max() <= 2e-5                     # vs the single-engine frame
    assert np.abs(out - ref["oracle_frame"]).max() <= 1e-3              # vs the oracle (the north_star gate)
    np.testing.assert_allclose(lm, ref["lightmap"], rtol=2e-5, atol=1e-9)
    stats = [h.e.stats() for h in engs]
    assert sum(s["occupied_mv"] for s in stats) == ref["stats"]["occupied_mv"] == 11325
    assert sum(s["pairs"] for s in stats) == ref["stats"]["pairs"] == 480441
    # every lattice sample is executed by exactly one slab: the shards partition the single-GPU work
    assert sum(s["samples"] for s in stats) == ref["stats"]["samples"] == ref["oracle_samples"]
    for h in engs:
        h.e.close()
    torch.cuda.empty_cache()


def test_fused_tau_product_is_bit_identical_to_the_explicit_product():
    sc = S.make_scene("C1")
    a = run_slabs(sc, 4, gathered=True)
    b = run_slabs(S.make_scene("C1"), 4, gathered=False)
    np.testing.assert_array_equal(a[0], b[0])
    np.testing.assert_array_equal(a[1], b[1])
