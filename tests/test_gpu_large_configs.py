"""GPU: the nv = 64 extension at scale -- BASELINE config 5 at FULL size on one GPU (64^3 metavoxels x 64^3 voxels, 1M particles,
3840x2160; ~168 GB of bricks in 288 GB of HBM) through size-independent properties plus an oracle comparison of one complete
metavoxel column, and a full oracle comparison of nv = 64 at a size the oracle covers (8^3 x 64^3)."""
import gc

import numpy as np
import pytest
import torch

from vpfx_amd import abi, engine as E, scene as S
from oracle import oracle as O

pytestmark = pytest.mark.gpu


def test_nv64_full_parity_at_8_cubed():
    """64^3-voxel bricks (beyond the reference's NUM_VOXELS 32) on an 8^3 grid: bins, every brick bit-identical
    (EXACT), light map, per-pixel RGBA and the sample count against the oracle."""
    sc = S.make_scene("n64", dims=(8, 64, 1000, 256, 256))
    o, g = O.Oracle(sc.config()), E.Engine(sc.config(), exact=True, early_out=False)
    for x in (o, g):
        x.set_frame(sc.light_to_world, sc.grid_center)
        x.bin(sc.particles, sc.layout, sc.psys_local_to_world)
        x.fill(sc.fill_params())
    cnt = o.bin_counts()
    np.testing.assert_array_equal(cnt, g.bin_counts())
    occ = list(zip(*np.nonzero(cnt)))
    assert len(occ) > 250                                               # C1-like occupancy (the bordered box differs with nv)
    for zz, yy, xx in occ:
        assert np.array_equal(o.read_brick(xx, yy, zz).view(np.uint16), g.read_brick(xx, yy, zz).view(np.uint16)), (xx, yy, zz)
    np.testing.assert_allclose(g.read_lightmap(), o.read_lightmap(), rtol=1e-5, atol=1e-9)
    io, ig = o.raymarch(sc.camera(), sc.raymarch_params()), g.raymarch(sc.camera(), sc.raymarch_params())
    assert np.abs(io - ig).max() <= 1e-3
    assert o.stats()["samples"] == g.stats()["samples"]
    # default (fast) math: <= 1 fp16 ulp
    g2 = E.Engine(sc.config())
    g2.set_frame(sc.light_to_world, sc.grid_center)
    g2.bin(sc.particles, sc.layout, sc.psys_local_to_world)
    g2.fill(sc.fill_params())
    for zz, yy, xx in occ[::7]:
        a, b = o.read_brick(xx, yy, zz).view(np.uint16).astype(np.int32), g2.read_brick(xx, yy, zz).view(np.uint16).astype(np.int32)
        assert np.abs(a - b).max() <= 1


def test_config5_full_size_on_one_gpu():
    """BASELINE.json configs[4] at full size.  The oracle cannot run 21 G voxels, so: (1) the workload statistics of SURVEY
    App. C, (2) range / finiteness properties, (3) bit-identical repeat, (4) ONE complete metavoxel column (64 slices deep)
    compared bit for bit with the oracle, which is fed exactly the particles that touch that column."""
    gc.collect()
    torch.cuda.empty_cache()
    free, total = torch.cuda.mem_get_info()
    if free < 200e9:
        # a 288-GB MI355X with < 200 GB free means something of THIS process (or a stale one) still holds memory: a bug, not an excuse -- fail.
        # Only a genuinely smaller device may skip.
        assert total < 250e9, f"config 5 needs ~190 GB of free HBM; only {free / 1e9:.0f} of {total / 1e9:.0f} GB are free on a device that should hold it"
        pytest.skip(f"device has {total / 1e9:.0f} GB in total: the 64^3 x 64^3 brick pool (~190 GB) cannot fit")
    sc = S.make_scene("C5")
    g = E.Engine(sc.config(), exact=True)
    g.set_frame(sc.light_to_world, sc.grid_center)
    g.bin(sc.particles, sc.layout, sc.psys_local_to_world)
    st = g.stats()
    assert st["occupied_mv"] == 80226
    assert abs(st["pairs"] - 4618872) <= 64 and 100 <= st["max_pairs_per_mv"] <= 110   # float64 probe vs fp32 binning: a few pairs
    g.fill(sc.fill_params())
    lm = g.read_lightmap()
    assert lm.shape == (4096, 4096) and lm.min() >= 0.0 and lm.max() <= 1.0 and np.isfinite(lm).all()
    img = g.raymarch(sc.camera(), sc.raymarch_params())
    assert img.shape == (2160, 3840, 4) and np.isfinite(img).all()
    assert img[..., 3].min() >= 0.0 and img[..., 3].max() <= 1.0 + 1e-6 and img[..., 3].mean() > 0.3
    s1 = g.stats()
    assert s1["samples"] > 5e8 and s1["brick_bytes"] >= 80226 * 64 ** 3 * 8
    # (3) determinism at full size
    cnt = g.bin_counts()
    col_pairs = cnt.sum(axis=0)
    yy, xx = np.unravel_index(np.argmax(col_pairs), col_pairs.shape)     # the heaviest metavoxel column
    zs = [int(z) for z in np.nonzero(cnt[:, yy, xx])[0]]
    bricks = {z: g.read_brick(xx, yy, z).copy() for z in zs[:: max(1, len(zs) // 6)]}
    g.fill(sc.fill_params())
    np.testing.assert_array_equal(g.read_lightmap(), lm)
    for z, b in bricks.items():
        assert np.array_equal(g.read_brick(xx, yy, z).view(np.uint16), b.view(np.uint16))
    np.testing.assert_array_equal(g.raymarch(sc.camera(), sc.raymarch_params()), img)
    # (4) the whole column (xx, yy, *) against the oracle: hand it the union of the column's particle lists, in particle order
    ids = np.unique(np.concatenate([g.bin_list(xx, yy, z) for z in zs]))
    sub = S.make_scene("C5")
    sub.particles = np.ascontiguousarray(sc.particles[ids])
    o = O.Oracle(sub.config())
    o.set_frame(sub.light_to_world, sub.grid_center)
    o.bin(sub.particles, sub.layout, sub.psys_local_to_world)
    np.testing.assert_array_equal(o.bin_counts()[:, yy, xx], cnt[:, yy, xx])
    o.fill(sub.fill_params())
    for z in zs:
        a, b = o.read_brick(xx, yy, z).view(np.uint16), g.read_brick(xx, yy, z).view(np.uint16)
        assert np.array_equal(a, b), (xx, yy, z)
    nv = 64
    np.testing.assert_allclose(lm[yy * nv:(yy + 1) * nv, xx * nv:(xx + 1) * nv],
                               o.read_lightmap()[yy * nv:(yy + 1) * nv, xx * nv:(xx + 1) * nv], rtol=1e-5, atol=1e-12)
    g.close()
    o.close()
    gc.collect()
    torch.cuda.empty_cache()


def _fp16_ulp_diff_max(a, b):
    """Largest distance in fp16 code points between two uint16 arrays of NON-NEGATIVE halves (positive halves are ordered like their codes)."""
    return int(np.abs(a.view(np.int16).astype(np.int32) - b.view(np.int16).astype(np.int32)).max())


def test_config5_through_the_fanout_slab_by_slab():
    """BASELINE.json configs[4] -- the only config assigned to 8 GPUs -- SHARDED, on one GPU: the eight light-axis slab
    contexts the library's planner cuts are created ONE AT A TIME (each ~1/8 of the 190 GB brick pool) next to the resident single-context C5,
    and driven through the fan-out's building blocks in rank order: vp_fill_local -> tau (carried on the HOST between contexts, as the
    all-gather would carry it between GPUs) -> vp_fill_finish_gathered -> vp_raymarch_partial_device.  Checked against the single context:
      (i)  EVERY brick of two layers per slab (its first and its last): densities bit-identical everywhere; lit colours bit-identical in the
           slab nearest the light and within 1 fp16 ulp behind it (T_in is a product of slab transmittances: fp32 reassociation; Fill.shader:224,250
           carries the same light through the UAV slice by slice); the final light map (the last slab's);
      (ii) the ORDERED blend of the partial images (VPR.cs:652-711 at slab granularity) against the single-context 4K frame (<= 2e-5), for the
           benchmark camera (all UNDER) and for a camera whose zBoundary falls inside the grid (OVER and UNDER phases, one straddling slab);
      and the shards PARTITION the work: occupied metavoxels, pairs and executed lattice samples add up to the single context's."""
    gc.collect()
    torch.cuda.empty_cache()
    free, total = torch.cuda.mem_get_info()
    if free < 240e9:
        assert total < 250e9, f"needs ~190 GB (whole grid) + ~30 GB (one slab) of free HBM; only {free / 1e9:.0f} of {total / 1e9:.0f} GB are free"
        pytest.skip(f"device has {total / 1e9:.0f} GB in total")
    dev = torch.device("cuda", 0)
    world = 8
    sc = S.make_scene("C5", cubemap="r8")                          # the benchmark's texel format: k_fill_lds<64>
    cams = [("benchmark", None), ("straddling", (24.0, 240.0, 16.0))]
    single = E.Engine(sc.config(), exact=True, early_out=False)
    single.set_frame(sc.light_to_world, sc.grid_center)
    single.upload_particles(sc.particles, sc.layout, sc.psys_local_to_world)
    hist = [float(x) for x in single.z_histogram()]
    single.bin_resident()
    single.fill(sc.fill_params())
    cnt = single.bin_counts()
    st1 = single.stats()
    assert st1["occupied_mv"] == 80226
    lm_single = single.read_lightmap()
    rp = sc.raymarch_params()
    frames, samples1, zbs, cam_structs = {}, {}, {}, {}
    for name, pos in cams:
        if pos is not None:
            sc.set_camera(pos)
        cam_structs[name] = sc.camera()
        img = torch.empty((sc.height, sc.width, 4), device=dev)
        single.raymarch_device(cam_structs[name], rp, img.data_ptr())
        single.sync()
        frames[name], samples1[name], zbs[name] = img, single.stats()["samples"], single.z_boundary(cam_structs[name])
    assert zbs["benchmark"] == -1 and 0 <= zbs["straddling"] < sc.N[2] - 1, zbs
    # the cut a fan-out context makes at its first vp_bin: the library's planner on the pair histogram
    bounds = E.plan_slabs(sc.N[2], world, fill_ms=hist, rm_ms=None)
    assert bounds[0][0] == 0 and bounds[-1][1] == sc.N[2] and all(z1 > z0 for z0, z1 in bounds) and all(bounds[i][1] == bounds[i + 1][0] for i in range(world - 1))
    lm_shape = (sc.N[1] * sc.nv, sc.N[0] * sc.nv)
    taus_host = []                                                   # what travels between the ranks: 64 MiB per slab
    parts = {name: {} for name, _ in cams}
    occupied = pairs = 0
    samples = {name: 0 for name, _ in cams}
    bricks_checked = worst_ulp = 0
    lm_last = None
    for r, (z0, z1) in enumerate(bounds):
        e = E.Engine(sc.config(device=0, slab=(z0, z1)), exact=True, early_out=False)
        e.set_frame(sc.light_to_world, sc.grid_center)
        e.upload_particles(sc.particles, sc.layout, sc.psys_local_to_world)
        e.bin_resident()
        tau = torch.empty(lm_shape, device=dev)
        e.fill_local(sc.fill_params(), tau.data_ptr())
        e.sync()
        taus_host.append(tau.cpu())
        tau_all = torch.ones((world,) + lm_shape, device=dev)      # slabs behind this one: never read by rank r's finish pass
        for j, t in enumerate(taus_host):
            tau_all[j].copy_(t)
        e.fill_finish_gathered(tau_all.data_ptr(), r, world)
        st = e.stats()
        occupied += st["occupied_mv"]; pairs += st["pairs"]
        assert st["brick_bytes"] < 0.25 * st1["brick_bytes"], "a slab context must hold its slab's bricks only"
        # (i) every brick of the slab's first and last layer
        for zz in sorted({z0, z1 - 1}):
            ys, xs = np.nonzero(cnt[zz])
            for yy, xx in zip(ys, xs):
                a, b = single.read_brick(xx, yy, zz).view(np.uint16), e.read_brick(xx, yy, zz).view(np.uint16)
                bricks_checked += 1
                if np.array_equal(a, b):
                    continue
                assert r > 0, f"slab 0 brick {(xx, yy, zz)} differs from the single context's"
                assert np.array_equal(a[..., 3], b[..., 3]), f"density of brick {(xx, yy, zz)} differs"
                u = _fp16_ulp_diff_max(a[..., :3], b[..., :3])
                worst_ulp = max(worst_ulp, u)
                assert u <= 1, f"brick {(xx, yy, zz)}: {u} fp16 ulps"
        # (ii) the slab's partial images for both cameras
        for name, _ in cams:
            over, under = torch.empty((sc.height, sc.width, 4), device=dev), torch.empty((sc.height, sc.width, 4), device=dev)
            e.raymarch_partial_device(cam_structs[name], rp, over.data_ptr(), under.data_ptr())
            e.sync()
            samples[name] += e.stats()["samples"]
            parts[name][r] = (over, under)
        if r == world - 1:
            lm_last = e.read_lightmap()
        e.close()
        del tau_all, tau
        torch.cuda.empty_cache()
    assert occupied == st1["occupied_mv"] and pairs == st1["pairs"]
    assert bricks_checked > 8000, bricks_checked
    np.testing.assert_allclose(lm_last, lm_single, rtol=2e-5, atol=1e-9)
    blender = E.Engine(S.make_scene("C5").config(device=0, slab=(0, 1)))      # any context of this screen size blends (no bricks needed)
    for name, _ in cams:
        assert samples[name] == samples1[name], (name, samples[name], samples1[name])      # every lattice sample is executed by exactly one slab
        chain, plan, straddler = E.blend_plan(bounds, zbs[name])
        assert (straddler is None) == (name == "benchmark")
        imgs = [parts[name][rk][kind] for rk, _, kind in plan]                 # kind 0: the slab's OVER image, 1: its UNDER image
        out = torch.empty((sc.height, sc.width, 4), device=dev)
        blender.blend_partials_device([t.data_ptr() for t in imgs], [k for _, _, k in plan], out.data_ptr())
        blender.sync()
        err = float((out - frames[name]).abs().max().item())
        assert err <= 2e-5, (name, err)
        assert float(frames[name][..., 3].mean().item()) > (0.3 if name == "benchmark" else 0.05)      # the frames are not empty
    print(f"[C5 slab by slab] cut {bounds}; bricks compared {bricks_checked}, worst {worst_ulp} fp16 ulp; samples {samples}")
    blender.close()
    single.close()
    gc.collect()
    torch.cuda.empty_cache()
