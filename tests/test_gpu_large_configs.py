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
