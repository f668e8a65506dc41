"""CPU: slab partitioning and the blend plan of the multi-GPU pipeline (pure host logic)."""
import pytest

import slab_reference as PAR


@pytest.mark.parametrize("nz,world", [(32, 1), (32, 2), (32, 4), (32, 8), (8, 8), (10, 3), (64, 8)])
def test_uniform_slabs_partition_the_axis(nz, world):
    b = PAR.slab_bounds(nz, world)
    assert len(b) == world and b[0][0] == 0 and b[-1][1] == nz
    assert all(z1 > z0 for z0, z1 in b)
    assert all(b[i][1] == b[i + 1][0] for i in range(world - 1))
    assert max(z1 - z0 for z0, z1 in b) - min(z1 - z0 for z0, z1 in b) <= 1


def test_too_many_ranks():
    with pytest.raises(ValueError):
        PAR.slab_bounds(4, 5)


def test_weighted_slabs_balance_work():
    w = [0, 0, 1, 5, 20, 40, 40, 20, 5, 1, 0, 0]
    b = PAR.slab_bounds(len(w), 4, w)
    assert b[0][0] == 0 and b[-1][1] == len(w) and all(z1 > z0 for z0, z1 in b)
    loads = [sum(w[z0:z1]) for z0, z1 in b]
    assert max(loads) <= 0.5 * sum(w)


@pytest.mark.parametrize("zb", [-1, 0, 3, 7, 15])
def test_blend_plan_orders_over_then_under(zb):
    bounds = PAR.slab_bounds(16, 4)
    plan, straddler = PAR.blend_plan(bounds, zb)
    kinds = [k for _, _, k in plan]
    assert kinds == sorted(kinds)                                   # every OVER before every UNDER
    overs = [r for r, w, k in plan if k == 0]
    unders = [r for r, w, k in plan if k == 1]
    assert overs == sorted(overs) and unders == sorted(unders)      # zz ascending inside each phase
    for r, (z0, z1) in enumerate(bounds):
        assert (r in overs) == (z0 <= zb)
        assert (r in unders) == (z1 - 1 > zb)
    if straddler is not None:
        z0, z1 = bounds[straddler]
        assert z0 <= zb < z1 - 1 and straddler in overs and straddler in unders
    else:
        assert not (set(overs) & set(unders))


def _check_partition(b, nz, world):
    assert len(b) == world and b[0][0] == 0 and b[-1][1] == nz
    assert all(z1 > z0 for z0, z1 in b), b
    assert all(b[i][1] == b[i + 1][0] for i in range(world - 1)), b


def test_weighted_slabs_are_always_a_valid_partition():
    """Property test: random, one-hot, tail-heavy and head-heavy histograms, world up to nz (the round-1 repair pass pushed
    cuts past nz: nz=8, world=4, all the work in slice 6 gave [(0,7),(7,8),(8,9),(9,8)])."""
    import random
    rng = random.Random(99)
    _check_partition(PAR.slab_bounds(8, 4, [0, 0, 0, 0, 0, 0, 19250, 0]), 8, 4)
    for nz in (1, 2, 3, 4, 8, 16, 32, 64):
        for world in sorted({1, 2, 3, 4, 8, nz // 2, nz - 1, nz}):
            if world < 1 or world > nz:
                continue
            cases = [[0.0] * nz, [1.0] * nz, [float(i) for i in range(nz)], [float(nz - i) for i in range(nz)]]
            for hot in range(nz):
                cases.append([1e6 if i == hot else 0.0 for i in range(nz)])
            for _ in range(40):
                cases.append([rng.choice([0.0, 0.0, rng.random(), 1000.0 * rng.random()]) for _ in range(nz)])
            for w in cases:
                _check_partition(PAR.slab_bounds(nz, world, w), nz, world)
    with pytest.raises(ValueError):
        PAR.slab_bounds(8, 2, [1.0] * 7)


def test_weighted_slabs_do_not_lose_balance_on_the_c3_histogram():
    # ball-shaped cloud: pairs per z-slice of the 32^3 benchmark grid (rounded), 2/4/8 ranks
    import math
    w = [max(0.0, 1.0 - ((z - 15.5) / 13.5) ** 2) * 25000 for z in range(32)]
    for world in (2, 4, 8):
        b = PAR.slab_bounds(32, world, w)
        _check_partition(b, 32, world)
        loads = [sum(w[z0:z1]) for z0, z1 in b]
        assert max(loads) <= 1.2 * sum(w) / world, (world, b, loads)       # the DP cut is optimal: 1.0 / 1.06 / 1.16 here


def test_weighted_slabs_minimise_the_heaviest_slab():
    """Brute force over every contiguous partition on small cases: slab_bounds returns an optimum."""
    import itertools, random
    rng = random.Random(5)
    for _ in range(60):
        nz, world = rng.randint(2, 9), rng.randint(1, 4)
        if world > nz:
            continue
        w = [rng.choice([0.0, rng.random(), 10 * rng.random()]) for _ in range(nz)]
        if sum(w) == 0:
            continue
        best = min(max(sum(w[a:b]) for a, b in zip((0,) + c, c + (nz,))) for c in itertools.combinations(range(1, nz), world - 1))
        b = PAR.slab_bounds(nz, world, w)
        _check_partition(b, nz, world)
        assert max(sum(w[z0:z1]) for z0, z1 in b) <= best * (1 + 1e-12) + 1e-12


def test_choose_slabs_minimises_the_sum_of_stage_maxima_over_its_candidates():
    import random
    rng = random.Random(3)
    for _ in range(30):
        nz, world = rng.randint(4, 32), rng.choice([2, 3, 4, 8])
        if world > nz:
            continue
        fill = [rng.random() for _ in range(nz)]
        rm = [4 * rng.random() * (nz - z) / nz for z in range(nz)]          # camera on the light side: near slices cost more
        b = PAR.choose_slabs(nz, world, fill, rm)
        _check_partition(b, nz, world)
        cost = lambda bb: 1.3 * max(sum(fill[a:c]) for a, c in bb) + max(sum(rm[a:c]) for a, c in bb)
        for alt in (PAR.slab_bounds(nz, world, fill), PAR.slab_bounds(nz, world, rm), PAR.slab_bounds(nz, world, [f + r for f, r in zip(fill, rm)])):
            assert cost(b) <= cost(alt) + 1e-12


def test_slice_costs_weigh_near_metavoxels_more():
    import numpy as np
    cnt = np.ones((4, 2, 2), dtype=np.int32)
    pos = np.zeros((4, 2, 2, 3))
    pos[..., 2] = np.arange(4)[:, None, None] * 3.0 + 10.0                   # slices at distance 10, 13, 16, 19 from a camera at the origin
    w, f, r = PAR.slice_costs(cnt, pos, (0, 0, 0), 3.0, 1080, np.radians(60.0), 64)
    assert f[0] == f[3] and r[0] > r[1] > r[2] > r[3] and abs(r[0] / r[3] - (19.0 / 10.0) ** 2) < 0.05


import numpy as np  # noqa: E402
import pytest  # noqa: E402


# ---- multi-GPU host logic of the library (csrc/host_logic.cpp: hl_plan_slabs, hl_blend_plan), no GPU needed
def test_library_slab_planner_is_valid_and_optimal():
    import itertools
    from vpfx_amd import engine as E
    rng = np.random.default_rng(7)
    assert E.plan_slabs(32, 4) == [(0, 8), (8, 16), (16, 24), (24, 32)]                # no costs: uniform
    assert E.plan_slabs(5, 5) == [(i, i + 1) for i in range(5)]
    with pytest.raises(E.VpfxError):
        E.plan_slabs(4, 5)                                                             # at most one rank per slice
    for trial in range(120):
        nz = int(rng.integers(3, 10)); world = int(rng.integers(2, min(nz, 5) + 1))
        kind = trial % 3
        F = rng.random(nz) * 3 if kind else np.eye(nz)[int(rng.integers(nz))] * 5       # random / one-hot histograms
        R = rng.random(nz) * 2 if kind != 2 else np.concatenate([rng.random(2) * 9, np.zeros(nz - 2)])   # ray-march work piled at the front
        b = E.plan_slabs(nz, world, F, R, 1)
        assert b[0][0] == 0 and b[-1][1] == nz and all(b0 < b1 for b0, b1 in b) and all(b[i][1] == b[i + 1][0] for i in range(world - 1))
        # the slabs behind the first: slowest local pass (it ends in the all-gather) + the slowest of [finish pass + ray-march]; the fused first
        # slab needs nobody's light and marches as soon as ITS fill is done (the all-gather runs beside its compute stream, round 5)
        obj = lambda bb: max(F[bb[0][0]:bb[0][1]].sum() + R[bb[0][0]:bb[0][1]].sum(),
                             max(F[a:c].sum() for a, c in bb) + max(0.42 * F[a:c].sum() + R[a:c].sum() for a, c in bb[1:]))
        best = min(obj(list(zip((0,) + c, c + (nz,)))) for c in itertools.combinations(range(1, nz), world - 1))
        assert abs(obj(b) - best) <= 1e-9                                              # exact optimum of the two-stage frame model
        # with hand-off groups the cut stays valid and is never worse than the fill-only cut under that model
        g = E.plan_slabs(nz, world, F, R, 2)
        assert g[0][0] == 0 and g[-1][1] == nz and all(b0 < b1 for b0, b1 in g)


def test_library_blend_plan_matches_the_python_pipeline_and_orders_front_to_back():
    from vpfx_amd import engine as E
    import slab_reference as PAR
    rng = np.random.default_rng(11)
    for trial in range(200):
        nz = int(rng.integers(2, 20)); world = int(rng.integers(1, min(nz, 8) + 1))
        cuts = [0] + sorted(rng.choice(np.arange(1, nz), size=world - 1, replace=False).tolist()) + [nz]
        bounds = list(zip(cuts, cuts[1:]))
        zb = int(rng.integers(-1, nz))
        chain, plan, strad = E.blend_plan(bounds, zb)
        ref_plan, ref_strad = PAR.blend_plan(bounds, zb)
        assert strad == ref_strad
        assert [(r, 1 if (which == "under" and r == ref_strad) else 0, k) for r, which, k in ref_plan] == plan
        assert sorted(chain) == list(range(world))
        # front to back: the straddler first, then phase-A-only slabs with zz descending, then phase-B-only slabs with zz ascending
        a_only = [r for r in range(world) if bounds[r][1] - 1 <= zb]
        b_only = [r for r in range(world) if bounds[r][0] > zb]
        assert chain == ([strad] if strad is not None else []) + sorted(a_only, reverse=True) + sorted(b_only)


def test_planner_stays_cheap_on_deep_grids():
    """the exact two-maxima search is ~world nz^4 / 2 steps (over half a billion at nz = 128); above 64 slices the planner keeps to its
    min-max candidates.  A 128-slice grid must still get a valid, balanced cut in well under a second."""
    import time
    import numpy as np
    from vpfx_amd import engine as E
    rng = np.random.default_rng(5)
    nz, world = 128, 8
    fill = np.exp(-((np.arange(nz) - 64.0) / 30.0) ** 2) + 0.01 * rng.random(nz)
    rm = np.exp(-np.arange(nz) / 20.0)
    t0 = time.perf_counter()
    b = E.plan_slabs(nz, world, fill_ms=fill, rm_ms=rm)
    assert time.perf_counter() - t0 < 2.0
    assert b[0][0] == 0 and b[-1][1] == nz and all(z1 > z0 for z0, z1 in b) and all(b[i][1] == b[i + 1][0] for i in range(world - 1))
    worst = max(fill[z0:z1].sum() for z0, z1 in b)
    assert worst <= 2.0 * fill.sum() / world                      # no slab carries more than twice the mean fill work
