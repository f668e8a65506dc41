"""GPU: random CALL SEQUENCES against the C ABI -- valid and invalid calls in any order (frames without particles, fills before bins, metavoxel
indices outside the grid, zero step counts, NaN parameters, per-metavoxel entry points in the middle of a whole-grid frame, occluders coming and
going).  The boundary's contract (include/vpfx.h): every entry point returns VP_OK or a negative status and leaves the context
usable -- no crash, no hang, no exception across the ABI -- and whatever happened before, the next complete, valid frame is the frame."""
import ctypes as C
import math

import numpy as np
import pytest

from vpfx_amd import abi, engine as E, scene as S

pytestmark = pytest.mark.gpu


def _ops(sc, rng, boxes):
    """name -> callable(engine); each may raise E.VpfxError (a negative status) and nothing else."""
    N = sc.N

    def idx():
        return tuple(int(rng.integers(-1, n + 1)) for n in N)            # now and then outside the grid

    def bad_fill():
        p = sc.fill_params()
        k = int(rng.integers(0, 4))
        if k == 0: p.opacity_factor = float("nan")
        elif k == 1: p.displacement_scale = -1.0
        elif k == 2: p.cubemap = None; p.cubemap_size = 0
        else: p.light_near, p.light_far = 5.0, 1.0
        return p

    def bad_rp():
        rp = sc.raymarch_params()
        if rng.random() < 0.5: rp.steps_per_mv = 0
        else: rp.soft_distance = 0
        return rp
    return {
        "set_frame": lambda e: e.set_frame(sc.light_to_world, sc.grid_center),
        "bin": lambda e: e.bin(sc.particles, sc.layout, sc.psys_local_to_world),
        "bin_empty": lambda e: e.bin(sc.particles[:0].copy(), sc.layout, sc.psys_local_to_world),
        "bin_resident": lambda e: e.bin_resident(),
        "fill": lambda e: e.fill(sc.fill_params()),
        "fill_bad": lambda e: e.fill(bad_fill()),
        "fill_begin": lambda e: e.fill_begin(sc.fill_params()),
        "fill_metavoxel": lambda e: e.fill_metavoxel(*idx()),
        "raymarch": lambda e: e.raymarch(sc.camera(), sc.raymarch_params()),
        "raymarch_bad": lambda e: e.raymarch(sc.camera(), bad_rp()),
        "render_metavoxel": lambda e: e.render_metavoxel(sc.camera(), sc.raymarch_params(), *idx(), blend_over=bool(rng.integers(0, 2))),
        "clear_rt": lambda e: e.clear_particles_rt(),
        "read_brick": lambda e: e.read_brick(*idx()),
        "read_lightmap": lambda e: e.read_lightmap(),
        "bin_list": lambda e: e.bin_list(*idx()),
        "stats": lambda e: e.stats(),
        "sync": lambda e: e.sync(),
        "occluders_on": lambda e: e.set_occluders(boxes),
        "occluders_off": lambda e: e.set_occluders([]),
        "solids_on": lambda e: e.set_occluders(boxes + [S.make_solid(abi.VP_OCC_CYLINDER, (0.0, 0.0, 0.0), (0.5, 1.0, 0.5)),
                                                        S.make_solid(abi.VP_OCC_ELLIPSOID, (1.0, 0.5, -1.0), (0.8, 0.4, 0.6))]),     # ABI 6: vp_set_occluders2
        "solids_bad": lambda e: e.set_occluders([S.make_solid(int(rng.choice([-1, 3, 99])), (0.0, 0.0, 0.0), (1.0, 1.0, 1.0))
                                                 if rng.random() < 0.5 else S.make_solid(abi.VP_OCC_CYLINDER, (0.0, 0.0, 0.0), (1.0, float(rng.choice([0.0, -1.0, np.nan])), 1.0))]),
        "rebalance": lambda e: e.rebalance(),
        "last_ms": lambda e: e.last_kernel_ms(int(rng.integers(-1, 5))),
    }


@pytest.mark.parametrize("seed", range(40))
def test_random_call_sequences_never_crash_and_the_next_valid_frame_is_the_frame(seed):
    rng = np.random.default_rng(1000 + seed)
    nv = int(rng.choice([8, 12, 16, 16, 32]))
    sc = S.make_scene(f"api{seed}", dims=(int(rng.integers(1, 5)), nv, int(rng.integers(1, 400)), int(rng.integers(9, 90)), int(rng.integers(5, 70))),
                      border=int(rng.choice([0, 1, 1])))
    D = 0.8 * max(sc.N) * sc.mv_scale
    boxes = [S.make_box((0.0, -0.3 * D, 0.0), (2 * D, 0.05 * D, 2 * D)), S.make_box((0.2 * D, 0.0, 0.1 * D), (0.1 * D, 0.2 * D, 0.1 * D))]
    fanout = seed % 3 == 2 and sc.N[2] >= 2
    cfg = sc.config(devices=[0] * min(sc.N[2], 3), multi_flags=abi.VP_MULTI_PEER_COPY) if fanout else sc.config()

    def valid_frame(e):
        e.set_occluders([])
        e.set_frame(sc.light_to_world, sc.grid_center)
        e.bin(sc.particles, sc.layout, sc.psys_local_to_world)
        e.fill(sc.fill_params())
        return e.raymarch(sc.camera(), sc.raymarch_params())
    ref_e = E.Engine(sc.config())
    ref = valid_frame(ref_e)
    ref_e.close()
    e = E.Engine(cfg)
    ops = _ops(sc, rng, boxes)
    names = sorted(ops)
    trace, statuses = [], set()
    for step in range(100):
        name = names[int(rng.integers(0, len(names)))]
        trace.append(name)
        try:
            ops[name](e)
        except E.VpfxError as ex:
            assert ex.code < 0, (trace[-8:], ex)
            statuses.add(ex.code)
            assert ex.code != abi.VP_ERR_HIP, (trace[-8:], str(ex))          # a device-side fault is never an acceptable answer to a bad call
        if step % 20 == 19:
            img = valid_frame(e)
            assert np.abs(img - ref).max() <= (2e-5 if fanout else 0.0), (trace[-25:], float(np.abs(img - ref).max()))
    assert statuses and statuses <= {abi.VP_ERR_BAD_ARG, abi.VP_ERR_STATE, abi.VP_ERR_UNSUPPORTED}, statuses      # (a hundred random calls always contain refused ones)
    e.close()
