"""GPU: contexts come and go without leaving device memory, host threads or file descriptors behind -- single contexts, fan-out contexts (worker
threads, two streams and their events per rank, exchange buffers), the emitter, contexts destroyed half-way through a frame and contexts that
were never used."""
import gc
import os
import threading

import numpy as np
import pytest
import torch

from vpfx_amd import abi, engine as E, scene as S

pytestmark = pytest.mark.gpu


def _free_bytes():
    torch.cuda.synchronize()
    return torch.cuda.mem_get_info()[0]


def _open_fds():
    return len(os.listdir("/proc/self/fd"))


def _frame(eng, sc):
    eng.set_frame(sc.light_to_world, sc.grid_center)
    eng.bin(sc.particles, sc.layout, sc.psys_local_to_world)
    eng.fill(sc.fill_params())
    return eng.raymarch(sc.camera(), sc.raymarch_params())


@pytest.mark.parametrize("kind", ["single", "fanout4", "fanout8_all_gather"])
def test_create_use_destroy_many_times_returns_everything(kind):
    sc = S.make_scene("C1", cubemap="r8")

    def make():
        if kind == "single":
            return E.Engine(sc.config())
        world = 4 if kind == "fanout4" else 8
        return E.Engine(sc.config(devices=[0] * world, multi_flags=abi.VP_MULTI_PEER_COPY | (abi.VP_MULTI_EXCHANGE_ALL_GATHER if world == 8 else 0)))
    # warm-up: the runtime's own pools, code objects and the library's lazily created state
    for _ in range(3):
        e = make(); ref = _frame(e, sc); e.close()
    gc.collect()
    free0, fds0, thr0 = _free_bytes(), _open_fds(), threading.active_count()
    n = 40 if kind == "single" else 15
    for i in range(n):
        e = make()
        if i % 5 == 4:
            pass                                          # created and never used
        elif i % 5 == 3:
            e.set_frame(sc.light_to_world, sc.grid_center)
            e.bin(sc.particles, sc.layout, sc.psys_local_to_world)
            e.fill(sc.fill_params())                      # destroyed with the fill still queued
        else:
            img = _frame(e, sc)
            assert np.array_equal(img, ref)               # and every incarnation renders the same bits
        e.close()
    gc.collect()
    free1, fds1 = _free_bytes(), _open_fds()
    assert free0 - free1 < 48 << 20, f"{(free0 - free1) / 2**20:.1f} MiB of device memory did not come back after {n} contexts"
    assert fds1 - fds0 <= 4, (fds0, fds1)
    nthreads = len(os.listdir("/proc/self/task"))
    assert nthreads < 200, nthreads                       # (worker threads of destroyed fan-out contexts have exited)
    assert threading.active_count() == thr0


def test_pinned_buffers_and_async_images_come_and_go():
    """vp_pin_host_buffer / vp_raymarch_async / vp_wait_image / vp_unpin_host_buffer in a loop with a fresh buffer every time: the registrations
    and the copy stream's events are released with the buffers."""
    sc = S.make_scene("T0")
    e = E.Engine(sc.config())
    ref = _frame(e, sc)
    free0, fds0 = _free_bytes(), _open_fds()
    for _ in range(60):
        img = np.empty((sc.height, sc.width, 4), dtype=np.float32)
        e.pin(img)
        e.raymarch_async(sc.camera(), sc.raymarch_params(), img)
        e.wait_image()
        assert np.array_equal(img, ref)
        e.unpin(img)
    assert free0 - _free_bytes() < 16 << 20 and _open_fds() - fds0 <= 4
    e.close()


def test_independent_contexts_on_concurrent_host_threads():
    """Four host threads, each with its OWN context (its own stream), different scenes, many frames at once on one GPU: no state is shared
    between contexts, so every thread must keep rendering exactly what its scene renders alone (the library's only process-wide state is
    read-only after load; vp_last_error is per context)."""
    scenes = [S.make_scene("C1", cubemap="r8"), S.make_scene("T0"), S.make_scene("a", dims=(4, 12, 300, 80, 60)), S.make_scene("b", dims=(3, 32, 200, 64, 64))]
    refs = []
    for sc in scenes:
        e = E.Engine(sc.config()); refs.append(_frame(e, sc)); e.close()
    errors = []

    def worker(i):
        try:
            sc = scenes[i]
            e = E.Engine(sc.config())
            for k in range(25):
                img = _frame(e, sc)
                if not np.array_equal(img, refs[i]):
                    errors.append((i, k, float(np.abs(img - refs[i]).max())))
                    break
            e.close()
        except Exception as ex:                            # noqa: BLE001 -- reported to the main thread
            errors.append((i, repr(ex)))
    ts = [threading.Thread(target=worker, args=(i,)) for i in range(len(scenes))]
    for t in ts:
        t.start()
    for t in ts:
        t.join(300)
        assert not t.is_alive()
    assert not errors, errors
