"""CPU, world_size 2 / 3 / 4 over gloo: the fan-out's sequencing AS THE LIBRARY DEFINES IT, executed on host buffers.

No GPU here, so the per-slab compute is the CPU oracle (test infrastructure).  Everything that decides what the ranks do with each other's
data comes out of libvpfx.so's host logic -- the same functions csrc/multi.cpp calls on the GPUs:
    vp_plan_slabs     the slab cut (from the (particle, metavoxel)-pair histogram, like the first vp_bin of a fan-out context)
    vp_blend_plan     the compositing order of the slabs and which partial image each contributes
    vp_exchange_plan  the message schedule of the image exchange: who sends which screen piece to whom, what is copied, in order
This file only EXECUTES that schedule (send / recv / copy / all-gather of numbered units over torch.distributed) -- it holds no second
statement of the exchange (the Python pipeline that used to be tested here was a duplicate of multi.cpp).  The frame
assembled on rank 0 must equal the single-process oracle frame, for both exchanges, with and without a slab that straddles zBoundary, with
pixel counts that do not divide by the world size.
"""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _scene(S, width):
    sc = S.make_scene("T0")
    if width is not None:
        sc.width = width                       # 95 x 64 pixels do not divide by 2, 3 or 4: padded pieces
    return sc


def _execute(plan, bufs, unit, world, rank):
    """Run one phase of the library's schedule on host buffers: bufs[id] is a [units, unit] float32 tensor."""
    from vpfx_amd import abi
    batch = []

    def flush():
        reqs = []
        for o in batch:
            t = bufs[o.buf][o.index]
            reqs.append(dist.isend(t, o.peer) if o.kind == abi.VP_XOP_SEND else dist.irecv(t, o.peer))
        for r in reqs:
            r.wait()
        batch.clear()

    for o in plan:
        if o.kind in (abi.VP_XOP_SEND, abi.VP_XOP_RECV):
            batch.append(o)
            continue
        flush()
        if o.kind == abi.VP_XOP_COPY:
            bufs[o.dst_buf][o.dst_index].copy_(bufs[o.buf][o.index])
        elif o.kind == abi.VP_XOP_ALL_GATHER:
            assert o.index == rank
            parts = list(bufs[o.buf][:world].unbind(0))
            dist.all_gather(parts, bufs[o.buf][rank].clone())
        else:
            raise AssertionError(f"unknown exchange operation {o.kind}")
    flush()


def _worker(rank, world, port, cam_pos, out_path, width, all_gather):
    sys.path.insert(0, ROOT)
    from __graft_entry__ import load_package
    load_package()
    from vpfx_amd import abi, engine as E, scene as S
    from oracle import oracle as O
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    sc = _scene(S, width)
    if cam_pos is not None:
        sc.set_camera(cam_pos)
    cam, rp = sc.camera(), sc.raymarch_params()
    # -- slab cut: the library's planner on the pair histogram of the whole grid (every rank has all particles: same cut everywhere)
    whole = O.Oracle(sc.config())
    whole.set_frame(sc.light_to_world, sc.grid_center)
    whole.bin(sc.particles, sc.layout, sc.psys_local_to_world)
    pairs_per_z = whole.bin_counts().sum(axis=(1, 2)).astype(np.float64)
    bounds = E.plan_slabs(sc.N[2], world, fill_ms=pairs_per_z)
    whole.close()
    o = O.Oracle(sc.config(slab=bounds[rank]))
    o.set_frame(sc.light_to_world, sc.grid_center)
    o.bin(sc.particles, sc.layout, sc.psys_local_to_world)
    # -- fill: slab-local pass, ONE all-gather of the transmittance maps, finish with the product of the maps nearer the light; the slab
    #    nearest the light runs the fused fill (multi.cpp: multi_fill)
    tau = torch.from_numpy(o.fill_local(sc.fill_params()))
    taus = [torch.empty_like(tau) for _ in range(world)]
    dist.all_gather(taus, tau)
    if rank == 0:
        o.fill(sc.fill_params())
    else:
        t_in = taus[0].clone()
        for r in range(1, rank):
            t_in.mul_(taus[r])
        o.fill_finish(t_in.numpy())
    # -- ray-march: partial images of the slab, then the library's exchange
    zb = o.z_boundary(cam)
    chain, plan, strad = E.blend_plan(bounds, zb)
    over, under, _ = o.raymarch_partial(cam, rp)
    npix = sc.width * sc.height
    piece = -(-npix // world)
    pixpad = piece * world
    unit = (pixpad if all_gather else piece) * 4
    units = 1 if all_gather else world

    def padded(img):
        t = torch.zeros(pixpad * 4, dtype=torch.float32)
        t[: npix * 4] = torch.from_numpy(np.ascontiguousarray(img)).reshape(-1)
        return t.view(units, unit)

    bufs = {abi.VP_XBUF_PRIMARY: padded(over if bounds[rank][0] <= zb else under),        # first image in blend order (multi.cpp: `primary`)
            abi.VP_XBUF_SECOND: padded(under),
            abi.VP_XBUF_PIECES: torch.zeros((world + 1, unit), dtype=torch.float32),
            abi.VP_XBUF_PIECE_OUT: torch.zeros((1, unit), dtype=torch.float32),
            abi.VP_XBUF_FINAL: torch.zeros((world if not all_gather else 1, unit), dtype=torch.float32)}
    _execute(E.exchange_plan(world, rank, strad, all_gather, 0), bufs, unit, world, rank)
    if not all_gather or rank == 0:
        images = [bufs[abi.VP_XBUF_PIECES][world if which else r].numpy().reshape(1, unit // 4, 4) for r, which, kind in plan]
        out = O.blend_partials(unit // 4, 1, images, [kind for _, _, kind in plan]).reshape(-1)
        (bufs[abi.VP_XBUF_FINAL] if all_gather else bufs[abi.VP_XBUF_PIECE_OUT])[0].copy_(torch.from_numpy(out))
    _execute(E.exchange_plan(world, rank, strad, all_gather, 1), bufs, unit, world, rank)
    if rank == 0:
        img = bufs[abi.VP_XBUF_FINAL].reshape(-1)[: npix * 4].numpy().reshape(sc.height, sc.width, 4)
        np.savez(out_path, img=img, strad=-1 if strad is None else strad, cuts=np.array([b[0] for b in bounds] + [bounds[-1][1]]))
    if rank == world - 1:
        np.savez(out_path + ".last.npz", lightmap=o.read_lightmap())
    dist.barrier()
    dist.destroy_process_group()


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


@pytest.mark.parametrize("world,cam_pos,width,all_gather", [(2, None, None, False), (3, None, 95, False), (2, (1.5, 14.0, 1.0), None, False),
                                                           (4, (1.5, 14.0, 1.0), 95, False), (3, (1.5, 14.0, 1.0), 95, True), (2, None, None, True),
                                                           (2, (1.5, 14.0, 1.0), 95, True)])
def test_the_librarys_exchange_schedule_over_gloo_matches_the_single_process_frame(tmp_path, world, cam_pos, width, all_gather):
    from vpfx_amd import scene as S
    from oracle import oracle as O
    sc = _scene(S, width)
    if cam_pos is not None:
        sc.set_camera(cam_pos)
    o = O.Oracle(sc.config())
    o.set_frame(sc.light_to_world, sc.grid_center)
    o.bin(sc.particles, sc.layout, sc.psys_local_to_world)
    o.fill(sc.fill_params())
    ref = o.raymarch(sc.camera(), sc.raymarch_params())
    out = str(tmp_path / "out.npz")
    mp.spawn(_worker, args=(world, _free_port(), cam_pos, out, width, all_gather), nprocs=world, join=True)
    got, last = np.load(out), np.load(out + ".last.npz")
    assert np.abs(got["img"] - ref).max() <= 1e-5
    np.testing.assert_allclose(last["lightmap"], o.read_lightmap(), rtol=1e-5, atol=1e-9)
    if cam_pos is not None and world == 2:
        assert got["strad"] >= 0               # the camera inside the grid: one slab contributes two images (the schedule's second-image path;
                                               # with 3 and 4 slabs zBoundary falls on a cut of this scene and nobody straddles it)


def test_exchange_plan_is_consistent_across_ranks():
    """Host only: for every world size, straddler and exchange form, every SEND of the schedule has exactly one matching RECV on the peer
    (same phase, same order per pair of ranks), every unit of VP_XBUF_PIECES a rank blends from is written exactly once, and rank 0 ends up
    with every piece of the final image."""
    from vpfx_amd import abi, engine as E
    for world in range(1, 9):
        for strad in [None] + list(range(world)):
            for ag in (False, True):
                for phase in (0, 1):
                    plans = [E.exchange_plan(world, r, strad, ag, phase) for r in range(world)]
                    for r in range(world):
                        for p in range(world):
                            sends = [o for o in plans[r] if o.kind == abi.VP_XOP_SEND and o.peer == p]
                            recvs = [o for o in plans[p] if o.kind == abi.VP_XOP_RECV and o.peer == r]
                            assert len(sends) == len(recvs), (world, strad, ag, phase, r, p)
                    for r in range(world):
                        written = [(o.buf, o.index) for o in plans[r] if o.kind == abi.VP_XOP_RECV] + \
                                  [(o.dst_buf, o.dst_index) for o in plans[r] if o.kind == abi.VP_XOP_COPY]
                        assert len(written) == len(set(written)), (world, strad, ag, phase, r)
                        if phase == 0 and not ag:
                            got = {i for b, i in written if b == abi.VP_XBUF_PIECES}
                            assert got == set(range(world)) | ({world} if strad is not None else set()), (world, strad, r, got)
                        if phase == 1 and not ag and r == 0:
                            assert {i for b, i in written if b == abi.VP_XBUF_FINAL} == set(range(world))
    with pytest.raises(E.VpfxError):
        E.exchange_plan(4, 4, None, False, 0)
