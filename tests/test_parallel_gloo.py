"""CPU, world_size 2 and 3 over gloo: the slab pipeline (vpfx_amd.parallel.SlabPipeline) with the CPU oracle plugged in as
the compute engine must reproduce the single-process result -- checks the two collectives and the ordered blend."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


class OracleSlabEngine:
    """Test-only adapter: the oracle behind the engine interface SlabPipeline expects (CPU tensors)."""

    def __init__(self, sc, slab):
        from oracle import oracle as O
        self.O = O
        self.sc = sc
        self.o = O.Oracle(sc.config(slab=slab))
        self.o.set_frame(sc.light_to_world, sc.grid_center)
        self._uploaded = False

    def bin_resident(self):
        self.o.bin(self.sc.particles, self.sc.layout, self.sc.psys_local_to_world)

    def fill(self, params):
        self.o.fill(params)

    def fill_local(self, params):
        return torch.from_numpy(self.o.fill_local(params))

    def fill_finish(self, t_in):
        self.o.fill_finish(None if t_in is None else t_in.numpy())

    def z_boundary(self, cam):
        return self.o.z_boundary(cam)

    def raymarch(self, cam, rp):
        return torch.from_numpy(self.o.raymarch(cam, rp))

    def raymarch_partial(self, cam, rp):
        over, under, _ = self.o.raymarch_partial(cam, rp)
        return torch.from_numpy(over), torch.from_numpy(under)

    def blend(self, images, kinds):
        npix = images[0].numel() // 4               # pieces of the image: the blend is per pixel
        out = self.O.blend_partials(npix, 1, [np.ascontiguousarray(t.numpy()).reshape(1, npix, 4) for t in images], kinds)
        return torch.from_numpy(out.reshape(tuple(images[0].shape)))


def _scene(S, width):
    sc = S.make_scene("T0")
    if width is not None:
        sc.width = width                       # 95 x 64 pixels do not divide by 2, 3 or 4: padded pieces
    return sc


def _worker(rank, world, port, cam_pos, out_path, width=None):
    sys.path.insert(0, ROOT)
    from __graft_entry__ import load_package
    load_package()
    from vpfx_amd import parallel as PAR, scene as S
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    sc = _scene(S, width)
    if cam_pos is not None:
        sc.set_camera(cam_pos)
    bounds = PAR.slab_bounds(sc.N[2], world)
    eng = OracleSlabEngine(sc, bounds[rank])
    pipe = PAR.SlabPipeline(eng, bounds, rank, world)
    pipe.fill(sc.fill_params())
    img = pipe.render(sc.camera(), sc.raymarch_params())                    # assembled on rank 0
    img_all = pipe.render(sc.camera(), sc.raymarch_params(), result="all")  # ... or everywhere
    whole = PAR.SlabPipeline(eng, bounds, rank, world, exchange="all_gather").render(sc.camera(), sc.raymarch_params())
    assert np.array_equal(whole.numpy(), img_all.numpy())                   # one all-gather of whole partial images
    assert (img is None) == (rank != 0)
    lm = eng.o.read_lightmap()
    if rank == world - 1:
        np.savez(out_path + ".last.npz", img=img_all.numpy(), lightmap=lm)
    if rank == 0:
        np.savez(out_path, img=img.numpy())
    dist.barrier()
    dist.destroy_process_group()


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


@pytest.mark.parametrize("world,cam_pos,width", [(2, None, None), (3, None, 95), (2, (1.5, 14.0, 1.0), None), (4, (1.5, 14.0, 1.0), 95)])
def test_slab_pipeline_matches_single_process(tmp_path, world, cam_pos, width):
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
    mp.spawn(_worker, args=(world, _free_port(), cam_pos, out, width), nprocs=world, join=True)
    got, last = np.load(out), np.load(out + ".last.npz")
    assert np.abs(got["img"] - ref).max() <= 1e-5
    assert np.array_equal(got["img"], last["img"])
    np.testing.assert_allclose(last["lightmap"], o.read_lightmap(), rtol=1e-5, atol=1e-9)
