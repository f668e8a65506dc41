"""RCCL API check on ONE GPU: a 1-rank "nccl" process group runs every collective of SlabPipeline's image exchange (all_to_all_single,
scatter, gather, all_gather_into_tensor, broadcast) and the fill's all-gather on the tensors / views the pipeline really passes --
catches dtype / contiguity / argument mistakes that gloo tolerates.  (Multi-GPU runs are the driver's; this box has one GPU.)"""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

sys.path.insert(0, '.')
from __graft_entry__ import load_package; load_package()
from vpfx_amd import scene as S, engine as E, parallel as PAR

os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29533")
torch.cuda.set_device(0)
dev = torch.device("cuda", 0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
sc = S.make_scene("C1")
sc.set_camera((1.0, 19.0, 0.6))                      # both blend phases -> the one slab straddles zBoundary: scatter path too
bounds = [(0, sc.N[2])]
eng = E.Engine(sc.config(device=0, slab=bounds[0]))
eng.set_frame(sc.light_to_world, sc.grid_center)
eng.upload_particles(sc.particles, sc.layout, sc.psys_local_to_world)
h = PAR.HipSlabEngine(eng, dev)
ref = E.Engine(sc.config())
ref.set_frame(sc.light_to_world, sc.grid_center)
ref.bin(sc.particles, sc.layout, sc.psys_local_to_world)
ref.fill(sc.fill_params())
want = ref.raymarch(sc.camera(), sc.raymarch_params())
for exchange in ("tiles", "all_gather"):
    pipe = PAR.SlabPipeline(h, bounds, 0, 1, exchange=exchange)
    pipe.world = 1
    # the world == 1 short cuts are bypassed on purpose: run the multi-rank code with one rank
    h.bin_resident()
    tau = h.fill_local(sc.fill_params())
    buf = torch.empty((1,) + tuple(tau.shape), dtype=tau.dtype, device=dev)
    taus = pipe._all_gather(buf, tau)
    assert torch.equal(taus[0], tau)
    h.fill_finish(None)
    cam, rp = sc.camera(), sc.raymarch_params()
    zb = h.z_boundary(cam)
    plan, straddler = PAR.blend_plan(bounds, zb)
    assert straddler == 0, (zb, plan)
    over, under = h.raymarch_partial(cam, rp)
    if exchange == "tiles":
        img = pipe._render_tiles(over, under, plan, straddler, "rank0")
        img_all = pipe._render_tiles(over, under, plan, straddler, "all")
        assert torch.equal(img, img_all)
    else:
        whole = torch.empty((1,) + tuple(over.shape), dtype=over.dtype, device=dev)
        prim = pipe._all_gather(whole, over)
        dist.broadcast(under, src=0)
        images = [under if (r == straddler and which == "under") else prim[r] for r, which, kind in plan]
        img = h.blend(images, [kind for _, _, kind in plan])
    err = float(np.abs(img.cpu().numpy() - want).max())
    print(f"exchange={exchange}: RCCL collectives ok, max |RGBA - single engine| = {err:.2e}")
    assert err <= 1e-5
dist.barrier()
dist.destroy_process_group()
print("ok")
