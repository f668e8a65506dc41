"""Predict the N-GPU frame time of the slab pipeline from ONE GPU: run the N slab engines of `bench.py --gpus N` one after the
other on this GPU (same slab cut, same kernels, each engine alone on the device), take the per-stage kernel times of every slab,
and combine them as the pipeline does (stage barriers at the two collectives):
    t_N = max_r(bin_r + fill_local_r) + t_allgather(tau) + max_r(finish_r) + max_r(raymarch_partial_r) + t_exchange(images) + blend
The collective terms are xGMI estimates (7 links x ~50 GB/s usable per GPU + ~40 us software latency per collective), stated in
the output; everything else is measured.  usage: scaling_model.py [C3] [r8|f32]"""
import sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from __graft_entry__ import load_package
load_package()
from vpfx_amd import engine as E, parallel as PAR, scene as S

name = sys.argv[1] if len(sys.argv) > 1 else "C3"
cube = sys.argv[2] if len(sys.argv) > 2 else "r8"
sc = S.make_scene(name, cubemap=cube)
dev = torch.device("cuda", 0)
probe = E.Engine(sc.config())
probe.set_frame(sc.light_to_world, sc.grid_center)
probe.bin(sc.particles, sc.layout, sc.psys_local_to_world)
counts = probe.bin_counts()
mvpos = probe.mv_positions()
probe.fill(sc.fill_params())
img = torch.empty((sc.height, sc.width, 4), device=dev)
for _ in range(3):
    probe.bin_resident(); probe.fill(sc.fill_params()); probe.raymarch_device(sc.camera(), sc.raymarch_params(), img.data_ptr())
probe.sync()
one = dict(bin=probe.last_kernel_ms(0), fill=probe.last_kernel_ms(1), rm=probe.last_kernel_ms(2), samples=probe.stats()["samples"])
probe.close()
weights, fill_w, rm_w = PAR.slice_costs(counts, mvpos, sc.cam_pos, sc.mv_scale, sc.height, np.radians(sc.fov_y_deg), sc.steps)
print(f"slice-cost model: fill {sum(fill_w):.2f} ms, ray-march {sum(rm_w):.2f} ms without cross-slab early-out")
npix = sc.width * sc.height
lm_bytes = sc.N[0] * sc.nv * sc.N[1] * sc.nv * 4
LINK = 7 * 50e9      # bytes/s a GPU can move over its seven xGMI links (conservative)
LAT = 40e-6          # software + launch latency per collective
out = {"config": name, "cubemap": cube, "one_gpu_ms": one, "predictions": {}}
for world in (2, 4, 8):
    bounds = PAR.choose_slabs(sc.N[2], world, fill_w, rm_w)
    rows = []
    for r in range(world):
        e = E.Engine(sc.config(device=0, slab=bounds[r]))
        e.set_frame(sc.light_to_world, sc.grid_center)
        e.upload_particles(sc.particles, sc.layout, sc.psys_local_to_world)
        h = PAR.HipSlabEngine(e, dev)
        tau_all = torch.ones((world,) + h.lm_shape, device=dev)
        for _ in range(3):
            h.bin_resident(); h.fill_local(sc.fill_params()); h.fill_finish_gathered(tau_all, r, world); h.raymarch_partial(sc.camera(), sc.raymarch_params())
        e.sync()
        st = e.stats()
        rows.append(dict(slab=bounds[r], bin=e.last_kernel_ms(0), fill_local=e.last_kernel_ms(1), finish=e.last_kernel_ms(3), rm=e.last_kernel_ms(2),
                         samples=st["samples"], occupied=st["occupied_mv"], pairs=st["pairs"]))
        e.close(); del h, tau_all
        torch.cuda.empty_cache()
    t_tau = LAT + (world - 1) * lm_bytes / LINK
    t_img = 2 * LAT + 2 * (world - 1) / world * npix * 16 / LINK      # all-to-all of pieces + gather of the finished pieces
    t_blend = 0.02e-3 * 1e3 / 1e3
    t = (max(x["bin"] + x["fill_local"] for x in rows) + max(x["finish"] for x in rows) + max(x["rm"] for x in rows)) * 1e-3 + t_tau + t_img + t_blend
    out["predictions"][world] = {
        "slabs": bounds, "per_rank": rows, "t_allgather_tau_ms": t_tau * 1e3, "t_image_exchange_ms": t_img * 1e3,
        "predicted_ms_per_step": t * 1e3, "speedup_vs_1gpu": (one["bin"] + one["fill"] + one["rm"]) / (t * 1e3),
        "samples_executed_all_ranks": sum(x["samples"] for x in rows),
        "host_overhead_note": "plus the Python/ctypes/torch.distributed host path per frame (~0.2-0.4 ms, not modelled)"}
    print(f"N={world}: predicted {t * 1e3:.2f} ms/step ({out['predictions'][world]['speedup_vs_1gpu']:.2f}x of the 1-GPU kernels {one['bin'] + one['fill'] + one['rm']:.2f} ms); "
          f"max fill_local {max(x['fill_local'] for x in rows):.2f} finish {max(x['finish'] for x in rows):.2f} rm {max(x['rm'] for x in rows):.2f}; "
          f"samples all ranks {sum(x['samples'] for x in rows) / 1e6:.0f} M vs {one['samples'] / 1e6:.0f} M on one GPU")
os.makedirs("gpurun_out", exist_ok=True)
json.dump(out, open(f"gpurun_out/scaling_model_{name}_{cube}.json", "w"), indent=1)
