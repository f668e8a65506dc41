"""Predict the N-GPU frame time of the in-library fan-out (csrc/multi.cpp) from ONE GPU.

For every (N, hand-off groups):
  1. the LIBRARY cuts the slabs: a fan-out context with N slabs on this GPU (VP_MULTI_PEER_COPY test hook) renders three frames,
     re-balancing twice from its measured work histograms -- exactly what `bench.py --gpus N` does in its warm-up; its image is checked
     against the single-GPU frame and its slab cut, compositing chain and hand-off groups are taken over;
  2. the N slab contexts are then run ONE AFTER THE OTHER, each alone on the GPU (the kernels of concurrent ranks would otherwise share
     the device and stretch each other), with the same cut, the same kernels and the real hand-off maps of the slabs in front
     (vp_raymarch_partial_handoff_device), front to back along the chain;
  3. the per-rank kernel times are combined as the pipeline does:
        t_N = max( bin_0 + fill_0 + raymarch_0 ,  max_r(bin_r + fill_local_r) + t(tau all-gather) + max_{r>0}(finish_r + raymarch_r) )
              + t(image exchange) + t(blend)                                                                                     [one group]
        (round 5: the all-gather of the transmittance maps runs on the library's exchange stream; slab 0 -- fused fill, nobody's light needed --
         marches as soon as its own fill is done; round 4's serial schedule, max_r(bin + fill_local) + t_tau + max_r(finish + raymarch), is
         printed beside it)
        t_N = max_r(bin_r + fill_local_r) + t(tau all-gather) + max_r(finish_r)
            + sum over hand-off groups g of max_{r in g}(raymarch_r) + (groups - 1) * t(hand-off hop) + t(image exchange) + t(blend)
     (slab 0, nearest the light, runs the fused fill: fill_local_0 = its fused kernel, finish_0 = 0)
The exchange terms are xGMI ESTIMATES (one link ~50 GB/s usable per direction, seven links per GPU, ~40 us software latency per RCCL call),
stated in the output; everything else is measured.  The host side is the library's worker threads (one per GPU), not modelled: each
issues ~25 launches per frame.  usage: scaling_model.py [C3|C5] [r8|f32]
C5 (round 6; BASELINE.json's only 8-GPU config: 64^3 x 64^3, 1 M particles, 3840 x 2160): the fan-out context of step 1 holds all N slabs on
this one GPU (the whole ~190 GB brick pool + the per-rank exchange buffers); worlds 2 / 4 / 8, one or two hand-off groups; the per-rank
brick bytes are recorded (what each of the N GPUs has to hold); the alternative 8-GPU plan is skipped."""
import sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from __graft_entry__ import load_package
load_package()
from vpfx_amd import abi, engine as E, scene as S

name = sys.argv[1] if len(sys.argv) > 1 else "C3"
cube = sys.argv[2] if len(sys.argv) > 2 else "r8"
sc = S.make_scene(name, cubemap=cube)
dev = torch.device("cuda", 0)
cam, rp = sc.camera(), sc.raymarch_params()
img = torch.empty((sc.height, sc.width, 4), device=dev)
probe = E.Engine(sc.config())
probe.set_frame(sc.light_to_world, sc.grid_center)
probe.bin(sc.particles, sc.layout, sc.psys_local_to_world)
probe.fill(sc.fill_params())
for _ in range(3):
    probe.bin_resident(); probe.fill(sc.fill_params()); probe.raymarch_device(cam, rp, img.data_ptr())
probe.sync()
one = dict(bin=probe.last_kernel_ms(0), fill=probe.last_kernel_ms(1), rm=probe.last_kernel_ms(2), samples=probe.stats()["samples"])
zb = probe.z_boundary(cam)
ref = img.clone()
BIG = sc.N[2] > 32
if BIG:
    # the per-slice cost profile the planner cuts from (what a fan-out context measures in its warm-up: pairs per slice for the fill, executed
    # samples per slice for the ray-march, scaled to the measured kernel times): taken from the ONE whole-grid context, because N slab contexts of
    # config 5 with their allocation head-room do not fit one GPU together
    hist = np.asarray(probe.z_histogram(), dtype=np.float64)
    probe.bin_resident(); probe.fill(sc.fill_params())           # (the histogram pass re-bins: the bricks count as stale until the next fill)
    o_, u_ = torch.empty_like(img), torch.empty_like(img)
    probe.raymarch_partial_handoff_device(cam, rp, o_.data_ptr(), u_.data_ptr(), 0, 0, 0, 0)
    probe.sync()
    zs = np.asarray(probe.zsamples(), dtype=np.float64)
    fill_w = list(hist / max(hist.sum(), 1.0) * one["fill"])
    rm_w = list(zs / max(zs.sum(), 1.0) * one["rm"])
    del o_, u_
probe.close()
torch.cuda.empty_cache()
one_ms = one["bin"] + one["fill"] + one["rm"]
print(f"1 GPU: bin {one['bin']:.3f} fill {one['fill']:.3f} ray-march {one['rm']:.3f} = {one_ms:.3f} ms, {one['samples'] / 1e6:.0f} M samples, zBoundary {zb}")
npix = sc.width * sc.height
lm = (sc.N[1] * sc.nv, sc.N[0] * sc.nv)
lm_bytes = lm[0] * lm[1] * 4
LINK1 = 50e9         # bytes/s over ONE xGMI link, one direction (conservative; 76.8 GB/s nominal)
LAT = 40e-6          # software + launch latency per RCCL call
out = {"config": name, "cubemap": cube, "one_gpu_ms": one, "assumptions": {"xgmi_link_GBps": LINK1 / 1e9, "rccl_call_latency_us": LAT * 1e6},
       "predictions": {}}


def frame(e, first=False):
    e.bin_resident(); e.fill(sc.fill_params()); e.raymarch_device(cam, rp, img.data_ptr())


for world in ((2, 4, 8) if BIG else (2, 4, 7, 8)):                # (7: the slab cut of the alternative 8-GPU plan below)
    for groups in ([1] if BIG else sorted({1, 2, world}) if world != 7 else [1]):
        if BIG:
            # 1'. the library's planner on the whole-grid context's cost profile (see above); no hand-off groups
            bounds = E.plan_slabs(sc.N[2], world, fill_ms=fill_w, rm_ms=rm_w, rm_groups=1)
            cuts = [b[0] for b in bounds] + [bounds[-1][1]]
            chain = E.blend_plan(bounds, zb)[0]
            group_of = [0] * world
            err = float("nan")
        else:
            # 1. the library's own cut, chain and groups
            m = E.Engine(sc.config(devices=[0] * world, multi_flags=abi.VP_MULTI_PEER_COPY, rm_groups=groups))
            m.set_frame(sc.light_to_world, sc.grid_center)
            m.upload_particles(sc.particles, sc.layout, sc.psys_local_to_world)
            frame(m)
            for _ in range(2):
                m.rebalance(); frame(m)
            frame(m)                       # (the cut asked for by the second vp_rebalance takes effect at this frame's bin)
            m.sync()
            err = float((img - ref).abs().max().item())
            info = m.multi_info()
            m.close()
            torch.cuda.empty_cache()
            cuts, chain, group_of = info["slab_cuts"], info["chain"], info["group_of"]
        # 2. every slab alone on the GPU, front to back, with the real hand-off maps
        rows, maps = {}, {}
        for r in chain:
            torch.cuda.empty_cache()
            free0 = torch.cuda.mem_get_info()[0]
            e = E.Engine(sc.config(device=0, slab=(cuts[r], cuts[r + 1])))
            e.set_frame(sc.light_to_world, sc.grid_center)
            e.upload_particles(sc.particles, sc.layout, sc.psys_local_to_world)
            tau_all = torch.ones((world,) + lm, device=dev)
            front = [s for s in chain if group_of[s] < group_of[r]]
            a_only = cuts[r + 1] - 1 <= zb
            t_in = torch.stack([maps[s][0 if a_only else 1] for s in front]).contiguous() if front else None
            over, under = torch.empty_like(img), torch.empty_like(img)
            t_out = torch.empty((2, sc.height, sc.width), device=dev, dtype=torch.uint8)
            for _ in range(3):
                e.bin_resident()
                if r == 0: e.fill(sc.fill_params())          # the slab nearest the light: fused fill, no finish pass (multi.cpp: multi_fill)
                else: e.fill_local(sc.fill_params(), tau_all[r].data_ptr()); e.fill_finish_gathered(tau_all.data_ptr(), r, world)
                e.raymarch_partial_handoff_device(cam, rp, over.data_ptr(), under.data_ptr(), t_in.data_ptr() if front else 0, len(front),
                                                  t_out[0].data_ptr(), t_out[1].data_ptr())
            e.sync()
            st = e.stats()
            device_bytes = free0 - torch.cuda.mem_get_info()[0]      # everything this rank holds: bricks + scratch + CSR + images + the all-gathered maps
            maps[r] = (t_out[0].clone(), t_out[1].clone())
            rows[r] = dict(slab=[cuts[r], cuts[r + 1]], group=group_of[r], bin=e.last_kernel_ms(0), fill_local=e.last_kernel_ms(1), finish=e.last_kernel_ms(3) if r else 0.0,
                           rm=e.last_kernel_ms(2), samples=st["samples"], occupied=st["occupied_mv"], pairs=st["pairs"], brick_bytes=st["brick_bytes"], device_bytes=int(device_bytes))
            e.close(); del tau_all, over, under
            torch.cuda.empty_cache()
        # 3. the pipeline
        t_tau = LAT + lm_bytes / LINK1                                   # every rank receives the other ranks' maps over separate links
        t_hop = LAT + npix / LINK1                                       # one byte per pixel per map, senders on separate links
        piece = npix * 16 / world
        t_img = 2 * (LAT + piece / LINK1)                                # all-to-all of pieces, then the gather on the display rank
        t_blend = 0.02e-3
        G = max(group_of) + 1
        rm_groups = [max(rows[r]["rm"] for r in chain if group_of[r] == g) for g in range(G)]
        fill_max = max(x["bin"] + x["fill_local"] for x in rows.values())
        if G == 1:      # after the tau all-gather every slab but slab 0 runs its finish pass and its ray-march back to back; slab 0 marches right after its fused fill
            after = max([x["finish"] + x["rm"] for r, x in rows.items() if r != 0] or [0.0])
            path0 = rows[0]["bin"] + rows[0]["fill_local"] + rows[0]["rm"]
            t = max(path0 * 1e-3, (fill_max + after) * 1e-3 + t_tau) + t_img + t_blend
            t_serial = (fill_max + max(x["finish"] + x["rm"] for x in rows.values())) * 1e-3 + t_tau + t_img + t_blend      # round 4's schedule
        else:
            after = max(x["finish"] for x in rows.values()) + sum(rm_groups)
            t = (fill_max + after) * 1e-3 + t_tau + (G - 1) * t_hop + t_img + t_blend
            t_serial = t
        key = f"{world}gpu_{groups}groups"
        out["predictions"][key] = {
            "world": world, "rm_groups": groups, "slab_cuts": cuts, "chain": chain, "group_of": group_of, "per_rank": [rows[r] for r in range(world)],
            "max_abs_rgba_diff_vs_1gpu_frame": err, "t_allgather_tau_ms": t_tau * 1e3, "t_handoff_hop_ms": t_hop * 1e3, "t_image_exchange_ms": t_img * 1e3,
            "raymarch_ms_per_group": rm_groups, "predicted_ms_per_step": t * 1e3, "predicted_ms_per_step_serial_schedule_r4": t_serial * 1e3,
            "speedup_vs_1gpu": one_ms / (t * 1e3),
            "samples_executed_all_ranks": sum(x["samples"] for x in rows.values())}
        print(f"N={world} groups={groups}: predicted {t * 1e3:.2f} ms/step ({one_ms / (t * 1e3):.2f}x; serial schedule {t_serial * 1e3:.2f}); cut {cuts}; "
              f"max bin+fill_local {max(x['bin'] + x['fill_local'] for x in rows.values()):.2f} finish {max(x['finish'] for x in rows.values()):.2f} max finish+rm {max(x['finish'] + x['rm'] for x in rows.values()):.2f} "
              f"rm per group {[round(x, 3) for x in rm_groups]} (max single {max(x['rm'] for x in rows.values()):.3f}); exchanges {1e3 * (t_tau + (G - 1) * t_hop + t_img):.2f}; "
              f"samples {sum(x['samples'] for x in rows.values()) / 1e6:.0f} M vs {one['samples'] / 1e6:.0f} M; fan-out image err {err:.1e}; "
              f"largest rank holds {max(x['device_bytes'] for x in rows.values()) / 2**30:.1f} GiB", flush=True)
# ---------------------------------------------------------------------------------------------------------------------------------------
# ONE alternative plan for 8 GPUs: the front slab -- the one that holds most of the ray-march when light and camera are on
# the same side -- is FILLED ON TWO RANKS (replicated: each fills the whole slab) which split its SCREEN, so that its ray-march halves; the other
# six ranks share the remaining slices.  Seven distinct slabs = the library's cut for world 7 (measured above); the two halves of the front
# slab's ray-march are measured for real: a scene-depth buffer of 0 on the other half of the screen rejects every metavoxel there (ZTest Less,
# RM.shader:14), the split column halves the executed samples (from the samples-per-ray view).
os.makedirs("gpurun_out/r6_scaling", exist_ok=True)
json.dump(out, open(f"gpurun_out/r6_scaling/scaling_model_{name}_{cube}.json", "w"), indent=1)
if BIG:
    sys.exit(0)
p7 = out["predictions"]["7gpu_1groups"]
cuts7 = p7["slab_cuts"]
e = E.Engine(sc.config(device=0, slab=(cuts7[0], cuts7[1])))
e.set_frame(sc.light_to_world, sc.grid_center)
e.upload_particles(sc.particles, sc.layout, sc.psys_local_to_world)
e.bin_resident(); e.fill(sc.fill_params())
rq = sc.raymarch_params(); rq.flags = abi.VP_RM_SHOW_RAY_SAMPLES
per_col = e.raymarch(cam, rq)[..., 0].astype(np.float64).sum(axis=0)
split = int(np.searchsorted(np.cumsum(per_col), per_col.sum() / 2.0))
halves = []
for side in (0, 1):
    depth = np.full((sc.height, sc.width), 3.0e38, dtype=np.float32)
    if side == 0: depth[:, split:] = 0.0
    else: depth[:, :split] = 0.0
    rh = sc.raymarch_params(); rh.scene_depth = depth.ctypes.data_as(abi.c_float_p)
    over, under = torch.empty_like(img), torch.empty_like(img)
    for _ in range(3):
        e.raymarch_partial_handoff_device(cam, rh, over.data_ptr(), under.data_ptr(), 0, 0, 0, 0)
    e.sync()
    halves.append(dict(rm=e.last_kernel_ms(2), samples=e.stats()["samples"]))
e.close()
rows7 = p7["per_rank"]
fill_max = max(x["bin"] + x["fill_local"] for x in rows7)
after7 = max([x["finish"] + x["rm"] for x in rows7[1:]] + [h["rm"] for h in halves])      # (serial schedule: a plan that replicates the front slab has no decoupled first rank)
piece8 = npix * 16 / 8
t_alt = (fill_max + after7) * 1e-3 + (LAT + lm_bytes / LINK1) + 2 * (LAT + piece8 / LINK1) + 0.02e-3
base8 = out["predictions"]["8gpu_1groups"]["predicted_ms_per_step"]
out["alternative_8gpu_front_slab_on_two_ranks"] = {
    "slab_cuts_7": cuts7, "screen_split_column": split, "front_halves": halves, "front_slab_whole": rows7[0],
    "max_bin_fill_local_ms": fill_max, "max_after_allgather_ms": after7, "predicted_ms_per_step": t_alt * 1e3,
    "baseline_8_slabs_ms_per_step": base8, "beats_baseline": bool(t_alt * 1e3 < base8)}
print(f"ALTERNATIVE at 8 GPUs -- front slab {cuts7[:2]} filled on two ranks, screen split at column {split}: halves' ray-march "
      f"{halves[0]['rm']:.3f} / {halves[1]['rm']:.3f} ms (whole: {rows7[0]['rm']:.3f}), max bin+fill_local {fill_max:.2f} (8 slabs: "
      f"{max(x['bin'] + x['fill_local'] for x in out['predictions']['8gpu_1groups']['per_rank']):.2f}), max after the all-gather {after7:.2f}: "
      f"predicted {t_alt * 1e3:.2f} ms/step vs {base8:.2f} for eight slabs -> {'BEATS' if t_alt * 1e3 < base8 else 'does NOT beat'} it", flush=True)
json.dump(out, open(f"gpurun_out/r6_scaling/scaling_model_{name}_{cube}.json", "w"), indent=1)
