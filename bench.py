#!/usr/bin/env python3
"""bench.py -- headline benchmark of the sparse-volumetric-particle hot path on MI355X.

One "step" = one full pass of the hot path over one batch of synthetic input that is already resident in HBM:
    bin (particles -> metavoxels)  ->  fill (+ light propagation)  ->  ray-march (+ inter-metavoxel blend).

Workload at N = 1: BASELINE.json configs[2], the configuration its metric is quoted on:
    32x32x32 metavoxels x 32^3 voxels, 100k particles, 1920x1080 (synthetic scene of SURVEY.md 8(d)).
N > 1 (launched by torch.distributed.run, one rank per GPU): the SAME grid split into light-axis slabs
(BASELINE.json configs[3]) -> strong scaling; two all-gathers per step (slab transmittance, partial images).

Prints ONE JSON line on rank 0.  `value` = (voxels filled + samples ray-marched) per second over the whole job,
in millions; the two halves are also reported separately.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

from __graft_entry__ import load_package  # noqa: E402

load_package()
from vpfx_amd import engine as E, parallel as PAR, scene as S  # noqa: E402

HBM_PEAK_GBS = 8000.0      # MI355X HBM3E spec peak (MI355X_MICROARCH.md); ~6300 GB/s achievable


def baseline_metric():
    """BASELINE.json's metric string verbatim (the file travels with the repo); the same text if it is missing."""
    try:
        return json.load(open(os.path.join(ROOT, "BASELINE.json")))["metric"]
    except Exception:
        return "Mvoxels/s filled + Msamples/s raymarched, 32\u00b3\u00d732\u00b3 grid @1080p"


def kernel_sources_sha():
    """Fingerprint of the kernel sources: profiles/traffic_<cfg>.json records the one it was measured on, and bench.py only
    reports that PMC measurement as `roofline.traffic` while it still matches (a stale number is worse than null)."""
    import hashlib
    h = hashlib.sha256()
    d = os.path.join(ROOT, "volumetric-particles-for-unity_amd", "csrc")
    for f in sorted(os.listdir(d)):
        if f.endswith((".hip", ".cpp", ".h")):
            h.update(f.encode() + b"\0" + open(os.path.join(d, f), "rb").read())
    return h.hexdigest()[:16]


def cpu_baseline(sc, threads=0):
    """Time the CPU oracle (a port of the reference's algorithm; the reference itself is C# + HLSL and cannot run
    here) on the GPU box's host cores.  Reported, not shipped: this is the only place bench.py touches oracle/."""
    from oracle import oracle as O
    o = O.Oracle(sc.config(), threads=threads)
    o.set_frame(sc.light_to_world, sc.grid_center)
    t0 = time.perf_counter()
    o.bin(sc.particles, sc.layout, sc.psys_local_to_world)
    t1 = time.perf_counter()
    o.fill(sc.fill_params())
    t2 = time.perf_counter()
    o.raymarch(sc.camera(), sc.raymarch_params())
    t3 = time.perf_counter()
    st = o.stats()
    units = st["voxels_filled"] + st["samples"]
    out = {
        "value": units / (t3 - t0) / 1e6, "unit": "M(voxels+samples)/s", "cores": threads or O.Oracle.max_threads(),
        "kind": "port",
        "sample": f"one full step of {sc.name} (bin {t1 - t0:.3f}s, fill {t2 - t1:.3f}s, raymarch {t3 - t2:.3f}s)",
        "fill_mvoxels_per_s": st["voxels_filled"] / (t2 - t1) / 1e6,
        "raymarch_msamples_per_s": st["samples"] / (t3 - t2) / 1e6,
        "seconds": t3 - t0,
    }
    o.close()
    # single-thread leg: ONE light-axis slice of the same workload (the middle one) on one thread, extrapolated
    # to the whole step with the per-unit rates (a full single-thread step would take minutes)
    zmid = sc.N[2] // 2
    o1 = O.Oracle(sc.config(slab=(zmid, zmid + 1)), threads=1)
    o1.set_frame(sc.light_to_world, sc.grid_center)
    o1.bin(sc.particles, sc.layout, sc.psys_local_to_world)
    s0 = time.perf_counter()
    o1.fill(sc.fill_params())
    s1 = time.perf_counter()
    o1.raymarch_partial(sc.camera(), sc.raymarch_params())
    s2 = time.perf_counter()
    st1 = o1.stats()
    fill_rate, rm_rate = st1["voxels_filled"] / (s1 - s0), st1["samples"] / max(s2 - s1, 1e-9)
    est = st["voxels_filled"] / fill_rate + st["samples"] / rm_rate
    out["single_thread"] = {
        "cores": 1, "sample": f"light-axis slice zz = {zmid} of {sc.N[2]}: {st1['voxels_filled']} voxels in {s1 - s0:.2f}s, "
                              f"{st1['samples']} samples in {s2 - s1:.2f}s",
        "fill_mvoxels_per_s": fill_rate / 1e6, "raymarch_msamples_per_s": rm_rate / 1e6,
        "seconds_1thread_full_step_extrapolated": est, "value": units / est / 1e6,
        "parallel_speedup_of_the_port": est / (t3 - t0),
    }
    o1.close()
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=300, help="timed steps (default: >= 1 s of timed region at ~5 ms per step)")
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--config", default="C3", help="scene name from vpfx_amd.scene.CONFIGS (default: the metric's config)")
    ap.add_argument("--cubemap", default="r8", choices=["r8", "f32"],
                    help="displacement cube map texel format: r8 = 8-bit like the reference's asset (LDS-resident in k_fill), f32 = float texels")
    ap.add_argument("--no-lds-cubemap", action="store_true", help="A/B: keep an R8 cube map on the global f32 footprint table")
    ap.add_argument("--no-grey", action="store_true", help="A/B: keep RGBA16F bricks although the ambient colour is grey")
    ap.add_argument("--displacement-scale", type=float, default=None,
                    help="override the scene's _DisplacementScale (default 0.7, scene:9016); 1.0 = the slider's maximum (smoothstep jump at net displacement 0)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-threads", type=int, default=0)
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend (nccl = RCCL; gloo only for functional tests)")
    ap.add_argument("--share-gpu", action="store_true", help="functional test only: all ranks on cuda:0")
    ap.add_argument("--exchange", default="tiles", choices=["tiles", "all_gather"],
                    help="N > 1 image exchange: all-to-all of screen pieces + sharded blend (default) or one all-gather of whole partial images")
    ap.add_argument("--uniform-slabs", action="store_true", help="equal-thickness slabs instead of pair-count balanced ones")
    ap.add_argument("--no-reference-frame", action="store_true", help="N > 1: do not render the 1-GPU frame on rank 0 for the shard check")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the product has no CPU path)")
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node {args.gpus}")
    if args.share_gpu:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    if world > 1:
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=device)
        else:
            dist.init_process_group(args.backend)

    sc = S.make_scene(args.config, cubemap=args.cubemap)
    if args.displacement_scale is not None:
        sc.displacement_scale = float(args.displacement_scale)
    weights, fill_w, rm_w, whole_occupied = None, None, None, None
    if world > 1:
        # every rank computes the same (particle, MV)-pair histogram along the light axis (balanced slabs) and the
        # whole-grid occupancy (sizes the optional 1-GPU reference frame below); binning allocates no bricks
        probe = E.Engine(sc.config(device=local_rank))
        probe.set_frame(sc.light_to_world, sc.grid_center)
        probe.bin(sc.particles, sc.layout, sc.psys_local_to_world)
        counts = probe.bin_counts()
        mvpos = probe.mv_positions()
        whole_occupied = probe.stats()["occupied_mv"]
        probe.close()
        if not args.uniform_slabs:
            # estimated ms per light-axis slice: fill (~ pairs) + this camera's ray-march (~ screen footprint of the occupied MVs)
            weights, fill_w, rm_w = PAR.slice_costs(counts, mvpos, sc.cam_pos, sc.mv_scale, sc.height, np.radians(sc.fov_y_deg), sc.steps)
    bounds = PAR.choose_slabs(sc.N[2], world, fill_w, rm_w) if weights is not None else PAR.slab_bounds(sc.N[2], world, None)
    # strong scaling: the unit of work is the SINGLE-GPU job (its voxels and its executed samples).  Sharded runs execute
    # more lattice samples in total (the saturation early-out only sees one slab), which must not inflate `value`.
    ref_units, ref_image, ref_skipped = None, None, None
    if world > 1 and rank == 0:
        # The 1-GPU reference frame is rendered on rank 0 only if the whole grid's brick pool fits NEXT TO this rank's slab
        # engine (bricks + 16 B/voxel split-fill scratch): at config 5 it is ~190 GB and must not be attempted.
        brick = 8 * sc.nv ** 3
        need_ref = whole_occupied * brick * 1.125 + 2e9
        need_slab = 3.0 * (whole_occupied / world) * 1.5 * brick * 1.125 + 2e9      # bricks + 2x scratch, 1.5x for an uneven slab
        free, _ = torch.cuda.mem_get_info()
        if need_ref + need_slab > 0.9 * free or args.no_reference_frame:
            ref_skipped = (f"1-GPU reference frame skipped: whole-grid brick pool {need_ref / 1e9:.0f} GB + slab engine "
                           f"{need_slab / 1e9:.0f} GB vs {free / 1e9:.0f} GB free" if not args.no_reference_frame else "--no-reference-frame")
            print(f"[bench] {ref_skipped}", file=sys.stderr)
        else:
            one = E.Engine(sc.config(device=local_rank))
            one.set_frame(sc.light_to_world, sc.grid_center)
            one.bin(sc.particles, sc.layout, sc.psys_local_to_world)
            one.fill(sc.fill_params())
            ref_image = torch.empty((sc.height, sc.width, 4), device=device)
            one.raymarch_device(sc.camera(), sc.raymarch_params(), ref_image.data_ptr())
            one.sync()
            st1 = one.stats()
            ref_units = (st1["voxels_filled"], st1["samples"])
            one.close()
            torch.cuda.empty_cache()
    cfg = sc.config(device=local_rank, slab=bounds[rank] if world > 1 else (0, 0))
    if args.no_lds_cubemap:
        cfg.reserved[0] = 1            # VPFX_CFG_NO_LDS_CUBEMAP
    if args.no_grey:
        cfg.reserved[1] = 1            # VPFX_CFG_NO_GREY_BRICKS
    eng = E.Engine(cfg)
    eng.set_frame(sc.light_to_world, sc.grid_center)
    eng.upload_particles(sc.particles, sc.layout, sc.psys_local_to_world)      # inputs resident in HBM from here on
    pipe = PAR.SlabPipeline(PAR.HipSlabEngine(eng, device), bounds, rank, world, exchange=args.exchange)
    fp_first, fp = sc.fill_params(), sc.fill_params()
    fp.cubemap = None                                                           # cubemap stays resident after the first fill
    cam, rp = sc.camera(), sc.raymarch_params()

    def step(first=False):
        pipe.fill(fp_first if first else fp)
        return pipe.render(cam, rp)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    step(first=True)
    for _ in range(max(args.warmup - 1, 0)):
        step()
    barrier()
    k_fill, k_rm, k_bin, k_fin = [], [], [], []
    t0 = time.perf_counter()
    image = None
    for _ in range(args.steps):
        image = step()
        # HIP-event durations of the dominant kernels, recorded on the stream they were launched on
        k_bin.append(eng.last_kernel_ms(0)); k_fill.append(eng.last_kernel_ms(1)); k_rm.append(eng.last_kernel_ms(2))
        if world > 1:
            k_fin.append(eng.last_kernel_ms(3))
    barrier()
    dt = time.perf_counter() - t0
    t = torch.tensor([dt], dtype=torch.float64, device=device)
    st = eng.stats()
    counts = torch.tensor([st["voxels_filled"], st["samples"], st["occupied_mv"], st["pairs"], st["bricks_sampled"]],
                          dtype=torch.float64, device=device)
    stage_max = torch.tensor([float(np.mean(k_bin)), float(np.mean(k_fill)), float(np.mean(k_rm)), float(np.mean(k_fin)) if k_fin else 0.0],
                             dtype=torch.float64, device=device)
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dist.all_reduce(counts, op=dist.ReduceOp.SUM)
        dist.all_reduce(stage_max, op=dist.ReduceOp.MAX)          # slowest rank per stage (kernel time, HIP events)
    dt = float(t.item())
    voxels, samples, occupied, pairs, bricks_sampled = [float(x) for x in counts.tolist()]

    if rank == 0:
        ms_step = dt / args.steps * 1e3
        nv = sc.nv
        fill_ms, rm_ms, bin_ms = float(np.mean(k_fill)), float(np.mean(k_rm)), float(np.mean(k_bin))
        # algorithmic bytes; rank-0 slab for N > 1
        # bricks are counted at the bytes per voxel the context actually stores (8 in both formats = SURVEY's figure)
        lds_path = args.cubemap == "r8" and not args.no_lds_cubemap
        bpv = st.get("brick_bytes_per_voxel", 8)
        fill_bytes = st["occupied_mv"] * (bpv * nv ** 3 + 8 * nv ** 2) + 84 * st["pairs"]
        rm_bytes = st["bricks_sampled"] * bpv * nv ** 3 + 16 * sc.width * sc.height
        roofs = {
            "fill": {"bound": "hbm", "kernel": "k_fill_lds" if lds_path else "k_fill", "achieved": fill_bytes / (fill_ms * 1e-3) / 1e9, "peak": HBM_PEAK_GBS,
                     "unit": "GB/s", "bytes_per_launch": fill_bytes, "avg_ms": fill_ms, "brick_bytes_per_voxel": bpv},
            "raymarch": {"bound": "hbm", "kernel": "k_raymarch", "achieved": rm_bytes / (rm_ms * 1e-3) / 1e9, "peak": HBM_PEAK_GBS,
                         "unit": "GB/s", "bytes_per_launch": rm_bytes, "avg_ms": rm_ms, "brick_bytes_per_voxel": bpv,
                         "requested_GBps_per_sample_footprint": st["samples"] * 8 * bpv / (rm_ms * 1e-3) / 1e9},
        }
        # HBM traffic per launch measured with rocprofv3 PMC passes (scripts/gpu_prof2.sh -> profiles/*traffic*.json);
        # bench.py cannot collect PMC counters itself, so the committed measurement of this same command is reported.
        traffic, tnote = {}, None
        tpath = os.path.join(ROOT, "profiles", f"traffic_{args.config}_{args.cubemap}.json")
        if world == 1 and os.path.exists(tpath):
            traffic = json.load(open(tpath))
            if traffic.get("kernel_sources_sha") != kernel_sources_sha() or args.no_lds_cubemap or args.no_grey:
                tnote = (f"profiles/{os.path.basename(tpath)} was measured on other kernel sources "
                         f"({traffic.get('kernel_sources_sha')} != {kernel_sources_sha()}) or another code path (A/B switch): not reported")
                traffic = {}
        for name, r in roofs.items():
            r["frac"] = r["achieved"] / r["peak"]
            t = traffic.get(r["kernel"])
            r["traffic"] = t["traffic_bytes"] if t else None
            if t:
                r["traffic_source"] = (f"profiles/{os.path.basename(tpath)} (rocprofv3 FETCH_SIZE x2 + WRITE_SIZE, separate PMC passes of this "
                                       f"command on these kernel sources, sha {traffic['kernel_sources_sha']})")
            elif tnote:
                r["traffic_note"] = tnote
        # what actually limits the two kernels (rocprofv3 PMC passes of this command, profiles/ + DESIGN.md 3.4): not HBM
        roofs["fill"]["limiter"] = (("VALU issue: ~2.0 G wave-level VALU per launch at ~4 cycles each = the whole kernel time (SQ_ACTIVE_INST_VALU x 4 / "
                                     "SIMD-cycles ~ 1.0); the cube map is LDS-resident (LDS pipe ~60 % busy, 3/4 of it bank conflicts); HBM at ~0.9 TB/s")
                                    if lds_path else
                                    ("the CU's single L1/TA path: one divergent wave-wide footprint gather per covered slice (~65 cycles per "
                                     "wave-slice per CU against ~38 of VALU); HBM at ~0.65 TB/s"))
        roofs["raymarch"]["limiter"] = (("VALU issue: ~51 VALU per wave-sample, SQ_ACTIVE_INST_VALU x 4 / SIMD-cycles ~ 0.78, 59 % of the lanes active; two 16-B "
                                         "footprint loads per sample (grey z-pair bricks), L1 at ~60 % of its access rate; HBM at ~2.5 TB/s")
                                        if st.get("brick_format", 0) == 1 else
                                        ("L1 (TCP): four 16-B footprint loads per sample, one lane quad per L1 cycle at best: TCP busy 96 %, ~26.5 L1 "
                                         "cycles per wave-load (profiles/r02_l1_gather_probe.txt); VALU hides underneath; HBM at ~1.6 TB/s"))
        smax = [float(x) for x in stage_max.tolist()]
        # whole-job algorithmic bytes (all ranks): bricks + light map + pair records; bricks sampled + the image
        fill_bytes_job = occupied * (bpv * nv ** 3 + 8 * nv ** 2) + 84 * pairs
        rm_bytes_job = bricks_sampled * bpv * nv ** 3 + 16 * sc.width * sc.height
        dom = "fill" if fill_ms >= rm_ms else "raymarch"
        executed = (voxels, samples)
        if ref_units is not None:
            voxels, samples = float(ref_units[0]), float(ref_units[1])
        # N > 1: the sharded frame against the single-GPU frame rendered on this rank before the timed region
        shard_err = float((image - ref_image).abs().max().item()) if (ref_image is not None and image is not None) else None
        out = {
            "metric": baseline_metric(),
            "value": (voxels + samples) / (dt / args.steps) / 1e6,
            "unit": "M(voxels+samples)/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms_step,
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
            "dtype": "f32 compute / f16 voxel storage", "data": "synthetic",
            "config": {"workload": f"{args.config}: {sc.N[0]}x{sc.N[1]}x{sc.N[2]} metavoxels x {nv}^3 voxels, "
                                   f"{len(sc.particles)} particles, {sc.width}x{sc.height}",
                       "cubemap": ("R8 (8-bit like the reference's asset; LDS-resident in k_fill)" if args.cubemap == "r8" and not args.no_lds_cubemap
                                   else "R8 on the global f32 footprint table" if args.cubemap == "r8" else "f32 texels, global footprint table"),
                       "brick_storage": ("grey z-pair entries: (luminance|density)(z), (luminance|density)(z+1), 8 B/voxel (grey ambient: r = g = b bit for bit)"
                                         if st.get("brick_format", 0) == 1 else "RGBA16F, 8 B/voxel"),
                       "parallelism": f"zslab{world}" + (f" ({args.exchange} exchange)" if world > 1 else ""), "slabs": bounds if world > 1 else None,
                       "occupied_mv": int(occupied), "pairs": int(pairs), "voxels_per_step": int(voxels),
                       "samples_per_step": int(samples),
                       "work_unit": ("voxels + executed samples of the 1-GPU job (fixed for every N)" if (world == 1 or ref_units is not None)
                                     else "voxels + samples executed on all ranks (1-GPU reference job not run: see reference_frame_skipped)"),
                       "samples_executed_all_ranks": int(executed[1]),
                       "max_abs_rgba_diff_vs_1gpu_frame": shard_err, "reference_frame_skipped": ref_skipped},
            # absolute rates: per stage against that stage's kernel time on the slowest rank (fill = local + finish for N > 1),
            # and for the whole frame (everything incl. the exchanges)
            "fill_mvoxels_per_s": voxels / ((smax[1] + smax[3]) * 1e-3) / 1e6,
            "raymarch_msamples_per_s": samples / (smax[2] * 1e-3) / 1e6,
            "frame_mvoxels_per_s": voxels / (dt / args.steps) / 1e6,
            "frame_msamples_per_s": samples / (dt / args.steps) / 1e6,
            "fill_frac_of_hbm_roofline": (fill_bytes_job / ((smax[1] + smax[3]) * 1e-3) / 1e9) / (HBM_PEAK_GBS * world),
            "raymarch_frac_of_hbm_roofline": (rm_bytes_job / (smax[2] * 1e-3) / 1e9) / (HBM_PEAK_GBS * world),
            "stage_ms": {"bin": bin_ms, "fill_kernel": fill_ms, "raymarch_kernel": rm_ms,
                         "fill_finish_kernel": float(np.mean(k_fin)) if k_fin else None},
            "stage_ms_slowest_rank": {"bin": smax[0], "fill_kernel": smax[1], "raymarch_kernel": smax[2],
                                      "fill_finish_kernel": smax[3] if world > 1 else None},
            "roofline": dict(roofs[dom], stage=dom),
            "roofline_all": roofs,
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(sc, args.cpu_threads)
            out["speedup_vs_cpu"] = out["value"] / out["cpu_baseline"]["value"]
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
