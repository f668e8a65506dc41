#!/usr/bin/env python3
"""bench.py -- headline benchmark of the sparse-volumetric-particle hot path on MI355X.

One "step" = one full pass of the hot path over one batch of synthetic input that is already resident in HBM:
    bin (particles -> metavoxels)  ->  fill (+ light propagation)  ->  ray-march (+ inter-metavoxel blend).

Workload at N = 1: BASELINE.json configs[2], the configuration its metric is quoted on:
    32x32x32 metavoxels x 32^3 voxels, 100k particles, 1920x1080 (synthetic scene of SURVEY.md 8(d)).
N > 1: the SAME grid cut into light-axis slabs (BASELINE.json configs[3]) -> strong scaling.  The multi-GPU fan-out lives INSIDE libvpfx
(csrc/multi.cpp: device list in vp_config, one worker thread + HIP stream + RCCL rank per GPU); this script only makes the same
vp_bin_resident / vp_fill / vp_raymarch calls as at N = 1.  Two launch styles, one code path in the library:
    python bench.py --gpus N                          one process drives the N GPUs (ncclCommInitAll) -- what the C# / C host does
    python -m torch.distributed.run ... bench.py --gpus N   one process per GPU (ncclCommInitRank; the unique id travels over a gloo
                                                      group, which also carries the barrier and the max-over-ranks of the timing)
torch.distributed never touches RCCL here: every data-path collective is issued by the library.

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
# RCCL shares device buffers between the ranks' processes through IPC handles; this pool's host driver supports dmabuf IPC only
# (without it: hipIpcGetMemHandle: invalid argument).  Read by the HSA runtime when it starts, i.e. before the first HIP call below.
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

from __graft_entry__ import load_package  # noqa: E402

load_package()
from vpfx_amd import abi, engine as E, scene as S  # noqa: E402

HBM_PEAK_GBS = 8000.0      # MI355X HBM3E spec peak (MI355X_MICROARCH.md); ~6300 GB/s achievable


def baseline_metric():
    """BASELINE.json's metric string verbatim (the file travels with the repo); the same text if it is missing."""
    try:
        return json.load(open(os.path.join(ROOT, "BASELINE.json")))["metric"]
    except Exception:
        return "Mvoxels/s filled + Msamples/s raymarched, 32³×32³ grid @1080p"


def kernel_sources_sha():
    """Fingerprint of the kernel sources: profiles/traffic_<cfg>.json and profiles/limiters_<cfg>.json record the one they were measured on,
    and bench.py only reports those PMC measurements while it still matches (a stale number is worse than null)."""
    import hashlib
    h = hashlib.sha256()
    d = os.path.join(ROOT, "volumetric-particles-for-unity_amd", "csrc")
    for f in sorted(os.listdir(d)):
        # the kernels and every instantiation unit (fill_generic.hip / raymarch_generic.hip decide which kernels run for voxel counts other than
        # 16 / 32 / 64); host-side .cpp files and occluders.hip (not on any profiled counter) do not change a counter
        if f in ("fill.hip", "fill_generic.hip", "fill_kernels.h", "raymarch.hip", "raymarch_generic.hip", "raymarch_kernels.h", "bin.hip", "vpfx_internal.h"):
            h.update(f.encode() + b"\0" + open(os.path.join(d, f), "rb").read())
    return h.hexdigest()[:16]


def cpu_quota_cores():
    """CPU bandwidth limit of this container in cores (cgroup v2 cpu.max / v1 cfs quota), or None: 128 OpenMP threads on a 16-core quota
    explain a poor parallel speed-up of the CPU leg better than anything in its code."""
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        return None if q == "max" else float(q) / float(per)
    except Exception:
        pass
    try:
        q = float(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
        per = float(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
        return None if q <= 0 else q / per
    except Exception:
        return None


def cpu_baseline(sc, threads=0, boxes=None):
    """Time the CPU oracle (a port of the reference's algorithm; the reference itself is C# + HLSL and cannot run
    here) on the GPU box's host cores, compiled -march=native on that box.  Reported, not shipped: this is the only place bench.py
    touches oracle/."""
    from oracle import oracle as O
    quota = cpu_quota_cores()
    if not threads and quota:
        # the container's CPU bandwidth is capped (16 cores on the GPU boxes, with 256 CPUs visible): more OpenMP threads than that only
        # take turns being throttled -- the honest core count of this leg is the quota
        threads = max(1, int(quota + 0.5))
    o = O.Oracle(sc.config(), threads=threads, native=True)
    if boxes is not None:
        o.set_occluders(boxes)
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
        "value": units / (t3 - t0) / 1e6, "unit": "M(voxels+samples)/s", "cores": threads or o.L.vpo_max_threads(),
        "kind": "port", "build": "gcc -O3 -march=native -fopenmp, built on this box",
        "cpus_visible": len(os.sched_getaffinity(0)), "cgroup_cpu_quota_cores": quota,
        "sample": f"one full step of {sc.name} (bin {t1 - t0:.3f}s, fill {t2 - t1:.3f}s, raymarch {t3 - t2:.3f}s)",
        "fill_mvoxels_per_s": st["voxels_filled"] / (t2 - t1) / 1e6,
        "raymarch_msamples_per_s": st["samples"] / (t3 - t2) / 1e6,
        "seconds": t3 - t0,
    }
    o.close()
    # single-thread leg: ONE light-axis slice of the same workload (the middle one) on one thread, extrapolated
    # to the whole step with the per-unit rates (a full single-thread step would take minutes)
    zmid = sc.N[2] // 2
    o1 = O.Oracle(sc.config(slab=(zmid, zmid + 1)), threads=1, native=True)
    if boxes is not None:
        o1.set_occluders(boxes)
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
        "note": ("the parallel speed-up of the port over its own single thread is bounded by the container's CPU quota (cgroup_cpu_quota_cores), "
                 "not by the box's CPU count; reported for the record -- the roofline fraction, not the GPU/CPU ratio, says how good the kernels are"),
    }
    o1.close()
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=300, help="timed steps (default: >= 1 s of timed region at ~4 ms per step)")
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--config", default="C3", help="scene name from vpfx_amd.scene.CONFIGS (default: the metric's config)")
    ap.add_argument("--cubemap", default="r8", choices=["r8", "f32"],
                    help="displacement cube map texel format: r8 = 8-bit like the reference's asset (LDS-resident in k_fill), f32 = float texels")
    ap.add_argument("--no-lds-cubemap", action="store_true", help="A/B: keep an R8 cube map on the global f32 footprint table")
    ap.add_argument("--no-grey", action="store_true", help="A/B: keep RGBA16F bricks although the ambient colour is grey")
    ap.add_argument("--displacement-scale", type=float, default=None,
                    help="override the scene's _DisplacementScale (default 0.7, scene:9016); 1.0 = the slider's maximum (smoothstep jump at net displacement 0)")
    ap.add_argument("--event-stride", type=int, default=0, help="read the stages' HIP-event kernel times every n-th timed step (0 = steps // 64, at least 1)")
    ap.add_argument("--no-formula-count", action="store_true", help="skip the one untimed ray-march without early-out that counts SURVEY 8(d)'s formula samples")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-threads", type=int, default=0)
    ap.add_argument("--share-gpu", action="store_true",
                    help="functional test only: all N slabs on cuda:0 through the library's peer-copy test hook (no RCCL, nothing to scale)")
    ap.add_argument("--exchange", default="tiles", choices=["tiles", "all_gather"],
                    help="N > 1 image exchange: all-to-all of screen pieces + sharded blend + gather (default) or one all-gather of whole partial images")
    ap.add_argument("--rm-groups", type=int, default=0, help="N > 1: groups of the ray-march saturation hand-off (1 = none, N = serial chain, 0 = library default)")
    ap.add_argument("--uniform-slabs", action="store_true", help="equal-thickness slabs instead of the work-balanced cut")
    ap.add_argument("--no-reference-frame", action="store_true", help="N > 1: do not render the 1-GPU frame on rank 0 for the shard check")
    ap.add_argument("--no-variants", action="store_true",
                    help="skip the untimed content variants after the timed region (float cube map, coloured ambient, +x view: `variants` in the JSON line)")
    args = ap.parse_args()
    # TEST HOOKS (tests/test_gpu_bench_cli.py runs the DRIVER'S command line byte for byte on a one-GPU box, so the switches cannot be flags):
    #   VPFX_BENCH_SHARE_GPU=1        = --share-gpu
    #   VPFX_BENCH_TEST_FAIL_RANK=r   rank r dies (os._exit) in the middle of the timed region: the job must exit non-zero, not hang
    if os.environ.get("VPFX_BENCH_SHARE_GPU") == "1":
        args.share_gpu = True
    fail_rank = int(os.environ.get("VPFX_BENCH_TEST_FAIL_RANK", "-1"))
    # stdout carries the JSON line and nothing else: everything any library prints there (gloo announces its connections on stdout from C++,
    # one line per rank, interleaved) is sent to stderr for the rest of the run; rank 0 writes the line to the saved descriptor at the end
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the product has no CPU path)")
    env_world = int(os.environ.get("WORLD_SIZE", "1"))
    N = args.gpus
    multi_process = env_world > 1
    if multi_process and env_world != N:
        raise SystemExit(f"--gpus {N} but WORLD_SIZE={env_world}")
    rank = int(os.environ.get("RANK", "0")) if multi_process else 0
    local_rank = int(os.environ.get("LOCAL_RANK", "0")) if multi_process else 0
    if args.share_gpu and multi_process:
        # rendezvous check only: every rank on cuda:0.  RCCL refuses two ranks on one GPU ("Duplicate GPU detected") -- AFTER its bootstrap has
        # exchanged the peers' information, so getting that error back as VP_ERR_RCCL proves that the unique id travelled, that the ranks found
        # each other and that failures surface instead of hanging (scripts/gpu_rendezvous_check.sh); the peer-copy hook needs one process
        local_rank = 0
    ngpu = torch.cuda.device_count()
    if N > 1 and not multi_process and not args.share_gpu and ngpu < N:
        raise SystemExit(f"--gpus {N} but only {ngpu} GPU(s) visible (functional run on one GPU: add --share-gpu)")
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    if multi_process:
        dist.init_process_group("gloo")          # control plane only: unique id, barrier, timing reduction.  RCCL belongs to the library.

    demo_boxes = None
    if args.config == "DEMO":
        # the reference's own scene (Assets/Volumetric_Particle_System.unity:9013-9026, 6118): 10^3 metavoxels x 32^3 voxels, the demo
        # emitter's <= 60 particles, 1024 x 768, ground / wall / cubes as occluders, bin + fill every 2nd frame (updateInterval 2, VPR.cs:186)
        sc, _, demo_boxes = S.make_demo_scene()
        if args.cubemap == "r8":
            sc.cubemap = S.make_cubemap_r8()
    else:
        sc = S.make_scene(args.config, cubemap=args.cubemap)
    if args.displacement_scale is not None:
        sc.displacement_scale = float(args.displacement_scale)

    flags = (abi.VP_MULTI_EXCHANGE_ALL_GATHER if args.exchange == "all_gather" else 0) | (abi.VP_MULTI_UNIFORM_SLABS if args.uniform_slabs else 0)
    launch = "single GPU"
    if N == 1:
        cfg = sc.config(device=local_rank)
    elif multi_process:
        uid = [E.rccl_unique_id() if rank == 0 else None]
        dist.broadcast_object_list(uid, src=0)
        cfg = sc.config(devices=[local_rank], world_size=N, first_rank=rank, multi_flags=flags, rm_groups=args.rm_groups, rccl_unique_id=uid[0])
        launch = f"one process per GPU ({N} ranks, torch.distributed.run), RCCL inside libvpfx (ncclCommInitRank)"
    elif args.share_gpu:
        cfg = sc.config(devices=[0] * N, multi_flags=flags | abi.VP_MULTI_PEER_COPY, rm_groups=args.rm_groups)
        launch = f"one process, {N} slabs on ONE GPU through the peer-copy test hook (functional run: no RCCL, no scaling)"
    else:
        cfg = sc.config(devices=list(range(N)), multi_flags=flags, rm_groups=args.rm_groups)
        launch = f"one process drives {N} GPUs, RCCL inside libvpfx (ncclCommInitAll)"
    if args.no_lds_cubemap:
        cfg.reserved[0] = 1            # VPFX_CFG_NO_LDS_CUBEMAP
    if args.no_grey:
        cfg.reserved[1] = 1            # VPFX_CFG_NO_GREY_BRICKS

    # strong scaling: the unit of work is the SINGLE-GPU job (its voxels and its executed samples).  Sharded runs may execute
    # more lattice samples in total (slabs of one hand-off group do not see each other's opacity), which must not inflate `value`.
    ref_units, ref_image, ref_skipped = None, None, None
    if N > 1 and rank == 0:
        # The 1-GPU reference frame is rendered on rank 0's GPU only if the whole grid's brick pool fits NEXT TO the slab engine(s)
        # that GPU also hosts: at config 5 it is ~190 GB and must not be attempted.
        probe = E.Engine(sc.config(device=local_rank))
        probe.set_frame(sc.light_to_world, sc.grid_center)
        probe.bin(sc.particles, sc.layout, sc.psys_local_to_world)
        whole_occupied = probe.stats()["occupied_mv"]
        probe.close()
        brick = 8 * sc.nv ** 3
        share = N if args.share_gpu else 1
        need_ref = whole_occupied * brick * 1.125 + 2e9
        need_slab = 3.0 * (whole_occupied / N) * share * 1.5 * brick * 1.125 + 2e9      # bricks + 2x scratch, 1.5x for an uneven slab
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

    eng = E.Engine(cfg)
    if demo_boxes is not None:
        eng.set_occluders(demo_boxes)
    eng.set_frame(sc.light_to_world, sc.grid_center)
    eng.upload_particles(sc.particles, sc.layout, sc.psys_local_to_world)      # inputs resident in HBM from here on
    fp_first, fp = sc.fill_params(), sc.fill_params()
    fp.cubemap = None                                                           # cubemap stays resident after the first fill
    cam, rp = sc.camera(), sc.raymarch_params()
    image = torch.empty((sc.height, sc.width, 4), device=device)                # particlesRT on the display GPU (rank 0)

    update_interval = 2 if args.config == "DEMO" else 1
    frame_no = [0]

    def step(first=False):
        if first or frame_no[0] % update_interval == 0:                         # Time.frameCount % updateInterval == 0   VPR.cs:186
            eng.bin_resident()
            eng.fill(fp_first if first else fp)
        eng.raymarch_device(cam, rp, image.data_ptr())
        frame_no[0] += 1

    sync_devices = list(range(N)) if (N > 1 and not multi_process and not args.share_gpu) else [local_rank]

    def barrier():
        eng.sync()
        if multi_process:
            dist.barrier()
        for d in sync_devices:
            torch.cuda.synchronize(d)

    step(first=True)
    # SURVEY 8(d) defines Msamples/s on the FORMULA count (sum over pixels and metavoxels of max(0, tExit - tEntry + 1)): one untimed ray-march
    # with the saturation early-out off (VP_RM_NO_EARLY_OUT) executes exactly those samples on the resident bricks.  The timed frames keep the
    # early-out (same image), so both counts and both rates are reported.
    samples_formula_local = 0
    if not args.no_formula_count:      # (profiling runs leave it out: the extra launch has the same kernel name and would sit in rocprofv3's per-kernel averages)
        rp_all = sc.raymarch_params()
        rp_all.flags |= abi.VP_RM_NO_EARLY_OUT
        eng.raymarch_device(cam, rp_all, image.data_ptr())
        eng.sync()
        samples_formula_local = eng.stats()["samples"]
    # N > 1: re-cut the slabs twice from the measured work (pairs per slice, samples executed per slice, kernel times), then keep the cut:
    # vp_rebalance makes the NEXT ray-march record its per-slice samples and the bin after that re-cut, hence the extra step at the end
    for i in range(max(args.warmup - 1, 3 if N > 1 else 0)):
        if N > 1 and i < 2 and not args.uniform_slabs:
            eng.rebalance()
        step()
    barrier()
    k_fill, k_rm, k_bin, k_fin = [], [], [], []
    frame_no[0] = 0
    t0 = time.perf_counter()
    # HIP-event durations of the dominant kernels are READ inside the timed region, on the stream they were launched on (N > 1: the slowest
    # local rank) -- but reading one waits for that launch (hipEventSynchronize), which stops the host from running ahead of the GPU: three
    # waits per frame are a third of a 0.3 ms frame.  So they are read on every stride-th step (>= 64 samples of every stage per run).
    stride = args.event_stride if args.event_stride > 0 else max(1, args.steps // 64)
    for i in range(args.steps):
        if rank == fail_rank and i == args.steps // 2:
            print(f"[bench] TEST HOOK: rank {rank} leaves the job at timed step {i}", file=sys.stderr, flush=True)
            os._exit(17)
        step()
        if i % stride == stride - 1 or i == args.steps - 1:
            k_bin.append(eng.last_kernel_ms(0)); k_fill.append(eng.last_kernel_ms(1)); k_rm.append(eng.last_kernel_ms(2))
            if N > 1:
                k_fin.append(eng.last_kernel_ms(3))
    barrier()
    dt = time.perf_counter() - t0
    st = eng.stats()
    info = eng.multi_info()
    t = torch.tensor([dt], dtype=torch.float64)
    counts = torch.tensor([st["voxels_filled"], st["samples"], st["occupied_mv"], st["pairs"], st["bricks_sampled"], samples_formula_local], dtype=torch.float64)
    stage_max = torch.tensor([float(np.mean(k_bin)), float(np.mean(k_fill)), float(np.mean(k_rm)), float(np.mean(k_fin)) if k_fin else 0.0],
                             dtype=torch.float64)
    per_rank = torch.zeros((max(N, 1), 5), dtype=torch.float64)                 # samples + the four stage kernel times of every rank
    if N > 1:
        for r in range(info["first_rank"], info["first_rank"] + info["num_local"]):
            per_rank[r, 0] = info["samples"][r]
            per_rank[r, 1:] = torch.tensor(info["stage_ms"][r], dtype=torch.float64)
    if multi_process:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dist.all_reduce(counts, op=dist.ReduceOp.SUM)
        dist.all_reduce(stage_max, op=dist.ReduceOp.MAX)          # slowest rank per stage (kernel time, HIP events)
        dist.all_reduce(per_rank, op=dist.ReduceOp.SUM)
    dt = float(t.item())
    voxels, samples, occupied, pairs, bricks_sampled, samples_formula = [float(x) for x in counts.tolist()]

    # The headline above is the friendliest content combination the reference's defaults allow (8-bit cube map that fits LDS, grey ambient,
    # the benchmark view).  The same frame under the inspector settings / views that take the other code paths, 20 steps each AFTER the timed
    # region (never part of `value`): float cube-map texels (k_fill on the global footprint table), a coloured ambientColor (VPR.cs:90,537:
    # RGBA16F bricks instead of grey z-pair entries -- four loads per sample in the ray-march instead of two), the view from +x (brick rows run
    # along grid x: the least coherent footprint loads).
    variants = None
    if N == 1 and not args.no_variants and args.config != "DEMO":
        variants = {}

        def run_variant(name, fill_first=None, fill_next=None, camera=None, note=""):
            cam_v = camera if camera is not None else cam
            f_first, f_next = fill_first or fp_first, fill_next or fp
            eng.bin_resident(); eng.fill(f_first); eng.raymarch_device(cam_v, rp, image.data_ptr())
            for _ in range(2):
                eng.bin_resident(); eng.fill(f_next); eng.raymarch_device(cam_v, rp, image.data_ptr())
            barrier()
            tv = time.perf_counter()
            kf, kr = [], []
            for i in range(20):
                eng.bin_resident(); eng.fill(f_next); eng.raymarch_device(cam_v, rp, image.data_ptr())
                if i % 5 == 4:
                    kf.append(eng.last_kernel_ms(1)); kr.append(eng.last_kernel_ms(2))
            barrier()
            ms = (time.perf_counter() - tv) / 20 * 1e3
            sv = eng.stats()
            variants[name] = {"fill_ms": float(np.mean(kf)), "raymarch_ms": float(np.mean(kr)), "ms_per_step": ms,
                              "samples_per_step": int(sv["samples"]), "brick_format": "grey z-pair" if sv.get("brick_format", 0) == 1 else "RGBA16F",
                              "value": (sv["voxels_filled"] + sv["samples"]) / (ms * 1e-3) / 1e6, "what": note}

        if args.cubemap == "r8":
            sc_f = S.make_scene(args.config, cubemap="f32")
            if args.displacement_scale is not None:
                sc_f.displacement_scale = float(args.displacement_scale)
            f1 = sc_f.fill_params()
            f2 = sc_f.fill_params(); f2.cubemap = None
            run_variant("cubemap_f32", f1, f2, note="float cube-map texels: k_fill on the global footprint table instead of the LDS-resident k_fill_lds")
        amb_first, amb_next = sc.fill_params(), sc.fill_params()
        for q in (amb_first, amb_next):
            q.ambient[0], q.ambient[1], q.ambient[2] = 0.25, 0.2, 0.15
        amb_next.cubemap = None
        run_variant("coloured_ambient", amb_first, amb_next, note="ambientColor (0.25, 0.2, 0.15): RGBA16F bricks, four footprint loads per sample")
        D = 0.8 * sc.N[0] * sc.mv_scale
        sc_x = S.make_scene(args.config, cubemap=args.cubemap)
        sc_x.set_camera((D, 0.05 * D, 0.125 * D))
        run_variant("view_plus_x", fp_first, fp, camera=sc_x.camera(), note="camera on the grid's +x axis (brick rows run along x)")
        eng.bin_resident(); eng.fill(fp_first)                      # leave the context as the timed region did
        barrier()
        # the same grid, particles and view with a voxel count that is none of 16 / 32 / 64 (VPR.cs:84 is a free inspector int): the run-time-nv
        # instantiations (fill_generic.hip / raymarch_generic.hip) on a second context, 20 steps
        if sc.nv in (16, 32, 64) and f"{args.config}nv24" in S.CONFIGS:
            sc_g = S.make_scene(f"{args.config}nv24", cubemap=args.cubemap)
            eg = E.Engine(sc_g.config(device=local_rank))
            eg.set_frame(sc_g.light_to_world, sc_g.grid_center)
            eg.upload_particles(sc_g.particles, sc_g.layout, sc_g.psys_local_to_world)
            g_first, g_next = sc_g.fill_params(), sc_g.fill_params()
            g_next.cubemap = None
            cam_g, rp_g = sc_g.camera(), sc_g.raymarch_params()
            eg.bin_resident(); eg.fill(g_first); eg.raymarch_device(cam_g, rp_g, image.data_ptr())
            for _ in range(2):
                eg.bin_resident(); eg.fill(g_next); eg.raymarch_device(cam_g, rp_g, image.data_ptr())
            eg.sync(); torch.cuda.synchronize()
            tv = time.perf_counter()
            kf, kr = [], []
            for i in range(20):
                eg.bin_resident(); eg.fill(g_next); eg.raymarch_device(cam_g, rp_g, image.data_ptr())
                if i % 5 == 4:
                    kf.append(eg.last_kernel_ms(1)); kr.append(eg.last_kernel_ms(2))
            eg.sync(); torch.cuda.synchronize()
            ms = (time.perf_counter() - tv) / 20 * 1e3
            sg = eg.stats()
            variants["generic_nv24"] = {
                "fill_ms": float(np.mean(kf)), "raymarch_ms": float(np.mean(kr)), "ms_per_step": ms, "samples_per_step": int(sg["samples"]),
                "voxels_per_step": int(sg["voxels_filled"]), "brick_format": "grey z-pair" if sg.get("brick_format", 0) == 1 else "RGBA16F",
                "value": (sg["voxels_filled"] + sg["samples"]) / (ms * 1e-3) / 1e6,
                "fill_ns_per_kvoxel": float(np.mean(kf)) * 1e6 / (sg["voxels_filled"] / 1e3), "headline_fill_ns_per_kvoxel": float(np.mean(k_fill)) * 1e6 / (voxels / 1e3),
                "what": f"{sc_g.N[0]}^3 metavoxels x 24^3 voxels (same particles and view): the run-time voxel-count kernels (GEN k_fill_lds / k_raymarch<0>)"}
            eg.close()

    if rank == 0:
        ms_step = dt / args.steps * 1e3
        nv = sc.nv
        fill_ms, rm_ms, bin_ms = float(np.mean(k_fill)), float(np.mean(k_rm)), float(np.mean(k_bin))
        # algorithmic bytes; bricks are counted at the bytes per voxel the context actually stores (8 in both formats)
        lds_path = args.cubemap == "r8" and not args.no_lds_cubemap
        bpv = st.get("brick_bytes_per_voxel", 8)
        fill_bytes = occupied * (bpv * nv ** 3 + 8 * nv ** 2) + 84 * pairs
        rm_bytes = bricks_sampled * bpv * nv ** 3 + 16 * sc.width * sc.height
        smax = [float(x) for x in stage_max.tolist()]
        fill_t = (smax[1] + smax[3]) * 1e-3                         # fill = local pass + finish pass on the slowest rank for N > 1
        roofs = {
            "fill": {"bound": "hbm", "kernel": "k_fill_lds" if lds_path else "k_fill", "achieved": fill_bytes / fill_t / 1e9 / N,
                     "peak": HBM_PEAK_GBS, "unit": "GB/s", "bytes_per_launch": fill_bytes / N, "avg_ms": fill_t * 1e3,
                     "brick_bytes_per_voxel": bpv},
            "raymarch": {"bound": "hbm", "kernel": "k_raymarch", "achieved": rm_bytes / (smax[2] * 1e-3) / 1e9 / N, "peak": HBM_PEAK_GBS,
                         "unit": "GB/s", "bytes_per_launch": rm_bytes / N, "avg_ms": smax[2], "brick_bytes_per_voxel": bpv,
                         "requested_GBps_per_sample_footprint": samples * 8 * bpv / (smax[2] * 1e-3) / 1e9 / N},
        }
        # HBM traffic per launch and the issue-side counters, measured with rocprofv3 PMC passes of this same command
        # (scripts/gpu_prof_r3.sh -> profiles/traffic_<cfg>_<cubemap>.json, limiters_<cfg>_<cubemap>.json); bench.py cannot collect PMC
        # counters itself, so the committed measurement is reported -- only while the kernel sources still have the fingerprint it was
        # taken on and no A/B switch changes the code path; otherwise null.
        sha = kernel_sources_sha()
        ab = args.no_lds_cubemap or args.no_grey or args.displacement_scale is not None

        def measured(kind):
            path = os.path.join(ROOT, "profiles", f"{kind}_{args.config}_{args.cubemap}.json")
            if N != 1 or not os.path.exists(path):
                return {}, None
            d = json.load(open(path))
            if d.get("kernel_sources_sha") != sha or ab:
                return {}, (f"profiles/{os.path.basename(path)} was measured on other kernel sources ({d.get('kernel_sources_sha')} != {sha}) "
                            f"or another code path (A/B switch): not reported")
            return d, None
        traffic, tnote = measured("traffic")
        limiters, lnote = measured("limiters")
        for name, r in roofs.items():
            r["frac"] = r["achieved"] / r["peak"]
            tr = traffic.get(r["kernel"])
            r["traffic"] = tr["traffic_bytes"] if tr else None
            # bench.py cannot collect PMC counters: everything below (traffic, limiter, valu_issue) is read from profiles/*.json, measured with
            # rocprofv3 on these kernel sources (fingerprint-checked) in an EARLIER run of this same command -- not in this run
            r["measured_in_this_run"] = {"achieved": True, "avg_ms": True, "traffic": False, "limiter": False, "valu_issue": False}
            if tr:
                r["traffic_source"] = (f"profiles/traffic_{args.config}_{args.cubemap}.json (rocprofv3 FETCH_SIZE x2 + WRITE_SIZE, separate PMC passes of "
                                       f"this command on these kernel sources, sha {sha})")
            elif tnote:
                r["traffic_note"] = tnote
            # what limits the kernel instead of HBM: counter-derived figures read from profiles/, or null (no prose that could go stale)
            r["limiter"] = limiters.get(r["kernel"])
            if lnote and not r["limiter"]:
                r["limiter_note"] = lnote
            # second roofline: VALU issue (what actually bounds these kernels) -- class counters x per-class issue cycles, <= 1
            r["valu_issue"] = (r["limiter"] or {}).get("valu_issue")
        dom = "fill" if fill_t * 1e3 >= smax[2] else "raymarch"
        executed = (voxels, samples)
        if ref_units is not None:
            voxels, samples = float(ref_units[0]), float(ref_units[1])
        # N > 1: the sharded frame against the single-GPU frame rendered on this rank before the timed region
        shard_err = float((image - ref_image).abs().max().item()) if ref_image is not None else None
        ex = info["exchange_ms"]
        out = {
            "metric": baseline_metric(),
            "value": (voxels / update_interval + samples) / (dt / args.steps) / 1e6,        # (DEMO: the fill runs every 2nd frame)
            "unit": "M(voxels+samples)/s",
            "n_gpus": N, "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms_step, "kernel_time_samples_per_stage": len(k_fill),
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
            "dtype": "f32 compute / f16 voxel storage", "data": "synthetic",
            "config": {"workload": f"{args.config}: {sc.N[0]}x{sc.N[1]}x{sc.N[2]} metavoxels x {nv}^3 voxels, "
                                   f"{len(sc.particles)} particles, {sc.width}x{sc.height}",
                       "update_interval": update_interval,
                       "cubemap": ("R8 (8-bit like the reference's asset; LDS-resident in k_fill)" if args.cubemap == "r8" and not args.no_lds_cubemap
                                   else "R8 on the global f32 footprint table" if args.cubemap == "r8" else "f32 texels, global footprint table"),
                       "displacement_scale": sc.displacement_scale,
                       "brick_storage": ("grey z-pair entries: (luminance|density)(z), (luminance|density)(z+1), 8 B/voxel (grey ambient: r = g = b bit for bit)"
                                         if st.get("brick_format", 0) == 1 else "RGBA16F, 8 B/voxel"),
                       "parallelism": f"zslab{N}" + (f" ({info['exchange']} exchange, {info['rm_groups']} hand-off group(s))" if N > 1 else ""),
                       "launch": launch,
                       "rccl_ranks": info["rccl_ranks"] if N > 1 else None,        # ncclCommCount of the library's communicator
                       "slabs": [[a, b] for a, b in zip(info["slab_cuts"], info["slab_cuts"][1:])] if N > 1 else None,
                       "occupied_mv": int(occupied), "pairs": int(pairs), "voxels_per_step": int(voxels),
                       "samples_per_step": int(samples),
                       "samples_executed": int(samples), "samples_formula": int(samples_formula) if samples_formula else None,
                       "work_unit": ("voxels + executed samples of the 1-GPU job (fixed for every N)" if (N == 1 or ref_units is not None)
                                     else "voxels + samples executed on all ranks (1-GPU reference job not run: see reference_frame_skipped)"),
                       "samples_executed_all_ranks": int(executed[1]),
                       "max_abs_rgba_diff_vs_1gpu_frame": shard_err, "reference_frame_skipped": ref_skipped},
            # absolute rates: per stage against that stage's kernel time on the slowest rank (fill = local + finish for N > 1),
            # and for the whole frame (everything incl. the exchanges)
            # Mvoxels/s filled = occupied_MVs * nv^3 / t_fill with t_fill = bin + density + propagate (device time)
            "fill_mvoxels_per_s": voxels / (smax[0] * 1e-3 + fill_t) / 1e6,
            "fill_kernel_only_mvoxels_per_s": voxels / fill_t / 1e6,
            "raymarch_msamples_per_s": samples / (smax[2] * 1e-3) / 1e6,                    # executed samples (the saturation early-out skips the hidden ones)
            "raymarch_msamples_per_s_formula": (samples_formula / (smax[2] * 1e-3) / 1e6) if samples_formula else None,    # SURVEY 8(d)'s count (what the CPU leg executes) over the same kernel time
            "value_formula_units": ((voxels / update_interval + samples_formula) / (dt / args.steps) / 1e6) if samples_formula else None,   # the frame's work in the units of the CPU leg
            "frame_mvoxels_per_s": voxels / (dt / args.steps) / 1e6,
            "frame_msamples_per_s": samples / (dt / args.steps) / 1e6,
            "fill_frac_of_hbm_roofline": roofs["fill"]["frac"],
            "raymarch_frac_of_hbm_roofline": roofs["raymarch"]["frac"],
            "stage_ms": {"bin": bin_ms, "fill_kernel": fill_ms, "raymarch_kernel": rm_ms,
                         "fill_finish_kernel": float(np.mean(k_fin)) if k_fin else None},
            "stage_ms_slowest_rank": {"bin": smax[0], "fill_kernel": smax[1], "raymarch_kernel": smax[2],
                                      "fill_finish_kernel": smax[3] if N > 1 else None},
            # N > 1: stream time of the exchanges on rank 0 (last step): tau all-gather, waiting for the saturation hand-off, image exchange + blend
            "exchange_ms": {"tau_all_gather": ex[0], "handoff_wait": ex[1], "t_blend_image_exchange_and_blend": ex[2]} if N > 1 else None,
            "per_rank": ({"samples": [int(x) for x in per_rank[:, 0].tolist()],
                          "kernel_ms_bin_fill_raymarch_finish": [[round(float(v), 4) for v in row[1:]] for row in per_rank.tolist()],
                          "handoff_chain_front_to_back": info["chain"], "handoff_group_of_rank": info["group_of"]} if N > 1 else None),
            "roofline": dict(roofs[dom], stage=dom),
            "roofline_all": roofs,
            # untimed content variants of the same frame (20 steps each after the timed region; see above) -- the cliffs next to the headline
            "variants": variants,
        }
        if N == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(sc, args.cpu_threads, demo_boxes)   # (one frame WITH bin + fill; DEMO refills every 2nd frame)
            # equal units on both sides: the CPU port executes every formula sample, so the GPU frame is credited with the same work
            out["speedup_vs_cpu"] = (out["value_formula_units"] or out["value"]) / out["cpu_baseline"]["value"]
            out["speedup_vs_cpu_note"] = ("(voxels + FORMULA samples) per second, GPU frame / CPU port; in executed samples the GPU figure is `value` "
                                          f"({out['value'] / out['cpu_baseline']['value']:.0f}x), which credits the early-out with nothing")
        sys.stdout.flush()
        os.write(json_fd, (json.dumps(out) + "\n").encode())
    eng.close()
    if multi_process:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
