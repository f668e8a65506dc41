"""GPU: bench.py as the driver runs it -- a bare `python bench.py ...` must start by itself for every --gpus N it accepts on this box, print ONE
JSON line with the contract's keys, and the N > 1 line must say which launch style ran, how many RCCL ranks the library's communicator reported
and how far the sharded frame is from the 1-GPU frame."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
KEYS = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config",
        "roofline")


def _bench(*args):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--config", "C1", "--steps", "4", "--warmup", "2"] + list(args), cwd=ROOT, env=env,
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-1500:]
    lines = r.stdout.splitlines()
    assert len(lines) == 1 and lines[0].startswith("{"), r.stdout[-1500:]     # the JSON line and nothing else on stdout
    d = json.loads(lines[0])
    for k in KEYS:
        assert k in d, k
    return d


def test_single_gpu_line_has_roofline_and_cpu_baseline():
    d = _bench()
    assert d["n_gpus"] == 1 and d["config"]["launch"] == "single GPU" and d["config"]["rccl_ranks"] is None
    rf = d["roofline"]
    assert rf["bound"] == "hbm" and rf["peak"] == 8000.0 and 0 < rf["frac"] < 1 and abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-12
    cb = d["cpu_baseline"]
    assert cb["kind"] == "port" and cb["cores"] >= 1 and cb["value"] > 0 and "single_thread" in cb
    assert d["stage_ms"]["bin"] + d["stage_ms"]["fill_kernel"] + d["stage_ms"]["raymarch_kernel"] <= d["ms_per_step"] * 1.02
    # the content variants next to the headline: the other code paths of the same frame, measured after the timed region
    v = d["variants"]
    assert set(v) == {"cubemap_f32", "coloured_ambient", "view_plus_x"}          # (+ generic_nv24 at C3, the only config with an nv24 twin)
    for name, x in v.items():
        assert x["fill_ms"] > 0 and x["raymarch_ms"] > 0 and x["ms_per_step"] >= x["fill_ms"] + x["raymarch_ms"], name
    assert v["coloured_ambient"]["brick_format"] == "RGBA16F" and v["cubemap_f32"]["brick_format"] == "grey z-pair"
    assert d["config"]["brick_storage"].startswith("grey")           # the headline's own storage is untouched by the variants


@pytest.mark.parametrize("n,exchange", [(2, "tiles"), (4, "all_gather")])
def test_self_launching_fanout_on_one_gpu(n, exchange):
    d = _bench("--gpus", str(n), "--share-gpu", "--exchange", exchange, "--no-cpu-baseline")
    c = d["config"]
    assert d["n_gpus"] == n and c["rccl_ranks"] == 0 and "peer-copy test hook" in c["launch"]
    assert len(c["slabs"]) == n and c["slabs"][0][0] == 0 and c["slabs"][-1][1] == 8
    assert c["max_abs_rgba_diff_vs_1gpu_frame"] is not None and c["max_abs_rgba_diff_vs_1gpu_frame"] <= 2e-5
    assert d["exchange_ms"]["t_blend_image_exchange_and_blend"] > 0 and len(d["per_rank"]["samples"]) == n
    assert d["scaling"] == "strong" and exchange in c["parallelism"]


def test_more_gpus_than_the_box_has_is_refused_loudly():
    import torch
    n = torch.cuda.device_count() + 1
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(n), "--config", "C1", "--steps", "2"], cwd=ROOT,
                       capture_output=True, text=True, timeout=300)
    assert r.returncode != 0 and "--share-gpu" in (r.stdout + r.stderr)


def test_one_process_per_gpu_launch_meets_in_rccl_and_fails_cleanly_on_one_gpu():
    """The driver's launch style (`python -m torch.distributed.run --nproc-per-node N bench.py --gpus N`) with two ranks on this ONE GPU: RCCL
    must refuse the communicator -- "Duplicate GPU detected", which it can only say after its bootstrap exchanged the ranks' information, i.e. after
    rank 0's unique id reached rank 1 over the gloo group and the two met.  So VP_ERR_RCCL from both ranks within seconds proves the rendezvous path
    (vp_rccl_unique_id -> broadcast -> vp_config.rccl_unique_id -> ncclCommInitRank) and that a communicator failure surfaces instead of hanging."""
    import socket
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    env = dict({k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}, MASTER_ADDR="127.0.0.1", NCCL_DEBUG="WARN")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                        "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--share-gpu", "--config", "C1", "--steps", "2"],
                       cwd=ROOT, env=env, capture_output=True, text=True, timeout=300)
    out = r.stdout + r.stderr
    assert r.returncode != 0
    assert out.count("VP_ERR_RCCL") >= 2, out[-3000:]                      # both ranks
    assert "Duplicate GPU detected" in out, out[-3000:]


TOOLS = os.path.join(ROOT, "tests", "tools")
MP_SHIM = os.path.join(TOOLS, "_build", "libfake_rccl_mp.so")


def _build_mp_shim():
    src = os.path.join(TOOLS, "fake_rccl_mp.cpp")
    if os.path.exists(MP_SHIM) and os.path.getmtime(MP_SHIM) >= os.path.getmtime(src):
        return
    os.makedirs(os.path.dirname(MP_SHIM), exist_ok=True)
    subprocess.run(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-D__HIP_PLATFORM_AMD__", "-I/opt/rocm/include", src, "-o", MP_SHIM,
                    "-L/opt/rocm/lib", "-lamdhip64", "-lpthread"], check=True)


@pytest.mark.parametrize("n,exchange", [(2, "tiles"), (4, "all_gather")])
def test_one_process_per_gpu_launch_runs_to_its_json_line_on_the_multiprocess_standin(n, exchange):
    """The command the driver measures scaling with -- `python -m torch.distributed.run --nproc-per-node N bench.py --gpus N` -- run to its end on
    this ONE GPU: N processes, each with its own context (num_devices 1, world_size N, first_rank = RANK), the unique id from rank 0 over gloo,
    ncclCommInitRank inside libvpfx.  The real librccl refuses ranks that share a device (previous test), so libvpfx is pointed (VPFX_RCCL_LIBRARY)
    at tests/tools/fake_rccl_mp.cpp, which stages the bytes through a shared segment between the processes under RCCL's matching rules.  Checked:
    one JSON line from rank 0 with the contract's keys, N RCCL ranks, the library's slab cut, per-rank figures from every process, and the sharded
    frame equal to the 1-GPU frame rendered on rank 0 before the timed region."""
    _build_mp_shim()
    env = dict({k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}, MASTER_ADDR="127.0.0.1",
               VPFX_RCCL_LIBRARY=MP_SHIM, FAKE_RCCL_TIMEOUT_MS="120000")
    r = _run_driver(lambda port: [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr", "127.0.0.1",
                                  "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", str(n), "--share-gpu", "--config", "C1", "--steps", "4",
                                  "--warmup", "2", "--exchange", exchange], env, 900)
    out = r.stdout + r.stderr
    assert r.returncode == 0, out[-4000:]
    lines = r.stdout.splitlines()
    assert len(lines) == 1 and lines[0].startswith("{"), r.stdout[-3000:]    # rank 0's JSON line and NOTHING else on stdout (gloo's chatter goes to stderr)
    d = json.loads(lines[0])
    for k in KEYS:
        assert k in d, k
    c = d["config"]
    assert d["n_gpus"] == n and c["rccl_ranks"] == n and "one process per GPU" in c["launch"] and "ncclCommInitRank" in c["launch"]
    assert len(c["slabs"]) == n and c["slabs"][0][0] == 0 and c["slabs"][-1][1] == 8
    assert c["max_abs_rgba_diff_vs_1gpu_frame"] is not None and c["max_abs_rgba_diff_vs_1gpu_frame"] <= 2e-5
    pr = d["per_rank"]
    assert len(pr["samples"]) == n and sum(pr["samples"]) >= c["samples_per_step"] > 0
    ms = pr["kernel_ms_bin_fill_raymarch_finish"]
    assert all(row[1] > 0 for row in ms) and ms[0][3] == 0 and all(row[3] > 0 for row in ms[1:])     # every process reported; finish pass on ranks > 0 only
    assert d["scaling"] == "strong" and d["value"] > 0 and exchange in c["parallelism"]


# ---- first-hardware-run hardening: the DRIVER'S command lines, byte for byte ------------------------------------------
def _driver_command(n, port, steps=20, warmup=5):
    """What the driver runs for N > 1 (task contract): `python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1
    --master-port P bench.py --gpus N --steps K --warmup W` -- default config (C3), default flags."""
    return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr", "127.0.0.1", "--master-port", str(port),
            "bench.py", "--gpus", str(n), "--steps", str(steps), "--warmup", str(warmup)]


def _driver_env(**extra):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    # one GPU under N ranks: every rank on cuda:0 (env form of --share-gpu) and the multi-process stand-in as the RCCL build libvpfx dlopens
    env.update(VPFX_BENCH_SHARE_GPU="1", VPFX_RCCL_LIBRARY=MP_SHIM, FAKE_RCCL_TIMEOUT_MS="120000", HSA_ENABLE_IPC_MODE_LEGACY="0")
    env.update(extra)
    return env


def _free_port():
    import socket
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    return port


RENDEZVOUS_TROUBLE = ("EADDRINUSE", "Address already in use", "address already in use", "RendezvousConnectionError", "RendezvousTimeoutError", "DistNetworkError",
                      "failed to bind", "Connection refused")


def _run_driver(make_cmd, env, timeout):
    """Run a torch.distributed.run command; a failure of the LAUNCHER's own rendezvous (the port picked a moment ago taken by somebody else: seen once in ~30
    suite passes) is not what these tests are about -- such a run is repeated on a fresh port, twice at most, and reported if it persists."""
    r = None
    for attempt in range(3):
        r = subprocess.run(make_cmd(_free_port()), cwd=ROOT, env=env, capture_output=True, text=True, timeout=timeout)
        if r.returncode == 0 or not any(t in r.stderr for t in RENDEZVOUS_TROUBLE) or "[bench]" in r.stderr:
            break
        print(f"[test] launcher rendezvous trouble, attempt {attempt + 1}:\n{r.stderr[-1500:]}", file=sys.stderr)
    return r


@pytest.mark.parametrize("n", [2, 4, 8])
def test_the_drivers_scaling_command_runs_to_one_json_line(n):
    """`--steps 20 --warmup 5`, everything else default: the metric's config (C3: 32^3 x 32^3, 100 k particles, 1920 x 1080) in N slabs, one process
    per rank.  ONE JSON line on stdout; N RCCL ranks; per-rank figures from every process; the sharded frame = the 1-GPU frame rendered on rank 0."""
    _build_mp_shim()
    r = _run_driver(lambda port: _driver_command(n, port), _driver_env(), 1200)
    assert r.returncode == 0, (r.stdout + r.stderr)[-4000:]
    lines = r.stdout.splitlines()
    assert len(lines) == 1 and lines[0].startswith("{"), r.stdout[-3000:]
    d = json.loads(lines[0])
    for k in KEYS:
        assert k in d, k
    c = d["config"]
    assert d["n_gpus"] == n and d["steps"] == 20 and d["warmup"] == 5 and d["scaling"] == "strong" and d["value"] > 0
    assert c["workload"].startswith("C3") and c["rccl_ranks"] == n and "one process per GPU" in c["launch"]
    assert len(c["slabs"]) == n and c["slabs"][0][0] == 0 and c["slabs"][-1][1] == 32
    assert c["max_abs_rgba_diff_vs_1gpu_frame"] is not None and c["max_abs_rgba_diff_vs_1gpu_frame"] <= 2e-5
    pr = d["per_rank"]
    assert len(pr["samples"]) == n and len(pr["kernel_ms_bin_fill_raymarch_finish"]) == n and all(row[1] > 0 for row in pr["kernel_ms_bin_fill_raymarch_finish"])
    assert d["roofline"]["frac"] > 0 and "cpu_baseline" not in d                       # the CPU leg is rank 0's at N = 1 only
    assert d["metric"].startswith("Mvoxels/s filled")


def test_a_rank_that_dies_mid_run_ends_the_job_nonzero_instead_of_hanging():
    """Rank 1 of 4 leaves in the middle of the timed region (os._exit: no clean-up, as a crashed process).  Its peers sit in exchanges with it:
    the library's bounded waits (and torchrun's worker watchdog) must end the whole command with a non-zero exit code well inside the time-out,
    and no JSON line may be printed."""
    import time
    _build_mp_shim()
    r = None
    for attempt in range(2):        # (a job that failed BEFORE the hook -- launcher trouble, see _run_driver -- says nothing about the hook: once more)
        t0 = time.time()
        r = _run_driver(lambda port: _driver_command(4, port, steps=10, warmup=3), _driver_env(VPFX_BENCH_TEST_FAIL_RANK="1"), 600)
        if "TEST HOOK: rank 1 leaves the job" in r.stderr:
            break
        print(f"[test] the job ended without reaching the hook (rc {r.returncode}), attempt {attempt + 1}:\n{r.stderr[-3000:]}", file=sys.stderr)
    # (the stand-in's time-out stays at its 120 s: with 20 s a slow start of rank 0 -- it renders the 1-GPU reference frame before it joins the
    #  communicator -- once made the OTHER ranks give up first, and the job failed without ever reaching the hook; torchrun ends the survivors as soon as
    #  rank 1 is gone, so the test does not wait for the time-out)
    took = time.time() - t0
    assert r.returncode != 0, (r.stdout + r.stderr)[-3000:]
    assert took < 300, took
    assert not any(l.startswith("{") for l in r.stdout.splitlines()), r.stdout[-2000:]
    assert "TEST HOOK: rank 1 leaves the job" in r.stderr, "the job failed, but not through the hook:\n" + r.stderr[-6000:]


def test_config5_shape_through_the_drivers_command_at_eight_ranks():
    """BASELINE config 5's SHAPE through `bench.py --gpus 8` as the driver launches it (VERDICT r5 next #1c): 64^3-voxel bricks, the 3840 x 2160 screen (133 MB
    partial images, 16.6 MB exchange pieces), eight processes.  Eight ranks of the full 64^3-metavoxel grid need 8 x 25-50 GiB (profiles/r06_scaling_model_C5_r8.json)
    and cannot share one 288-GB test GPU, so the grid is one eighth of config 5's (32^3 metavoxels, 125 k particles: 'C5e'); the full grid's sharded parity is
    tests/test_gpu_large_configs.py::test_config5_through_the_fanout_slab_by_slab."""
    _build_mp_shim()
    r = _run_driver(lambda port: _driver_command(8, port, steps=4, warmup=3) + ["--config", "C5e"], _driver_env(), 1500)
    assert r.returncode == 0, (r.stdout + r.stderr)[-4000:]
    lines = r.stdout.splitlines()
    assert len(lines) == 1 and lines[0].startswith("{"), r.stdout[-3000:]
    d = json.loads(lines[0])
    c = d["config"]
    assert d["n_gpus"] == 8 and c["rccl_ranks"] == 8 and c["workload"].startswith("C5e: 32x32x32 metavoxels x 64^3 voxels")
    assert len(c["slabs"]) == 8 and c["slabs"][0][0] == 0 and c["slabs"][-1][1] == 32
    assert c["max_abs_rgba_diff_vs_1gpu_frame"] is not None and c["max_abs_rgba_diff_vs_1gpu_frame"] <= 2e-5
    pr = d["per_rank"]
    assert len(pr["samples"]) == 8 and all(row[1] > 0 for row in pr["kernel_ms_bin_fill_raymarch_finish"])
    assert d["roofline_all"]["fill"]["kernel"] == "k_fill_lds" and d["value"] > 0
