"""The RCCL branch of the fan-out (csrc/multi.cpp: ncclCommInitAll / ncclCommInitRank, grouped ncclSend / ncclRecv, in-place ncclAllGather,
ncclCommAbort, ncclCommGetAsyncError) with 2 / 3 / 4 / 8 ranks on ONE GPU.

The GPU boxes have one GPU, the real librccl wants one GPU per rank, and VP_MULTI_PEER_COPY replaces exactly the calls in question by mail-box
copies matched by (source, destination, sequence) -- looser than RCCL's matching in issue order.  So the fan-out tests are re-run in a CHILD
process whose `librccl.so.1` is tests/tools/fake_rccl.cpp: a stand-in that lets ranks share a device, moves the bytes with stream-ordered
copies and ENFORCES RCCL's rules (operations of a communicator run in issue order, sends match receives first to first per pair and only
across the two current operations, equal byte counts, all-gathers only match all-gathers, the in-place rule) -- reporting ncclInvalidUsage or a
time-out where the real library would hang.  multi.cpp dlopen()s the library by name, so LD_LIBRARY_PATH selects it; the child never imports
torch (torch maps its own librccl.so.1).  The stand-in's own checks are tested in tests/tools/rccl_shim_cases.py, also run here.
"""
import json
import os
import re
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TOOLS = os.path.join(ROOT, "tests", "tools")
SHIM_DIR = os.path.join(TOOLS, "_build")
SHIM = os.path.join(SHIM_DIR, "librccl.so.1")


def build_shim():
    src = os.path.join(TOOLS, "fake_rccl.cpp")
    if os.path.exists(SHIM) and os.path.getmtime(SHIM) >= os.path.getmtime(src):
        return
    os.makedirs(SHIM_DIR, exist_ok=True)
    subprocess.run(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-D__HIP_PLATFORM_AMD__", "-I/opt/rocm/include", src, "-o", SHIM,
                    "-Wl,-soname,librccl.so.1", "-L/opt/rocm/lib", "-lamdhip64", "-lpthread"], check=True)


def run_child(args, tmp_path, timeout=1500):
    build_shim()
    stats = str(tmp_path / "fake_rccl_stats.json")
    env = dict(os.environ, VPFX_TEST_RCCL_SHIM="1", FAKE_RCCL_STATS_FILE=stats,
               LD_LIBRARY_PATH=SHIM_DIR + os.pathsep + os.environ.get("LD_LIBRARY_PATH", ""))
    r = subprocess.run([sys.executable, "-m", "pytest", "-q", "-x", "-m", "gpu", "-p", "no:cacheprovider"] + args, cwd=ROOT, env=env,
                       capture_output=True, text=True, timeout=timeout)
    tail = (r.stdout[-6000:] + "\n" + r.stderr[-3000:])
    assert r.returncode == 0, tail
    m = re.search(r"(\d+) passed", r.stdout)
    assert m and "failed" not in r.stdout and "skipped" not in r.stdout, tail
    return int(m.group(1)), json.load(open(stats)), tail


def test_fanout_tests_on_the_rccl_branch_with_the_checking_standin(tmp_path):
    """The fan-out tests of tests/test_gpu_multi.py -- worlds of 2, 3, 4 and 8 ranks, both exchange forms, hand-off groups, cameras with and
    without a straddling slab, slab re-cuts (the profile all-gather), 300 frames on 8 ranks, the degenerate slabs, images smaller than one exchange
    piece per rank, the C3 grid in eight slabs --
    with use_rccl = true; communicators alternately from ncclCommInitAll and from ncclGetUniqueId + grouped ncclCommInitRank."""
    sel = ("test_fanout_matches_single_context_and_oracle or test_serial_handoff_chain or test_handoff_with_a_camera_inside or "
           "test_config4_benchmark_grid_in_eight_slabs or test_fanout_many_frames or test_fanout_edge_cases or test_fanout_at_the_extremes")
    passed, st, tail = run_child(["tests/test_gpu_multi.py", "-k", sel], tmp_path)
    assert passed == 4 + 1 + 3 + 1 + 1 + 4 + 8, tail
    # the RCCL branch really ran, on the stand-in, and the stand-in had nothing to report
    assert st["communicators"] >= 60 and st["p2p_groups"] > 1000 and st["all_gathers"] > 100 and st["bytes"] > 1 << 30, st
    assert st["errors"] == 0 and st["aborts"] == 0, st


def test_the_standin_reports_what_hangs_on_rccl_and_failure_paths_of_the_fanout(tmp_path):
    passed, st, tail = run_child(["tests/tools/rccl_shim_cases.py"], tmp_path, timeout=600)
    assert passed == 8 + 3, tail
    assert st["errors"] >= 8 and st["aborts"] >= 1, st              # every negative case was reported by the stand-in; the injected fault aborted


def test_an_unloadable_rccl_library_is_an_error_code_not_a_crash():
    """VPFX_RCCL_LIBRARY pointing at something dlopen() refuses must come back from vp_create as VP_ERR_RCCL with dlopen's reason in the
    message (the message used to be built from a second dlerror() call, which returns NULL: undefined behaviour).  Child process: the loader's verdict is
    cached per process."""
    code = ("import sys; sys.path.insert(0, %r)\n"
            "from __graft_entry__ import load_package; load_package()\n"
            "from vpfx_amd import abi, engine as E, scene as S\n"
            "sc = S.make_scene('T0')\n"
            "try:\n"
            "    E.Engine(sc.config(devices=[0, 0], multi_flags=abi.VP_MULTI_TEST_HOOKS | abi.VP_MULTI_TEST_SHARED_DEVICE))\n"
            "    print('CREATED')\n"
            "except E.VpfxError as ex:\n"
            "    print('CODE', ex.code, str(ex))\n") % ROOT
    env = dict(os.environ, VPFX_RCCL_LIBRARY="/nonexistent/librccl_not_here.so")
    r = subprocess.run([sys.executable, "-c", code], cwd=ROOT, env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert "CODE -7" in r.stdout and "VPFX_RCCL_LIBRARY=/nonexistent/librccl_not_here.so cannot be loaded" in r.stdout, r.stdout[-2000:]
    assert "No such file" in r.stdout or "cannot open shared object" in r.stdout, r.stdout[-2000:]
