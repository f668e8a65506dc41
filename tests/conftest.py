import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
if os.path.dirname(os.path.abspath(__file__)) not in sys.path:
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))      # tests/slab_reference.py (test infrastructure)

from __graft_entry__ import load_package  # noqa: E402

load_package()      # importable as `vpfx_amd`


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with `-m gpu` on the GPU box)")


def _gpu_available():
    # tests/test_gpu_rccl_shim.py re-runs the fan-out tests in a child process whose librccl.so.1 is the checking stand-in of tests/tools: that
    # process must not import torch (torch maps its own librccl.so.1, and a later dlopen("librccl.so.1") would return THAT copy); the parent
    # test is GPU-marked, so a GPU is there.
    if os.environ.get("VPFX_TEST_RCCL_SHIM") == "1":
        return True
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False


def pytest_collection_modifyitems(config, items):
    if _gpu_available():
        return
    skip = pytest.mark.skip(reason="no GPU visible")
    for it in items:
        if "gpu" in it.keywords:
            it.add_marker(skip)
