"""A slice of the randomised parity sweep (scripts/fuzz_parity.py; thousands of scenes were run during development) as a regression
test: random grids, borders, transforms, cameras and options; bit-level checks on bins / bricks / sample counts, 1e-3 on RGBA, plus
the slab-sharded and literal-order paths on every third seed.  Seed 5598 is the scene that exposed the skipped cell crossings of
the first ray-march traversal."""
import importlib.util
import os

import pytest
import torch  # noqa: F401  (before libvpfx, see engine._share_hip_runtime_with_torch)

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _fuzz():
    spec = importlib.util.spec_from_file_location("fuzz_parity", os.path.join(ROOT, "scripts", "fuzz_parity.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


@pytest.mark.parametrize("first", [5598, 1000, 1010, 1020, 40416])
def test_random_scenes(first):
    fz = _fuzz()
    for seed in range(first, first + (1 if first == 5598 else 10)):
        r = fz.one_case(seed)
        assert r["rgba_err"] <= 1e-3
