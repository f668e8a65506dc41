"""CPU: the C-ABI library loads, exports every symbol include/vpfx.h declares, its structs match the ctypes mirror,
and -- without a GPU -- it fails loudly instead of falling back to a CPU path."""
import ctypes as C
import os
import re
import subprocess
import tempfile

import pytest

from vpfx_amd import abi, engine as E, scene as S

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HDR = os.path.join(ROOT, "include", "vpfx.h")


def header_functions():
    src = open(HDR).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(vp_[a-z_0-9]+)\s*\(", src)))


def test_header_and_mirror_list_the_same_functions():
    assert header_functions() == sorted(abi.EXPORTED_SYMBOLS)


def test_library_exports_every_declared_symbol():
    lib = E.lib()
    for name in header_functions():
        assert hasattr(lib, name), name
    assert lib.vp_abi_version() == abi.VPFX_ABI_VERSION == 6


def test_header_constants_match_the_mirror():
    """Every numeric #define of the header that the ctypes mirror also names has the same value (flags, enums of the exchange plan, ABI)."""
    src = re.sub(r"/\*.*?\*/", "", open(HDR).read(), flags=re.S)
    defs = {m.group(1): int(m.group(2), 0) for m in re.finditer(r"#define\s+(VP[A-Z_0-9]*)\s+(0x[0-9a-fA-F]+|-?\d+)\b", src)}
    checked = 0
    for name, value in defs.items():
        if hasattr(abi, name):
            assert getattr(abi, name) == value, name
            checked += 1
    assert checked >= 25, checked
    for name in ("VP_RM_NO_EARLY_OUT", "VP_MULTI_TEST_HOOKS", "VP_MULTI_TEST_DROP_SEND", "VP_MULTI_TEST_SHARED_DEVICE", "VP_XOP_ALL_GATHER", "VP_XBUF_FINAL", "VPFX_ABI_VERSION"):
        assert name in defs and hasattr(abi, name), name


def test_struct_layouts_match_header():
    code = r'''
#include <stdio.h>
#include <stddef.h>
#include "vpfx.h"
int main(void) {
  printf("%zu %zu %zu %zu %zu %zu\n", sizeof(vp_config), sizeof(vp_particle_layout), sizeof(vp_fill_params), sizeof(vp_camera),
         sizeof(vp_raymarch_params), sizeof(vp_stats));
  printf("%zu %zu %zu %zu %zu\n", offsetof(vp_fill_params, cubemap), offsetof(vp_fill_params, light_depth_map), offsetof(vp_fill_params, cubemap_format),
         offsetof(vp_raymarch_params, scene_depth), offsetof(vp_camera, fov_y));
  return 0; }
'''
    with tempfile.TemporaryDirectory() as d:
        src = os.path.join(d, "t.c")
        open(src, "w").write(code)
        exe = os.path.join(d, "t")
        subprocess.run(["gcc", "-I", os.path.join(ROOT, "include"), src, "-o", exe], check=True)
        out = subprocess.run([exe], check=True, capture_output=True, text=True).stdout.split()
    sizes = [int(x) for x in out]
    mirror = [C.sizeof(abi.vp_config), C.sizeof(abi.vp_particle_layout), C.sizeof(abi.vp_fill_params), C.sizeof(abi.vp_camera),
              C.sizeof(abi.vp_raymarch_params), C.sizeof(abi.vp_stats),
              abi.vp_fill_params.cubemap.offset, abi.vp_fill_params.light_depth_map.offset, abi.vp_fill_params.cubemap_format.offset,
              abi.vp_raymarch_params.scene_depth.offset, abi.vp_camera.fov_y.offset]
    assert sizes == mirror


def _has_gpu():
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False


@pytest.mark.skipif(_has_gpu(), reason="GPU present: vp_create succeeds")
def test_no_device_fails_loudly_no_cpu_fallback():
    sc = S.make_scene("T0")
    with pytest.raises(E.VpfxError) as ei:
        E.Engine(sc.config())
    assert ei.value.code == abi.VP_ERR_NO_DEVICE
    assert "no CPU fallback" in str(ei.value)


def test_bad_arguments_are_rejected_before_touching_a_device():
    lib = E.lib()
    h = C.c_void_p()
    assert lib.vp_create(None, C.byref(h)) == abi.VP_ERR_BAD_ARG
    cfg = S.make_scene("T0").config()
    for bad_nv in (65, 1, 0, -16):                         # [2, 64] is the whole domain (24, 12, 13 ... run the run-time-nv kernels)
        cfg.num_voxels = bad_nv
        cfg.num_border = 0
        assert lib.vp_create(C.byref(cfg), C.byref(h)) in (abi.VP_ERR_UNSUPPORTED, abi.VP_ERR_BAD_ARG)
    if not _has_gpu():                                     # an accepted voxel count gets as far as looking for a device
        cfg.num_voxels = 24
        assert lib.vp_create(C.byref(cfg), C.byref(h)) == abi.VP_ERR_NO_DEVICE
    cfg = S.make_scene("T0").config()
    cfg.num_border = 8                                     # 2b >= nv
    assert lib.vp_create(C.byref(cfg), C.byref(h)) == abi.VP_ERR_BAD_ARG
    cfg = S.make_scene("T0").config(slab=(3, 2))
    assert lib.vp_create(C.byref(cfg), C.byref(h)) == abi.VP_ERR_BAD_ARG
    assert lib.vp_set_frame(None, None, None) == abi.VP_ERR_BAD_ARG
    assert b"vp_create" in lib.vp_last_error(None)


def test_product_sources_never_reference_the_oracle():
    """The product (package + csrc + bench GPU leg) must not import, link or call anything under oracle/."""
    pkg = os.path.join(ROOT, "volumetric-particles-for-unity_amd")
    for dp, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".cpp", ".hip", ".h", ".cs")) or f == "Makefile":
                txt = open(os.path.join(dp, f), errors="ignore").read()
                assert "vp_oracle" not in txt and "vporacle" not in txt and "from oracle" not in txt and "import oracle" not in txt, os.path.join(dp, f)
