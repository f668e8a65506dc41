"""The C ABI used from C, without ctypes: include/vpfx.h must compile as C99, its struct layouts must be the ones the ctypes
mirror (vpfx_amd.abi) uses, and examples/demo_frame.c (one frame through the ABI) must link against libvpfx.so.  On the GPU the
demo's output is compared with the same scene run through the Python binding and the oracle."""
import ctypes as C
import os
import re
import subprocess

import numpy as np
import pytest

from vpfx_amd import abi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "volumetric-particles-for-unity_amd")
STRUCTS = ["vp_config", "vp_particle_layout", "vp_fill_params", "vp_camera", "vp_raymarch_params", "vp_obb", "vp_occluder", "vp_stats", "vp_multi_info", "vp_xop", "vp_emitter_config"]


def _cc(args, **kw):
    return subprocess.run(["gcc", "-std=c99", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include")] + args, capture_output=True, text=True, **kw)


def test_header_is_c99_and_struct_layouts_match_the_ctypes_mirror(tmp_path):
    lines = ['#include <stdio.h>', '#include <stddef.h>', '#include "vpfx.h"', 'int main(void) {']
    for name in STRUCTS:
        st = getattr(abi, name)
        lines.append(f'  printf("{name} %zu\\n", sizeof({name}));')
        for field, _ in st._fields_:
            lines.append(f'  printf("{name}.{field} %zu\\n", offsetof({name}, {field}));')
    lines += ['  printf("abi %d\\n", VPFX_ABI_VERSION);', '  return 0; }']
    src, exe = tmp_path / "layout.c", tmp_path / "layout"
    src.write_text("\n".join(lines))
    r = _cc([str(src), "-o", str(exe)])
    assert r.returncode == 0, r.stderr
    out = dict(l.split() for l in subprocess.run([str(exe)], capture_output=True, text=True, check=True).stdout.splitlines())
    for name in STRUCTS:
        st = getattr(abi, name)
        assert int(out[name]) == C.sizeof(st), name
        for field, _ in st._fields_:
            assert int(out[f"{name}.{field}"]) == getattr(st, field).offset, f"{name}.{field}"


def _build_demo(tmp_path, name="demo_frame"):
    exe = tmp_path / name
    r = _cc(["-D_POSIX_C_SOURCE=199309L", os.path.join(ROOT, "examples", name + ".c"), "-L", PKG, "-lvpfx", "-lm", f"-Wl,-rpath,{PKG}", "-o", str(exe)])
    assert r.returncode == 0, r.stderr
    return str(exe)


def test_demo_links_and_fails_loudly_without_a_device(tmp_path):
    r = subprocess.run([_build_demo(tmp_path)], capture_output=True, text=True)
    assert r.returncode in (0, 3), r.stdout + r.stderr                 # 3 = VP_ERR_NO_DEVICE reported, no CPU fallback
    if r.returncode == 3:
        assert "no CPU fallback" in r.stderr


def test_demo_scene_links_and_fails_loudly_without_a_device(tmp_path):
    r = subprocess.run([_build_demo(tmp_path, "demo_scene"), "2", "64", "48"], capture_output=True, text=True)
    assert r.returncode in (0, 3), r.stdout + r.stderr
    if r.returncode == 3:
        assert "no CPU fallback" in r.stderr


@pytest.mark.gpu
def test_demo_scene_from_c_is_the_scene_the_python_binding_renders(tmp_path):
    """examples/demo_scene.c: the reference's scene + the library's emitter, bin + fill every 2nd frame, from plain C.  The particle cloud is the
    one scene.make_demo_scene builds (same emitter, same warm-up), so the binned statistics of frame 0 must be the binding's."""
    from vpfx_amd import engine as E, scene as S
    r = subprocess.run([_build_demo(tmp_path, "demo_scene"), "5", "256", "192"], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    nums = {k: float(v) for k, v in re.findall(r"(\w+) (-?[\d.]+)", r.stdout)}
    sc, em, solids = S.make_demo_scene(width=256, height=192)
    em.step(1.0 / 30.0)                                                  # the C loop steps the emitter before frame 0's refill
    parts = em.particles()
    eng = E.Engine(sc.config())
    eng.set_frame(sc.light_to_world, sc.grid_center)
    eng.set_occluders(solids)                                            # the same eight solids the C file spells out
    assert abs(nums["light_depth_shadowed"] - int((eng.render_light_depth() < 1).sum())) <= 2
    eng.bin(parts, sc.layout, sc.psys_local_to_world)
    st = eng.stats()
    assert nums["particles"] == len(parts) and nums["occupied_mv"] == st["occupied_mv"] > 0 and nums["pairs"] == st["pairs"]
    assert 0.01 < nums["covered"] < 1.0 and nums["sum_rgb"] > 0


def _lcg_scene(P):
    """The scene of examples/demo_frame.c (same LCG, same draw order)."""
    state = [12345]

    def frand():
        state[0] = (state[0] * 1664525 + 1013904223) & 0xFFFFFFFF
        return np.float32((state[0] >> 8) / 16777216.0)

    N, S = 4, 16
    rec = np.zeros((P, 10), dtype=np.float32)
    for i in range(P):
        for k in range(3):
            rec[i, k] = (frand() - np.float32(0.5)) * np.float32(0.7) * np.float32(N) * np.float32(3.0)
        rec[i, 6] = np.float32(3.0) * (np.float32(0.6) + np.float32(0.8) * frand())
        rec[i, 7] = np.float32(360.0) * frand()
        rec[i, 9] = 6.0
        rec[i, 8] = np.float32(6.0) * frand()
    cube = np.array([np.float32(0.1) + np.float32(0.85) * frand() for _ in range(6 * S * S)], dtype=np.float32).reshape(6, S, S)
    return rec, cube


@pytest.mark.gpu
def test_demo_frame_matches_the_python_binding_and_the_oracle(tmp_path):
    from vpfx_amd import engine as E
    from oracle import oracle as O
    P, N, NV, W, H = 150, 4, 16, 96, 64
    r = subprocess.run([_build_demo(tmp_path), str(P)], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    nums = {k: float(v) for k, v in re.findall(r"(\w+) (-?[\d.]+)", r.stdout)}
    rec, cube = _lcg_scene(P)
    cfg = abi.vp_config()
    cfg.num_mv[0] = cfg.num_mv[1] = cfg.num_mv[2] = N
    cfg.num_voxels, cfg.num_border, cfg.mv_scale, cfg.width, cfg.height, cfg.device = NV, 1, 3.0, W, H, -1
    lay = abi.vp_particle_layout()
    lay.stride, lay.off_position, lay.off_size, lay.off_rotation, lay.off_lifetime, lay.off_start_lifetime = 40, 0, 24, 28, 32, 36
    l2w = np.array([1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, -40, 1], dtype=np.float32)
    ident = np.eye(4, dtype=np.float32).reshape(16)
    fp = abi.vp_fill_params()
    fp.opacity_factor, fp.displacement_scale = 0.04, 0.7
    fp.ambient[0] = fp.ambient[1] = fp.ambient[2] = 0.2
    fp.init_light_intensity, fp.light_near, fp.light_far, fp.light_cam_distance = 1.0, 0.3, 1000.0, 200.0
    abi.set_cubemap(fp, cube)
    D = np.float32(0.8 * N * 3.0)
    cam = abi.vp_camera()
    m = [1, 0, 0, 0, 0, 1, 0, 0, 0, 0, -1, 0, 0, 0, -float(D), 1]
    for i in range(16):
        cam.camera_to_world[i] = cam.world_to_camera[i] = m[i]
    cam.cam_pos[2] = -float(D)
    cam.fov_y, cam.near_clip, cam.far_clip = float(np.float32(60.0 * 3.14159265358979 / 180.0)), 0.3, 1000.0
    rp = abi.vp_raymarch_params()
    rp.steps_per_mv, rp.soft_distance = 64, 20
    imgs = []
    for eng in (E.Engine(cfg), O.Oracle(cfg)):
        eng.set_frame(l2w, np.zeros(3, dtype=np.float32))
        eng.bin(rec, lay, ident)
        eng.fill(fp)
        imgs.append(eng.raymarch(cam, rp))
        if isinstance(eng, E.Engine):
            st, lm = eng.stats(), eng.read_lightmap()
    for key in ("particles", "occupied_mv", "pairs", "samples"):
        assert nums[key] == st[key], key
    assert nums["voxels"] == st["voxels_filled"]
    assert nums["occupied_mv"] > 10 and nums["samples"] > 1e5          # the scene is not degenerate
    assert abs(nums["sum_alpha"] - float(imgs[0][..., 3].astype(np.float64).sum())) <= 1e-3
    assert abs(nums["sum_rgb"] - float(imgs[0][..., :3].astype(np.float64).sum())) <= 1e-3
    assert abs(nums["sum_lightmap"] - float(lm.astype(np.float64).sum())) <= 1e-3
    assert np.abs(imgs[0] - imgs[1]).max() <= 1e-3                      # ... and the C caller's frame is the oracle's frame
    # the same C program on a fan-out context: the device list is the only thing the host adds (here: 2 slabs on this GPU through the
    # library's peer-copy test hook); same statistics, same frame up to the reassociated light product (<= 2e-5 per pixel)
    r2 = subprocess.run([_build_demo(tmp_path), str(P), "2", "share"], capture_output=True, text=True)
    assert r2.returncode == 0, r2.stdout + r2.stderr
    nums2 = {k: float(v) for k, v in re.findall(r"(\w+) (-?[\d.]+)", r2.stdout)}
    for key in ("particles", "occupied_mv", "pairs", "voxels"):
        assert nums2[key] == nums[key], key
    assert nums2["ranks"] == 2 and nums2["rccl_ranks"] == 0 and "slabs [0," in r2.stdout
    assert abs(nums2["sum_alpha"] - nums["sum_alpha"]) <= 2e-5 * W * H and abs(nums2["sum_rgb"] - nums["sum_rgb"]) <= 6e-5 * W * H
    assert abs(nums2["sum_lightmap"] - nums["sum_lightmap"]) <= 1e-4 * abs(nums["sum_lightmap"])
