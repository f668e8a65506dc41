"""The Unity native-plugin entry points (include/vpfx.h "Unity native-plugin hookup", csrc/unity_plugin.cpp): UnityPluginLoad, a frame
description per slot, and the render-event callback that Unity would invoke on its render thread via GL.IssuePluginEvent -- driven here from a
second host thread.  The frame it produces must be the one the direct calls produce (VPR.cs:181-220: set-up, bin + fill every updateInterval
frames, ray-march every frame)."""
import ctypes as C
import os
import threading

import numpy as np
import pytest
import torch

from vpfx_amd import abi, engine as E, scene as S

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

pytestmark = pytest.mark.gpu


def _desc(sc, eng, flags, particles):
    f = abi.vp_unity_frame()
    f.ctx = eng.h.value
    f.flags = flags
    f.particle_count = len(particles)
    l2w = np.ascontiguousarray(sc.light_to_world, dtype=np.float32).reshape(-1)
    p2w = np.ascontiguousarray(sc.psys_local_to_world, dtype=np.float32).reshape(-1)
    for i in range(16):
        f.light_to_world[i] = float(l2w[i]); f.psys_local_to_world[i] = float(p2w[i])
    for i in range(3):
        f.grid_center[i] = float(sc.grid_center[i])
    f.particles = particles.ctypes.data
    f.layout = sc.layout
    f.fill = sc.fill_params()
    f.camera = sc.camera()
    f.raymarch = sc.raymarch_params()
    return f


def _issue_plugin_event(fn, slot):
    """GL.IssuePluginEvent: Unity calls the function on its render thread, not on the thread that asked for it."""
    t = threading.Thread(target=fn, args=(slot,))
    t.start(); t.join()


@pytest.mark.parametrize("fanout", [False, True])
def test_render_event_runs_the_frame_on_another_thread(fanout):
    L = E.lib()
    L.vp_unity_render_event_func.restype = C.CFUNCTYPE(None, C.c_int)
    sc = S.make_scene("C1", cubemap="r8")
    direct = E.Engine(sc.config())
    direct.set_frame(sc.light_to_world, sc.grid_center)
    direct.bin(sc.particles, sc.layout, sc.psys_local_to_world)
    direct.fill(sc.fill_params())
    ref = direct.raymarch(sc.camera(), sc.raymarch_params())

    eng = E.Engine(sc.config(devices=[0, 0], multi_flags=abi.VP_MULTI_PEER_COPY) if fanout else sc.config())
    L.UnityPluginLoad(None)
    fn = L.vp_unity_render_event_func()
    slot = 3
    ev = C.c_uint64(0)
    assert L.vp_unity_last_status(slot, C.byref(ev)) == abi.VP_ERR_STATE and ev.value == 0          # nothing has run yet
    _issue_plugin_event(fn, slot)                                                                    # no description: an error, not a crash
    assert L.vp_unity_last_status(slot, None) == abi.VP_ERR_STATE
    particles = np.ascontiguousarray(sc.particles)
    frame = _desc(sc, eng, abi.VP_UNITY_SET_FRAME | abi.VP_UNITY_BIN_AND_FILL, particles)
    assert L.vp_unity_set_frame_desc(slot, C.byref(frame)) == 0
    _issue_plugin_event(fn, slot)
    assert L.vp_unity_last_status(slot, None) == abi.VP_ERR_STATE                                    # no output registered
    host = np.zeros((sc.height, sc.width, 4), dtype=np.float32)
    dev = torch.zeros((sc.height, sc.width, 4), device="cuda")
    assert L.vp_unity_register_output(slot, C.c_void_p(dev.data_ptr()), host.ctypes.data_as(C.POINTER(C.c_float))) == 0
    _issue_plugin_event(fn, slot)
    assert L.vp_unity_last_status(slot, C.byref(ev)) == 0 and ev.value == 3, E.lib().vp_last_error(eng.h)
    tol = 2e-5 if fanout else 0.0
    assert np.abs(host - ref).max() <= tol and np.abs(dev.cpu().numpy() - ref).max() <= tol
    # a frame between two fills (updateInterval): ray-march only, other camera, host output only
    sc.set_camera((6.0, 3.0, -14.0))
    ref2 = direct.raymarch(sc.camera(), sc.raymarch_params())
    frame2 = _desc(sc, eng, 0, particles)
    assert L.vp_unity_set_frame_desc(slot, C.byref(frame2)) == 0
    assert L.vp_unity_register_output(slot, None, host.ctypes.data_as(C.POINTER(C.c_float))) == 0
    _issue_plugin_event(fn, slot)
    assert L.vp_unity_last_status(slot, None) == 0
    assert np.abs(host - ref2).max() <= tol
    assert L.vp_unity_set_frame_desc(99, C.byref(frame2)) == abi.VP_ERR_BAD_ARG
    # teardown order of a host: detach the slot BEFORE freeing the arrays and the context; an event Unity delivers afterwards is a
    # no-op -- it must neither touch the old buffer nor the destroyed context -- and the counter keeps counting
    L.vp_unity_last_status(slot, C.byref(ev))
    before = ev.value
    assert L.vp_unity_clear_slot(slot) == 0 and L.vp_unity_clear_slot(99) == abi.VP_ERR_BAD_ARG
    host[...] = -7.0
    _issue_plugin_event(fn, slot)
    assert L.vp_unity_last_status(slot, C.byref(ev)) == abi.VP_ERR_STATE and ev.value == before + 1
    assert np.all(host == -7.0)
    L.UnityPluginUnload()
    assert L.vp_unity_last_status(slot, C.byref(ev)) == abi.VP_ERR_STATE and ev.value == 0          # unloading forgets every slot
    eng.close(); direct.close()


def test_render_target_shared_through_an_exported_fd(tmp_path):
    """Texture interop, the native half (vp_unity_register_output_fd): the memory behind the host's render texture arrives as a POSIX fd exported by
    the graphics API; the library imports it through HIP's external-memory API and the ray-march writes it directly (VPR.cs:204-210 blits particlesRT
    on the GPU; without this the frame crosses PCIe).  No Vulkan on the box: the producer is HIP's own virtual-memory API exporting a dma-buf fd
    (tests/tools/extmem_producer.cpp) -- the import path is the one a Vulkan export takes.  The producer's own view of the memory must hold the frame."""
    import subprocess
    so = str(tmp_path / "libextmem_producer.so")
    subprocess.run(["g++", "-shared", "-fPIC", "-D__HIP_PLATFORM_AMD__", "-I/opt/rocm/include", os.path.join(ROOT, "tests", "tools", "extmem_producer.cpp"),
                    "-L/opt/rocm/lib", "-lamdhip64", "-o", so], check=True)
    P = C.CDLL(so)
    P.producer_create.argtypes = [C.c_size_t, C.POINTER(C.c_int), C.POINTER(C.c_void_p), C.POINTER(C.c_size_t)]
    P.producer_read.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t]
    L = E.lib()
    L.vp_unity_render_event_func.restype = C.CFUNCTYPE(None, C.c_int)
    L.vp_unity_register_output_fd.argtypes = [C.c_int32, C.c_void_p, C.c_int32, C.c_uint64, C.c_uint64]
    sc = S.make_scene("C1", cubemap="r8")
    direct = E.Engine(sc.config())
    direct.set_frame(sc.light_to_world, sc.grid_center)
    direct.bin(sc.particles, sc.layout, sc.psys_local_to_world)
    direct.fill(sc.fill_params())
    ref = direct.raymarch(sc.camera(), sc.raymarch_params())
    eng = E.Engine(sc.config())
    need, offset = sc.width * sc.height * 16, 1 << 16
    fd, view, size = C.c_int(-1), C.c_void_p(), C.c_size_t(0)
    assert P.producer_create(need + offset, C.byref(fd), C.byref(view), C.byref(size)) == 0
    L.UnityPluginLoad(None)
    slot = 2
    assert L.vp_unity_register_output_fd(slot, eng.h, fd.value, size.value, size.value) == abi.VP_ERR_BAD_ARG      # no room behind that offset
    assert L.vp_unity_register_output_fd(slot, eng.h, -1, size.value, 0) == abi.VP_ERR_BAD_ARG
    assert L.vp_unity_register_output_fd(slot, eng.h, fd.value, size.value, offset) == 0, L.vp_last_error(eng.h)
    particles = np.ascontiguousarray(sc.particles)
    frame = _desc(sc, eng, abi.VP_UNITY_SET_FRAME | abi.VP_UNITY_BIN_AND_FILL, particles)
    assert L.vp_unity_set_frame_desc(slot, C.byref(frame)) == 0
    _issue_plugin_event(L.vp_unity_render_event_func(), slot)
    assert L.vp_unity_last_status(slot, None) == 0, L.vp_last_error(eng.h)
    eng.sync()
    got = np.zeros((sc.height, sc.width, 4), dtype=np.float32)
    assert P.producer_read(view, offset, got.ctypes.data_as(C.c_void_p), got.nbytes) == 0
    assert np.array_equal(got, ref)                                  # the "texture" holds the frame, written by the ray-march itself
    # a plain device / host output registered AFTER an fd import replaces it (the import is released, not leaked) and survives:
    # the next event writes the new host buffer, the shared memory keeps the previous frame
    host = np.zeros((sc.height, sc.width, 4), dtype=np.float32)
    L.vp_unity_register_output.argtypes = [C.c_int32, C.c_void_p, C.c_void_p]
    assert L.vp_unity_register_output(slot, None, host.ctypes.data_as(C.c_void_p)) == 0
    sc2 = S.make_scene("C1", cubemap="r8")
    sc2.set_camera((2.0, 1.0, -9.0))
    frame2 = _desc(sc2, eng, 0, particles)
    assert L.vp_unity_set_frame_desc(slot, C.byref(frame2)) == 0
    _issue_plugin_event(L.vp_unity_render_event_func(), slot)
    assert L.vp_unity_last_status(slot, None) == 0, L.vp_last_error(eng.h)
    assert np.array_equal(host, direct.raymarch(sc2.camera(), sc2.raymarch_params()))
    assert L.vp_unity_clear_slot(slot) == 0                          # a late event is a no-op
    _issue_plugin_event(L.vp_unity_render_event_func(), slot)
    assert L.vp_unity_last_status(slot, None) == abi.VP_ERR_STATE
    L.UnityPluginUnload()
    eng.close(); direct.close()

