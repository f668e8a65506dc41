"""ctypes mirror of include/vpfx.h (the C ABI structs).  Keep in lock-step with the header.

The same Structures are passed to libvpfx (the product) and, in tests only, to the CPU oracle,
whose entry points deliberately take the same structs.
"""
import ctypes as C

VP_OK = 0
VP_ERR_BAD_ARG = -1
VP_ERR_HIP = -2
VP_ERR_OOM = -3
VP_ERR_STATE = -4
VP_ERR_NO_DEVICE = -5
VP_ERR_UNSUPPORTED = -6
VP_ERR_RCCL = -7

VP_MAX_LOCAL_DEVICES = 8
VP_MAX_RANKS = 16
VP_MULTI_PEER_COPY = 1             # test hook: no RCCL, device-to-device copies between the local slab contexts (devices may repeat)
VP_MULTI_EXCHANGE_ALL_GATHER = 2
VP_MULTI_UNIFORM_SLABS = 4
VP_MULTI_FORCE = 8
VP_MULTI_TEST_DROP_SEND = 16       # test hook: the last rank skips its first send (time-out / abort test)
VP_MULTI_TEST_SHARED_DEVICE = 32   # test hook: devices[] may repeat on the RCCL path (needs the checking stand-in librccl of tests/tools)
VP_MULTI_TEST_HOOKS = 0x40000000   # opt-in: vp_create honours the VPFX_TEST_* environment switches

VP_RM_QUANTIZE_UNORM8 = 1
VP_RM_SHOW_NUM_SAMPLES = 2
VP_RM_SHOW_BLEND_FUNC = 4
VP_RM_SHOW_DRAW_ORDER = 8
VP_RM_SHOW_RAY_SAMPLES = 16
VP_RM_NO_EARLY_OUT = 32

VP_BRICKS_RGBA16F, VP_BRICKS_GREY_ZPAIR = 0, 1
VP_CUBEMAP_F32 = 0
VP_CUBEMAP_R8 = 1

VPFX_ABI_VERSION = 6

STATUS_NAMES = {
    0: "VP_OK", -1: "VP_ERR_BAD_ARG", -2: "VP_ERR_HIP", -3: "VP_ERR_OOM",
    -4: "VP_ERR_STATE", -5: "VP_ERR_NO_DEVICE", -6: "VP_ERR_UNSUPPORTED", -7: "VP_ERR_RCCL",
}

c_float_p = C.POINTER(C.c_float)


def set_cubemap(params, array):
    """Point vp_fill_params.cubemap at a numpy cube map [6, S, S]: float32 -> VP_CUBEMAP_F32, uint8 -> VP_CUBEMAP_R8.
    `array` must be C-contiguous and stay alive for the call (None clears the pointer: keep the resident map)."""
    if array is None:
        params.cubemap = None
        return
    import numpy as np
    if array.dtype == np.uint8:
        params.cubemap_format = VP_CUBEMAP_R8
    elif array.dtype == np.float32:
        params.cubemap_format = VP_CUBEMAP_F32
    else:
        raise TypeError(f"cube map dtype {array.dtype}: float32 or uint8")
    if not array.flags.c_contiguous or array.ndim != 3 or array.shape[0] != 6 or array.shape[1] != array.shape[2]:
        raise ValueError("cube map must be a C-contiguous [6, S, S] array")
    params.cubemap_size = array.shape[1]
    params.cubemap = array.ctypes.data


class vp_config(C.Structure):
    _fields_ = [
        ("num_mv", C.c_int32 * 3),
        ("num_voxels", C.c_int32),
        ("num_border", C.c_int32),
        ("mv_scale", C.c_float),
        ("width", C.c_int32),
        ("height", C.c_int32),
        ("device", C.c_int32),
        ("slab_z0", C.c_int32),
        ("slab_z1", C.c_int32),
        ("exact_math", C.c_int32),
        ("no_early_out", C.c_int32),
        ("reserved", C.c_int32 * 3),
        # ABI 3: multi-GPU fan-out inside the library
        ("num_devices", C.c_int32),
        ("devices", C.c_int32 * 8),
        ("world_size", C.c_int32),
        ("first_rank", C.c_int32),
        ("multi_flags", C.c_int32),
        ("rm_groups", C.c_int32),
        ("rccl_unique_id", C.c_uint8 * 128),
    ]


class vp_multi_info(C.Structure):
    _fields_ = [
        ("world_size", C.c_int32),
        ("num_local", C.c_int32),
        ("first_rank", C.c_int32),
        ("rccl_ranks", C.c_int32),
        ("exchange", C.c_int32),
        ("rm_groups", C.c_int32),
        ("slab_cuts", C.c_int32 * (VP_MAX_RANKS + 1)),
        ("chain", C.c_int32 * VP_MAX_RANKS),
        ("group_of", C.c_int32 * VP_MAX_RANKS),
        ("samples", C.c_int64 * VP_MAX_RANKS),
        ("stage_ms", (C.c_float * 4) * VP_MAX_RANKS),
        ("exchange_ms", C.c_float * 4),
    ]


VP_UNITY_MAX_SLOTS = 8
VP_UNITY_SET_FRAME, VP_UNITY_BIN_AND_FILL = 1, 2


class vp_particle_layout(C.Structure):
    _fields_ = [
        ("stride", C.c_int32),
        ("off_position", C.c_int32),
        ("off_size", C.c_int32),
        ("off_rotation", C.c_int32),
        ("off_lifetime", C.c_int32),
        ("off_start_lifetime", C.c_int32),
        ("rotation_in_radians", C.c_int32),
        ("reserved", C.c_int32),
    ]


class vp_fill_params(C.Structure):
    _fields_ = [
        ("opacity_factor", C.c_float),
        ("displacement_scale", C.c_float),
        ("fade_out_particles", C.c_int32),
        ("ambient", C.c_float * 3),
        ("init_light_intensity", C.c_float),
        ("light_near", C.c_float),
        ("light_far", C.c_float),
        ("light_cam_distance", C.c_float),
        ("cubemap_size", C.c_int32),
        ("cubemap_format", C.c_int32),
        ("cubemap", C.c_void_p),
        ("light_depth_map", c_float_p),
    ]


class vp_camera(C.Structure):
    _fields_ = [
        ("world_to_camera", C.c_float * 16),
        ("camera_to_world", C.c_float * 16),
        ("cam_pos", C.c_float * 3),
        ("fov_y", C.c_float),
        ("near_clip", C.c_float),
        ("far_clip", C.c_float),
    ]


class vp_raymarch_params(C.Structure):
    _fields_ = [
        ("steps_per_mv", C.c_int32),
        ("soft_distance", C.c_int32),
        ("scene_depth", c_float_p),
        ("flags", C.c_int32),
        ("reserved", C.c_int32 * 3),
    ]


class vp_obb(C.Structure):
    _fields_ = [("center", C.c_float * 3), ("axes", C.c_float * 9), ("half_extent", C.c_float * 3)]


VP_OCC_BOX, VP_OCC_CYLINDER, VP_OCC_ELLIPSOID = 0, 1, 2


class vp_occluder(C.Structure):
    """ABI 6: typed occluder solid; the first 60 bytes are a vp_obb."""
    _fields_ = [("center", C.c_float * 3), ("axes", C.c_float * 9), ("half_extent", C.c_float * 3), ("type", C.c_int32)]


def as_occluders(solids):
    """A ctypes array of vp_occluder from a mixed list of vp_obb (-> VP_OCC_BOX) and vp_occluder records."""
    arr = (vp_occluder * len(solids))()
    for i, b in enumerate(solids):
        C.memmove(C.byref(arr[i]), C.byref(b), C.sizeof(vp_obb))
        arr[i].type = b.type if isinstance(b, vp_occluder) else VP_OCC_BOX
    return arr


class vp_emitter_config(C.Structure):
    """Parameters of the library's particle source (the demo scene's ParticleSystem, scene:2264-2620)."""
    _fields_ = [("seed", C.c_uint64), ("rate", C.c_float), ("lifetime", C.c_float), ("speed", C.c_float), ("size", C.c_float),
                ("cone_angle_deg", C.c_float), ("cone_radius", C.c_float), ("angular_velocity_deg", C.c_float), ("max_particles", C.c_int32),
                ("reserved", C.c_int32 * 6)]


class vp_stats(C.Structure):
    _fields_ = [
        ("particles", C.c_int64),
        ("occupied_mv", C.c_int64),
        ("pairs", C.c_int64),
        ("voxels_filled", C.c_int64),
        ("samples", C.c_int64),
        ("brick_bytes", C.c_int64),
        ("max_pairs_per_mv", C.c_int64),
        ("bricks_sampled", C.c_int64),
        ("brick_bytes_per_voxel", C.c_int64),
        ("brick_format", C.c_int64),
        ("reserved", C.c_int64 * 2),
    ]


class vp_unity_frame(C.Structure):
    _fields_ = [
        ("ctx", C.c_void_p),
        ("flags", C.c_int32),
        ("particle_count", C.c_int32),
        ("light_to_world", C.c_float * 16),
        ("grid_center", C.c_float * 3),
        ("psys_local_to_world", C.c_float * 16),
        ("particles", C.c_void_p),
        ("layout", vp_particle_layout),
        ("fill", vp_fill_params),
        ("camera", vp_camera),
        ("raymarch", vp_raymarch_params),
    ]


# every symbol include/vpfx.h declares (checked by tests/test_abi.py against the built library)
EXPORTED_SYMBOLS = [
    "vp_create", "vp_destroy", "vp_last_error", "vp_abi_version", "vp_set_stream", "vp_sync", "vp_pin_host_buffer", "vp_unpin_host_buffer",
    "vp_set_frame", "vp_bin", "vp_upload_particles", "vp_bin_resident", "vp_fill", "vp_fill_begin", "vp_fill_metavoxel",
    "vp_raymarch", "vp_raymarch_device", "vp_raymarch_async", "vp_wait_image", "vp_clear_particles_rt", "vp_render_metavoxel", "vp_read_particles_rt", "vp_composite_device",
    "vp_fill_local", "vp_fill_finish", "vp_fill_finish_gathered", "vp_raymarch_partial_device", "vp_blend_partials_device",
    "vp_blend_partials_range_device",
    "vp_z_boundary", "vp_z_histogram", "vp_set_occluders", "vp_set_occluders2", "vp_render_light_depth", "vp_render_scene_depth",
    "vp_get_mv_positions", "vp_read_binlist", "vp_read_bincounts", "vp_read_brick",
    "vp_read_lightmap", "vp_get_stats", "vp_last_kernel_ms",
    "vp_raymarch_partial_handoff_device", "vp_read_zsamples",
    "vp_get_multi_info", "vp_rebalance", "vp_rccl_unique_id", "vp_plan_slabs", "vp_blend_plan", "vp_exchange_plan",
    "vp_unity_render_event_func", "vp_unity_set_frame_desc", "vp_unity_register_output", "vp_unity_register_output_fd", "vp_unity_last_status", "vp_unity_clear_slot",
    "vp_emitter_default_config", "vp_emitter_create", "vp_emitter_destroy", "vp_emitter_step", "vp_emitter_count", "vp_emitter_write_particles",
]
class vp_xop(C.Structure):
    """One operation of the image exchange's message schedule (vp_exchange_plan)."""
    _fields_ = [("kind", C.c_int32), ("peer", C.c_int32), ("buf", C.c_int32), ("index", C.c_int32), ("dst_buf", C.c_int32), ("dst_index", C.c_int32)]


VP_XOP_RECV, VP_XOP_SEND, VP_XOP_COPY, VP_XOP_ALL_GATHER = 0, 1, 2, 3
VP_XBUF_PRIMARY, VP_XBUF_SECOND, VP_XBUF_PIECES, VP_XBUF_PIECE_OUT, VP_XBUF_FINAL = 0, 1, 2, 3, 4

UNITY_LOADER_SYMBOLS = ["UnityPluginLoad", "UnityPluginUnload"]        # the two names Unity's plugin loader looks up
