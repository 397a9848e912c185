"""ctypes mirror of include/slhip.h and loader of the C-ABI library ``lib/libslhip.so``.

The product path has NO fallback: if the library is missing or no gfx950 device is present the
calls raise (see :func:`lib`).  The struct definitions are also used by the tests to drive the
CPU oracle (``oracle/libslref.so``) with the very same binary scene description.
"""
import ctypes as C
import os

import numpy as np

NUM_LIGHTS = 3
CHUNK_TRIS = 256

DRAW_HAS_BASE_TEX = 1
DRAW_VERTEX_COLORS = 2
DRAW_CASTS_SHADOW = 4
DRAW_ALPHA_TEST = 8
DRAW_NO_VERTEX_ID = 16
DRAW_HAS_NORMAL_TEX = 32
DRAW_HAS_MR_TEX = 64
DRAW_HAS_OCCLUSION_TEX = 128
DRAW_HAS_EMISSIVE_TEX = 256
DRAW_HAS_STICKER = 512
# SLHIP_SAMPLER_*: wrap S | wrap T << 2 (0 repeat, 1 clamp, 2 mirror) | mag linear 0x10 | min linear 0x20 | mip mode << 6
SAMPLER_DEFAULT = 0x10 | 0x20 | (2 << 6)

OUT_RGB = 0x01
OUT_COORD = 0x02
OUT_CLASS = 0x04
OUT_INSTANCE = 0x08
OUT_NORMALS = 0x10
OUT_VERTEX_IDX = 0x20
OUT_BARY = 0x40
OUT_CAM_COORD = 0x80
OUT_ALL = 0xFF
OUT_GT6 = 0x1F
RENDER_SSAO = 0x100
RENDER_SHADOWS = 0x200
RENDER_SHADOW_RESET = 0x400
RENDER_KEEP_HDR = 0x800
ABI_VERSION = 5
DEFAULT_HULL_PAIRS, DEFAULT_CONTACTS = 2048, 1024   # SLHIP_DEFAULT_HULL_PAIRS / SLHIP_DEFAULT_CONTACTS of include/slhip.h
COMM_ID_BYTES = 128


class MeshPool(C.Structure):
    _fields_ = [
        ("d_pos", C.c_void_p),
        ("d_nrm", C.c_void_p),
        ("d_uv", C.c_void_p),
        ("d_col", C.c_void_p),
        ("d_tan", C.c_void_p),
        ("d_idx", C.c_void_p),
        ("d_tex", C.c_void_p),
        ("n_vertices", C.c_uint64),
        ("n_indices", C.c_uint64),
        ("n_tex_bytes", C.c_uint64),
        ("d_light_maps", C.c_void_p),
        ("n_light_maps", C.c_uint64),
    ]


class LightMapRec(C.Structure):
    """slhip_light_map: device pointers + sizes of one image-based-lighting texture set."""

    _fields_ = [
        ("d_env", C.c_void_p),
        ("d_irradiance", C.c_void_p),
        ("d_prefilter", C.c_void_p),
        ("d_brdf_lut", C.c_void_p),
        ("env_size", C.c_uint32),
        ("env_levels", C.c_uint32),
        ("irr_size", C.c_uint32),
        ("pre_size", C.c_uint32),
        ("pre_levels", C.c_uint32),
        ("lut_size", C.c_uint32),
    ]


# numpy dtypes with the exact layout of slhip_draw / slhip_scene / slhip_chunk
DRAW_DTYPE = np.dtype(
    [
        ("mesh_to_object", np.float32, (16,)),
        ("object_to_world", np.float32, (16,)),
        ("normal_to_world", np.float32, (12,)),
        ("base_color", np.float32, (4,)),
        ("emissive", np.float32, (4,)),
        ("alpha_cutoff", np.float32),
        ("metallic", np.float32),
        ("roughness", np.float32),
        ("scene", np.uint32),
        ("class_index", np.uint32),
        ("instance_index", np.uint32),
        ("flags", np.uint32),
        ("n_verts", np.uint32),
        ("vtx_base", np.uint32),
        ("idx_base", np.uint32),
        ("n_tris", np.uint32),
        ("prim_base", np.uint32),
        ("tex_offset", np.uint32),
        ("tex_w", np.uint32),
        ("tex_h", np.uint32),
        ("clip_base", np.uint32),
        ("normal_tex", np.uint32, (3,)),      # offset, w, h
        ("mr_tex", np.uint32, (3,)),
        ("occlusion_tex", np.uint32, (3,)),
        ("emissive_tex", np.uint32, (3,)),
        ("sticker_tex", np.uint32, (3,)),
        ("tex_sampler", np.uint8, (8,)),     # base, normal, metallic-roughness, occlusion, emissive
        ("_pad", np.uint32, (3,)),
        ("sticker_projection", np.float32, (16,)),
        ("sticker_range", np.float32, (4,)),
    ],
    align=False,
)
assert DRAW_DTYPE.itemsize == 432, DRAW_DTYPE.itemsize

SCENE_DTYPE = np.dtype(
    [
        ("proj", np.float32, (16,)),
        ("world_to_cam", np.float32, (16,)),
        ("cam_position", np.float32, (4,)),
        ("light_dir", np.float32, (NUM_LIGHTS, 4)),
        ("light_color", np.float32, (NUM_LIGHTS, 4)),
        ("shadow_mat", np.float32, (NUM_LIGHTS, 16)),
        ("ambient", np.float32, (4,)),
        ("manual_exposure", np.float32),
        ("draw_begin", np.uint32),
        ("draw_end", np.uint32),
        ("n_prims", np.uint32),
        ("light_map", np.uint32),     # 1 + index into the pool's light maps, 0 = none
        ("bg_tex", np.uint32, (3,)),  # background image: offset, w, h (w = 0: none)
    ],
    align=False,
)
assert SCENE_DTYPE.itemsize == 480, SCENE_DTYPE.itemsize

CHUNK_DTYPE = np.dtype(
    [("scene", np.uint32), ("draw", np.uint32), ("first_tri", np.uint32), ("count", np.uint32)]
)

# slhip_host_object / slhip_host_scene (include/slhip.h): flat descriptors for the C++ record assembly
HOST_OBJECT_DTYPE = np.dtype([
    ("pose", np.float32, (16,)), ("bbox_center", np.float32, (4,)), ("color", np.float32, (4,)), ("force_color", np.uint32),
    ("tmpl_begin", np.uint32), ("tmpl_count", np.uint32), ("instance_index", np.uint32), ("metallic", np.float32),
    ("roughness", np.float32), ("casts_shadows", np.uint32), ("_pad", np.uint32)])
assert HOST_OBJECT_DTYPE.itemsize == 128
HOST_SCENE_DTYPE = np.dtype([
    ("proj", np.float32, (16,)), ("proj_inv", np.float32, (16,)), ("camera_pose", np.float32, (16,)),
    ("light_dir", np.float32, (NUM_LIGHTS, 4)), ("light_color", np.float32, (NUM_LIGHTS, 4)), ("ambient", np.float32, (4,)),
    ("plane_pose", np.float32, (16,)), ("plane_size", np.float32, (2,)), ("plane_template", np.int32),
    ("manual_exposure", np.float32), ("obj_begin", np.uint32), ("obj_end", np.uint32), ("light_map", np.uint32),
    ("bg_tex", np.uint32, (3,))])
assert HOST_SCENE_DTYPE.itemsize == 408


class RenderOut(C.Structure):
    _fields_ = [
        ("d_rgb", C.c_void_p),
        ("d_coord", C.c_void_p),
        ("d_class", C.c_void_p),
        ("d_instance", C.c_void_p),
        ("d_normals", C.c_void_p),
        ("d_vertex_idx", C.c_void_p),
        ("d_bary", C.c_void_p),
        ("d_cam_coord", C.c_void_p),
    ]


class RenderScratch(C.Structure):
    _fields_ = [
        ("d_vis", C.c_void_p),
        ("d_hdr", C.c_void_p),
        ("d_ao", C.c_void_p),
        ("d_shadow", C.c_void_p),
        ("d_queue", C.c_void_p),
        ("d_lum", C.c_void_p),
        ("d_clip", C.c_void_p),
        ("d_shadow_tiles", C.c_void_p),
        ("queue_capacity", C.c_uint32),
        ("shadow_res", C.c_uint32),
        ("n_clip_verts", C.c_uint32),
        ("shadow_lights", C.c_uint32),
        ("d_vattr", C.c_void_p),
    ]


_LIB = None
# slhip_camera_params (include/slhip.h), 268 bytes
CAMERA_DTYPE = np.dtype([
    ("translation", np.float32, (6,)), ("scaling", np.float32, (3,)),
    ("blur_kernel", np.float32, (25,)), ("post_kernel", np.float32, (25,)),
    ("exposure_gain", np.float32), ("noise_a", np.float32), ("noise_b", np.float32), ("hue_shift", np.float32),
    ("blur_enabled", np.uint32), ("noise_enabled", np.uint32), ("seed_lo", np.uint32), ("seed_hi", np.uint32),
])
assert CAMERA_DTYPE.itemsize == 268

# slhip_asset / slhip_synth_params / slhip_synth_object / slhip_synth_scene (include/slhip.h)
ASSET_DTYPE = np.dtype([
    ("mesh_to_object", np.float32, (16,)), ("bbox_min", np.float32, (4,)), ("bbox_max", np.float32, (4,)),
    ("com", np.float32, (4,)), ("inv_inertia", np.float32, (12,)), ("mass", np.float32), ("mu_s", np.float32),
    ("mu_d", np.float32), ("restitution", np.float32), ("bsphere", np.float32, (4,)), ("hull_begin", np.uint32),
    ("hull_end", np.uint32), ("draw_begin", np.uint32), ("draw_count", np.uint32), ("n_verts", np.uint32),
    ("n_chunks", np.uint32), ("_pad", np.uint32, (2,)),
])
assert ASSET_DTYPE.itemsize == 224
SYNTH_PARAMS_DTYPE = np.dtype([
    ("n_scenes", np.uint32), ("n_objects", np.uint32), ("n_assets", np.uint32), ("flags", np.uint32),
    ("seed_lo", np.uint32), ("seed_hi", np.uint32), ("scene_id_base", np.uint32), ("render_chunk", np.uint32),
    ("max_draws_per_scene", np.uint32), ("max_chunks_per_scene", np.uint32), ("max_clip_verts_per_scene", np.uint32),
    ("plane_z", np.float32), ("proj", np.float32, (16,)), ("proj_inv", np.float32, (16,)), ("plane_size", np.float32, (2,)),
    ("manual_exposure", np.float32), ("_pad0", np.float32), ("light_color", np.float32, (4,)), ("ambient", np.float32, (4,)),
])
assert SYNTH_PARAMS_DTYPE.itemsize == 224
SYNTH_OBJECT_DTYPE = np.dtype([("asset", np.uint32), ("instance_index", np.uint32), ("metallic", np.float32),
                               ("roughness", np.float32)])
SYNTH_SCENE_DTYPE = np.dtype([("plane_pose", np.float32, (16,)), ("camera_pose", np.float32, (16,))])
SYNTH_SAMPLE_DISTINCT = 1
SYNTH_RANDOM_PBR = 2
SYNTH_SHADOWS = 4
SYNTH_MAX_ASSETS = 1024

# SLHIP_LIB selects another build of the same library (developer A/B runs); there is no fallback
_LIB_PATH = os.environ.get("SLHIP_LIB") or os.path.join(os.path.dirname(os.path.abspath(__file__)), "lib", "libslhip.so")


class SlhipError(RuntimeError):
    pass


def lib_path():
    return _LIB_PATH


def lib():
    """Loads lib/libslhip.so (built by ``__graft_entry__.build()``); raises if absent."""
    global _LIB
    if _LIB is not None:
        return _LIB
    if not os.path.exists(_LIB_PATH):
        raise SlhipError(
            "stillleben_amd: %s is missing -- build it with `python -c 'import __graft_entry__ as g; "
            "g.build()'` (there is no CPU fallback)" % _LIB_PATH
        )
    L = C.CDLL(_LIB_PATH)
    L.slhip_abi_version.restype = C.c_int
    L.slhip_last_error.restype = C.c_char_p
    L.slhip_device_init.argtypes = [C.c_int]
    L.slhip_render.argtypes = [
        C.POINTER(MeshPool), C.c_void_p, C.c_void_p, C.c_void_p,
        C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32,
        C.c_void_p, C.POINTER(RenderOut), C.POINTER(RenderScratch), C.c_void_p,
    ]
    L.slhip_render_scratch_bytes.argtypes = [
        C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.POINTER(C.c_uint64 * 7)
    ]
    L.slhip_settle_solver_wave_lds.restype = C.c_int
    L.slhip_settle_solver_wave_lds.argtypes = []
    L.slhip_timing_enable.argtypes = [C.c_int]
    L.slhip_render_timings.argtypes = [C.POINTER(C.c_float * 8)]
    L.slhip_diff_sobel_valid.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
    L.slhip_diff_dilate.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
    L.slhip_diff_image_gradients.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
    L.slhip_diff_pose_backward.argtypes = [C.c_void_p] * 7 + [C.c_int, C.c_int, C.c_int] + [C.c_void_p] * 4
    L.slhip_diff_pose_backward_batch.argtypes = [C.c_void_p] * 4 + [C.c_uint64] + [C.c_void_p] * 3 + [C.c_int] * 4 + [C.c_void_p] * 4
    L.slhip_diff_vertex_backward.argtypes = [C.c_void_p] * 8 + [C.c_int, C.c_int, C.c_int] + [C.c_void_p] * 4
    L.slhip_settle.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                               C.c_void_p, C.c_uint64, C.c_void_p]
    L.slhip_settle_status.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p, C.POINTER(C.c_uint32), C.c_void_p]
    L.slhip_host_shadow_matrices.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
    L.slhip_host_normal_matrix.argtypes = [C.c_void_p, C.c_void_p]
    L.slhip_host_convex_hull.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint32, C.POINTER(C.c_uint32)]
    L.slhip_host_fill_holes.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32]
    L.slhip_records_count.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]
    L.slhip_records_build_render.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p,
                                             C.c_uint32, C.c_void_p, C.c_uint32]
    L.slhip_render_ssao_skipped.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, C.POINTER(C.c_uint64 * 2), C.c_void_p]
    L.slhip_settle_caps.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p, C.POINTER(C.c_uint64 * 10), C.c_void_p]
    L.slhip_settle_timing_enable.argtypes = [C.c_int]
    L.slhip_settle_timings.argtypes = [C.POINTER(C.c_float * 5), C.POINTER(C.c_uint32 * 5)]
    if hasattr(L, "slhip_settle_timing_every"):      # (absent from older builds selected through SLHIP_LIB for A/B runs)
        L.slhip_settle_timing_every.argtypes = [C.c_uint32]
        L.slhip_settle_timings_by_step.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.POINTER(C.c_uint32)]
    L.slhip_settle_scratch_bytes.argtypes = [C.c_uint32, C.c_void_p, C.POINTER(C.c_uint64)]
    L.slhip_overlap_any.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    L.slhip_light_map_floats.argtypes = [C.c_uint32] * 6 + [C.POINTER(C.c_uint64 * 4)]
    L.slhip_light_map_build.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
    L.slhip_camera_model.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
    L.slhip_stream_create_cu_range.argtypes = [C.c_uint32, C.c_uint32, C.POINTER(C.c_void_p)]
    L.slhip_stream_destroy.argtypes = [C.c_void_p]
    L.slhip_synth_stage.argtypes = [C.c_void_p] * 8
    L.slhip_synth_place.argtypes = [C.c_void_p] * 10
    L.slhip_comm_unique_id.argtypes = [C.c_void_p]
    L.slhip_comm_create.argtypes = [C.c_void_p, C.c_int, C.c_int, C.POINTER(C.c_void_p)]
    L.slhip_comm_destroy.argtypes = [C.c_void_p]
    L.slhip_comm_info.argtypes = [C.c_void_p, C.POINTER(C.c_int), C.POINTER(C.c_int)]
    L.slhip_allgather.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p]
    L.slhip_allgather_group.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    _LIB = L
    return L


def check(status, what):
    if status != 0:
        msg = lib().slhip_last_error()
        raise SlhipError("%s failed (%d): %s" % (what, status, msg.decode() if msg else "?"))
