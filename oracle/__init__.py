"""ORACLE -- test infrastructure only.

CPU restatements of the reference's hot path (oracle/*.c -> oracle/libslref.so).  Only
tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this package; the
product package (stillleben_amd) never does."""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def build(force=False):
    so = os.path.join(_HERE, "libslref.so")
    srcs = [os.path.join(_HERE, f) for f in os.listdir(_HERE) if f.endswith((".c", ".h"))]
    srcs.append(os.path.join(_HERE, "..", "include", "slhip.h"))
    if force or not os.path.exists(so) or any(os.path.getmtime(s) > os.path.getmtime(so) for s in srcs):
        subprocess.run(["make", "-C", _HERE], check=True, stdout=subprocess.DEVNULL)
    return so


def lib():
    global _LIB
    if _LIB is None:
        so = os.path.join(_HERE, "libslref.so")
        if not os.path.exists(so):
            build()
        _LIB = C.CDLL(so)
    return _LIB


def _p(a):
    return C.c_void_p(a.ctypes.data) if a is not None else C.c_void_p(0)


class RenderResult:
    pass


def render(pool_arrays, srec, drec, W, H, flags, depth_peel=None, shadow_res=2048, want_hdr=False,
           want_shadow=False, light_maps=None):
    """Runs the CPU reference renderer on the binary scene description produced by
    stillleben_amd._batch.build_batch.  Returns numpy arrays shaped like the device buffers."""
    from stillleben_amd import _abi

    L = lib()
    pos, nrm, uv, col, idx, tex = [np.ascontiguousarray(a) for a in pool_arrays[:6]]
    tan = np.ascontiguousarray(pool_arrays[6]) if len(pool_arrays) > 6 else None
    pool = _abi.MeshPool()
    pool.d_pos, pool.d_nrm, pool.d_uv, pool.d_col, pool.d_idx, pool.d_tex = (_p(a) for a in (pos, nrm, uv, col, idx, tex))
    pool.n_vertices, pool.n_indices, pool.n_tex_bytes = len(pos), len(idx), tex.size
    pool.d_tan = _p(tan)
    if light_maps:
        # light_maps: list of (buffers dict as returned by light_map_build, sizes dict) -- host memory
        recs = (_abi.LightMapRec * len(light_maps))()
        keep = []
        for r, (bufs, sizes) in zip(recs, light_maps):
            b = {k: np.ascontiguousarray(v, dtype=np.float32) for k, v in bufs.items()}
            keep.append(b)
            r.d_env, r.d_irradiance, r.d_prefilter, r.d_brdf_lut = (b[k].ctypes.data for k in ("env", "irradiance", "prefilter", "brdf_lut"))
            for k, v in sizes.items():
                setattr(r, k, int(v))
        pool.d_light_maps, pool.n_light_maps = C.cast(recs, C.c_void_p).value, len(light_maps)
    B = len(srec)
    r = RenderResult()
    r.rgb = np.zeros((B, H, W, 4), np.uint8)
    r.coord = np.zeros((B, H, W, 4), np.float32)
    r.cls = np.zeros((B, H, W, 1), np.uint16)
    r.instance = np.zeros((B, H, W, 1), np.uint16)
    r.normals = np.zeros((B, H, W, 4), np.float32)
    r.vertex_idx = np.zeros((B, H, W, 4), np.uint32)
    r.bary = np.zeros((B, H, W, 4), np.float32)
    r.cam_coord = np.zeros((B, H, W, 4), np.float32)
    r.hdr = np.zeros((B, H, W, 4), np.float32) if want_hdr else None
    r.shadow = np.zeros((B, 3, shadow_res, shadow_res), np.float32) if want_shadow else None
    out = _abi.RenderOut()
    out.d_rgb, out.d_coord, out.d_class, out.d_instance = _p(r.rgb), _p(r.coord), _p(r.cls), _p(r.instance)
    out.d_normals, out.d_vertex_idx, out.d_bary, out.d_cam_coord = _p(r.normals), _p(r.vertex_idx), _p(r.bary), _p(r.cam_coord)
    srec = np.ascontiguousarray(srec)
    drec = np.ascontiguousarray(drec)
    L.slref_render.argtypes = [C.POINTER(_abi.MeshPool), C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32,
                               C.c_uint32, C.c_void_p, C.POINTER(_abi.RenderOut), C.c_uint32, C.c_void_p, C.c_void_p]
    st = L.slref_render(C.byref(pool), _p(srec), _p(drec), B, W, H, flags, _p(depth_peel), C.byref(out), shadow_res,
                        _p(r.hdr), _p(r.shadow))
    if st != 0:
        raise RuntimeError("slref_render failed: %d" % st)
    return r


def ssao_tables():
    L = lib()
    noise = np.zeros(48, np.float32)
    kern = np.zeros(192, np.float32)
    L.slref_ssao_tables(_p(noise), _p(kern))
    return noise, kern


def ssao_pass(proj, cam_coord, normals):
    """The SSAO pass alone (before the blur): proj float32[4, 4] row-major, cam_coord / normals float32[H, W, 4] -> ao float32[H, W]."""
    L = lib()
    H, W = cam_coord.shape[:2]
    proj = np.ascontiguousarray(proj, np.float32).reshape(16)
    cam = np.ascontiguousarray(cam_coord, np.float32)
    nrm = np.ascontiguousarray(normals, np.float32)
    ao = np.zeros((H, W), np.float32)
    L.slref_ssao_pass.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p]
    L.slref_ssao_pass.restype = None
    L.slref_ssao_pass(_p(proj), _p(cam), _p(nrm), W, H, _p(ao))
    return ao


class SettleState:
    """Contact state of a settle that outlives a call (slref_settle_ex): created by a call with params['resume'] == 0,
    continued by calls with params['resume'] == the steps run so far."""

    def __init__(self):
        self.handle = C.c_void_p(0)

    def __del__(self):
        try:
            if self.handle:
                L = lib()
                L.slref_settle_state_free.argtypes = [C.c_void_p]
                L.slref_settle_state_free(self.handle)
        except Exception:
            pass


def settle(srec, bodies, hulls, hull_verts, params, state=None):
    """Runs the CPU reference stepper in place on `bodies` (numpy structured array)."""
    L = lib()
    srec = np.ascontiguousarray(srec)
    hulls = np.ascontiguousarray(hulls)
    hull_verts = np.ascontiguousarray(hull_verts, dtype=np.float32)
    params = np.ascontiguousarray(params)
    assert bodies.flags["C_CONTIGUOUS"]
    L.slref_settle_ex.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    st = L.slref_settle_ex(_p(srec), len(srec), _p(bodies), _p(hulls), _p(hull_verts), _p(params),
                           C.byref(state.handle) if state is not None else None)
    if st != 0:
        raise RuntimeError("slref_settle failed: %d" % st)
    return bodies


def overlap_any(srec, bodies, hulls, hull_verts):
    L = lib()
    srec = np.ascontiguousarray(srec)
    flags = np.zeros(len(bodies), np.uint8)
    L.slref_overlap_any.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    st = L.slref_overlap_any(_p(srec), len(srec), _p(np.ascontiguousarray(bodies)), _p(np.ascontiguousarray(hulls)),
                             _p(np.ascontiguousarray(hull_verts, dtype=np.float32)), _p(flags))
    if st != 0:
        raise RuntimeError("slref_overlap_any failed: %d" % st)
    return flags


def debug_contacts(srec, bodies, hulls, hull_verts, params, max_rows=4096):
    L = lib()
    out = np.zeros((max_rows, 12), np.float32)
    L.slref_debug_contacts.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]
    n = L.slref_debug_contacts(_p(np.ascontiguousarray(srec)), _p(np.ascontiguousarray(bodies)),
                               _p(np.ascontiguousarray(hulls)), _p(np.ascontiguousarray(hull_verts, dtype=np.float32)),
                               _p(np.ascontiguousarray(params)), _p(out), max_rows)
    return out[:n]


# ---- sl.diff oracle (oracle/diff_ref.c) ------------------------------------------------------
def sobel_valid(inst, depth):
    L = lib()
    H, W = inst.shape
    inst = np.ascontiguousarray(inst, dtype=np.int16)
    depth = np.ascontiguousarray(depth, dtype=np.float32)
    out = np.zeros((H, W), np.uint8)
    L.slref_sobel_valid.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p]
    L.slref_sobel_valid(_p(inst), _p(depth), H, W, _p(out))
    return out.astype(bool)


def dilate(mask, valid, coords):
    L = lib()
    H, W = mask.shape
    m = np.ascontiguousarray(mask, dtype=np.uint8)
    v = np.ascontiguousarray(valid, dtype=np.uint8)
    c = np.ascontiguousarray(coords, dtype=np.float32)
    om = np.zeros((H, W), np.uint8)
    oc = np.zeros((H, W, 3), np.float32)
    L.slref_dilate.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
    L.slref_dilate(_p(m), _p(v), _p(c), H, W, _p(om), _p(oc))
    return om.astype(bool), oc


def image_gradients(rgb, valid):
    L = lib()
    H, W = valid.shape
    rgb = np.ascontiguousarray(rgb, dtype=np.uint8)
    v = np.ascontiguousarray(valid, dtype=np.uint8)
    gx = np.zeros((3, H, W), np.float32)
    gy = np.zeros((3, H, W), np.float32)
    L.slref_image_gradients.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
    L.slref_image_gradients(_p(rgb), _p(v), H, W, _p(gx), _p(gy))
    return gx, gy


def pose_backward(rgb, coord, inst, grad_img, P, poses, obj_inst):
    L = lib()
    H, W = inst.shape
    rgb = np.ascontiguousarray(rgb, dtype=np.uint8)
    coord = np.ascontiguousarray(coord, dtype=np.float32)
    inst = np.ascontiguousarray(inst, dtype=np.int16)
    grad_img = np.ascontiguousarray(grad_img, dtype=np.float32)
    P = np.ascontiguousarray(P, dtype=np.float32)
    poses = np.ascontiguousarray(poses, dtype=np.float32).reshape(-1, 16)
    obj_inst = np.ascontiguousarray(obj_inst, dtype=np.int32)
    out = np.zeros((len(poses), 6), np.float32)
    L.slref_pose_backward.argtypes = [C.c_void_p] * 7 + [C.c_int, C.c_int, C.c_int, C.c_void_p]
    L.slref_pose_backward(_p(rgb), _p(coord), _p(inst), _p(grad_img), _p(P), _p(poses), _p(obj_inst), len(poses), H, W, _p(out))
    return out


def apply_pose_delta(pose, delta, orthonormalize=True):
    """D5 (diff.py:525-590), numpy restatement."""
    pose = np.asarray(pose, dtype=np.float32).reshape(-1, 4, 4)
    delta = np.asarray(delta, dtype=np.float32).reshape(-1, 6)
    D = np.zeros((len(pose), 4, 4), np.float32)
    D[:, 0, 0] = D[:, 1, 1] = D[:, 2, 2] = D[:, 3, 3] = 1.0
    D[:, 0, 1], D[:, 0, 2] = -delta[:, 2], delta[:, 1]
    D[:, 1, 0], D[:, 1, 2] = delta[:, 2], -delta[:, 0]
    D[:, 2, 0], D[:, 2, 1] = -delta[:, 1], delta[:, 0]
    D[:, :3, 3] = delta[:, 3:]
    out = np.matmul(pose, D)
    if orthonormalize:
        for b in range(len(out)):
            u, _, vt = np.linalg.svd(out[b, :3, :3].astype(np.float64))
            out[b, :3, :3] = (u @ vt).astype(np.float32)
    return out


def camera_model(rgb, params, stage=4):
    """Deterministic stages of the camera model (oracle/camera_ref.c).  rgb f32[B,3,H,W], params =
    array of stillleben_amd._abi.CAMERA_DTYPE records; stage 0..4 (see camera_ref.c)."""
    from stillleben_amd import _abi

    L = lib()
    rgb = np.ascontiguousarray(rgb, dtype=np.float32)
    B, _, H, W = rgb.shape
    rec = np.ascontiguousarray(np.asarray(params, dtype=_abi.CAMERA_DTYPE).reshape(B))
    out, tmp = np.zeros_like(rgb), np.zeros_like(rgb)
    L.slref_camera_model.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_int, C.c_int, C.c_void_p, C.c_int]
    st = L.slref_camera_model(_p(rgb), _p(out), _p(tmp), B, H, W, _p(rec), int(stage))
    if st != 0:
        raise ValueError("oracle.camera_model: the random noise stage is not restated")
    return out


def light_map_build(equirect, sizes):
    """IBL precompute (oracle/ibl_ref.c).  equirect f32[H,W,3]; sizes = dict(env_size, env_levels, irr_size,
    pre_size, pre_levels, lut_size).  Returns dict of numpy buffers laid out like slhip_light_map's."""
    from stillleben_amd import _abi

    L = lib()
    eq = np.ascontiguousarray(equirect, dtype=np.float32)
    H, W, _ = eq.shape

    def cube_floats(size, levels):
        return sum(24 * (size >> l) ** 2 for l in range(levels))

    bufs = {
        "env": np.zeros(cube_floats(sizes["env_size"], sizes["env_levels"]), np.float32),
        "irradiance": np.zeros(cube_floats(sizes["irr_size"], 1), np.float32),
        "prefilter": np.zeros(cube_floats(sizes["pre_size"], sizes["pre_levels"]), np.float32),
        "brdf_lut": np.zeros(2 * sizes["lut_size"] ** 2, np.float32),
    }
    rec = _abi.LightMapRec()
    rec.d_env, rec.d_irradiance, rec.d_prefilter, rec.d_brdf_lut = (bufs[k].ctypes.data for k in ("env", "irradiance", "prefilter", "brdf_lut"))
    for k, v in sizes.items():
        setattr(rec, k, int(v))
    L.slref_light_map_build.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p]
    st = L.slref_light_map_build(_p(eq), H, W, C.byref(rec))
    if st != 0:
        raise RuntimeError("slref_light_map_build failed")
    return bufs


def vertex_backward(rgb, coord, inst, bary, grad_img, P, poses, obj_inst):
    """D6 dense form (oracle/diff_ref.c slref_vertex_backward): returns (grad_vertices, grad_colors) f32[H,W,3,3]."""
    L = lib()
    H, W = inst.shape
    rgb = np.ascontiguousarray(rgb, dtype=np.uint8)
    coord = np.ascontiguousarray(coord, dtype=np.float32)
    inst = np.ascontiguousarray(inst, dtype=np.int16)
    bary = np.ascontiguousarray(bary, dtype=np.float32)
    if bary.shape[-1] == 3:
        bary = np.ascontiguousarray(np.concatenate([bary, np.zeros(bary.shape[:-1] + (1,), np.float32)], axis=-1))
    grad_img = np.ascontiguousarray(grad_img, dtype=np.float32)
    P = np.ascontiguousarray(P, dtype=np.float32)
    poses = np.ascontiguousarray(poses, dtype=np.float32).reshape(-1, 16)
    obj_inst = np.ascontiguousarray(obj_inst, dtype=np.int32)
    gv = np.zeros((H, W, 3, 3), np.float32)
    gc = np.zeros((H, W, 3, 3), np.float32)
    L.slref_vertex_backward.argtypes = [C.c_void_p] * 8 + [C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
    L.slref_vertex_backward(_p(rgb), _p(coord), _p(inst), _p(bary), _p(grad_img), _p(P), _p(poses), _p(obj_inst), len(poses), H, W,
                            _p(gv), _p(gc))
    return gv, gc


def synth_stage(params, assets, asset_ids=None):
    """oracle/synth_ref.c: tabletop set-up of a batch.  Returns (bodies, settle_scenes, objects, scenes)."""
    from stillleben_amd import _abi
    from stillleben_amd import _settle_batch as SB

    L = lib()
    p = np.array(params)
    n, m = int(p["n_scenes"]), int(p["n_objects"])
    bodies = np.zeros(n * m, dtype=SB.BODY_DTYPE)
    ss = np.zeros(n, dtype=SB.SETTLE_SCENE_DTYPE)
    objs = np.zeros(n * m, dtype=_abi.SYNTH_OBJECT_DTYPE)
    scs = np.zeros(n, dtype=_abi.SYNTH_SCENE_DTYPE)
    assets = np.ascontiguousarray(assets)
    ids = None if asset_ids is None else np.ascontiguousarray(asset_ids, dtype=np.uint16)
    L.slref_synth_stage.argtypes = [C.c_void_p] * 7
    st = L.slref_synth_stage(_p(p), _p(assets), _p(ids) if ids is not None else None, _p(bodies), _p(ss), _p(objs), _p(scs))
    if st != 0:
        raise RuntimeError("slref_synth_stage failed: %d" % st)
    return bodies, ss, objs, scs


def synth_place(params, assets, templates, bodies, objects, scenes):
    """oracle/synth_ref.c: camera / light / shadow matrix / render records.  `scenes` gets camera_pose filled.
    Returns (scene_records, draw_records, chunk_records)."""
    from stillleben_amd import _abi

    L = lib()
    p = np.array(params)
    n = int(p["n_scenes"])
    srec = np.zeros(n, dtype=_abi.SCENE_DTYPE)
    drec = np.zeros(n * int(p["max_draws_per_scene"]), dtype=_abi.DRAW_DTYPE)
    crec = np.zeros(n * int(p["max_chunks_per_scene"]), dtype=_abi.CHUNK_DTYPE)
    assets, templates = np.ascontiguousarray(assets), np.ascontiguousarray(templates)
    bodies, objects = np.ascontiguousarray(bodies), np.ascontiguousarray(objects)
    assert scenes.flags["C_CONTIGUOUS"]
    L.slref_synth_place.argtypes = [C.c_void_p] * 9
    st = L.slref_synth_place(_p(p), _p(assets), _p(templates), _p(bodies), _p(objects), _p(scenes), _p(srec), _p(drec), _p(crec))
    if st != 0:
        raise RuntimeError("slref_synth_place failed: %d" % st)
    return srec, drec, crec


def synth_draws(params, scene):
    """The raw random draws of one scene: dict(yaw, azimuth, elevation, light_normals[3], quat[n,4], pbr[n,2])."""
    L = lib()
    p = np.array(params)
    m = int(p["n_objects"])
    out = np.zeros(6 + 6 * m, np.float32)
    L.slref_synth_draws.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p]
    L.slref_synth_draws(_p(p), scene, _p(out))
    per = out[6:].reshape(m, 6)
    return {"yaw": out[0], "azimuth": out[1], "elevation": out[2], "light_normals": out[3:6].copy(),
            "quat": per[:, :4].copy(), "pbr": per[:, 4:].copy()}


def det_logf(x):
    L = lib()
    L.slref_det_logf.restype = C.c_float
    L.slref_det_logf.argtypes = [C.c_float]
    return L.slref_det_logf(float(x))


def det_sincosf(x):
    L = lib()
    L.slref_det_sincosf.argtypes = [C.c_float, C.POINTER(C.c_float), C.POINTER(C.c_float)]
    s, c = C.c_float(), C.c_float()
    L.slref_det_sincosf(float(x), C.byref(s), C.byref(c))
    return s.value, c.value


def settle_stats(srec, bodies, hulls, hull_verts, params):
    """settle() with the solver's work statistics switched on (single-threaded use only): returns a dict of four
    256-bin histograms over the steps -- 'active' contacts, friction 'anchors', 'colours', 'chain' (rows a lane pair walks
    in sequence per sweep)."""
    L = lib()
    h = np.zeros(1096, np.uint64)
    L.slref_settle_set_stats.argtypes = [C.c_void_p]
    L.slref_settle_set_stats(_p(h))
    try:
        settle(srec, bodies, hulls, hull_verts, params)
    finally:
        L.slref_settle_set_stats(None)
    return {"active": h[:256], "anchors": h[256:512], "colours": h[512:768], "chain": h[768:1024], "gjk_iters": h[1024:1088], "tilt_iters": h[1088:]}
