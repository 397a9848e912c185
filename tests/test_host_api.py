"""CPU tests of the host layer: the C-ABI library loads and exports every symbol declared in
include/slhip.h, record layouts match, API surface/error behaviour mirror the reference
(reference tests/basic.cpp:62-86,119-132,309-373; python/src/py_*.cpp)."""
import ctypes
import os
import re

import numpy as np
import pytest
import torch

import scenes as S

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_abi_library_exports_every_declared_symbol():
    import __graft_entry__ as g

    g.build()
    from stillleben_amd import _abi

    hdr = open(os.path.join(ROOT, "include", "slhip.h")).read()
    names = set(re.findall(r"\b(slhip_[a-z_0-9]+)\s*\(", hdr))
    assert len(names) >= 14
    L = ctypes.CDLL(_abi.lib_path())
    for n in sorted(names):
        assert hasattr(L, n), "libslhip.so does not export %s" % n
    assert _abi.lib().slhip_abi_version() == _abi.ABI_VERSION == 5
    assert _abi.lib().slhip_last_error() is not None


def test_no_cpu_fallback_without_device(sl):
    from stillleben_amd import _abi
    from stillleben_amd._context import engine

    if torch.cuda.is_available():
        pytest.skip("a HIP device is present")
    scene = S.cube_lookat_scene(sl)
    with pytest.raises(_abi.SlhipError):
        sl.RenderPass().render(scene)
    with pytest.raises(_abi.SlhipError):
        scene.simulate(0.01)
    with pytest.raises(_abi.SlhipError):
        engine()


def test_mesh_bbox_and_pretransform(sl):
    # basic.cpp:62-86, :119-132
    m = sl.Mesh(S.BUNNY, physics=False)
    assert torch.isfinite(m.bbox.min).all() and torch.isfinite(m.bbox.max).all()
    m.center_bbox()
    assert m.bbox.center.abs().max() < 1e-4
    m.scale_to_bbox_diagonal(0.5)
    assert abs(m.bbox.diagonal - 0.5) < 1e-5
    pre = m.pretransform.clone()
    m.pretransform = pre
    assert torch.allclose(m.pretransform, pre, atol=1e-6)
    with pytest.raises(ValueError):
        m.pretransform = torch.diag(torch.tensor([1.0, 2.0, 1.0, 1.0]))   # non-uniform scale (mesh.cpp:1063-1067)
    m.scale_to_bbox_diagonal(0.5, 'order_of_magnitude')
    assert m._scale in (np.float32(0.01), np.float32(0.1), np.float32(1.0))
    with pytest.raises(ValueError):
        m.class_index = 70000


def test_object_and_scene_rules(sl):
    m = sl.Mesh(S.CUBE, physics=False)
    assert m.class_index == 1                                  # mesh.h:300
    scene = sl.Scene((640, 480))
    a, b = sl.Object(m), sl.Object(m)
    b.instance_index = 15
    scene.add_object(a)
    scene.add_object(b)
    assert a.instance_index == 1 and b.instance_index == 15   # scene.cpp:285-287
    with pytest.raises(ValueError):
        a.instance_index = 1 << 16
    with pytest.raises(ValueError):
        scene.light_directions = torch.zeros(4, 3)             # scene.cpp:418-419
    with pytest.raises(ValueError):
        sl.RenderPass("toon")
    assert sl.RenderPass().ssao_enabled is True                # render_pass.h:150
    assert scene.manual_exposure == -1.0
    # light tensors alias scene memory (py_scene.cpp:284-309)
    scene.light_directions[0, 2] = -1.0
    assert scene.light_directions[0, 2] == -1.0
    with pytest.warns(UserWarning):
        scene.choose_random_light_position()                   # quirk q3
    P = scene.projection_matrix()
    fx = 640 / (2 * np.tan(np.radians(58.0) / 2))
    assert abs(P[0, 0] * 320 - fx) < 1e-2 and P[3, 2] == 1.0


def test_serialization_round_trip(sl):
    # basic.cpp:309-373, step by step
    m = sl.Mesh(S.BUNNY, physics=False)
    scene = sl.Scene((640, 480))
    o = sl.Object(m)
    distance = scene.min_dist_for_object_diameter(float(m.bbox.diagonal))
    pose = torch.eye(4)
    pose[2, 3] = distance
    o.set_pose(pose)
    o.instance_index = 15
    scene.add_object(o)
    scene.set_camera_look_at([0.3, -0.2, 0.5], [0.0, 0.1, 0.0])
    scene.light_directions = torch.tensor([[0.1, -0.4, -0.9], [0.5, 0.5, -0.2]])
    scene.light_colors = torch.tensor([[300.0, 290.0, 280.0], [10.0, 20.0, 30.0]])
    text = scene.serialize()
    cache = sl.MeshCache()     # empty, as in the reference test: the mesh is loaded from its file name
    s0 = sl.Scene((640, 480))
    s0.deserialize(text, cache)
    assert len(s0.objects) == 1
    o0 = s0.objects[0]
    assert float(o0.mesh._scale) == float(m._scale)
    assert (o0.mesh.pretransform - m.pretransform).norm() < 1e-5
    t = o0.pose()[:3, 3]
    assert t[0] == 0 and t[1] == 0 and abs(float(t[2]) - distance) < 1e-6
    assert o0.instance_index == 15
    assert torch.equal(o0.pose(), o.pose())                        # 9 significant digits: exact
    assert torch.allclose(s0.camera_pose(), scene.camera_pose(), atol=1e-6)
    assert torch.equal(s0.light_directions, scene.light_directions) and torch.equal(s0.light_colors, scene.light_colors)
    assert torch.equal(s0.projection_matrix(), scene.projection_matrix())
    # cache functionality: a second scene shares the mesh object
    s1 = sl.Scene((640, 480))
    s1.deserialize(text, cache)
    assert len(s1.objects) == 1 and s1.objects[0].mesh is o0.mesh


def test_serialization_document_structure(sl):
    """The document is the reference's: keys and order of scene.cpp:761-796 / object.cpp:384-406 /
    mesh.cpp:1091-1097, values before sub-groups, [light] x 3, [object] + [object/mesh]."""
    from stillleben_amd import serialization

    m = sl.Mesh(S.CUBE, physics=False)
    scene = sl.Scene((320, 240))
    for k in range(2):
        ob = sl.Object(m)
        ob.static = k == 1
        scene.add_object(ob)
    doc = serialization.to_document(scene)
    assert [k for k, _ in doc.values] == ["viewport", "projection", "cameraPosition", "cameraRotation", "ambientLight",
                                          "numObjects", "backgroundPlanePose", "backgroundPlaneSize", "manualExposure"]
    assert [n for n, _ in doc.groups] == ["light"] * 3 + ["object"] * 2
    og = doc.groups_named("object")[1]
    assert [k for k, _ in og.values] == ["pose", "instanceIndex", "specularColor", "shininess", "roughness", "metallic",
                                         "casts_shadows", "stickerRange", "stickerRotation", "static", "density",
                                         "linear_velocity_limit"]
    assert og.value("static") == "true" and doc.value("numObjects") == "2" and doc.value("viewport") == "320 240"
    assert [k for k, _ in og.group("mesh").values] == ["filename", "classIndex", "scale", "rigidPretransform"]
    lines = scene.serialize().splitlines()
    assert lines.index("[object]") < lines.index("[object/mesh]") and lines[0].startswith("viewport=")
    # a Matrix4 is written row by row: the translation of a pose sits at positions 3, 7, 11
    ob = scene.objects[0]
    pose = torch.eye(4)
    pose[0, 3], pose[1, 3], pose[2, 3] = 1.0, 2.0, 3.0
    ob.set_pose(pose)
    vals = serialization.to_document(scene).groups_named("object")[0].value("pose").split()
    assert [vals[3], vals[7], vals[11]] == ["1", "2", "3"]


def test_deserialize_reference_style_document(sl):
    """A document as the reference writes it (6 significant digits, its key spelling) plus the legacy
    `lightPosition` form (scene.cpp:817-821)."""
    text = "\n".join([
        "viewport=640 480",
        "cameraPosition=0 0 0",
        "cameraRotation=0 0 0 1",
        "lightPosition=0 0 2",
        "ambientLight=0.1 0.2 0.3",
        "numObjects=1",
        "manualExposure=1.5",
        "[object]",
        "pose=1 0 0 0.25 0 1 0 -0.5 0 0 1 1.75 0 0 0 1",
        "instanceIndex=7",
        "static=true",
        "linearVelocityLimit=2.5",
        "[object/mesh]",
        "filename=" + S.CUBE,
        "classIndex=3",
        "scale=0.1",
        "rigidPretransform=1 0 0 -0.5 0 1 0 0 0 0 1 0 0 0 0 1",
    ]) + "\n"
    scene = sl.Scene((64, 48))
    scene.deserialize(text)
    assert scene.viewport == (640, 480) and scene.manual_exposure == 1.5
    ob = scene.objects[0]
    assert ob.instance_index == 7 and ob.static and ob.mesh.class_index == 3
    assert torch.allclose(ob.pose()[:3, 3], torch.tensor([0.25, -0.5, 1.75]))
    assert abs(float(ob._linear_velocity_limit) - 2.5) < 1e-7
    assert abs(float(ob.mesh.pretransform[0, 3]) - (-0.05)) < 1e-7       # scale * rigid translation
    assert torch.allclose(scene.light_directions[0], torch.tensor([0.0, 0.0, -1.0]))
    assert torch.allclose(scene.light_colors[0], torch.tensor([0.0, 0.8, 0.0]))


def test_quaternion_helpers_and_alias_package(sl):
    import stillleben

    assert stillleben.Scene is sl.Scene and stillleben.diff is sl.diff
    q = torch.tensor([0.1, -0.3, 0.2, 0.9])
    q = q / q.norm()
    R = sl.quat_to_matrix(q)
    assert torch.allclose(R @ R.T, torch.eye(3), atol=1e-6)
    q2 = sl.matrix_to_quat(R)
    assert torch.allclose(q2 * torch.sign(q2[3]), q * torch.sign(q[3]), atol=1e-6)


def test_out_of_scope_shims_fail_loudly(sl):
    with pytest.raises(RuntimeError):
        sl.LightMap("x.ibl")          # image-based lighting exists (row f1): a missing file is reported like the reference does
    with pytest.raises(NotImplementedError):
        sl.Viewer()
    scene = sl.Scene((64, 48))
    with pytest.warns(UserWarning):
        sl.view(scene)


def test_pose_samplers(sl):
    # pose.h:56-218: sampled positions project inside 80 % of the frustum at a distance in
    # [1.2 d, d / 0.4]; the viewpoint sampler turns the chosen object axis towards the camera
    from stillleben_amd import pose_sampling as ps

    scene = sl.Scene((640, 480))
    P = scene.projection_matrix().numpy()
    rng = np.random.default_rng(0)
    d = ps.minimum_distance_for_object_diameter(0.3, P)
    assert d == pytest.approx(max(P[0, 0], P[1, 1]) * 0.15)
    pos = ps.RandomPositionSampler(P, 0.3)
    for _ in range(50):
        p = pos(rng)
        assert 1.2 * d - 1e-5 <= p[2] <= d / 0.4 + 1e-5
        assert abs(P[0, 0] * p[0] / p[2]) <= 0.8 + 1e-5 and abs(P[1, 1] * p[1] / p[2]) <= 0.8 + 1e-5
    vp = ps.ViewPointPoseSampler(pos, (0.0, 0.0, 1.0))
    for _ in range(10):
        T = vp(rng)
        R, t = T[:3, :3], T[:3, 3]
        assert np.allclose(R @ R.T, np.eye(3), atol=1e-5)
        assert np.allclose(R @ np.array([0, 0, 1.0]), -t / np.linalg.norm(t), atol=1e-5)
    vc = ps.ViewCorrectedPoseSampler(pos, np.eye(3))
    T = vc(rng)
    # the corrected orientation maps the viewing ray back onto +z
    assert np.allclose(T[:3, :3].T @ (T[:3, 3] / np.linalg.norm(T[:3, 3])), [0, 0, 1], atol=1e-5)
    assert np.allclose(ps.rotation_correction_for_translation(np.array([0, 0, 2.0])), np.eye(3))


def test_hull_cache_invalidation_keys(sl, tmp_path):
    """`.sl_mesh`-style cache (mesh.cpp:94-172, :490-511): hit when version, flags, vertex / index
    digests match and the cache is newer than the source; miss otherwise; atomic rewrite."""
    import os
    import shutil
    import time

    from stillleben_amd import hulls

    # MurmurHash64A known answers (seed 23 = Corrade's default): empty input and block + tail paths
    assert hulls.murmur64a(b"") == hulls.murmur64a(b"", 23)
    assert hulls.murmur64a(b"abcdefgh") != hulls.murmur64a(b"abcdefgi")
    assert hulls.murmur64a(b"abcdefghijk") != hulls.murmur64a(b"abcdefghijl")
    ref = 0
    for seed, data in ((23, b"12345678"), (23, b"1234567890abc")):
        ref ^= hulls.murmur64a(data, seed)
    assert 0 < ref < 2 ** 64

    src = tmp_path / "cube.glb"
    shutil.copy(S.CUBE, src)
    old = time.time() - 100
    os.utime(src, (old, old))
    m = sl.Mesh(str(src))                      # physics=True: computes the hulls and writes the cache
    cache = str(src) + hulls.CACHE_SUFFIX
    assert os.path.exists(cache)
    got = hulls.read_cache(cache, str(src), m._data, m._flags)
    assert got is not None and len(got) == len(m._hulls) == 1
    assert np.array_equal(got[0].vertices, m._hulls[0].vertices) and np.array_equal(got[0].triangles, m._hulls[0].triangles)
    # flags are a key
    assert hulls.read_cache(cache, str(src), m._data, sl.Mesh.Flag.PHYSICS_FORCE_CONVEX_HULL) is None
    # geometry digests are keys
    m2 = sl.Mesh(str(src), physics=False)
    m2._data.positions[0, 0] += 1e-3
    assert hulls.read_cache(cache, str(src), m2._data, m2._flags) is None
    m2._data.positions[0, 0] -= 1e-3
    m2._data.indices[:3] = m2._data.indices[:3][::-1].copy()
    assert hulls.read_cache(cache, str(src), m2._data, m2._flags) is None
    # a source newer than the cache makes it stale
    new = time.time() + 100
    os.utime(src, (new, new))
    assert hulls.read_cache(cache, str(src), m._data, m._flags) is None
    # a truncated / foreign file is a miss, not an error
    with open(cache, "wb") as f:
        f.write(b"garbage")
    os.utime(src, (old, old))
    assert hulls.read_cache(cache, str(src), m._data, m._flags) is None
    # and the next load rewrites it
    m3 = sl.Mesh(str(src))
    assert hulls.read_cache(cache, str(src), m3._data, m3._flags) is not None
    # in-memory / primitive meshes never touch the disk
    prim = sl.Mesh("primitive://cube")
    assert prim._hulls is not None and not os.path.exists("primitive://cube" + hulls.CACHE_SUFFIX)


# ---- 'next' row f4: asynchronous ImageSaver, ImageLoader; Animator (reference src/image_saver.cpp,
# src/image_loader.cpp, src/animator.cpp) --------------------------------------------------------------
def test_image_saver_round_trip(tmp_path):
    import stillleben_amd as sl
    from PIL import Image

    g = torch.Generator().manual_seed(3)
    rgb = torch.randint(0, 256, (48, 64, 3), dtype=torch.uint8, generator=g)
    rgba = torch.randint(0, 256, (48, 64, 4), dtype=torch.uint8, generator=g)
    gray = torch.randint(0, 256, (48, 64), dtype=torch.uint8, generator=g)
    depth = torch.randint(0, 30000, (48, 64), dtype=torch.int16, generator=g)
    saver = sl.ImageSaver()
    with pytest.raises(RuntimeError):
        saver.save(rgb, str(tmp_path / "early.png"))          # py_image_saver.cpp:36-37
    with saver:
        for i in range(20):                                    # more jobs than the queue bound
            saver.save(rgb, str(tmp_path / ("rgb%02d.png" % i)))
        saver.save(rgba, str(tmp_path / "rgba.png"))
        saver.save(gray, str(tmp_path / "gray.png"))
        saver.save(depth, str(tmp_path / "depth.png"))
        rgb_copy = rgb.clone()
        rgb.zero_()                                            # the job owns its data once save() returned
        with pytest.raises(ValueError):
            saver.save(torch.zeros(4, 4, 2, dtype=torch.uint8), str(tmp_path / "bad.png"))
        with pytest.raises(ValueError):
            saver.save(torch.zeros(4, 4, 3), str(tmp_path / "bad.png"))
        with pytest.raises(ValueError):
            saver.save(torch.zeros(4, 4), str(tmp_path / "bad.png"))
    for i in range(20):
        assert np.array_equal(np.asarray(Image.open(tmp_path / ("rgb%02d.png" % i))), rgb_copy.numpy())
    assert np.array_equal(np.asarray(Image.open(tmp_path / "rgba.png")), rgba.numpy())
    assert np.array_equal(np.asarray(Image.open(tmp_path / "gray.png")), gray.numpy())
    assert np.array_equal(np.asarray(Image.open(tmp_path / "depth.png")).astype(np.int16), depth.numpy())


def test_image_loader_random_order(tmp_path):
    import stillleben_amd as sl
    from PIL import Image

    imgs = {}
    for i in range(5):
        a = np.full((8 + i, 12, 3), 10 * i + 1, np.uint8)
        Image.fromarray(a).save(tmp_path / ("im%d.png" % i))
        imgs[a.shape[0]] = a
    Image.fromarray(np.zeros((4, 4), np.uint8)).save(tmp_path / "gray.png")   # skipped: not RGB/RGBA
    (tmp_path / "broken.jpg").write_bytes(b"not an image")                    # skipped: unreadable
    loader = sl.ImageLoader(str(tmp_path), seed=1)
    seen = set()
    for _ in range(60):
        t = loader.next()
        assert isinstance(t, sl.Texture)
        arr = t._rgba
        assert arr.shape[2] == 4 and np.array_equal(arr[..., :3], imgs[arr.shape[0]])
        seen.add(arr.shape[0])
    assert seen == set(imgs)
    assert isinstance(loader.next_texture2d(), sl.Texture2D)
    assert isinstance(loader.next_rectangle_texture(), sl.Texture)
    loader.close()
    empty = tmp_path / "empty"
    empty.mkdir()
    with pytest.raises(RuntimeError):
        sl.ImageLoader(str(empty))


def test_animator_interpolates_poses():
    import stillleben_amd as sl
    p1 = torch.eye(4)
    p2 = p1.clone()
    p2[:3, 3] = torch.tensor([0.0, 1.0, 0.0])
    p2[:3, :3] = sl.quat_to_matrix([0.0, 0.0, np.sin(np.pi / 4), np.cos(np.pi / 4)])   # 90 deg about z
    anim = sl.Animator([p1, p2], 100)
    assert len(anim) == 100
    poses = list(anim)
    assert len(poses) == 100
    assert torch.allclose(poses[0], p1, atol=1e-6)
    assert torch.allclose(poses[50][:3, 3], torch.tensor([0.0, 0.5, 0.0]), atol=1e-6)
    # nlerp at the midpoint of a 90 degree turn is exactly the 45 degree rotation
    assert torch.allclose(poses[50][:3, :3], sl.quat_to_matrix([0.0, 0.0, np.sin(np.pi / 8), np.cos(np.pi / 8)]), atol=1e-6)
    for p in poses:
        assert torch.allclose(p[:3, :3] @ p[:3, :3].t(), torch.eye(3), atol=1e-5)
    with pytest.raises(ValueError):
        sl.Animator([p1], 10)
    three = list(sl.Animator([p1, p2, p1], 10))                # keyframes at ticks 0, 5, 10
    assert torch.allclose(three[5], p2, atol=1e-6)


def test_pools_pin_the_objects_their_caches_are_keyed_by():
    """The texture cache of HostPool and the hull cache of HullPool are keyed by id(); a collected array's
    id can be handed to a new object (found by the render soak: one object per scene drew another mesh's
    texture), so the pools keep the keyed objects alive."""
    import gc

    from stillleben_amd._batch import HostPool

    pool = HostPool()
    seen = {}
    for i in range(64):
        tex = np.full((4, 4, 4), i, np.uint8)       # same shape every time: a freed block is reused at once
        off, w, h = pool.add_texture(tex, mips=False)
        assert off not in seen, "texture %d was given the pool slot of texture %d" % (i, seen[off])
        seen[off] = i
        del tex
        gc.collect()
    data = np.concatenate(pool.tex)
    for off, i in seen.items():
        assert (data[off:off + 64] == i).all()


def test_pybind11_diff_module_is_the_reference_boundary():
    """python/src/bridge_diff.cpp:160-180: `libstillleben_diff_python` with generate_sobel_valid_mask / dilate_object_mask,
    importable the way the reference's diff.py:22-30 imports it; argument errors are ValueError (std::invalid_argument) with the
    reference's texts; without a HIP device a compute call raises (no CPU path)."""
    import torch

    from stillleben.lib import libstillleben_diff_python as m

    assert callable(m.generate_sobel_valid_mask) and callable(m.dilate_object_mask)
    with pytest.raises(ValueError, match="two-dimensional"):
        m.generate_sobel_valid_mask(torch.zeros(4, dtype=torch.int16), torch.zeros(4, 4))
    with pytest.raises(ValueError, match="same height and width"):
        m.generate_sobel_valid_mask(torch.zeros(4, 4, dtype=torch.int16), torch.zeros(4, 5))
    with pytest.raises(ValueError, match="three-dimensional"):
        m.dilate_object_mask(torch.zeros(4, 4, dtype=torch.bool), torch.zeros(4, 4, dtype=torch.bool), torch.zeros(4, 4))
    if not torch.cuda.is_available():
        with pytest.raises(RuntimeError, match="no HIP device"):
            m.generate_sobel_valid_mask(torch.zeros(4, 4, dtype=torch.int16), torch.zeros(4, 4))


def test_main_module_answers_to_the_reference_name():
    """python/src/bridge.cpp:23-42: the main extension module is `libstillleben_python`; the reference's package imports it as
    `from .lib.libstillleben_python import *` + `import _set_install_prefix` (python/stillleben/__init__.py:12-13) and its diff
    module as `from .lib.libstillleben_python import Scene, RenderPassResult` (diff.py:22).  The module exports exactly the names
    bridge.cpp's thirteen init(m) calls register (py_*.cpp: py::class_ / m.def), and they are the package's objects."""
    import importlib

    ns = {}
    exec("from stillleben.lib.libstillleben_python import *\n"
         "from stillleben.lib.libstillleben_python import _set_install_prefix\n"
         "from stillleben.lib.libstillleben_python import Scene, RenderPassResult\n", ns)
    m = importlib.import_module("stillleben.lib.libstillleben_python")
    registered = {
        "init", "init_cuda", "_set_install_prefix",                                   # py_context.cpp:82-102
        "Range3D", "quat_to_matrix", "matrix_to_quat", "Texture", "Texture2D",        # py_magnum.cpp:51-157
        "Mesh", "MeshCache", "Object", "LightMap", "Scene",                           # py_mesh / py_object / py_light_map / py_scene
        "RenderPassResult", "RenderPass", "render_debug_image",                       # py_render_pass.cpp:81-282
        "ImageLoader", "ImageSaver", "Animator", "Viewer", "view", "JobQueue", "ManipulationSim",
    }
    public = {n for n in vars(m) if not n.startswith("_")} | {"_set_install_prefix"}
    assert public == registered
    assert set(m.__all__) == registered - {"_set_install_prefix"}
    import stillleben as sl
    import stillleben_amd
    for n in registered:
        assert getattr(m, n) is getattr(stillleben_amd, n) is getattr(sl, n), n
        assert n.startswith("_") or ns[n] is getattr(m, n)
    # the reference's own __all__ (python/stillleben/__init__.py:15-42) resolves on the alias package
    for n in ("init", "init_cuda", "render_debug_image", "Animator", "ImageLoader", "ImageSaver", "LightMap", "Mesh", "MeshCache",
              "Object", "Range3D", "RenderPass", "RenderPassResult", "Scene", "Texture", "Texture2D", "Viewer", "view",
              "camera_model", "diff", "extension", "losses", "quat_to_matrix", "matrix_to_quat"):
        assert hasattr(sl, n), n


def test_kernel_source_fingerprint_counts_code_only(tmp_path, monkeypatch):
    """bench.py quotes per-instruction roofline figures only when profiles/rNN/counters.json was taken from the kernel sources of
    the tree (kernel_source_sha): the fingerprint must move with the code and stay put when a comment is reworded."""
    import os
    import shutil
    import sys

    root = os.path.join(os.path.dirname(__file__), "..")
    sys.path.insert(0, root)
    import bench

    a = bench.kernel_source_sha()
    assert len(a) == 16 and a == bench.kernel_source_sha()
    # a copy of the tree's sources with (1) a reworded comment, (2) a changed constant
    for sub in ("stillleben_amd/csrc", "include"):
        shutil.copytree(os.path.join(root, sub), tmp_path / sub)
    monkeypatch.setattr(bench, "ROOT", str(tmp_path))
    assert bench.kernel_source_sha() == a
    f = tmp_path / "include" / "slhip.h"
    text = f.read_text()
    f.write_text(text.replace("/*", "/* reworded:", 1) + "\n// a trailing remark\n")
    assert bench.kernel_source_sha() == a
    assert "#define SLHIP_ABI_VERSION 5" in text
    f.write_text(text.replace("#define SLHIP_ABI_VERSION 5", "#define SLHIP_ABI_VERSION 6", 1))
    assert bench.kernel_source_sha() != a
