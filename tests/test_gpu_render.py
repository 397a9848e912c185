"""GPU parity tests of the render half: HIP path (through the C-ABI) vs the CPU oracle on the
same seeded scenes.  Bar: integer outputs, depth and every geometric float output bit-exact; colour: the HDR float image
that enters the tone map within 1e-3 relative (the north star's bound, test_hdr_colour_within_1e3_relative), the 8-bit RGB
within 1 LSB on all but 1e-4 of the values and never more than 2 (the tone map's ACES curve is steep near black: one HDR ulp
can move a channel across two quantisation steps)."""
import numpy as np
import pytest
import torch

import scenes as S
from stillleben_amd import _abi
from stillleben_amd._batch import HostPool, build_batch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def eng(sl):
    from stillleben_amd._context import engine

    return engine()


def both(eng, oracle, scene_list, mask=_abi.OUT_ALL, ssao=True, shadows=True, depth_peel=None, shadow_res=None):
    from stillleben_amd import _engine

    W, H = scene_list[0].viewport
    bufs = eng.render(scene_list, mask, ssao=ssao, shadows=shadows,
                      depth_peel=None if depth_peel is None else torch.from_numpy(depth_peel).to(eng.device))
    torch.cuda.synchronize()
    want_rgb = bool(mask & _abi.OUT_RGB)
    flags = mask | (_abi.RENDER_SSAO if ssao and want_rgb else 0) | (_abi.RENDER_SHADOWS if shadows and want_rgb else 0)
    pool = HostPool()
    srec, drec, _ = build_batch(scene_list, pool, with_shadows=shadows and want_rgb)
    ref = oracle.render(pool.arrays(), srec, drec, W, H, flags, depth_peel=depth_peel,
                        shadow_res=_engine.SHADOW_RES)
    return bufs, ref


def assert_geometry_equal(bufs, ref, mask=_abi.OUT_ALL):
    def eq(name, a, b):
        a = a.cpu().numpy()
        if a.dtype == np.int16:
            a = a.view(np.uint16)
        if a.dtype == np.int32:
            a = a.view(np.uint32)
        if not np.array_equal(a.view(np.uint8), np.ascontiguousarray(b).view(np.uint8)):
            bad = np.argwhere(a != b)
            raise AssertionError("%s differs at %d elements, first %s: %s vs %s"
                                 % (name, len(bad), bad[0], a[tuple(bad[0])], b[tuple(bad[0])]))

    if mask & _abi.OUT_INSTANCE:
        eq("instance", bufs.instance, ref.instance)
    if mask & _abi.OUT_CLASS:
        eq("class", bufs.cls, ref.cls)
    if mask & _abi.OUT_VERTEX_IDX:
        eq("vertex_idx", bufs.vertex_idx, ref.vertex_idx)
    if mask & _abi.OUT_COORD:
        eq("coord", bufs.coord, ref.coord)
    if mask & _abi.OUT_BARY:
        eq("bary", bufs.bary, ref.bary)
    if mask & _abi.OUT_CAM_COORD:
        eq("cam_coord", bufs.cam_coord, ref.cam_coord)
    if mask & _abi.OUT_NORMALS:
        eq("normals", bufs.normals, ref.normals)


def assert_rgb_close(bufs, ref):
    a = bufs.rgb.cpu().numpy().astype(np.int32)
    b = ref.rgb.astype(np.int32)
    d = np.abs(a - b)
    assert d.max() <= 2, "rgb max diff %d" % d.max()
    assert (d > 1).mean() < 1e-4
    assert (d > 0).mean() < 0.02


@pytest.mark.parametrize("ssao", [False, True])
def test_hdr_colour_within_1e3_relative(sl, oracle, eng, ssao):
    """The north star's colour bound on the FLOAT image: what k_shade (+ k_ssao / k_ssao_apply) hand to the tone map against the
    oracle's, 1e-3 relative per channel.  The kernels evaluate sRGB decode, Fresnel power, BRDF divisions and light / half
    vector normalisation through the hardware's log2 / exp2 / rcp / rsq (1 ulp each); the oracle uses libm and IEEE division."""
    from stillleben_amd import _engine

    scene = S.clutter_scene(sl, 21, n_objects=8, size=(320, 240), with_bunny=True)
    scene.manual_exposure = 1.0
    W, H = scene.viewport
    mask = _abi.OUT_ALL
    bufs = eng.render([scene], mask, ssao=ssao, shadows=True, keep_hdr=True)
    torch.cuda.synchronize()
    keep = bufs._keepalive[0]
    hdr = keep["hdr"].view(torch.float32)[: 2 * H * W * 4].reshape(2, H, W, 4)[1 if ssao else 0].cpu().numpy()
    pool = HostPool()
    srec, drec, _ = build_batch([scene], pool, with_shadows=True)
    flags = mask | (_abi.RENDER_SSAO if ssao else 0) | _abi.RENDER_SHADOWS
    ref = oracle.render(pool.arrays(), srec, drec, W, H, flags, shadow_res=_engine.SHADOW_RES, want_hdr=True)
    r = ref.hdr[0]
    lit = r[..., :3].max(axis=-1) > 0
    assert lit.mean() > 0.5                                   # the frame is lit (plane + objects)
    err = np.abs(hdr[..., :3] - r[..., :3]) / np.maximum(np.abs(r[..., :3]), 1e-4)
    assert err.max() <= 1e-3, "HDR colour off by %.2e relative" % err.max()
    assert np.array_equal(hdr[..., 3], r[..., 3])             # alpha is exact


def test_fronto_parallel_lambert_known_answer(sl, eng):
    """The analytic radiance of tests/test_oracle_render.py (computed by hand from render_shader.frag:272-373: no oracle involved)
    on the kernels' float image."""
    from test_oracle_render import lambert_kat_expected, lambert_kat_scene

    scene = lambert_kat_scene(sl)
    W, H = scene.viewport
    bufs = eng.render([scene], _abi.OUT_ALL, ssao=False, shadows=False, keep_hdr=True)
    torch.cuda.synchronize()
    hdr = bufs._keepalive[0]["hdr"].view(torch.float32)[: H * W * 4].reshape(H, W, 4).cpu().numpy()
    want = lambert_kat_expected()
    assert np.allclose(hdr[H // 2, W // 2, :3], want, rtol=1e-4), (hdr[H // 2, W // 2], want)
    assert hdr[H // 2, W // 2, 3] == 1.0


def test_auto_exposure_of_a_two_tone_picture_known_answer(sl, eng):
    """The hand-computed tone map of tests/test_oracle_render.py (a uniformly lit grey face under the reference's auto exposure renders
    at 207 whatever the light's intensity: no oracle involved) on the kernels' 8-bit picture (k_lum_reduce + k_tonemap)."""
    from test_oracle_render import auto_exposure_kat_expected, auto_exposure_kat_scene

    want = auto_exposure_kat_expected()
    pics = []
    for light in (3.0, 30.0):
        scene = auto_exposure_kat_scene(sl, light)
        W, H = scene.viewport
        bufs = eng.render([scene], _abi.OUT_ALL, ssao=False, shadows=False)
        torch.cuda.synchronize()
        rgb, inst = bufs.rgb.cpu().numpy()[0], bufs.instance.cpu().numpy()[0, :, :, 0]
        face = inst == 1
        assert abs(int(rgb[H // 2, W // 2, 0]) - want) <= 1 and np.abs(rgb[face][:, :3].astype(int) - want).max() <= 2
        assert (rgb[~face] == 0).all()
        pics.append(rgb.copy())
    assert np.abs(pics[0].astype(int) - pics[1].astype(int)).max() <= 1


def test_depth_peel_second_layer_known_answer(sl, eng):
    """Two fronto-parallel sheets (tests/test_oracle_render.py): the layer peeled off behind the near sheet IS the far sheet, at its
    analytic depth -- on the kernels' outputs, no oracle involved."""
    from test_oracle_render import depth_peel_kat_check, depth_peel_kat_scene

    scene = depth_peel_kat_scene(sl)
    mask = _abi.OUT_COORD | _abi.OUT_INSTANCE
    b0 = eng.render([scene], mask, ssao=False, shadows=False)
    torch.cuda.synchronize()
    first = (b0.instance.cpu().numpy()[0, :, :, 0].copy(), b0.coord.cpu().numpy()[0, :, :, 3].copy())
    b1 = eng.render([scene], mask, ssao=False, shadows=False, depth_peel=b0.coord.clone())
    torch.cuda.synchronize()
    second = (b1.instance.cpu().numpy()[0, :, :, 0].copy(), b1.coord.cpu().numpy()[0, :, :, 3].copy())
    depth_peel_kat_check(scene, first, second)


def test_cast_shadow_known_answer(sl, eng):
    """The geometric shadow of tests/test_oracle_render.py (a slab test in numpy: no oracle involved) on the kernels' float image:
    inside the analytic shadow the ambient-only picture, outside it the picture without shadows, bit for bit; the outline within
    a few pixels."""
    from test_oracle_render import cast_shadow_kat_check, cast_shadow_kat_scene

    def hdr_of(scene, shadows):
        W, H = scene.viewport
        bufs = eng.render([scene], _abi.OUT_ALL, ssao=False, shadows=shadows, keep_hdr=True)
        torch.cuda.synchronize()
        hdr = bufs._keepalive[0]["hdr"].view(torch.float32)[: H * W * 4].reshape(H, W, 4).cpu().numpy().copy()
        return hdr[:, :, :3], bufs.cam_coord.cpu().numpy()[0].copy(), bufs.instance.cpu().numpy()[0, :, :, 0].copy()

    lit_scene, ld = cast_shadow_kat_scene(sl, 3.0)
    hs, cam, inst = hdr_of(lit_scene, True)
    hn, _, _ = hdr_of(lit_scene, False)
    ha, _, _ = hdr_of(cast_shadow_kat_scene(sl, 0.0)[0], False)
    cast_shadow_kat_check(lit_scene, ld, cam, inst, hs, hn, ha)


def test_ssao_inner_corner_known_answer(sl, eng):
    """The analytic occlusion of an inner right-angle corner (tests/test_oracle_render.py: closed-form depth, float64 numpy, the
    generator's sample tables -- no oracle involved) on the occlusion plane k_ssao_tiled writes."""
    from test_oracle_render import inner_corner_kat_check, inner_corner_kat_scene

    scene = inner_corner_kat_scene(sl)
    W, H = scene.viewport
    bufs = eng.render([scene], _abi.OUT_ALL, ssao=True, shadows=False)
    torch.cuda.synchronize()
    ao = bufs._keepalive[0]["ao"].view(torch.float32)[: H * W].reshape(H, W).cpu().numpy()
    inner_corner_kat_check(scene, ao, bufs.cam_coord.cpu().numpy()[0], bufs.instance.cpu().numpy()[0, :, :, 0])


def test_cube_lookat(sl, oracle, eng):
    scene = S.cube_lookat_scene(sl)
    bufs, ref = both(eng, oracle, [scene])
    assert_geometry_equal(bufs, ref)
    assert_rgb_close(bufs, ref)
    vi = bufs.vertex_idx.cpu().numpy()[0, :, :, :3]
    assert len(np.unique(vi)) == 5


@pytest.mark.parametrize("seed", [0, 1, 2, 3])
def test_clutter_full_pipeline(sl, oracle, eng, seed):
    scene = S.clutter_scene(sl, seed, n_objects=8, size=(320, 240), with_bunny=(seed % 2 == 1))
    if seed >= 2:
        scene.manual_exposure = 1.0
    bufs, ref = both(eng, oracle, [scene])
    assert_geometry_equal(bufs, ref)
    assert_rgb_close(bufs, ref)


def test_bunny_640x480(sl, oracle, eng):
    m = sl.Mesh(S.BUNNY, physics=False)
    m.center_bbox()
    m.scale_to_bbox_diagonal(0.5)
    scene = sl.Scene((640, 480))
    obj = sl.Object(m)
    scene.add_object(obj)
    obj.instance_index = 0xFFFF
    pose = torch.eye(4)
    pose[2, 3] = scene.min_dist_for_object_diameter(0.5)
    obj.set_pose(pose)
    scene.light_directions = torch.tensor([[0.3, 0.4, 0.8]])
    bufs, ref = both(eng, oracle, [scene])
    assert_geometry_equal(bufs, ref)
    assert_rgb_close(bufs, ref)
    assert int(bufs.instance.min()) == -1  # R16UI 65535 read as int16


def test_near_plane_clipping(sl, oracle, eng):
    # camera inside the extent of the background plane and very close to a cube: triangles cross
    # the near plane (z = 0.1) and must be clipped identically
    scene = S.clutter_scene(sl, 11, n_objects=3, size=(320, 240))
    c = scene.objects[0].pose()[:3, 3]
    scene.set_camera_look_at(c + torch.tensor([0.16, 0.05, 0.06]), c)
    bufs, ref = both(eng, oracle, [scene])
    assert (ref.instance[0] != 0).mean() > 0.5
    assert abs(float(ref.coord[0, :, :, 3].min()) - 0.1) < 1e-6  # geometry cut exactly at the near plane
    assert_geometry_equal(bufs, ref)
    assert_rgb_close(bufs, ref)


def test_batch_of_scenes(sl, oracle, eng):
    scs = [S.clutter_scene(sl, 20 + i, n_objects=5, size=(160, 120)) for i in range(5)]
    bufs, ref = both(eng, oracle, scs)
    assert_geometry_equal(bufs, ref)
    assert_rgb_close(bufs, ref)


def test_gt6_subset_without_post(sl, oracle, eng):
    scene = S.clutter_scene(sl, 4, n_objects=6)
    mask = _abi.OUT_GT6 & ~_abi.OUT_RGB
    bufs, ref = both(eng, oracle, [scene], mask=mask)
    assert bufs.rgb is None and bufs.bary is None
    assert_geometry_equal(bufs, ref, mask)


def test_depth_peel(sl, oracle, eng):
    scene = S.clutter_scene(sl, 5, n_objects=6, plane=False)
    mask = _abi.OUT_COORD | _abi.OUT_INSTANCE
    b0, r0 = both(eng, oracle, [scene], mask=mask)
    assert_geometry_equal(b0, r0, mask)
    b1, r1 = both(eng, oracle, [scene], mask=mask, depth_peel=r0.coord)
    assert_geometry_equal(b1, r1, mask)
    assert (r1.instance != r0.instance).any()


def test_public_api_roundtrip(sl, eng):
    scene = S.cube_lookat_scene(sl)
    rp = sl.RenderPass()
    res = rp.render(scene)
    assert res.rgb().shape == (480, 640, 4) and res.rgb().dtype == torch.uint8
    assert res.instance_index().dtype == torch.int16 and res.instance_index().shape == (480, 640, 1)
    assert res.coordinates().shape == (480, 640, 3)
    assert res.depth().shape == (480, 640)
    assert res.vertex_indices().dtype == torch.int32 and res.vertex_indices().shape == (480, 640, 3)
    assert res.barycentric_coeffs().shape == (480, 640, 3)
    assert res.cam_coordinates().shape == (480, 640, 4)
    assert res.normals().shape == (480, 640, 4)


def test_c4_bunny_x50_raster_stress(sl, oracle, eng):
    # BASELINE config 4: stanford bunny x50 (3.47 M triangles) on a seeded 5x10 grid at z in [1,3] m,
    # 640x480, all 8 outputs, render only -- bit-exact against the oracle
    m = sl.Mesh(S.BUNNY, physics=False)
    m.center_bbox()
    m.scale_to_bbox_diagonal(0.5)
    scene = sl.Scene((640, 480))
    rng = np.random.default_rng(4)
    for i in range(50):
        o = sl.Object(m)
        pose = np.eye(4, dtype=np.float32)
        pose[:3, :3] = S.random_rotation(rng)
        gx, gy = i % 10, i // 10
        z = 1.0 + 2.0 * (i / 49.0)
        pose[:3, 3] = [(gx - 4.5) * 0.11 * z, (gy - 2.0) * 0.16 * z, z]
        o.set_pose(torch.from_numpy(pose))
        scene.add_object(o)
    scene.light_directions = torch.tensor([[0.2, 0.5, 0.8]])
    bufs, ref = both(eng, oracle, [scene], ssao=False, shadows=False)
    assert len(np.unique(ref.instance)) == 51
    assert_geometry_equal(bufs, ref)
    assert_rgb_close(bufs, ref)


def test_batch_equals_single_scene_renders(sl, eng):
    # size-independent property for the batched path (BASELINE config 3): rendering B scenes in one
    # launch sequence is identical to rendering each scene alone
    scs = [S.clutter_scene(sl, 300 + i, n_objects=6, size=(320, 240), with_bunny=(i % 3 == 0)) for i in range(12)]
    for s in scs:
        s.manual_exposure = 1.0
    batch = eng.render(scs, _abi.OUT_ALL, ssao=True, shadows=True)
    torch.cuda.synchronize()
    for i, s in enumerate(scs):
        one = eng.render([s], _abi.OUT_ALL, ssao=True, shadows=True)
        for name in ("rgb", "coord", "cls", "instance", "normals", "vertex_idx", "bary", "cam_coord"):
            assert torch.equal(getattr(batch, name)[i], getattr(one, name)[0]), (name, i)


def test_c2_bench_workload_end_to_end(sl, oracle, eng):
    """BASELINE config C2 exactly as bench.py drives it: 20 procedural YCB-like objects (8k vertices /
    16k triangles, textured) per scene, settled on the GPU, random camera + light, 640x480, shadows +
    SSAO, auto exposure -- rendered in one launch and compared with the oracle: every geometric
    output bit for bit (instance mask included), rgb to the 8-bit tolerance."""
    import bench
    from stillleben_amd import physics, synthetic

    meshes = synthetic.ycb_like_meshes(seed=0, tex_size=256)
    scs = [bench.make_scene(sl, meshes, 777 + i) for i in range(3)]
    physics.settle_batch(scs)
    for s in scs:
        s.choose_random_camera_pose()
        s.choose_random_light_direction()
    bufs, ref = both(eng, oracle, scs, mask=_abi.OUT_GT6 | _abi.OUT_CAM_COORD)
    assert (ref.instance != 0).mean() > 0.02   # the heap is small in the YCB camera (f = 1067 px)
    assert len(np.unique(ref.instance[0])) > 10   # most of the 20 objects are visible
    assert_geometry_equal(bufs, ref, mask=_abi.OUT_GT6 | _abi.OUT_CAM_COORD)
    assert_rgb_close(bufs, ref)


def test_watertight_sheet_on_the_gpu(sl, oracle, eng):
    """along the shared edges of a Delaunay sheet no pixel is owned twice and none never -- the same
    check as tests/test_oracle_render.py::test_watertight_sheet_no_pixel_owned_twice_or_never, on the kernels' output
    (visibility raster: 64-bit atomicMin of depth | triangle id), plus bit-equality with the oracle."""
    from types import SimpleNamespace

    from test_oracle_render import check_watertight

    mask = _abi.OUT_INSTANCE | _abi.OUT_VERTEX_IDX | _abi.OUT_COORD

    def render(scene):
        bufs, ref = both(eng, oracle, [scene], mask=mask)
        assert_geometry_equal(bufs, ref, mask)
        return SimpleNamespace(instance=bufs.instance.cpu().numpy().view(np.uint16), vertex_idx=bufs.vertex_idx.cpu().numpy(),
                               coord=bufs.coord.cpu().numpy())

    for n_points, seed, tilt in ((600, 1, 0.0), (15000, 2, 0.0), (90000, 3, 0.0), (60000, 5, 50.0)):
        check_watertight(render, sl, n_points, seed, tilt)



def test_reference_format_scene_file_renders_like_the_oracle(sl, oracle, eng):
    """SURVEY 8f row f3 through the device (Scene::deserialize src/scene.cpp:802-869, Object::deserialize object.cpp:415-452,
    Mesh::deserialize mesh.cpp:1104-1115): the hand-written reference-format document tests/golden/reference_format_scene.ini --
    a scene "settled by the reference" -- is deserialised, rendered through slhip_render and every output compared with the oracle on
    the records of the same scene; then serialised, read back into a fresh scene and rendered again: the same bits."""
    import os

    from conftest import GOLDEN

    text = open(os.path.join(GOLDEN, "reference_format_scene.ini")).read().replace("@CUBE@", S.CUBE)
    scene = sl.Scene((64, 48))
    scene.deserialize(text)
    assert scene.viewport == (320, 240) and len(scene.objects) == 2
    bufs, ref = both(eng, oracle, [scene])
    assert_geometry_equal(bufs, ref)
    assert_rgb_close(bufs, ref)
    inst = bufs.instance.cpu().numpy()[0, :, :, 0]
    cls = bufs.cls.cpu().numpy()[0, :, :, 0]
    assert {1, 2} <= set(np.unique(inst).tolist())                       # both objects of the file are in the picture ...
    assert set(np.unique(cls[inst > 0]).tolist()) == {4}                 # ... with the mesh group's classIndex
    rgb = bufs.rgb.cpu().numpy()
    assert rgb[0, :, :, :3].max() > 0                                    # lit by the file's one light
    # write -> read -> render: the document we write carries everything the picture depends on.  Floats are written with six
    # significant digits (Corrade's ConfigurationValue<float> [ext]), so the first write rounds the poses the file's decimal
    # literals gave; from then on write -> read is the identity: the same text, the same bits in every output
    again = sl.Scene((64, 48))
    again.deserialize(scene.serialize())
    third = sl.Scene((64, 48))
    third.deserialize(again.serialize())
    assert third.serialize() == again.serialize()                       # (the camera's matrix -> quaternion -> text has settled)
    fourth = sl.Scene((64, 48))
    fourth.deserialize(third.serialize())
    bufs2, ref2 = both(eng, oracle, [again])
    assert_geometry_equal(bufs2, ref2)
    assert_rgb_close(bufs2, ref2)
    bufs3 = eng.render([third], _abi.OUT_ALL, ssao=True, shadows=True)
    bufs4 = eng.render([fourth], _abi.OUT_ALL, ssao=True, shadows=True)
    torch.cuda.synchronize()
    for name in ("instance", "cls", "vertex_idx", "coord", "bary", "cam_coord", "normals", "rgb"):
        a, b = getattr(bufs3, name).cpu().numpy(), getattr(bufs4, name).cpu().numpy()
        assert np.array_equal(a.view(np.uint8), b.view(np.uint8)), name
    assert (bufs3.instance.cpu().numpy() != bufs2.instance.cpu().numpy()).mean() < 1e-3
    inst2 = bufs2.instance.cpu().numpy()[0, :, :, 0]
    assert (inst2 != inst).mean() < 1e-3                                 # the rounding moves a silhouette pixel at most
    # and through the public API, as a user of the reference would (py_scene.cpp: Scene.deserialize; py_render_pass.cpp: render)
    res = sl.RenderPass().render(again)
    assert np.array_equal(res.instance_index().cpu().numpy()[:, :, 0], inst2)
