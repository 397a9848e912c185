"""Edge cases of the hot path on the GPU, each compared with the oracle: empty and ragged inputs,
odd viewports, large heaps (the lists of a scene's hull pairs and contacts have no cap but the caller's capacities), scenes
too large for the 8-per-CU LDS share, and the loud error paths of the C-ABI."""
import ctypes as C

import numpy as np
import pytest
import torch

import scenes as S
from stillleben_amd import _abi
from stillleben_amd import _settle_batch as SB
from test_gpu_render import assert_geometry_equal, assert_rgb_close, both
from test_gpu_settle import assert_bodies_equal, heap, run_both, scaled

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def eng(sl):
    from stillleben_amd._context import engine

    return engine()


# ---- render ---------------------------------------------------------------------------------------
def test_empty_scene_and_ragged_batch(sl, oracle, eng):
    """A scene without objects renders the clear values; a batch mixes it with populated scenes."""
    empty = sl.Scene((160, 120))
    full = S.clutter_scene(sl, 3, n_objects=4, size=(160, 120))
    one = S.clutter_scene(sl, 4, n_objects=1, size=(160, 120))
    bufs, ref = both(eng, oracle, [empty, full, one, sl.Scene((160, 120))])
    assert_geometry_equal(bufs, ref)
    assert_rgb_close(bufs, ref)
    inst = bufs.instance.cpu().numpy()
    assert (inst[0] == 0).all() and (inst[3] == 0).all() and (inst[1] != 0).any()
    coord = bufs.coord.cpu().numpy()
    assert (coord[0] == 3000.0).all()                      # render_pass.cpp:316 clear value


@pytest.mark.parametrize("size", [(1, 1), (7, 5), (333, 97), (641, 479)])
def test_odd_viewports(sl, oracle, eng, size):
    scene = S.clutter_scene(sl, 9, n_objects=3, size=size)
    bufs, ref = both(eng, oracle, [scene])
    assert_geometry_equal(bufs, ref)
    assert_rgb_close(bufs, ref)


def test_render_rejects_mixed_viewports_and_missing_library_state(sl, eng):
    with pytest.raises(ValueError):
        eng.render([sl.Scene((64, 48)), sl.Scene((32, 24))])
    L = _abi.lib()
    # null arguments are refused with an error string, not a crash
    assert L.slhip_settle(None, 1, None, None, None, None, None, 0, None) != 0
    assert b"null" in L.slhip_last_error()
    assert L.slhip_camera_model(None, None, None, 1, 4, 4, None, None) != 0


# ---- settle ---------------------------------------------------------------------------------------
def test_empty_and_single_body_scenes(sl, oracle):
    cube = scaled(sl, S.CUBE, 0.2)
    empty = sl.Scene((64, 48), seed=1)
    single = heap(sl, 2, 1, cube)
    gpu, ref = run_both(oracle, [empty, single, sl.Scene((64, 48), seed=3), heap(sl, 4, 2, cube)], frames=40)
    assert_bodies_equal(gpu, ref)
    assert len(gpu) == 3


def test_large_heaps(sl, oracle):
    """64 bodies dropped as one heap (round 3's body cap; its candidate hull pairs and contacts ran into the old 512 / 255
    caps), and 100 bodies (the reference has no limit on the objects of a scene, scene.cpp:278-288): every contact is
    taken, the GPU agrees with the oracle bit for bit."""
    from stillleben_amd import physics

    cube = scaled(sl, S.CUBE, 0.12)
    bunny = scaled(sl, S.BUNNY, 0.15)           # 121 hulls each
    big = heap(sl, 11, 64, cube, bunny)
    small = heap(sl, 12, 3, cube)
    gpu, ref = run_both(oracle, [big, small], frames=12)
    assert_bodies_equal(gpu, ref)
    caps = physics.settle_engine().caps(2)
    assert caps["contact_drop_steps"] == 0 and caps["pair_drop_steps"] == 0
    hundred = heap(sl, 13, 100, cube)
    gpu, ref = run_both(oracle, [hundred, small], frames=25)
    assert_bodies_equal(gpu, ref)
    caps = physics.settle_engine().caps(2)
    assert caps["contact_drop_steps"] == 0 and caps["pair_drop_steps"] == 0
    # SLHIP_MAX_BODIES = 400: the working bodies (60 KB) and the groups of six body pairs per body share a workgroup's LDS
    four_hundred = heap(sl, 14, 400, cube)
    gpu, ref = run_both(oracle, [four_hundred, small], frames=10)
    assert_bodies_equal(gpu, ref)
    caps = physics.settle_engine().caps(2)
    assert caps["scenes_dropped"] == 0 and caps["max_contacts"] > 0 and caps["group_drop_steps"] == 0


def test_hundred_bodies_settle(sl):
    """A 100-body scene settles through the public API: nothing falls through the table, (almost) everything comes to rest."""
    cube = scaled(sl, S.CUBE, 0.1)
    scene = sl.Scene((320, 240), seed=77)
    for _ in range(100):
        scene.add_object(sl.Object(cube))
    scene.simulate_tabletop_scene()
    z = np.array([float(o.pose()[2, 3]) for o in scene.objects])
    v = np.array([float(o.linear_velocity.norm()) for o in scene.objects])
    assert (z > 0.04).all()
    assert (v < 0.05).mean() > 0.9


def test_more_than_max_bodies_is_refused(sl):
    from stillleben_amd import physics

    cube = scaled(sl, S.CUBE, 0.1)
    scene = heap(sl, 5, SB.MAX_BODIES + 1, cube)
    se = physics.settle_engine()
    with pytest.raises(RuntimeError) as e:          # the host refuses before anything is launched
        SB.build_settle_batch([scene], se.pool, [(True, 0.04)])
    assert str(SB.MAX_BODIES) in str(e.value)
    # and the C-ABI refuses a hand-made oversized scene as well
    ok = heap(sl, 6, 2, cube)
    srec, bodies = SB.build_settle_batch([ok], se.pool, [(True, 0.04)])
    prm = SB.default_params(frames=1)
    prm["max_bodies_per_scene"] = SB.MAX_BODIES + 1
    L = _abi.lib()
    d = se.eng.upload_records(bodies)
    d_s = se.eng.upload_records(srec)
    hulls_d, verts_d = se.hulls_dev()
    need = C.c_uint64()
    L.slhip_settle_scratch_bytes(1, C.c_void_p(np.ascontiguousarray(prm).ctypes.data), C.byref(need))
    scr = torch.empty(int(need.value), dtype=torch.uint8, device=se.eng.device)
    st = L.slhip_settle(C.c_void_p(d_s.data_ptr()), 1, C.c_void_p(d.data_ptr()), C.c_void_p(hulls_d.data_ptr()),
                        C.c_void_p(verts_d.data_ptr()), C.c_void_p(np.ascontiguousarray(prm).ctypes.data),
                        C.c_void_p(scr.data_ptr()), scr.numel(), None)
    assert st != 0 and str(SB.MAX_BODIES).encode() in L.slhip_last_error()


def test_many_hull_scene_next_to_small_scenes(sl, oracle):
    """Sizing hints are batch maxima: one scene of 12 bunnies (1452 hulls) next to tiny scenes."""
    cube = scaled(sl, S.CUBE, 0.15)
    bunny = scaled(sl, S.BUNNY, 0.2)
    scene = sl.Scene((64, 48), seed=21)
    for _ in range(12):
        scene.add_object(sl.Object(bunny))
    from stillleben_amd import physics

    physics.prepare_tabletop(scene)
    gpu, ref = run_both(oracle, [heap(sl, 22, 2, cube), scene, heap(sl, 23, 5, cube)], frames=15)
    assert_bodies_equal(gpu, ref)


def test_static_only_and_sleeping_scene(sl, oracle):
    """All bodies static, and a scene left to sleep: nothing moves, flags and counters still agree."""
    cube = scaled(sl, S.CUBE, 0.2)
    scene = sl.Scene((64, 48), seed=31)
    for k in range(3):
        o = sl.Object(cube)
        o.static = True
        p = torch.eye(4)
        p[0, 3] = 0.5 * k
        o.set_pose(p)
        scene.add_object(o)
    gpu, ref = run_both(oracle, [scene], plane=False, frames=5)
    assert_bodies_equal(gpu, ref)
    rest = heap(sl, 32, 2, cube)
    gpu, ref = run_both(oracle, [rest], frames=250)      # long enough for every body to fall asleep
    assert_bodies_equal(gpu, ref)
    assert (gpu["flags"] & 2).all() or np.abs(gpu["lin_vel"]).max() < 1e-3


# ---- post-render I/O ('next' row f4) -----------------------------------------------------------------
def test_image_saver_takes_device_tensors(sl, tmp_path):
    """ImageSaver copies render outputs off the device on a side stream; the files hold the exact bytes."""
    from PIL import Image

    scene = S.clutter_scene(sl, 5, n_objects=4, size=(160, 120))
    result = sl.RenderPass().render(scene)
    rgb = result.rgb().cuda()                    # device tensors, whatever the context's output mode
    inst = result.instance_index().cuda().reshape(120, 160)
    assert rgb.is_cuda and rgb.dtype == torch.uint8 and inst.dtype == torch.int16
    with sl.ImageSaver() as saver:
        for i in range(12):
            saver.save(rgb, str(tmp_path / ("rgb%d.png" % i)))
        saver.save(inst, str(tmp_path / "inst16.png"))
        saver.save(inst.to(torch.uint8), str(tmp_path / "inst.png"))
    for i in range(12):
        assert np.array_equal(np.asarray(Image.open(tmp_path / ("rgb%d.png" % i))), rgb.cpu().numpy())
    assert np.array_equal(np.asarray(Image.open(tmp_path / "inst.png")), inst.to(torch.uint8).cpu().numpy())
    assert np.array_equal(np.asarray(Image.open(tmp_path / "inst16.png")).astype(np.int16), inst.cpu().numpy())
    assert inst.max() > 0


def test_cu_range_stream_runs_the_path(sl, oracle):
    """slhip_stream_create_cu_range: a settle launched on a stream confined to 32 CUs gives the same
    bits as the oracle (placement never affects results); bad ranges fail loudly."""
    from stillleben_amd import physics
    from stillleben_amd.parallel import cu_partition_streams

    cube = scaled(sl, S.CUBE, 0.15)
    scs = [heap(sl, 900 + i, 5, cube) for i in range(8)]
    se = physics.settle_engine()
    srec, bodies = SB.build_settle_batch(scs, se.pool, [(True, 0.04)] * len(scs))
    prm = SB.default_params(frames=20)
    settle_streams, render_stream = cu_partition_streams(32, 1)
    with torch.cuda.stream(settle_streams[0]):
        gpu = se.run(srec, bodies.copy(), prm)
    torch.cuda.synchronize()
    hulls, verts = se.pool.arrays()
    ref = bodies.copy()
    oracle.settle(srec, ref, hulls, verts, prm)
    assert_bodies_equal(gpu, ref)
    assert render_stream.cuda_stream != settle_streams[0].cuda_stream
    L = _abi.lib()
    h = C.c_void_p()
    assert L.slhip_stream_create_cu_range(250, 100, C.byref(h)) != 0
    assert b"exceed" in L.slhip_last_error()
    with pytest.raises(ValueError):
        cu_partition_streams(0)


def test_ssao_samples_behind_the_camera(sl, oracle, eng):
    """Surfaces 0.11-0.2 m from the eye: part of the 0.1 m SSAO hemisphere lies behind the camera plane, the
    perspective division of those samples overflows, and the z fetch is clamped to the edge texel on both
    sides (float clamp before the integer conversion) -- geometry bit-exact, RGB within the usual bar."""
    m = sl.Mesh(S.CUBE, physics=False)      # 2 m cube at the origin
    scene = sl.Scene((160, 120))
    scene.add_object(sl.Object(m))
    scene.ambient_light = torch.tensor([0.3, 0.3, 0.3])
    scene.manual_exposure = 1.0
    # eye 0.13 m off the +x face, looking along the face at a grazing angle
    scene.set_camera_look_at(torch.tensor([1.13, 0.0, 0.0]), torch.tensor([1.0, 0.6, 0.2]))
    bufs, ref = both(eng, oracle, [scene])
    assert_geometry_equal(bufs, ref)
    assert_rgb_close(bufs, ref)
    z = ref.cam_coord[0, :, :, 2]
    assert (z[z > 0] < 0.2).any()          # the case is exercised: fragments closer than 0.2 m


def test_fragment_at_the_far_plane_loses(sl, oracle, eng):
    """Found by tests/soak/soak_render.py (seed 610184, round 1): a background-plane fragment whose 24-bit depth rounds to
    0xFFFFFF equals the cleared depth, and GL_LESS rejects it -- the visibility key must not accept it.  The situation is built
    directly (a large background plane seen at a grazing angle crosses the far plane at 10 m): the soak's scene depended on where
    that round's settle left its objects."""
    scene = S.clutter_scene(sl, 11, n_objects=3, size=(640, 480))
    scene.background_plane_size = torch.tensor([40.0, 40.0])
    # rolled, so that the cut line crosses the pixel rows and fragments with z just below 10 m exist
    scene.set_camera_look_at(torch.tensor([-1.0, 0.0, 1.5]), torch.tensor([1.0, 0.0, 1.3]), torch.tensor([0.0, 0.6, 1.0]))
    mask = _abi.OUT_GT6 | _abi.OUT_CAM_COORD
    bufs, ref = both(eng, oracle, [scene], mask=mask, ssao=False, shadows=False)
    assert_geometry_equal(bufs, ref, mask=mask)
    z = ref.cam_coord[0, :, :, 2]
    assert z[(z < 2000.0)].max() > 9.9995     # the far plane (10 m) is in view ...
    assert (z > 2000.0).any()                 # ... and cuts the plane: background beyond it


def test_long_lived_scene_edge_cases(sl):
    """Scene.simulate on an empty scene, on a scene of static bodies only, after an object was added (cold start), and with a
    velocity set from outside between two calls (the body wakes, the contact state stays)."""
    cube = scaled(sl, S.CUBE, 0.1)
    empty = sl.Scene((64, 48))
    empty.simulate(0.01)
    empty.simulate(0.01)
    assert empty._phys_state.steps == 2
    st = sl.Scene((64, 48))
    o = sl.Object(cube)
    o.static = True
    st.add_object(o)
    st.simulate(0.01)
    st.simulate(0.01)
    assert np.array_equal(o.pose().numpy(), np.eye(4, dtype=np.float32))
    free = sl.Object(cube)
    p = torch.eye(4)
    p[2, 3] = 0.3
    free.set_pose(p)
    st.add_object(free)
    st.simulate(0.01)
    assert st._phys_state.steps == 1                         # the scene changed: cold start
    for _ in range(60):
        st.simulate(0.01)
    assert st._phys_state.steps == 61
    z = float(free.pose()[2, 3])
    assert 0.058 < z < 0.064                                 # at rest on the static cube: one edge of 0.0577 m + the rest offsets above its centre
    free.linear_velocity = torch.tensor([0.0, 0.0, 1.0])     # kicked from outside: wakes, keeps the state
    st.simulate(0.01)
    assert st._phys_state.steps == 62 and float(free.linear_velocity[2]) > 0.5


def test_oversized_scratch_is_refused_with_a_clear_message(sl):
    """A batch whose scratch cannot fit the device says so before allocating (which knobs to turn), instead of failing inside
    torch's allocator: 2^22 scenes of 20 objects would need terabytes of settle scratch."""
    from stillleben_amd import _settle_batch as SB
    from stillleben_amd import physics

    se = physics.settle_engine()
    prm = SB.default_params()
    prm["max_bodies_per_scene"], prm["max_hulls_per_scene"], prm["max_hull_verts_per_scene"] = 20, 160, 4000
    prm["max_hull_pairs_per_scene"], prm["max_contacts_per_scene"] = 4096, 1024
    with pytest.raises(RuntimeError, match="settle scratch of 4194304 scenes.*GB are free"):
        se.scratch(1 << 22, stream=12345, params=prm)
