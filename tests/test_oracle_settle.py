"""CPU tests pinning the settle ORACLE against the known-answer tests of SURVEY.md 8c
(k1..k7) -- the reference pins only k1 (tests/test_python.py:111-130)."""
import math

import numpy as np
import pytest
import torch

import scenes as S
from stillleben_amd import _settle_batch as SB

HALF = 0.2 / math.sqrt(3.0) / 2.0  # half edge of the cube scaled to bbox diagonal 0.2
TABLE = 0.04


def scaled_cube(sl, diag=0.2):
    m = sl.Mesh(S.CUBE)
    m.center_bbox()
    m.scale_to_bbox_diagonal(diag)
    return m


def run(oracle, scenes_, plane=True, **kw):
    pool = SB.HullPool()
    srec, bodies = SB.build_settle_batch(scenes_, pool, [(plane, TABLE)] * len(scenes_))
    hulls, verts = pool.arrays()
    oracle.settle(srec, bodies, hulls, verts, SB.default_params(**kw))
    return bodies, (srec, hulls, verts)


def test_k7_mass_properties_unit_cube(sl):
    o = sl.Object(sl.Mesh(S.CUBE))
    assert o.volume == pytest.approx(8.0, rel=1e-5)
    assert o.mass == pytest.approx(8000.0, rel=1e-5)          # density 1000 (object.h:287)
    assert np.allclose(o.inertia.numpy(), 8000.0 * 8.0 / 12.0, rtol=1e-4)
    assert np.allclose(o.inertial_frame.numpy()[:3, 3], 0.0, atol=1e-6)
    o.mass = 4000.0                                            # setMass rescales density (object.cpp:223-229)
    assert o.density == pytest.approx(500.0, rel=1e-5)


def test_k1_free_flight(sl, oracle):
    # tests/test_python.py:111-130: v = (100,0,0), one simulate(0.002) without a table
    m = sl.Mesh(S.BUNNY)
    m.center_bbox()
    m.scale_to_bbox_diagonal(0.5)
    scene = sl.Scene((640, 480))
    obj = sl.Object(m)
    obj.linear_velocity = torch.tensor([100.0, 0.0, 0.0])
    scene.add_object(obj)
    b, _ = run(oracle, [scene], plane=False, tabletop=False, dt=0.002, frames=1, substeps=1)
    v = b[0]["lin_vel"]
    assert v[0] == pytest.approx(100.0, abs=1e-7)
    assert v[1] == pytest.approx(0.0, abs=1e-7)
    assert v[2] == pytest.approx(-9.81 * 0.002, rel=1e-6)
    assert b[0]["pose"][3] == pytest.approx(100.0 * 0.002, rel=1e-5)  # semi-implicit Euler


def test_k2_resting_cube_stays(sl, oracle):
    cube = scaled_cube(sl)
    scene = sl.Scene((320, 240))
    o = sl.Object(cube)
    scene.add_object(o)
    p = torch.eye(4)
    p[2, 3] = TABLE + HALF + 0.0015  # rest offset (object.cpp:201)
    o.set_pose(p)
    b, _ = run(oracle, [scene])
    pose = b[0]["pose"].reshape(4, 4)
    assert abs(pose[2, 3] - (TABLE + HALF + 0.0015)) < 1e-3
    assert np.abs(pose[:2, 3]).max() < 1e-3
    ang = math.degrees(math.acos(min(1.0, (np.trace(pose[:3, :3]) - 1.0) / 2.0)))
    assert ang < 1.0
    assert b[0]["flags"] & SB.BODY_ASLEEP


def test_k3_two_cube_stack_is_stable(sl, oracle):
    cube = scaled_cube(sl)
    scene = sl.Scene((320, 240))
    zs = [TABLE + HALF + 0.0015, TABLE + 3 * HALF + 0.0015 + 0.003]
    for z in zs:
        o = sl.Object(cube)
        scene.add_object(o)
        p = torch.eye(4)
        p[2, 3] = z
        o.set_pose(p)
    b, _ = run(oracle, [scene])
    assert np.allclose(b["pose"][:, 11], zs, atol=2e-3)
    assert np.abs(b["pose"][:, [3, 7]]).max() < 0.01
    assert np.all(b["flags"] & SB.BODY_ASLEEP)


@pytest.mark.parametrize("seed", [0, 1, 2])
def test_k4_tilted_cube_settles_on_a_face(sl, oracle, seed):
    cube = scaled_cube(sl)
    scene = sl.Scene((320, 240))
    o = sl.Object(cube)
    scene.add_object(o)
    p = np.eye(4, dtype=np.float32)
    p[:3, :3] = S.random_rotation(np.random.default_rng(seed))
    p[2, 3] = 0.25
    o.set_pose(torch.from_numpy(p))
    b, _ = run(oracle, [scene])
    R = b[0]["pose"].reshape(4, 4)[:3, :3]
    assert np.abs(R[2, :]).max() > 0.9995           # one cube axis is vertical
    assert abs(b[0]["pose"][11] - (TABLE + HALF + 0.0015)) < 1e-3
    assert np.abs(b[0]["lin_vel"]).max() < 0.05


def test_k5_energy_does_not_increase_on_impact(sl, oracle):
    # restitution 0.1/0 average = 0.05 with the table: rebound speed << impact speed
    cube = scaled_cube(sl)
    scene = sl.Scene((320, 240))
    o = sl.Object(cube)
    scene.add_object(o)
    p = torch.eye(4)
    p[2, 3] = 1.0
    o.set_pose(p)
    pool = SB.HullPool()
    srec, bodies = SB.build_settle_batch([scene], pool, [(True, TABLE)])
    hulls, verts = pool.arrays()
    prm = SB.default_params(tabletop=False, frames=1, substeps=1)
    vmin, vmax_after = 0.0, 0.0
    hit = False
    for _ in range(80):
        oracle.settle(srec, bodies, hulls, verts, prm)
        vz = float(bodies[0]["lin_vel"][2])
        vmin = min(vmin, vz)
        if hit:
            vmax_after = max(vmax_after, vz)
        if vz > vmin + 1.0:
            hit = True
    assert hit and vmin < -3.5
    assert vmax_after < 0.1 * abs(vmin) + 0.05
    assert float(bodies[0]["pose"][11]) > TABLE + HALF - 2e-3   # no tunnelling (speculative contacts)


@pytest.mark.parametrize("seed", range(8))
def test_k6_tabletop_invariants_c1(sl, oracle, seed):
    # BASELINE config 1: 4 cubes (bbox diagonal 0.2, tools/display_mesh.py:175-178) drop-settle
    from stillleben_amd import physics

    cube = scaled_cube(sl)
    scene = sl.Scene((320, 240), seed=seed)
    for _ in range(4):
        scene.add_object(sl.Object(cube))
    has_plane = physics.prepare_tabletop(scene)
    assert has_plane
    b, (srec, hulls, verts) = run(oracle, [scene])
    assert np.all(b["pose"][:, 11] > -0.5)
    assert np.all(b["separation"] >= -0.01)
    assert np.abs(b["lin_vel"]).max() < 0.05
    assert np.all(np.isfinite(b["pose"]))
    assert oracle.overlap_any(srec, b, hulls, verts).sum() == 0
    # camera placement: all bbox corners inside the frustum, elevation in [30, 60] degrees
    SB.write_back([scene], b)
    scene.choose_random_camera_pose()
    cam = scene.camera_pose().numpy()
    w2c = np.linalg.inv(cam)
    P = scene.projection_matrix().numpy()
    for o in scene.objects:
        c = o.mesh.bbox.corners() @ o.pose().numpy()[:3, :3].T + o.pose().numpy()[:3, 3]
        cc = c @ w2c[:3, :3].T + w2c[:3, 3]
        clip = np.concatenate([cc, np.ones((8, 1))], axis=1) @ P.T
        ndc = clip[:, :2] / clip[:, 3:4]
        assert np.all(np.abs(ndc) <= 1.0 + 1e-3) and np.all(cc[:, 2] > 0)
    elev = math.degrees(math.asin(-cam[2, 2]))  # camera looks along +z of its frame
    assert 30.0 - 1e-3 <= elev <= 60.0 + 1e-3


def test_heap_of_20_mixed_objects(sl, oracle):
    from stillleben_amd import physics

    cube = scaled_cube(sl, 0.15)
    bunny = sl.Mesh(S.BUNNY)
    bunny.center_bbox()
    bunny.scale_to_bbox_diagonal(0.2)
    scene = sl.Scene((640, 480), seed=3)
    for i in range(20):
        scene.add_object(sl.Object(bunny if i % 5 == 4 else cube))
    physics.prepare_tabletop(scene)
    b, (srec, hulls, verts) = run(oracle, [scene])
    assert np.all(b["pose"][:, 11] > 0.0)
    assert np.abs(b["lin_vel"]).max() < 0.05
    assert np.all(b["separation"] >= -0.01)
    assert oracle.overlap_any(srec, b, hulls, verts).sum() == 0


def test_overlap_query(sl, oracle):
    cube = scaled_cube(sl)
    scene = sl.Scene((320, 240))
    a, b_ = sl.Object(cube), sl.Object(cube)
    scene.add_object(a)
    scene.add_object(b_)
    p = torch.eye(4)
    p[0, 3] = 0.05
    b_.set_pose(p)
    pool = SB.HullPool()
    srec, bodies = SB.build_settle_batch([scene], pool, [(False, 0.0)])
    hulls, verts = pool.arrays()
    assert oracle.overlap_any(srec, bodies, hulls, verts).tolist() == [1, 1]
    p[0, 3] = 0.3
    b_.set_pose(p)
    srec, bodies = SB.build_settle_batch([scene], pool, [(False, 0.0)])
    assert oracle.overlap_any(srec, bodies, hulls, verts).tolist() == [0, 0]


def _drive_scene(sl):
    cube = scaled_cube(sl, 0.1)
    scene = sl.Scene((320, 240))
    tool, target = sl.Object(cube), sl.Object(cube)
    scene.add_object(target)
    p = torch.eye(4)
    p[0, 3] = 0.15
    target.set_pose(p)
    init = torch.eye(4)
    sim = sl.ManipulationSim(scene, tool, init)
    return scene, sim, tool, target


def test_manipulation_drive_spring(sl, oracle):
    # S10: the manipulator is pulled to the goal by the spring drive, keeps its orientation
    # (locked axes) and pushes a free object out of the way; the force limit caps its acceleration.
    scene, sim, tool, target = _drive_scene(sl)
    pool = SB.HullPool()
    tool._drive["target"] = np.array([0.3, 0.0, 0.0], np.float32)
    srec, bodies = SB.build_settle_batch([scene], pool, [(False, 0.0)])
    hulls, verts = pool.arrays()
    prm = SB.default_params(tabletop=False, dt=0.005, frames=1, substeps=1)
    prm["gravity"] = (0.0, 0.0, 0.0)
    mass = tool.mass
    vmax = 0.0
    for i in range(400):
        oracle.settle(srec, bodies, hulls, verts, prm)
        v = float(bodies[1]["lin_vel"][0])
        if i == 0:
            # force limit 60 N: dv <= F dt / m (manipulation_sim.cpp:55)
            assert 0 < v <= 60.0 * 0.005 / mass * 1.0001
        vmax = max(vmax, v)
    tool_x, target_x = bodies[1]["pose"][3], bodies[0]["pose"][3]
    assert abs(tool_x - 0.3) < 0.05                      # reached the goal neighbourhood
    assert target_x > 0.3                                # the free cube was pushed ahead
    R = bodies[1]["pose"].reshape(4, 4)[:3, :3]
    assert np.allclose(R, np.eye(3), atol=2e-2)          # rotation stayed locked
    assert np.all(np.isfinite(bodies["pose"]))


def _heap(sl, seed, n=6):
    rng = np.random.default_rng(seed)
    cube = scaled_cube(sl)
    scene = sl.Scene((320, 240), seed=seed)
    for i in range(n):
        o = sl.Object(cube)
        p = np.eye(4, dtype=np.float32)
        p[:3, :3] = S.random_rotation(rng)
        p[:3, 3] = [rng.uniform(-0.05, 0.05), rng.uniform(-0.05, 0.05), TABLE + 0.1 + 0.13 * i]
        o.set_pose(torch.from_numpy(p))
        scene.add_object(o)
    return scene


def test_contact_state_outlives_the_call(sl, oracle):
    """slhip_settle_params.resume: k calls of one frame == one call of k frames, bit for bit (PhysX keeps its contact cache for
    the life of the PxScene, scene.cpp:720-739) -- and a cold start per call is NOT the same thing (what rounds 1-3 did)."""
    frames = 30
    pool = SB.HullPool()
    srec, b0 = SB.build_settle_batch([_heap(sl, 3), _heap(sl, 4)], pool, [(True, TABLE)] * 2)
    hulls, verts = pool.arrays()
    one = b0.copy()
    oracle.settle(srec, one, hulls, verts, SB.default_params(frames=frames))
    many, cold = b0.copy(), b0.copy()
    st = oracle.SettleState()
    for f in range(frames):
        prm = SB.default_params(frames=1)
        prm["resume"] = 4 * f
        oracle.settle(srec, many, hulls, verts, prm, state=st)
        oracle.settle(srec, cold, hulls, verts, SB.default_params(frames=1))
    assert one.tobytes() == many.tobytes()
    assert one.tobytes() != cold.tobytes()
    # a resume that does not match the steps run so far is refused
    prm = SB.default_params(frames=1)
    prm["resume"] = 7
    with pytest.raises(RuntimeError):
        oracle.settle(srec, many, hulls, verts, prm, state=st)
