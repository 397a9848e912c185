"""GPU parity tests of the settle half: slhip_settle / slhip_overlap_any (through the C-ABI)
vs the CPU oracle.  Bar: BIT-EXACT body state (pose, velocities, separation, flags) after the
full 400-step settle -- the algorithm uses only exactly rounded operations in a fixed order."""
import math

import numpy as np
import pytest
import torch

import scenes as S
from stillleben_amd import _settle_batch as SB

pytestmark = pytest.mark.gpu

TABLE = 0.04


@pytest.fixture(autouse=True, params=["lockstep", "persistent"])
def launch_form(request, monkeypatch):
    """Every test of this file runs in both launch forms of slhip_settle: six launches per step over the whole batch (what large
    batches -- the benchmark's -- take) and the one-launch persistent kernel (what batches of this size take by default)."""
    monkeypatch.setenv("SLHIP_SETTLE_PERSISTENT", "1" if request.param == "persistent" else "0")
    return request.param


def scaled(sl, path, diag):
    m = sl.Mesh(path)
    m.center_bbox()
    m.scale_to_bbox_diagonal(diag)
    return m


def heap(sl, seed, n, cube, bunny=None):
    from stillleben_amd import physics

    scene = sl.Scene((320, 240), seed=seed)
    for i in range(n):
        scene.add_object(sl.Object(bunny if (bunny is not None and i % 5 == 4) else cube))
    physics.prepare_tabletop(scene)
    return scene


def run_both(oracle, scenes_, plane=True, **kw):
    from stillleben_amd import physics

    se = physics.settle_engine()
    srec, bodies = SB.build_settle_batch(scenes_, se.pool, [(plane, TABLE)] * len(scenes_))
    prm = SB.default_params(**kw)
    gpu = se.run(srec, bodies.copy(), prm)
    hulls, verts = se.pool.arrays()
    ref = bodies.copy()
    prm["max_hull_pairs_per_scene"] = se.last_params["max_hull_pairs_per_scene"]      # (the host path grows the lists when a heap
    prm["max_contacts_per_scene"] = se.last_params["max_contacts_per_scene"]          #  needs it: the oracle gets the same)
    prm["max_body_pairs_per_scene"] = se.last_params["max_body_pairs_per_scene"]
    oracle.settle(srec, ref, hulls, verts, prm)
    return gpu, ref


def assert_bodies_equal(gpu, ref):
    for name in ("pose", "lin_vel", "ang_vel", "separation", "flags", "stuck_counter", "wake_counter", "stab"):
        a, b = np.ascontiguousarray(gpu[name]), np.ascontiguousarray(ref[name])
        if not np.array_equal(a.view(np.uint8), b.view(np.uint8)):
            bad = np.argwhere(a != b)
            raise AssertionError("%s differs for %d entries; first %s: %r vs %r"
                                 % (name, len(bad), bad[0], a[tuple(bad[0])], b[tuple(bad[0])]))


def test_free_flight_step(sl, oracle):
    m = scaled(sl, S.BUNNY, 0.5)
    scene = sl.Scene((640, 480))
    o = sl.Object(m)
    o.linear_velocity = torch.tensor([100.0, 0.0, 0.0])
    scene.add_object(o)
    gpu, ref = run_both(oracle, [scene], plane=False, tabletop=False, dt=0.002, frames=1, substeps=1)
    assert_bodies_equal(gpu, ref)
    assert gpu[0]["lin_vel"][0] == pytest.approx(100.0, abs=1e-7)
    assert gpu[0]["lin_vel"][2] < -1e-4


@pytest.mark.parametrize("seed", [0, 1, 2, 3])
def test_c1_four_cubes_full_settle(sl, oracle, seed):
    cube = scaled(sl, S.CUBE, 0.2)
    gpu, ref = run_both(oracle, [heap(sl, seed, 4, cube)])
    assert_bodies_equal(gpu, ref)
    assert np.abs(gpu["lin_vel"]).max() < 0.05


@pytest.mark.parametrize("frames", [3, 25, 100])
def test_c2_twenty_objects(sl, oracle, frames):
    cube = scaled(sl, S.CUBE, 0.15)
    bunny = scaled(sl, S.BUNNY, 0.2)
    gpu, ref = run_both(oracle, [heap(sl, 7, 20, cube, bunny)], frames=frames)
    assert_bodies_equal(gpu, ref)


def test_batch_of_scenes(sl, oracle):
    cube = scaled(sl, S.CUBE, 0.15)
    bunny = scaled(sl, S.BUNNY, 0.2)
    scs = [heap(sl, 100 + i, 3 + 2 * i, cube, bunny) for i in range(6)]
    gpu, ref = run_both(oracle, scs, frames=40)
    assert_bodies_equal(gpu, ref)


@pytest.mark.parametrize("form", ["lockstep", "persistent", "lockstep, then persistent from frame 25"])
def test_both_forms_of_the_step_give_the_same_bits(sl, oracle, monkeypatch, form, launch_form):
    if launch_form != "lockstep":
        pytest.skip("sets the form itself")
    """slhip_settle has two launch forms -- six launches per step over the whole batch (large batches), one launch in which a
    wave takes a scene through every step (k_w_persistent: small batches; the default of every other test of this file) -- built
    from the same per-scene and per-pair functions.  Both against the oracle on a batch of heaps with ragged sizes, a resumed
    second call included."""
    monkeypatch.setenv("SLHIP_SETTLE_PERSISTENT", "1" if form == "persistent" else "0")
    if "then" in form:
        monkeypatch.setenv("SLHIP_SETTLE_SWITCH_FRAME", "25")      # (both forms work on the same state: a call may change between them)
    cube = scaled(sl, S.CUBE, 0.15)
    bunny = scaled(sl, S.BUNNY, 0.2)
    scs = [heap(sl, 300 + i, 2 + 3 * i, cube, bunny) for i in range(7)]
    gpu, ref = run_both(oracle, scs, frames=60)
    assert_bodies_equal(gpu, ref)
    assert np.abs(gpu["lin_vel"]).max() > 0.0


_NATIVE = {}


def test_natively_decomposed_meshes_settle_bit_exact(sl, oracle):
    """Row S1 feeding the settle: collision hulls made by the IN-TREE decomposition (hulls._compute_hulls: no V-HACD fixture, no
    cache) -- a power drill (10 hulls) and a mug (34) -- dropped with cubes; the kernels against the oracle on those hulls."""
    from stillleben_amd import hulls as H
    from stillleben_amd import physics, synthetic
    from stillleben_amd.mesh import Mesh

    if not _NATIVE:      # (the decomposition takes seconds: once for both launch forms)
        for name in ("035_power_drill", "025_mug"):
            cm, _ = synthetic.make_class_mesh(name, 0, 8192, 64)
            hs = H._compute_hulls(cm, False)
            assert len(hs) > 1
            _NATIVE[name] = Mesh.from_data(cm, hs, "memory://%s_native" % name)
    cube = scaled(sl, S.CUBE, 0.12)
    scene = sl.Scene((320, 240), seed=11)
    for m in (_NATIVE["035_power_drill"], cube, _NATIVE["025_mug"], cube, _NATIVE["035_power_drill"]):
        scene.add_object(sl.Object(m))
    physics.prepare_tabletop(scene)
    gpu, ref = run_both(oracle, [scene], frames=50)
    assert_bodies_equal(gpu, ref)
    assert (gpu["pose"][:, 11] > TABLE).all()


def test_static_object_and_no_plane(sl, oracle):
    cube = scaled(sl, S.CUBE, 0.2)
    big = scaled(sl, S.CUBE, 1.0)
    scene = sl.Scene((320, 240), seed=5)
    base = sl.Object(big)
    base.static = True
    scene.add_object(base)
    for _ in range(3):
        scene.add_object(sl.Object(cube))
    from stillleben_amd import physics

    assert physics.prepare_tabletop(scene) is False  # a static object => no table (scene.cpp:629)
    gpu, ref = run_both(oracle, [scene], plane=False, frames=60)
    assert_bodies_equal(gpu, ref)
    assert np.array_equal(gpu[0]["pose"], np.eye(4, dtype=np.float32).reshape(-1))


def test_overlap_query(sl, oracle):
    from stillleben_amd import physics

    cube = scaled(sl, S.CUBE, 0.2)
    bunny = scaled(sl, S.BUNNY, 0.2)
    rng = np.random.default_rng(0)
    scene = sl.Scene((320, 240))
    for i in range(12):
        o = sl.Object(bunny if i % 3 == 0 else cube)
        p = np.eye(4, dtype=np.float32)
        p[:3, :3] = S.random_rotation(rng)
        p[:3, 3] = rng.uniform(-0.25, 0.25, 3)
        o.set_pose(torch.from_numpy(p))
        scene.add_object(o)
    se = physics.settle_engine()
    srec, bodies = SB.build_settle_batch([scene], se.pool, [(False, 0.0)])
    flags = se.overlap(srec, bodies)
    hulls, verts = se.pool.arrays()
    ref = oracle.overlap_any(srec, bodies, hulls, verts)
    assert np.array_equal(flags, ref)
    assert 0 < ref.sum() < 12


def test_public_api(sl):
    cube = scaled(sl, S.CUBE, 0.2)
    scene = sl.Scene((320, 240), seed=1)
    for _ in range(4):
        scene.add_object(sl.Object(cube))
    calls = []
    scene.simulate_tabletop_scene(vis_cb=lambda i: calls.append(i))
    assert calls == list(range(100))
    for o in scene.objects:
        assert o.pose()[2, 3] > 0.04
        assert float(o.linear_velocity.abs().max()) < 0.05
    scene.check_collisions()
    assert all(o.separation == 0.0 for o in scene.objects)


def test_manipulation_sim_parity_and_api(sl, oracle):
    from stillleben_amd import physics

    cube = scaled(sl, S.CUBE, 0.1)
    scene = sl.Scene((320, 240))
    target = sl.Object(cube)
    scene.add_object(target)
    p = torch.eye(4)
    p[0, 3] = 0.15
    target.set_pose(p)
    tool = sl.Object(cube)
    sim = sl.ManipulationSim(scene, tool, torch.eye(4))
    assert tool in scene.objects
    tool._drive["target"] = np.array([0.3, 0.0, 0.0], np.float32)
    se = physics.settle_engine()
    srec, bodies = SB.build_settle_batch([scene], se.pool, [(False, 0.0)])
    prm = SB.default_params(tabletop=False, dt=0.005, frames=50, substeps=1)
    gpu = se.run(srec, bodies.copy(), prm)
    hulls, verts = se.pool.arrays()
    ref = bodies.copy()
    oracle.settle(srec, ref, hulls, verts, prm)
    assert_bodies_equal(gpu, ref)
    # public API: step towards a goal
    goal = torch.eye(4)
    goal[0, 3] = 0.05
    x0 = float(tool.pose()[0, 3])
    for _ in range(20):
        sim.step(goal, 0.005)
    assert float(tool.pose()[0, 3]) > x0


def test_batch_settle_equals_single(sl, oracle):
    # BASELINE config 3 property: a scene's result does not depend on what else is in the launch
    from stillleben_amd import physics

    cube = scaled(sl, S.CUBE, 0.15)
    scs = [heap(sl, 500 + i, 6, cube) for i in range(32)]
    se = physics.settle_engine()
    planes = [(True, TABLE)] * len(scs)
    srec, bodies = SB.build_settle_batch(scs, se.pool, planes)
    prm = SB.default_params(frames=30)
    all_ = se.run(srec, bodies.copy(), prm)
    for i in (0, 13, 31):
        s1, b1 = SB.build_settle_batch([scs[i]], se.pool, [planes[i]])
        one = se.run(s1, b1.copy(), prm)
        lo, hi = int(srec[i]["body_begin"]), int(srec[i]["body_end"])
        assert_bodies_equal(all_[lo:hi], one)


def test_c2_bench_workload_full_launch(sl, oracle):
    """BASELINE config C2 at the benchmark's launch size: 2048 scenes of 20 procedural YCB-like
    objects in ONE slhip_settle launch (8 resident scenes per CU).  (i) the first scenes are
    compared bit for bit with the oracle after the full 400 steps; (ii) copies of scene 0 planted
    across the launch (other CUs, other launch rounds) must reproduce it exactly -- a scene's result
    may not depend on placement, neighbours or launch order."""
    import bench
    from stillleben_amd import physics, synthetic

    meshes = synthetic.ycb_like_meshes(seed=0, tex_size=64)
    distinct = [bench.make_scene(sl, meshes, 4242 + i) for i in range(12)]
    se = physics.settle_engine()
    planes_d = [(physics.prepare_tabletop(s), physics.PLANE_HALF_Z) for s in distinct]
    n_launch = 2048
    copies = {257: 0, 1031: 0, 2047: 0, 1500: 3}
    scs = [distinct[copies[i]] if i in copies else distinct[i % len(distinct)] for i in range(n_launch)]
    planes = [planes_d[copies[i]] if i in copies else planes_d[i % len(distinct)] for i in range(n_launch)]
    srec, bodies = SB.build_settle_batch(scs, se.pool, planes)
    prm = SB.default_params(tabletop=True)
    gpu = se.run(srec, bodies.copy(), prm)
    hulls, verts = se.pool.arrays()
    n_ref = 6
    hi = int(srec[n_ref - 1]["body_end"])
    ref = bodies[:hi].copy()
    oracle.settle(srec[:n_ref], ref, hulls, verts, prm)
    assert_bodies_equal(gpu[:hi], ref)
    for i, src in copies.items():
        a0, a1 = int(srec[i]["body_begin"]), int(srec[i]["body_end"])
        b0, b1 = int(srec[src]["body_begin"]), int(srec[src]["body_end"])
        assert_bodies_equal(gpu[a0:a1], gpu[b0:b1])
    # the piles are at rest (tests/test_oracle_k6.py holds the 64-scene figure: 97 %)
    speed = np.linalg.norm(gpu["lin_vel"][:, :3], axis=1)
    assert np.mean(speed < 0.05) >= 0.9


def test_c2_soak_distinct_scenes(sl, oracle):
    """24 further C2 scenes (other seeds than the launch test), full 400-step settle, every body bit for bit:
    exercises the pair cache, the warm-started and capped tilt runs and the size-ordered colouring on
    many more contact configurations than the hand-built heaps."""
    import bench
    from stillleben_amd import physics, synthetic

    meshes = synthetic.ycb_like_meshes(seed=0, tex_size=64)
    scs = [bench.make_scene(sl, meshes, 77000 + i) for i in range(24)]
    se = physics.settle_engine()
    planes = [(physics.prepare_tabletop(s), physics.PLANE_HALF_Z) for s in scs]
    srec, bodies = SB.build_settle_batch(scs, se.pool, planes)
    prm = SB.default_params(tabletop=True)
    gpu = se.run(srec, bodies.copy(), prm)
    hulls, verts = se.pool.arrays()
    ref = bodies.copy()
    oracle.settle(srec, ref, hulls, verts, prm)
    assert_bodies_equal(gpu, ref)


def _yawed(pose, angle):
    c, s_ = math.cos(angle), math.sin(angle)
    pose[:2, :2] = [[c, -s_], [s_, c]]
    return pose


def test_physics_kat_scenarios_match_the_oracle(sl, oracle):
    """The scenarios of tests/test_oracle_physics_kat.py (tilted gravity, impacts above / below the bounce threshold,
    free spin, clamped spin, cube columns, head-on collision) through slhip_settle: bit-exact with the oracle, so the
    known answers checked there on the CPU hold for the HIP path as well."""
    import math

    import test_oracle_physics_kat as K
    from stillleben_amd import physics

    se = physics.settle_engine()
    h = K.half_edge()
    th = math.atan(0.5)
    cases = [
        (K.build(sl, [K.at(0, 0, K.TABLE + h + 0.0015)]), dict(frames=100, gravity=(K.G * math.sin(th), 0.0, -K.G * math.cos(th)))),
        (K.build(sl, [K.at(0, 0, K.TABLE + h + 0.05)], vel=[(0, 0, -3.9)]), dict(frames=40, dt=0.0025)),
        (K.build(sl, [K.at(0, 0, 1.0)], plane=False, vel=[(0.3, -0.2, 0.1)], ang=[(300.0, 0.0, 400.0)]), dict(frames=50, gravity=(0.0, 0.0, 0.0))),
        (K.build(sl, [K.at(0, 0, K.TABLE + h + 0.0015 + k * (2 * h + 0.003)) for k in range(6)]), dict(frames=400)),   # the column of six: warm start + persistent manifolds
        (K.build(sl, [K.at(-h - 0.02, 0, 1.0), K.at(h + 0.02, 0, 1.0)], plane=False, vel=[(1.5, 0, 0), (-1.5, 0, 0)]), dict(frames=30, dt=0.002, gravity=(0.0, 0.0, 0.0))),
        (K.build(sl, [K.at((c - 1) * (2 * h + 0.001), 0, K.TABLE + h + 0.0015 + k * (2 * h + 0.003)) for k in range(4) for c in range(3)]), dict(frames=400)),   # the wall
        (K.build(sl, [K.at(0, 0, K.TABLE + h + 0.0015)], vel=[(0.8, 0, 0)]), dict(frames=120, gravity=(K.G * math.sin(math.atan(0.15)), 0.0, -K.G * math.cos(math.atan(0.15))))),   # the sliding box
        (K.build(sl, [K.at(-h - 0.0005, 0, K.TABLE + h + 0.0015), K.at(h + 0.0005, 0, K.TABLE + h + 0.0015), K.at(0, 0, K.TABLE + 3 * h + 0.0045)]), dict(frames=150)),   # the pile of three
        (K.build(sl, [K.at(0, 0, K.TABLE + h + 0.0015), _yawed(K.at(0, 0, K.TABLE + 3 * h + 0.0045), math.pi / 4)]), dict(frames=60)),   # face manifold = corners of an octagon
        (K.build(sl, [K.at(0, 0, K.TABLE + h + 0.0015 + k * (2 * h + 0.003)) for k in range(8)]), dict(frames=400)),   # the column of eight: stands since round 6 (the patches' centre rows)
        (K.build(sl, [K.at(0, 0, K.TABLE + h + 0.0015 + k * (2 * h + 0.003)) for k in range(12)]), dict(frames=200)),   # ... and of twelve
        # the stabilisation (scene.cpp:163): a cube kept awake is lightened, damped and, after 1.5 s, frozen; a pile of two beside a falling cube
        (K.build(sl, [K.at(0, 0, K.TABLE + h + 0.0015)]), dict(frames=300, sleep_threshold=0.0, frozen=True)),
        (K.build(sl, [K.at(0, 0, K.TABLE + h + 0.0015), K.at(0, 0, K.TABLE + 3 * h + 0.0045), K.at(1.0, 0, 3.0)]), dict(frames=30, sleep_threshold=0.0)),
        (K.build(sl, [K.at(0, 0, K.TABLE + h + 0.0015 + k * (2 * h + 0.003)) for k in range(4)]), dict(frames=200, stabilization_threshold=0.0)),   # ... and switched off
    ]
    for (srec, bodies, hulls, verts), kw in cases:
        prm = SB.default_params(tabletop=False, dt=kw.get("dt", 0.01), frames=kw["frames"], substeps=1)
        prm["gravity"] = kw.get("gravity", (0.0, 0.0, -K.G))
        for key in ("sleep_threshold", "stabilization_threshold"):
            if key in kw:
                prm[key] = kw[key]
        # the KAT builder used its own hull pool: rebuild the batch on the engine's pool (same hulls, other indices)
        gb = bodies.copy()
        cube = K.cube_mesh(sl)
        hb, he, _, _ = se.pool.register(cube)
        gb["hull_begin"], gb["hull_end"] = hb, he
        gpu = se.run(srec, gb, prm)
        ref = bodies.copy()
        oracle.settle(srec, ref, hulls, verts, prm)
        for name in ("pose", "lin_vel", "ang_vel", "separation", "flags", "wake_counter", "stab"):
            a, b = np.ascontiguousarray(gpu[name]), np.ascontiguousarray(ref[name])
            assert np.array_equal(a.view(np.uint8), b.view(np.uint8)), name
        if kw.get("frozen"):
            assert int(gpu[0]["flags"]) & SB.BODY_FROZEN
    # the 3 : 1 box on an incline, just beyond its toppling threshold (tan = 0.37 > 1/3): it goes over on the device as well
    a_ = 0.02
    th = math.atan(0.37)
    scene = sl.Scene((64, 48))
    o = sl.Object(K.box_mesh(sl, (a_, a_, 3 * a_)))
    scene.add_object(o)
    o.set_pose(torch.from_numpy(K.at(0, 0, K.TABLE + 3 * a_ + 0.0015)))
    srec, bodies = SB.build_settle_batch([scene], se.pool, [(True, K.TABLE)])
    prm = SB.default_params(tabletop=False, dt=0.01, frames=180, substeps=1)
    prm["gravity"] = (K.G * math.sin(th), 0.0, -K.G * math.cos(th))
    gpu = se.run(srec, bodies.copy(), prm)
    hulls, verts = se.pool.arrays()
    ref = bodies.copy()
    oracle.settle(srec, ref, hulls, verts, prm)
    assert_bodies_equal(gpu, ref)
    assert math.degrees(math.acos(max(-1.0, min(1.0, float(gpu[0]["pose"][10]))))) > 45.0


def test_rolling_and_drive_known_answers_match_the_oracle(sl, oracle):
    """The rolling cylinder and the D6 drive's equilibrium / force limit of tests/test_oracle_physics_kat.py through slhip_settle: the
    oracle's bits, and the known answers themselves on the device's results."""
    import math

    import test_oracle_physics_kat as K
    from stillleben_amd import physics

    se = physics.settle_engine()
    # the cylinder: v = omega r, (2/3) g sin(theta) to within the prism's facets
    th = math.atan(0.1)
    r = 0.05
    scene = sl.Scene((64, 48))
    o = sl.Object(K.prism_mesh(sl, 32, r, 0.04))
    scene.add_object(o)
    o.set_pose(torch.from_numpy(K.at(0, 0, K.TABLE + r + 0.0015)))
    srec, bodies = SB.build_settle_batch([scene], se.pool, [(True, K.TABLE)])
    hulls, verts = se.pool.arrays()
    prm = SB.default_params(tabletop=False, dt=0.01, frames=60, substeps=1)
    prm["gravity"] = (0.0, K.G * math.sin(th), -K.G * math.cos(th))
    gpu = se.run(srec, bodies.copy(), prm)
    ref = bodies.copy()
    oracle.settle(srec, ref, hulls, verts, prm)
    assert_bodies_equal(gpu, ref)
    v, w = float(gpu[0]["lin_vel"][1]), float(gpu[0]["ang_vel"][0])
    assert v > 0.2 and abs(v + w * r) < 0.02 * v
    # the drive: m g / k below the target; 60 dt / m per step at the force limit
    cube = K.cube_mesh(sl, 0.1)
    for target, z0, frames, gravity in (((0.0, 0.0, 1.0), 1.0, 1500, (0.0, 0.0, -K.G)), ((5.0, 0.0, 0.0), 0.0, 5, (0.0, 0.0, 0.0))):
        scene = sl.Scene((64, 48))
        tool = sl.Object(cube)
        p = torch.eye(4)
        p[2, 3] = z0
        sl.ManipulationSim(scene, tool, p)
        tool._drive["target"] = np.asarray(target, np.float32)
        se.pool.__dict__.pop("_body_templates", None)
        srec, bodies = SB.build_settle_batch([scene], se.pool, [(False, 0.0)])
        hulls, verts = se.pool.arrays()
        prm = SB.default_params(tabletop=False, dt=0.01, frames=frames, substeps=1)
        prm["gravity"] = gravity
        gpu = se.run(srec, bodies.copy(), prm)
        ref = bodies.copy()
        oracle.settle(srec, ref, hulls, verts, prm)
        assert_bodies_equal(gpu, ref)
        m = 1.0 / float(bodies[0]["inv_mass"])
        if frames > 100:
            assert abs(float(gpu[0]["pose"][11]) - (1.0 - m * K.G / 600.0)) < 2e-6
        else:
            assert abs(float(gpu[0]["lin_vel"][0]) - 60.0 / m * 0.01 * frames) < 1e-4 * 60.0 / m * 0.01 * frames


def test_solver_wave_packing_and_odd_batches(sl, oracle, monkeypatch):
    """The lockstep pipeline on a batch of mixed scenes (tabletop settle): solver waves of one, two or four cost-sorted scenes by
    need (default), at most one and at most four per wave give the oracle's bits, also when the last multi-scene solver wave holds
    a single scene."""
    cube = scaled(sl, S.CUBE, 0.15)
    bunny = scaled(sl, S.BUNNY, 0.2)
    scs = [heap(sl, 300 + i, 4 + 3 * i, cube, bunny) for i in range(6)]
    gpu, ref = run_both(oracle, scs, frames=60)
    assert_bodies_equal(gpu, ref)
    for spw in ("1", "4"):
        monkeypatch.setenv("SLHIP_SOLVE_SPW", spw)
        gpu2s, _ = run_both(oracle, scs, frames=60)
        assert_bodies_equal(gpu2s, ref)
    gpu2, ref2 = run_both(oracle, scs + [heap(sl, 310, 9, cube, bunny)], frames=60)   # odd number: the last solver wave holds one scene
    assert_bodies_equal(gpu2, ref2)


def test_refused_first_scene_leaves_the_others_exact(sl, oracle):
    """A scene that exceeds the sizing hints is left untouched and reported -- also when it is scene 0, whose block clears the
    per-step cost classes (round-2 advisor finding: the clear must not sit behind the refusal's early return)."""
    from stillleben_amd import physics

    cube = scaled(sl, S.CUBE, 0.15)
    bunny = scaled(sl, S.BUNNY, 0.2)
    scs = [heap(sl, 400, 6, cube, bunny)] + [heap(sl, 401 + i, 5, cube) for i in range(5)]   # scene 0 carries the many-hull bunny
    se = physics.settle_engine()
    srec, bodies = SB.build_settle_batch(scs, se.pool, [(True, TABLE)] * len(scs))
    prm = SB.default_params(frames=30)
    hulls, verts = se.pool.arrays()
    n_h = (bodies["hull_end"] - bodies["hull_begin"]).astype(np.int64)
    per_scene = [int(n_h[int(r["body_begin"]):int(r["body_end"])].sum()) for r in srec]
    assert per_scene[0] > max(per_scene[1:])
    status = se.run_with_status(srec, bodies.copy(), prm, max_hulls_per_scene=max(per_scene[1:]))
    gpu, st = status
    assert st[0] != 0 and (st[1:] == 0).all()
    ref = bodies.copy()
    oracle.settle(srec[1:], ref, hulls, verts, prm)      # scene 0 untouched on both sides
    assert_bodies_equal(gpu, ref)


def _bunny_pile_batch(sl, n_scenes=4):
    from stillleben_amd import physics

    cube = scaled(sl, S.CUBE, 0.12)
    bunny = scaled(sl, S.BUNNY, 0.25)
    scs = []
    for i in range(n_scenes):                             # bunnies (121 hulls each) dropped on each other: many hull pairs
        scene = sl.Scene((320, 240), seed=500 + i)
        for k in range(12):                               # (six bunnies + six cubes: ~300 contacts offered in a step)
            scene.add_object(sl.Object(bunny if k < 6 else cube))
        physics.prepare_tabletop(scene)
        scs.append(scene)
    return scs


def _oracle_caps(oracle, srec, ref, hulls, verts, prm):
    import ctypes as C

    L = oracle.lib()
    oc = np.zeros((len(srec), 7), np.uint32)
    L.slref_settle_set_caps.argtypes = [C.c_void_p]
    L.slref_settle_set_caps(C.c_void_p(oc.ctypes.data))
    try:
        oracle.settle(srec, ref, hulls, verts, prm)
    finally:
        L.slref_settle_set_caps(None)
    return oc


def test_no_contact_is_dropped(sl, oracle):
    """PhysX has no caps (scene.cpp:738-739).  The pile that saturated round 3's 255-contact / 512-pair caps: with the default
    list capacities NOTHING is dropped (slhip_settle_caps says so), the contacts beyond the solver's LDS-resident part are swept
    from global memory, and the bodies are the oracle's, bit for bit."""
    from stillleben_amd import physics

    scs = _bunny_pile_batch(sl)
    se = physics.settle_engine()
    srec, bodies = SB.build_settle_batch(scs, se.pool, [(True, TABLE)] * len(scs))
    prm = SB.default_params(frames=40)
    prm["pair_contact_budget"] = 0                                                   # every point goes to the solver, as in PhysX
    prm["max_hull_pairs_per_scene"], prm["max_contacts_per_scene"] = 16384, 4096     # (bunny on bunny: thousands of candidate hull pairs)
    gpu, caps = se.run_with_caps(srec, bodies.copy(), prm)
    hulls, verts = se.pool.arrays()
    ref = bodies.copy()
    oc = _oracle_caps(oracle, srec, ref, hulls, verts, SB.sizing_hints(prm, srec, bodies, hulls))
    assert_bodies_equal(gpu, ref)
    assert caps["contact_drop_steps"] == 0 and caps["pair_drop_steps"] == 0 and caps["scenes_dropped"] == 0
    assert int(oc[:, 0].sum()) == 0 and int(oc[:, 1].sum()) == 0
    assert caps["max_contacts"] == int(oc[:, 2].max()) and caps["max_hull_pairs"] == int(oc[:, 3].max())
    assert caps["contact_sum"] == int(oc[:, 6].astype(np.uint64).sum()) > 0 and caps["group_drop_steps"] == int(oc[:, 5].sum()) == 0
    assert caps["max_contacts"] > 255                      # beyond round 3's cap ...
    assert caps["spill_steps"] > 0 and caps["scenes_spilled"] > 0      # ... and beyond the LDS-resident part: the case is exercised
    assert caps["reduced_steps"] == 0
    # the same pile with the optional compound manifold reduction (pair_contact_budget = 32): bunny on bunny offers hundreds of one-point
    # manifolds per body pair, the deepest 64 go to the solver -- counted, and identical on both sides
    prm["pair_contact_budget"] = SB.PAIR_CONTACT_BUDGET_FAST
    gpu, caps = se.run_with_caps(srec, bodies.copy(), prm)
    ref = bodies.copy()
    oc = _oracle_caps(oracle, srec, ref, hulls, verts, SB.sizing_hints(prm, srec, bodies, hulls))
    assert_bodies_equal(gpu, ref)
    assert caps["reduced_steps"] == int(oc[:, 4].sum()) > 0
    assert caps["scenes_dropped"] == 0


def test_undersized_capacities_are_counted(sl, oracle):
    """A caller that sizes the lists too small loses contacts / hull pairs in list order -- counted, identically on both
    sides, never silently: the next call with larger capacities gives the uncapped result."""
    from stillleben_amd import physics

    scs = _bunny_pile_batch(sl, 2)
    se = physics.settle_engine()
    srec, bodies = SB.build_settle_batch(scs, se.pool, [(True, TABLE)] * len(scs))
    prm = SB.default_params(frames=30)
    prm["pair_contact_budget"] = 0
    prm["max_hull_pairs_per_scene"], prm["max_contacts_per_scene"] = 300, 150
    gpu, caps = se.run_with_caps(srec, bodies.copy(), prm)
    hulls, verts = se.pool.arrays()
    ref = bodies.copy()
    oc = _oracle_caps(oracle, srec, ref, hulls, verts, SB.sizing_hints(prm, srec, bodies, hulls))
    assert_bodies_equal(gpu, ref)
    assert caps["contact_drop_steps"] == int(oc[:, 0].sum()) > 0
    assert caps["pair_drop_steps"] == int(oc[:, 1].sum()) > 0
    assert caps["scenes_dropped"] == int(((oc[:, 0] + oc[:, 1]) > 0).sum())
    # the host path grows the capacities by itself: the result is the one of lists that never ran out
    prm0 = SB.default_params(frames=30)
    prm0["pair_contact_budget"] = 0
    prm0["max_hull_pairs_per_scene"], prm0["max_contacts_per_scene"] = 300, 150
    grown = se.run(srec, bodies.copy(), prm0)
    big = SB.default_params(frames=30)
    big["pair_contact_budget"] = 0
    big["max_hull_pairs_per_scene"], big["max_contacts_per_scene"] = 32768, 8192
    ref2 = bodies.copy()
    oracle.settle(srec, ref2, hulls, verts, big)
    assert_bodies_equal(grown, ref2)
    assert se.caps(len(srec))["scenes_dropped"] == 0


def test_body_pair_list_is_a_capacity_the_host_grows(sl, oracle):
    """The list of touching body pairs (one solver group each) is sized by the caller (max_body_pairs_per_scene): a pair beyond it
    is dropped AND counted in a word of its own -- identically on both sides -- and the host path settles again with a list
    that holds what the scene offers: the result is the one of a list that never ran out."""
    from stillleben_amd import physics

    cube = scaled(sl, S.CUBE, 0.15)
    scs = [heap(sl, 1300 + i, 9, cube) for i in range(4)]
    se = physics.settle_engine()
    srec, bodies = SB.build_settle_batch(scs, se.pool, [(True, TABLE)] * len(scs))
    prm = SB.default_params(frames=25)
    prm["max_body_pairs_per_scene"] = 3
    gpu, caps = se.run_with_caps(srec, bodies.copy(), prm)
    hulls, verts = se.pool.arrays()
    ref = bodies.copy()
    oc = _oracle_caps(oracle, srec, ref, hulls, verts, SB.sizing_hints(prm, srec, bodies, hulls))
    assert_bodies_equal(gpu, ref)
    assert caps["group_drop_steps"] == int(oc[:, 5].sum()) > 0
    assert caps["pair_drop_steps"] == int(oc[:, 1].sum()) == 0           # ... and the hull-pair statistics stay what they are
    assert caps["max_hull_pairs"] == int(oc[:, 3].max())
    assert caps["scenes_dropped"] == int((oc[:, 5] > 0).sum()) > 0
    small = SB.default_params(frames=25)
    small["max_body_pairs_per_scene"] = 3
    grown = se.run(srec, bodies.copy(), small)
    ref2 = bodies.copy()
    oracle.settle(srec, ref2, hulls, verts, SB.default_params(frames=25))
    assert_bodies_equal(grown, ref2)
    assert se.caps(len(srec))["scenes_dropped"] == 0 and int(np.asarray(se.last_params["max_body_pairs_per_scene"]).reshape(-1)[0]) >= 12


# ---- the contact state outlives the call (slhip_settle_params.resume; PhysX: one PxScene per sl.Scene) ----------------------
def _object_state(scene):
    return np.stack([np.concatenate([o._pose.reshape(-1), o._linear_velocity, o._angular_velocity, [o._separation]]) for o in scene._objects]).astype(np.float32)


def test_resume_equals_one_call(sl, oracle):
    """k calls of one frame on the same scratch == one call of k frames, bit for bit, on a batch (C-ABI level), and equal to the
    oracle's; cold starts per call give something else."""
    from stillleben_amd import physics

    cube = scaled(sl, S.CUBE, 0.15)
    scs = [heap(sl, 900 + i, 7, cube) for i in range(8)]
    se = physics.settle_engine()
    srec, bodies = SB.build_settle_batch(scs, se.pool, [(True, TABLE)] * len(scs))
    frames = 25
    one = se.run(srec, bodies.copy(), SB.default_params(frames=frames))
    d_b = se.eng.upload_records(bodies.copy())
    cold = bodies.copy()
    for f in range(frames):
        prm = SB.default_params(frames=1)
        prm["resume"] = 4 * f
        prm = SB.sizing_hints(prm, srec, bodies, se.pool.arrays()[0])
        d_b = se.run_device(srec, None, prm, d_bodies=d_b)
    many = np.frombuffer(d_b.cpu().numpy().tobytes(), dtype=SB.BODY_DTYPE).copy()
    assert_bodies_equal(many, one)
    hulls, verts = se.pool.arrays()
    ref = bodies.copy()
    oracle.settle(srec, ref, hulls, verts, SB.default_params(frames=frames))
    assert_bodies_equal(many, ref)
    for f in range(frames):
        cold = se.run(srec, cold, SB.default_params(frames=1))
    assert cold["pose"].tobytes() != one["pose"].tobytes()


def test_vis_cb_settle_equals_plain_settle(sl, oracle):
    """simulate_tabletop_scene(vis_cb=f) steps the same long-lived scene as simulate_tabletop_scene() (scene.cpp:720-739):
    same bodies, bit for bit."""
    cube = scaled(sl, S.CUBE, 0.2)
    bunny = scaled(sl, S.BUNNY, 0.25)
    out = []
    for cb in (None, lambda i: None):
        scene = sl.Scene((320, 240), seed=21)
        for i in range(6):
            scene.add_object(sl.Object(bunny if i == 2 else cube))
        scene.simulate_tabletop_scene(vis_cb=cb)
        out.append(_object_state(scene))
        assert scene._phys_state is None
    assert out[0].tobytes() == out[1].tobytes()
    # ... and what the oracle's single call gives
    from stillleben_amd import physics

    scene = sl.Scene((320, 240), seed=21)
    for i in range(6):
        scene.add_object(sl.Object(bunny if i == 2 else cube))
    hp = physics.prepare_tabletop(scene)
    se = physics.settle_engine()
    srec, bodies = SB.build_settle_batch([scene], se.pool, [(hp, physics.PLANE_HALF_Z)])
    hulls, verts = se.pool.arrays()
    oracle.settle(srec, bodies, hulls, verts, SB.default_params(tabletop=True))
    assert np.array_equal(bodies["pose"].reshape(-1, 16), out[0][:, :16])


def _column(sl, n):
    """n cubes in a column on a static slab (Scene::simulate has no table, scene.cpp:903-912)."""
    cube = scaled(sl, S.CUBE, 0.2)
    slab = scaled(sl, S.CUBE, 1.5)
    h = 0.2 / np.sqrt(3.0) / 2.0
    H = 1.5 / np.sqrt(3.0) / 2.0
    scene = sl.Scene((320, 240))
    base = sl.Object(slab)
    base.static = True
    p = torch.eye(4)
    p[2, 3] = -H
    base.set_pose(p)
    scene.add_object(base)
    zs = [h + 0.003 + k * (2 * h + 0.003) for k in range(n)]
    for z in zs:
        o = sl.Object(cube)
        p = torch.eye(4)
        p[2, 3] = float(z)
        o.set_pose(p)
        scene.add_object(o)
    return scene, zs


def test_column_stands_through_scene_simulate(sl, oracle):
    """A column of 6 cubes stepped by 400 Scene.simulate(0.01) calls stands (a cold start per call topples it: round 3), and is
    bit for bit what ONE 400-step call gives."""
    from stillleben_amd import physics

    scene, zs = _column(sl, 6)
    se = physics.settle_engine()
    scene.load_physics()
    srec, bodies = SB.build_settle_batch([scene], se.pool, [(False, 0.0)])
    for _ in range(400):
        scene.simulate(0.01)
    assert scene._phys_state.steps == 400
    z = np.array([float(o.pose()[2, 3]) for o in scene.objects[1:]])
    assert np.allclose(z, zs, atol=4e-3), z
    for o in scene.objects[1:]:
        R = o.pose().numpy()[:3, :3]
        assert np.degrees(np.arccos(min(1.0, float(R[2, 2])))) < 1.0
    prm = SB.default_params(tabletop=False, dt=0.01, frames=400, substeps=1)
    hulls, verts = se.pool.arrays()
    ref = bodies.copy()
    oracle.settle(srec, ref, hulls, verts, prm)
    assert np.array_equal(ref["pose"].reshape(-1, 4, 4), np.stack([o._pose for o in scene._objects]))
    assert np.array_equal(ref["flags"], scene._phys_state.bodies["flags"])
    # a pose set from outside is a teleport: the next call starts cold
    scene.objects[3].set_pose(scene.objects[3].pose())
    scene.simulate(0.01)
    assert scene._phys_state.steps == 1


def test_scene_simulate_with_150_bodies(sl):
    """A long-lived scene of 150 bodies (round-5 advisor): Scene.simulate sizes the scene's own scratch -- every body pair would not
    fit the kernels' LDS beyond ~100 bodies, so the list of touching body pairs is capped at 12 per body there -- and steps it; the
    grid of cubes on a slab stays where it is, resumed calls continue the state."""
    cube = scaled(sl, S.CUBE, 0.05)
    slab = scaled(sl, S.CUBE, 3.0)
    h = 0.05 / np.sqrt(3.0) / 2.0
    H = 3.0 / np.sqrt(3.0) / 2.0
    scene = sl.Scene((160, 120))
    base = sl.Object(slab)
    base.static = True
    p = torch.eye(4)
    p[2, 3] = -H
    base.set_pose(p)
    scene.add_object(base)
    for k in range(150):
        o = sl.Object(cube)
        p = torch.eye(4)
        p[0, 3], p[1, 3], p[2, 3] = (k % 15 - 7) * 0.04, (k // 15 - 5) * 0.04, h + 0.003
        o.set_pose(p)
        scene.add_object(o)
    for _ in range(20):
        scene.simulate(0.01)
    st = scene._phys_state
    assert st.steps == 20 and len(st.bodies) == 151
    z = np.array([float(o.pose()[2, 3]) for o in scene.objects[1:]])
    assert np.allclose(z, h + 0.003, atol=2e-3), (z.min(), z.max())
    assert float(st.bodies["stab"][1:, 1].min()) > 0.0            # every cube rests on the static slab: lightened by the stabilisation


def test_manipulation_steps_equal_one_call(sl, oracle):
    """50 ManipulationSim.step calls == one 50-frame slhip_settle (manipulation_sim.cpp:83-93 steps one PxScene), bit for bit --
    and a plain object sharing the manipulator's mesh does not inherit its spring drive (round-3 advisor)."""
    from stillleben_amd import physics

    cube = scaled(sl, S.CUBE, 0.1)

    def make():
        scene = sl.Scene((320, 240))
        tool = sl.Object(cube)
        sim = sl.ManipulationSim(scene, tool, torch.eye(4))     # the manipulator is the FIRST object seen for its template key
        for k in range(3):
            o = sl.Object(cube)
            p = torch.eye(4)
            p[0, 3] = 0.105 * (k + 1)
            o.set_pose(p)
            scene.add_object(o)
        return scene, tool, sim

    se = physics.settle_engine()
    se.pool.__dict__.pop("_body_templates", None)
    scene, tool, sim = make()
    goal = torch.eye(4)
    goal[0, 3] = 0.2
    tool._drive["target"] = goal[:3, 3].numpy().copy()
    srec, bodies = SB.build_settle_batch([scene], se.pool, [(False, 0.0)])
    assert (bodies["drive_flags"] != 0).sum() == 1 and bodies["drive_flags"][0] == 15
    prm = SB.default_params(tabletop=False, dt=0.005, frames=50, substeps=1)
    one = se.run(srec, bodies.copy(), prm)
    hulls, verts = se.pool.arrays()
    ref = bodies.copy()
    oracle.settle(srec, ref, hulls, verts, prm)
    assert_bodies_equal(one, ref)
    for _ in range(50):
        sim.step(goal, 0.005)
    assert scene._phys_state.steps == 50
    assert np.array_equal(one["pose"].reshape(-1, 4, 4), np.stack([o._pose for o in scene._objects]))
    assert np.array_equal(one["lin_vel"][:, :3], np.stack([o._linear_velocity for o in scene._objects]))
    assert float(scene.objects[1].pose()[0, 3]) > 0.105      # the tool pushed its neighbour
