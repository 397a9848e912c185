"""CPU tests pinning oracle/synth_ref.c (the checker of slhip_synth_stage / slhip_synth_place):
(i) against the per-scene Python mirror of the same reference functions -- Scene::simulateTableTopScene's
set-up (reference src/scene.cpp:612-678), chooseRandomCameraPose (:472-610), chooseRandomLightDirection
(:453-470), the shadow matrices (render_pass.cpp:69-211) and the per-drawable uniforms
(render_pass.cpp:534-621) -- fed with the oracle's own random draws; (ii) analytic properties the
reference's algorithm guarantees; (iii) accuracy of the deterministic log / sin / cos."""
import math

import numpy as np
import pytest
import torch

import scenes as S
from stillleben_amd import _abi, _math as M
from stillleben_amd import _settle_batch as SB


def make_table(sl, n_cubes=3, bunny=True):
    from stillleben_amd._batch import HostPool
    from stillleben_amd.scene_batch import AssetTable

    meshes = []
    for i in range(n_cubes):
        m = sl.Mesh(S.CUBE)
        m.center_bbox()
        m.scale_to_bbox_diagonal(0.12 + 0.05 * i)
        m.class_index = i + 1
        meshes.append(m)
    if bunny:
        b = sl.Mesh(S.BUNNY)
        b.center_bbox()
        b.scale_to_bbox_diagonal(0.25)
        b.class_index = 9
        meshes.append(b)
    pool, hulls = HostPool(), SB.HullPool()
    return AssetTable(meshes, mesh_pool=pool, hull_pool=hulls), pool, hulls


def make_params(table, n_scenes, n_objects, distinct, seed=5, render_chunk=0, flags_extra=_abi.SYNTH_RANDOM_PBR | _abi.SYNTH_SHADOWS):
    from stillleben_amd.scene import Scene

    sc = Scene((640, 480))
    sc.set_camera_intrinsics(1066.778, 1067.487, 312.9869, 241.3109)
    p = np.zeros((), dtype=_abi.SYNTH_PARAMS_DTYPE)
    p["n_scenes"], p["n_objects"], p["n_assets"] = n_scenes, n_objects, len(table)
    p["flags"] = (_abi.SYNTH_SAMPLE_DISTINCT if distinct else 0) | flags_extra
    p["seed_lo"], p["seed_hi"], p["scene_id_base"] = seed, 77, 1000
    p["render_chunk"] = render_chunk or n_scenes
    p["max_draws_per_scene"] = table.bound(table.n_draws, n_objects, distinct) + 1
    p["max_chunks_per_scene"] = table.bound(table.n_chunks, n_objects, distinct) + 1
    p["max_clip_verts_per_scene"] = table.bound(table.n_clip, n_objects, distinct) + 4
    p["plane_z"] = 0.04
    p["proj"] = sc._projection.reshape(-1)
    p["proj_inv"] = np.linalg.inv(sc._projection.astype(np.float64)).astype(np.float32).reshape(-1)
    p["plane_size"] = (3.0, 3.0)
    p["manual_exposure"] = 1.0
    p["light_color"][:3] = 300.0
    p["ambient"][:3] = 0.05
    return p, sc


def test_deterministic_transcendentals(oracle):
    rng = np.random.default_rng(0)
    xs = np.concatenate([rng.uniform(-2 * math.pi, 2 * math.pi, 4000), [0.0, math.pi, -math.pi, math.pi / 2, -math.pi / 2, 2 * math.pi]])
    err = 0.0
    for x in xs.astype(np.float32):
        s, c = oracle.det_sincosf(x)
        err = max(err, abs(s - math.sin(float(x))), abs(c - math.cos(float(x))))
    assert err < 2.5e-7
    us = np.concatenate([rng.uniform(0, 1, 4000), [2.0 ** -25, 1 - 2.0 ** -25, 0.5, 0.70710678]]).astype(np.float32)
    for u in us[us > 0]:
        assert abs(oracle.det_logf(u) - math.log(float(u))) <= 3e-7 * max(1.0, abs(math.log(float(u))))


def test_random_draws_have_the_reference_distributions(sl, oracle):
    table, _, _ = make_table(sl, bunny=False)
    p, _ = make_params(table, 1, 3, True)
    p["n_objects"] = 60   # many draws per scene
    yaw, az, el, nrm, quat = [], [], [], [], []
    for s in range(300):
        d = oracle.synth_draws(p, s)
        yaw.append(d["yaw"]); az.append(d["azimuth"]); el.append(d["elevation"]); nrm.append(d["light_normals"]); quat.append(d["quat"])
    yaw, az, el = np.array(yaw), np.array(az), np.array(el)
    assert -math.pi <= yaw.min() and yaw.max() <= math.pi and abs(yaw.mean()) < 0.35 and yaw.std() == pytest.approx(2 * math.pi / math.sqrt(12), rel=0.1)
    assert az.std() == pytest.approx(2 * math.pi / math.sqrt(12), rel=0.1) and abs(np.corrcoef(yaw, az)[0, 1]) < 0.2
    assert math.radians(30) <= el.min() and el.max() <= math.radians(60) and el.mean() == pytest.approx(math.radians(45), abs=0.02)
    q = np.concatenate(quat).reshape(-1)            # 300 * 60 * 4 N(0,1) draws
    assert abs(q.mean()) < 0.02 and q.std() == pytest.approx(1.0, abs=0.02)
    assert np.mean(np.abs(q) > 1.959964) == pytest.approx(0.05, abs=0.006)    # tails
    assert abs(np.mean(q ** 3)) < 0.05 and np.mean(q ** 4) == pytest.approx(3.0, abs=0.15)
    n = np.array(nrm)
    assert abs(n.mean()) < 0.1 and n.std() == pytest.approx(1.0, abs=0.08)


@pytest.mark.parametrize("distinct", [True, False])
def test_stage_matches_the_per_scene_mirror(sl, oracle, distinct):
    """Bodies written by the oracle == physics.prepare_tabletop + body_record of an sl.Scene whose random
    draws are the oracle's (yaw, quaternions)."""
    from stillleben_amd import physics

    table, pool, hulls = make_table(sl)
    n_scenes, n_obj = 6, 4 if distinct else 7
    p, _ = make_params(table, n_scenes, n_obj, distinct)
    ids = None
    if not distinct:
        ids = np.random.default_rng(3).integers(0, len(table), (n_scenes, n_obj)).astype(np.uint16)
    bodies, ss, objs, scs = oracle.synth_stage(p, table.records, ids)
    assert np.array_equal(ss["body_begin"], np.arange(n_scenes) * n_obj) and (ss["has_plane"] == 1).all()
    for s in range(n_scenes):
        d = oracle.synth_draws(p, s)
        chosen = objs["asset"][s * n_obj:(s + 1) * n_obj]
        if distinct:
            assert len(set(chosen.tolist())) == n_obj          # without replacement
        else:
            assert np.array_equal(chosen, ids[s])
        scene = sl.Scene((640, 480))
        for a in chosen:
            scene.add_object(sl.Object(table.meshes[int(a)]))

        class FakeRng:          # feeds physics.prepare_tabletop the oracle's draws, in its call order
            def __init__(self):
                self.q = iter(d["quat"])

            def uniform(self, lo, hi):
                return float(d["yaw"])

            def standard_normal(self, n):
                return next(self.q).astype(np.float64)

        scene._rng = FakeRng()
        assert physics.prepare_tabletop(scene) is True
        srec, ref = SB.build_settle_batch([scene], hulls, [(True, physics.PLANE_HALF_Z)])
        got = bodies[s * n_obj:(s + 1) * n_obj]
        assert np.allclose(got["pose"], ref["pose"], atol=2e-6)
        for f in ("com", "inv_inertia", "bsphere", "bbox_center", "mu_s", "mu_d", "restitution", "wake_counter", "hull_begin", "hull_end", "flags", "lin_vel", "ang_vel", "stuck_counter"):
            assert np.array_equal(got[f], ref[f]), f
        assert np.allclose(got["inv_mass"], ref["inv_mass"], rtol=1e-6) and np.isinf(got["separation"]).all()
        assert np.allclose(scs[s]["plane_pose"].reshape(4, 4), scene._background_plane_pose, atol=1e-6)
        assert np.array_equal(objs["instance_index"][s * n_obj:(s + 1) * n_obj], np.arange(1, n_obj + 1))
        pbr = np.stack([objs["metallic"], objs["roughness"]], axis=1)[s * n_obj:(s + 1) * n_obj]
        assert np.array_equal(pbr, d["pbr"]) and (pbr > 0).all() and (pbr < 1).all()
        # the stack: every object above the previous one, rotations orthonormal
        R = got["pose"].reshape(-1, 4, 4)[:, :3, :3].astype(np.float64)
        assert np.allclose(R @ np.transpose(R, (0, 2, 1)), np.eye(3), atol=1e-5)


def test_place_matches_the_per_scene_mirror_and_keeps_objects_in_view(sl, oracle):
    """Camera pose, light direction, shadow matrix and draw records == the per-scene Python path on an sl.Scene
    holding the same poses; every bbox corner projects inside the image; light from above / camera side."""
    from stillleben_amd import camera_placement
    from stillleben_amd._batch import build_batch

    table, pool, hulls = make_table(sl)
    n_scenes, n_obj = 8, 4
    p, proto = make_params(table, n_scenes, n_obj, True, render_chunk=4)
    bodies, ss, objs, scs = oracle.synth_stage(p, table.records)
    # perturb the poses as a settle would (the place step must use the CURRENT poses)
    rng = np.random.default_rng(1)
    for b in bodies:
        pose = b["pose"].reshape(4, 4).copy()
        pose[:3, 3] = [rng.uniform(-0.3, 0.3), rng.uniform(-0.3, 0.3), rng.uniform(0.05, 0.2)]
        b["pose"] = pose.reshape(-1)
    srec, drec, crec = oracle.synth_place(p, table.records, table.templates, bodies, objs, scs)
    md, mk, mv = int(p["max_draws_per_scene"]), int(p["max_chunks_per_scene"]), int(p["max_clip_verts_per_scene"])
    for s in range(n_scenes):
        d = oracle.synth_draws(p, s)
        local = s % 4
        scene = sl.Scene((640, 480))
        scene._projection = proto._projection.copy()
        for o in range(n_obj):
            k = s * n_obj + o
            obj = sl.Object(table.meshes[int(objs[k]["asset"])])
            obj.metallic, obj.roughness = float(objs[k]["metallic"]), float(objs[k]["roughness"])
            scene.add_object(obj)
            obj._pose = bodies[k]["pose"].reshape(4, 4).copy()
        scene._background_plane_pose = scs[s]["plane_pose"].reshape(4, 4).copy()
        scene.background_plane_size = torch.tensor([3.0, 3.0])
        scene.ambient_light = torch.tensor([0.05, 0.05, 0.05])
        scene.manual_exposure = 1.0
        scene._camera_pose = camera_placement.choose_camera_pose(scene, np.float32(d["azimuth"]), np.float32(d["elevation"]))
        cam = scs[s]["camera_pose"].reshape(4, 4)
        assert np.allclose(cam, scene._camera_pose, atol=5e-5)

        class FakeRng:
            def __init__(self):
                self.n = iter(d["light_normals"])

            def standard_normal(self):
                return float(next(self.n))

        scene._rng = FakeRng()
        scene.choose_random_light_direction()
        rs, rd, rc = build_batch([scene], pool, with_shadows=True)
        got = srec[s]
        assert np.allclose(got["light_dir"][0][:3], rs[0]["light_dir"][0][:3], atol=2e-6)
        for f in ("proj", "light_color", "ambient", "manual_exposure", "light_map", "bg_tex", "n_prims"):
            assert np.array_equal(got[f], rs[0][f]), f
        assert np.allclose(got["world_to_cam"], rs[0]["world_to_cam"], atol=5e-5)
        assert np.allclose(got["cam_position"], rs[0]["cam_position"], atol=5e-5)
        ref_sm, got_sm = rs[0]["shadow_mat"][0].reshape(4, 4), got["shadow_mat"][0].reshape(4, 4)
        assert np.allclose(got_sm, ref_sm, rtol=2e-3, atol=2e-3)
        assert np.array_equal(got["shadow_mat"][1], np.eye(4, dtype=np.float32).reshape(-1))
        # draw records: same list, chunk-relative indices
        nd = int(got["draw_end"] - got["draw_begin"])
        assert got["draw_begin"] == local * md and nd == len(rd)
        gd = drec[s * md:s * md + nd]
        for f in ("mesh_to_object", "object_to_world", "base_color", "emissive", "alpha_cutoff", "class_index", "instance_index",
                  "flags", "n_verts", "vtx_base", "idx_base", "n_tris", "prim_base", "tex_offset", "tex_w", "tex_h", "tex_sampler"):
            assert np.array_equal(gd[f], rd[f]), f
        assert np.allclose(gd["metallic"], rd["metallic"]) and np.allclose(gd["roughness"], rd["roughness"])
        assert np.allclose(gd["normal_to_world"], rd["normal_to_world"], rtol=1e-5, atol=1e-6)
        assert (gd["scene"] == local).all()
        assert np.array_equal(gd["clip_base"], local * mv + rd["clip_base"])
        assert (drec[s * md + nd:(s + 1) * md]["n_tris"] == 0).all()
        gc = crec[s * mk:(s + 1) * mk]
        used = gc[gc["count"] > 0]
        assert len(used) == len(rc) and np.array_equal(used["draw"], local * md + rc["draw"])
        assert np.array_equal(used["first_tri"], rc["first_tri"]) and np.array_equal(used["count"], rc["count"])
        assert (gc["scene"] == local).all()
        # analytic: all bbox corners inside the image, elevation in [30, 60] degrees, light from above and from the
        # camera side (scene.cpp:453-470, :472-610)
        P, w2c = proto._projection, M.inverted_rigid(cam)
        for o in scene._objects:
            c = o._mesh.bbox.corners()
            pc = (w2c @ o._pose @ np.concatenate([c, np.ones((8, 1), np.float32)], axis=1).T).T
            clip = (P @ pc.T).T
            ndc = clip[:, :2] / clip[:, 3:4]
            assert (np.abs(ndc) <= 1.0 + 2e-4).all() and (pc[:, 2] > 0).all()
        view = cam[:3, 2]                              # camera looks along +z of its frame
        assert math.radians(29.9) <= math.asin(-view[2]) <= math.radians(60.1)
        l_cam = cam[:3, :3].T @ got["light_dir"][0][:3]
        assert l_cam[1] >= 0 and l_cam[2] >= 0         # direction of travel: downwards (+y is down) and away from the camera


def test_stride_overflow_is_reported(sl, oracle):
    table, _, _ = make_table(sl)
    p, _ = make_params(table, 2, 4, True)
    bodies, ss, objs, scs = oracle.synth_stage(p, table.records)
    p["max_chunks_per_scene"] = 2
    with pytest.raises(RuntimeError):
        oracle.synth_place(p, table.records, table.templates, bodies, objs, scs)
