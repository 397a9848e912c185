"""CPU tests of host-side semantics the reference pins in C++: vertex updates ADD and recompute normals
(reference src/mesh.cpp:763-886, python/src/py_mesh.cpp:69-160), Mesh.load_threaded (mesh.cpp:923-999)
and the JobQueue's submission-order contract (src/job_queue.cpp:55-82)."""
import numpy as np
import pytest
import torch

import scenes as S


def test_update_positions_adds_accumulates_and_recomputes_normals(sl):
    m = sl.Mesh(S.CUBE, physics=False)
    before = m.points.clone()
    ids = torch.tensor([1, 1, 5], dtype=torch.int32)           # 1-based, id 1 twice: both updates count
    upd = torch.tensor([[0.1, 0.0, 0.0], [0.2, 0.0, 0.0], [0.0, 0.0, -0.5]])
    m.update_positions(ids, upd)
    after = m.points
    assert torch.allclose(after[0], before[0] + torch.tensor([0.3, 0.0, 0.0]), atol=1e-7)
    assert torch.allclose(after[4], before[4] + torch.tensor([0.0, 0.0, -0.5]), atol=1e-7)
    untouched = [i for i in range(len(before)) if i not in (0, 4)]
    assert torch.equal(after[untouched], before[untouched])
    # normals: unit length, equal to the area-weighted face-normal sum of mesh.cpp:763-815
    p = after.numpy()
    idx = m.faces.numpy().reshape(-1, 3)
    acc = np.zeros_like(p)
    for f in idx:
        cr = np.cross(p[f[0]] - p[f[1]], p[f[0]] - p[f[2]])
        for v in f:
            acc[v] += cr
    ref = acc / np.linalg.norm(acc, axis=1, keepdims=True)
    assert np.allclose(m.normals.numpy(), ref, atol=1e-5)
    assert np.allclose(np.linalg.norm(m.normals.numpy(), axis=1), 1.0, atol=1e-5)


def test_update_colors_adds_and_set_new_positions_recomputes(sl):
    m = sl.Mesh(S.CUBE, physics=False)
    c0 = m.colors.clone()
    m.update_colors(torch.tensor([2, 2], dtype=torch.int32), torch.tensor([[-0.25, 0, 0, 0], [-0.25, 0, 0, 0]], dtype=torch.float32))
    assert torch.allclose(m.colors[1], c0[1] + torch.tensor([-0.5, 0, 0, 0]))
    n0 = m.normals.clone()
    m.set_new_positions(m.points * torch.tensor([1.0, 1.0, 3.0]))      # stretched box: smooth normals change
    assert m.normals.shape == n0.shape and np.allclose(np.linalg.norm(m.normals.numpy(), axis=1), 1.0, atol=1e-5)
    with pytest.raises(ValueError):
        m.set_new_positions(torch.zeros(3, 3))
    with pytest.raises(ValueError):
        m.set_new_colors(torch.zeros(3, 4))


def test_update_argument_checks_follow_py_mesh(sl):
    m = sl.Mesh(S.CUBE, physics=False)
    ok_i, ok_p = torch.tensor([1, 2], dtype=torch.int32), torch.zeros(2, 3)
    with pytest.raises(ValueError, match="one dimensional"):
        m.update_positions(ok_i.reshape(1, 2), ok_p)
    with pytest.raises(ValueError, match="two dimensional"):
        m.update_positions(ok_i, torch.zeros(6))
    with pytest.raises(ValueError, match="same size"):
        m.update_positions(ok_i, torch.zeros(3, 3))
    with pytest.raises(ValueError, match=r"\(N,3\)"):
        m.update_positions(ok_i, torch.zeros(2, 4))
    with pytest.raises(ValueError, match=r"\(N,4\)"):
        m.update_colors(ok_i, torch.zeros(2, 3))
    with pytest.raises(ValueError):
        m.update_positions(torch.tensor([0, 99], dtype=torch.int32), ok_p)


def test_load_threaded_keeps_order_and_reports_failures(sl, tmp_path):
    paths = [S.CUBE, S.BUNNY, S.CUBE]
    meshes = sl.Mesh.load_threaded(paths, visual=True, physics=False)
    assert [m.filename for m in meshes] == paths
    assert meshes[0].points.shape[0] == 24 and meshes[1].points.shape[0] == 41210
    flags = [sl.Mesh.Flag.NONE, sl.Mesh.Flag.PHYSICS_FORCE_CONVEX_HULL, sl.Mesh.Flag.NONE]
    assert len(sl.Mesh.load_threaded(paths, True, False, flags)) == 3
    with pytest.raises(ValueError):
        sl.Mesh.load_threaded(paths, True, False, flags[:2])
    with pytest.raises(RuntimeError, match="Could not load one of the meshes"):
        sl.Mesh.load_threaded([S.CUBE, str(tmp_path / "missing.obj")], True, False)


def test_job_queue_returns_scenes_in_submission_order(sl, monkeypatch):
    """job_queue.cpp:55-82 with the settle itself stubbed (the GPU version of this test is in
    tests/test_gpu_acceptance.py): add_scene loads physics, retrieve_scene hands scenes back first-in
    first-out regardless of which finishes first, and an empty queue raises the reference's message."""
    from stillleben_amd import physics

    launches = []

    def fake_settle(batch, frames=None):
        launches.append(list(batch))
        for s in reversed(batch):
            s._settled = True

    monkeypatch.setattr(physics, "settle_batch", fake_settle)
    q = sl.JobQueue(num_threads=3)
    assert q.num_threads == 3 and sl.JobQueue().num_threads >= 1
    with pytest.raises(RuntimeError, match="No scenes in work queue"):
        q.retrieve_scene()
    cube = sl.Mesh(S.CUBE, physics=True)
    scenes = []
    for i in range(5):
        s = sl.Scene((64, 48), seed=i)
        s.add_object(sl.Object(cube))
        scenes.append(s)
    for s in scenes[:3]:
        q.add_scene(s)
    assert all(s._physics_loaded for s in scenes[:3])
    assert q.retrieve_scene() is scenes[0]
    q.add_scene(scenes[3])
    q.add_scene(scenes[4])
    assert [q.retrieve_scene() for _ in range(4)] == scenes[1:]
    assert launches == [scenes[:3], scenes[3:]]          # queued scenes settle together in one launch
    with pytest.raises(RuntimeError):
        q.retrieve_scene()
    q.stop()


def test_hull_fixture_is_rejected_after_a_geometry_edit(sl, tmp_path):
    import shutil

    from stillleben_amd import hulls

    dst = tmp_path / "cube.glb"
    shutil.copy(S.CUBE, dst)
    shutil.copy(S.CUBE + ".hulls.npz", str(dst) + ".hulls.npz")
    m = sl.Mesh(str(dst), physics=False)
    good = hulls.hulls_for_mesh(m, use_cache=False)
    assert len(good) == 1 and np.abs(good[0].vertices).max() == pytest.approx(1.0)
    m._data.positions = (m._data.positions * 2.0).astype(np.float32)       # the asset changed, the fixture did not
    after = hulls.hulls_for_mesh(m, use_cache=False)
    assert np.abs(after[0].vertices).max() == pytest.approx(2.0)           # recomputed, not the stale fixture


def test_synthetic_set_ships_the_vhacd_hulls(sl):
    from stillleben_amd import synthetic

    meshes = synthetic.ycb_like_meshes(seed=0, tex_size=64)
    n = {m.filename.split("/")[-1]: len(m._hulls) for m in meshes}
    assert n["011_banana"] == 44 and n["024_bowl"] == 28 and n["025_mug"] == 28 and n["003_cracker_box"] == 1
    assert all(len(h.vertices) <= 64 for m in meshes for h in m._hulls)
    parts = synthetic.ycb_like_meshes(seed=0, tex_size=64, hulls="parts")
    assert len(parts[9]._hulls) != 44
