"""GPU parity tests of the device-side scene synthesis (slhip_synth_stage / slhip_synth_place through the
C-ABI) against oracle/synth_ref.c -- bar: BIT-EXACT records -- and of the whole batched path
stage -> settle -> place -> render -> gather on BASELINE config C3's per-GPU shard (64 scenes of 20 YCB-like
objects, 640x480, 6-channel ground truth), with oracle parity of settle and render on sampled scenes."""
import numpy as np
import pytest
import torch

import scenes as S
from stillleben_amd import _abi
from stillleben_amd import _settle_batch as SB

pytestmark = pytest.mark.gpu


def small_table(sl):
    meshes = []
    for i in range(3):
        m = sl.Mesh(S.CUBE)
        m.center_bbox()
        m.scale_to_bbox_diagonal(0.12 + 0.05 * i)
        m.class_index = i + 1
        meshes.append(m)
    b = sl.Mesh(S.BUNNY)
    b.center_bbox()
    b.scale_to_bbox_diagonal(0.25)
    b.class_index = 9
    meshes.append(b)
    return sl.AssetTable(meshes)


@pytest.fixture(scope="module")
def ycb_table(sl):
    from stillleben_amd import synthetic

    return sl.AssetTable(synthetic.ycb_like_meshes(seed=0, tex_size=256))


def assert_records_equal(name, got, ref):
    for f in got.dtype.names:
        a, b = np.ascontiguousarray(got[f]), np.ascontiguousarray(ref[f])
        if not np.array_equal(a.view(np.uint8), b.view(np.uint8)):
            bad = np.argwhere(a != b)
            raise AssertionError("%s.%s differs at %d entries, first %s: %r vs %r" % (name, f, len(bad), bad[0], a[tuple(bad[0])], b[tuple(bad[0])]))


@pytest.mark.parametrize("distinct", [True, False])
def test_stage_and_place_records_bit_exact(sl, oracle, distinct):
    table = small_table(sl)
    n_scenes, n_obj = 37, 4 if distinct else 9
    ids = None if distinct else np.random.default_rng(2).integers(0, len(table), (n_scenes, n_obj))
    batch = sl.SceneBatch(table, n_scenes, n_obj, seed=(77 << 32) | 5, asset_ids=ids, render_chunk=16, manual_exposure=1.0,
                          scene_id_base=1000)
    batch.set_camera_intrinsics(1066.778, 1067.487, 312.9869, 241.3109)
    batch.stage()
    torch.cuda.synchronize()
    rb, rss, robj, rsc = oracle.synth_stage(batch.params, table.records, batch.asset_ids)
    assert_records_equal("body", batch.host_bodies(), rb)
    assert_records_equal("settle_scene", batch.host_settle_scenes(), rss)
    assert_records_equal("object", batch.host_objects(), robj)
    assert_records_equal("scene", batch.host_scenes(), rsc)
    # a short settle moves everything; the place step must follow the CURRENT poses
    batch.settle(frames=5)
    batch.check_settled()
    batch.place()
    torch.cuda.synchronize()
    bodies = batch.host_bodies()
    assert not np.array_equal(bodies["pose"], rb["pose"])
    srec, drec, crec = oracle.synth_place(batch.params, table.records, table.templates, bodies, robj, rsc)
    g_s, g_d, g_c = batch.host_render_records()
    assert_records_equal("slhip_scene", g_s, srec)
    assert_records_equal("slhip_draw", g_d, drec)
    assert_records_equal("slhip_chunk", g_c, crec)
    assert_records_equal("scene(camera)", batch.host_scenes(), rsc)


def test_stage_rejects_bad_arguments(sl):
    table = small_table(sl)
    with pytest.raises(ValueError):
        sl.SceneBatch(table, 4, 5)              # 5 distinct classes from a table of 4
    with pytest.raises(ValueError):
        sl.SceneBatch(table, 4, 2, asset_ids=np.full((4, 2), 99))
    batch = sl.SceneBatch(table, 4, 2)
    batch.params["flags"] = 0                    # no ids and no sampling: the C-ABI must refuse
    with pytest.raises(_abi.SlhipError):
        batch.stage()


def _one_scene_records(batch, srec, drec, s):
    """Scene `s` of the batch records as a stand-alone single-scene description for the CPU oracle."""
    md = int(batch.params["max_draws_per_scene"])
    nd = int(srec[s]["draw_end"] - srec[s]["draw_begin"])
    rs = srec[s:s + 1].copy()
    rd = drec[s * md:s * md + nd].copy()
    rs["draw_begin"], rs["draw_end"] = 0, nd
    rd["scene"] = 0
    rd["clip_base"] = np.concatenate([[0], np.cumsum(rd["n_verts"][:-1], dtype=np.uint64)]).astype(np.uint32)
    return rs, rd


def test_c3_shard_end_to_end_with_gather(sl, oracle, ycb_table):
    """BASELINE config C3, one GPU's share: 64 scenes x 20 YCB-like objects, staged, settled (400 steps), placed and
    rendered at 640x480 (6-channel GT, shadows + SSAO) in two 32-scene chunks, gathered through the C-ABI
    all-gather (RCCL, one rank).  Oracle parity: the settle of 4 scenes bit-exact, the render of 4 scenes
    (integer outputs, coordinates / depth, normals bit-exact; rgb to the 8-bit bar)."""
    from stillleben_amd import parallel
    from stillleben_amd._engine import SHADOW_RES

    n_scenes, n_obj, rc = 64, 20, 32
    batch = sl.SceneBatch(ycb_table, n_scenes, n_obj, seed=2024, render_chunk=rc, scene_id_base=512)
    batch.set_camera_intrinsics(1066.778, 1067.487, 312.9869, 241.3109)
    batch.stage()
    torch.cuda.synchronize()
    staged = batch.host_bodies()
    batch.settle()
    batch.check_settled()
    batch.place()
    comm = parallel.SlhipComm(0, 1)
    gather = parallel.BatchGatherer(None, 1, depth=2, comm=comm)
    outs, gathered = [], []
    for c in range(batch.n_render_chunks()):
        buf = batch.render(c)
        views = gather([buf.rgb, buf.coord, buf.cls, buf.instance, buf.normals])
        outs.append(buf)
        gathered.append([v.clone() for v in views])
    torch.cuda.synchronize()
    comm.close()
    for buf, views in zip(outs, gathered):
        for t, v in zip((buf.rgb, buf.coord, buf.cls, buf.instance, buf.normals), views):
            assert v.shape == (1,) + tuple(t.shape) and torch.equal(v[0], t)
    W, H = batch.resolution
    bodies = batch.host_bodies()
    objs = batch.host_objects()
    # every scene: 20 distinct classes, nothing fell through the table, all objects in the picture
    for s in range(n_scenes):
        assert len(set(objs["asset"][s * n_obj:(s + 1) * n_obj].tolist())) == n_obj
    assert (bodies["pose"].reshape(-1, 4, 4)[:, 2, 3] > 0.0).all()
    inst = torch.cat([b.instance for b in outs]).cpu().numpy().view(np.uint16)[..., 0]
    seen = [len(set(np.unique(inst[s]).tolist()) - {0}) for s in range(n_scenes)]
    # occlusion hides a few of the 20 at most; a scene whose pile threw an object far off is framed from far away
    # (chooseRandomCameraPose keeps EVERY object in view, scene.cpp:525-573) and may show almost nothing
    assert np.percentile(seen, 10) >= 12 and np.mean(seen) >= 16, sorted(seen)
    # settle parity on 4 scenes (full 400 steps)
    hulls, verts = batch.se.pool.arrays()
    prm = np.array(batch.settle_params)
    for s in (0, 21, 42, 63):
        ref = staged[s * n_obj:(s + 1) * n_obj].copy()
        srec1 = np.zeros(1, dtype=SB.SETTLE_SCENE_DTYPE)
        srec1["body_end"], srec1["has_plane"], srec1["plane_z"] = n_obj, 1, 0.04
        oracle.settle(srec1, ref, hulls, verts, prm)
        got = bodies[s * n_obj:(s + 1) * n_obj]
        for f in ("pose", "lin_vel", "ang_vel", "separation", "flags", "stuck_counter", "wake_counter"):
            assert np.array_equal(np.ascontiguousarray(got[f]).view(np.uint8), np.ascontiguousarray(ref[f]).view(np.uint8)), (s, f)
    # render parity on 4 scenes
    srec, drec, _ = batch.host_render_records()
    flags = _abi.OUT_GT6 | _abi.OUT_CAM_COORD | _abi.RENDER_SSAO | _abi.RENDER_SHADOWS
    pool_arrays = batch.eng.pool.arrays()
    for s in (1, 22, 43, 62):
        rs, rd = _one_scene_records(batch, srec, drec, s)
        ref = oracle.render(pool_arrays, rs, rd, W, H, flags, shadow_res=SHADOW_RES)
        buf, k = outs[s // rc], s % rc
        assert np.array_equal(buf.instance[k].cpu().numpy().view(np.uint16), ref.instance[0]), s
        assert np.array_equal(buf.cls[k].cpu().numpy().view(np.uint16), ref.cls[0]), s
        assert np.array_equal(buf.coord[k].cpu().numpy().view(np.uint32), ref.coord[0].view(np.uint32)), s
        assert np.array_equal(buf.normals[k].cpu().numpy().view(np.uint32), ref.normals[0].view(np.uint32)), s
        d = np.abs(buf.rgb[k].cpu().numpy().astype(np.int32) - ref.rgb[0].astype(np.int32))
        assert d.max() <= 2 and (d > 1).mean() < 1e-4 and (d > 0).mean() < 0.02, (s, d.max())
    # ... with the SSAO taps run only where something can occlude (k_ssao_mask): a third of the tiles and more are open plane
    tiles, skipped = batch.eng.ssao_skipped(outs[0], W, H)
    assert tiles == outs[0].B * (W // 8) * (H // 8) and skipped > 0.3 * tiles, (tiles, skipped)


def test_batch_scene_hand_over_renders_the_same_picture(sl, ycb_table):
    """SceneBatch.scene(i) rebuilds an ordinary sl.Scene (objects, poses, camera, light, plane); rendering it through
    sl.RenderPass reproduces the batch's picture.  The per-scene host path derives world-to-camera, normal and shadow
    matrices with numpy instead of the kernels' fmaf chains, so silhouettes may move by a last-bit rounding: masks agree
    on > 99.9 % of the pixels, coordinates to 0.3 mm where both see the same object (the outcome depends on the settled heap:
    1.0e-4 on a silhouette pixel of round 6's heaps, 4e-5 before)."""
    batch = sl.SceneBatch(ycb_table, 6, 20, resolution=(320, 240), seed=9, manual_exposure=1.0)
    batch.set_camera_intrinsics(533.4, 533.7, 156.5, 120.6)
    batch.stage()
    batch.settle(frames=20)
    batch.place()
    buf = batch.render(0, mask=_abi.OUT_ALL)
    torch.cuda.synchronize()
    rp = sl.RenderPass()
    for i in (0, 5):
        scene = batch.scene(i)
        assert len(scene.objects) == 20 and [o.instance_index for o in scene.objects] == list(range(1, 21))
        res = rp.render(scene)
        a, b = res.instance_index().cpu(), buf.instance[i].cpu()
        same = (a == b)
        assert float(same.float().mean()) > 0.999 and int((b != 0).sum()) > 2000
        assert torch.equal(res.class_index().cpu()[same], buf.cls[i].cpu()[same])
        m = same[..., 0]
        assert float((res.coordDepth().cpu()[m] - buf.coord[i].cpu()[m]).abs().max()) < 3e-4
        d = (res.rgb().cpu().int() - buf.rgb[i].cpu().int()).abs()[m]
        assert float((d > 2).float().mean()) < 5e-3
