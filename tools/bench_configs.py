#!/usr/bin/env python3
"""Developer tool: the BASELINE.json configurations other than the headline one (bench.py measures C2),
each timed on one MI355X with HIP events and priced against the algorithmic bytes of SURVEY.md 8d.
Prints one JSON line per configuration; `profiles/` keeps the output.

  C1  4 cubes, 320x240: settle + instance-mask render (plumbing case, GPU path)
  C3  512 C2 scenes in one batch (64 per GPU x 8 in the reference's layout; here all 512 on one GPU)
  C4  stanford bunny x50 (3.47 M triangles), 640x480, all 8 outputs, render only
  C5  sl.diff: 64 objects x 32 pose hypotheses, render + backward pass per hypothesis"""
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import bench  # noqa: E402
import scenes as S  # noqa: E402
import stillleben_amd as sl  # noqa: E402
from stillleben_amd import _abi, physics, synthetic  # noqa: E402
from stillleben_amd._context import engine  # noqa: E402

PEAK = 8000.0


def timed(fn, reps=5, warm=1):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


QUIET = False     # bench.py --config: the records are wrapped into the bench contract's line instead of printed


def emit(name, **kw):
    rec = dict(config=name, **kw)
    if not QUIET:
        print(json.dumps(rec))
    return rec


def c1(B=256):
    m = sl.Mesh(S.CUBE)
    m.center_bbox()
    m.scale_to_bbox_diagonal(0.2)
    scs = []
    for i in range(B):
        s = sl.Scene((320, 240), seed=i)
        for _ in range(4):
            s.add_object(sl.Object(m))
        scs.append(s)
    t0 = time.perf_counter()
    physics.settle_batch(scs)
    torch.cuda.synchronize()
    t_settle = time.perf_counter() - t0
    eng = engine()
    ms = timed(lambda: eng.render(scs, _abi.OUT_INSTANCE, ssao=False, shadows=False))
    return emit("C1 4 cubes 320x240 (batch of %d scenes)" % B, settle_s_per_batch=t_settle, render_ms_per_batch=ms,
         scenes_per_s_render=B / (ms * 1e-3), scenes_per_s_settle_incl_host=B / t_settle)


def c3():
    meshes = synthetic.ycb_like_meshes(seed=0)
    B = 512
    scs = [bench.make_scene(sl, meshes, 10000 + i) for i in range(B)]
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    physics.settle_batch(scs)
    torch.cuda.synchronize()
    t_settle = time.perf_counter() - t0
    for s in scs:
        s.choose_random_light_direction()
    eng = engine()
    chunks = [scs[i:i + 128] for i in range(0, B, 128)]
    bufs = [None] * len(chunks)

    def render():
        for k, ch in enumerate(chunks):
            bufs[k] = eng.render(ch, _abi.OUT_GT6, ssao=True, shadows=True, buffers=bufs[k])

    ms = timed(render, reps=3)
    alg = B * (20 * 8192 * 68 + 20 * 16384 * 12 + 307200 * 40)   # 27.4 MB per scene
    return emit("C3 512 C2 scenes, one GPU", settle_s_incl_host_glue=t_settle, render_ms=ms, scenes_per_s_render_incl_host_glue=B / (ms * 1e-3),
         roofline={"bound": "hbm", "algorithmic_bytes": alg, "achieved": alg / (ms * 1e-3) / 1e9, "peak": PEAK, "unit": "GB/s",
                   "frac": alg / (ms * 1e-3) / 1e9 / PEAK, "note": "render incl. per-call host batch assembly (engine.render)"})


def c4(B=1):
    """Bunny x 50 (3.47 M triangles per scene), 640x480, all 8 outputs, render only.  B > 1: B such scenes (their own random
    rotations) in ONE launch sequence -- the form in which the configuration stresses bandwidth and not launch latency."""
    import ctypes as C

    m = sl.Mesh(S.BUNNY, physics=False)
    m.center_bbox()
    m.scale_to_bbox_diagonal(0.5)
    rng = np.random.default_rng(4)
    scs = []
    for b in range(B):
        scene = sl.Scene((640, 480))
        for i in range(50):
            o = sl.Object(m)
            pose = np.eye(4, dtype=np.float32)
            pose[:3, :3] = S.random_rotation(rng)
            z = 1.0 + 2.0 * (i / 49.0)
            pose[:3, 3] = [((i % 10) - 4.5) * 0.11 * z, ((i // 10) - 2.0) * 0.16 * z, z]
            o.set_pose(torch.from_numpy(pose))
            scene.add_object(o)
        scene.light_directions = torch.tensor([[0.2, 0.5, 0.8]])
        scs.append(scene)
    eng = engine()
    from stillleben_amd._batch import build_batch

    srec, drec, crec = build_batch(scs, eng.pool, None, with_shadows=False)
    eng.L.slhip_timing_enable(0)
    buf = [None]

    def render():
        buf[0] = eng.render_records(srec, drec, crec, 640, 480, _abi.OUT_ALL, ssao=False, shadows=False, buffers=buf[0])

    reps = 10 if B == 1 else 5
    ms = timed(render, reps=reps)
    # the visibility raster's share: the library's own HIP events around each phase of one more sequence
    eng.L.slhip_timing_enable(1)
    render()
    torch.cuda.synchronize()
    ph = (C.c_float * 8)()
    eng.L.slhip_render_timings(C.byref(ph))
    eng.L.slhip_timing_enable(0)
    names = ["shadow_raster", "shadow_large", "vis_raster", "vis_large", "shade", "ssao", "ssao_apply", "tonemap"]
    phases = {n: float(ph[i]) for i, n in enumerate(names)}
    tris = int(drec["n_tris"].sum())
    verts = int(drec["n_verts"].sum())
    alg = verts * 68 + tris * 12 + B * 307200 * 88
    raster_ms = phases["vis_raster"] + phases["vis_large"]
    alg_raster = verts * 16 + tris * 12 + B * 307200 * 8          # positions + indices in, one 64-bit key per pixel out
    return emit("C4 bunny x50 raster stress" + (" (batch of %d scenes per launch sequence)" % B if B > 1 else ""), batch=B,
         triangles=tris, vertices=verts, render_ms=ms, render_ms_per_scene=ms / B, mtris_per_s=tris / (ms * 1e-3) / 1e6,
         phases_ms=phases, raster_share=raster_ms / max(1e-9, sum(phases.values())),
         raster={"ms": raster_ms, "mtris_per_s": tris / max(1e-9, raster_ms * 1e-3) / 1e6, "algorithmic_bytes": alg_raster,
                 "achieved_GBps": alg_raster / max(1e-9, raster_ms * 1e-3) / 1e9, "frac": alg_raster / max(1e-9, raster_ms * 1e-3) / 1e9 / PEAK},
         roofline={"bound": "hbm", "algorithmic_bytes": alg, "achieved": alg / (ms * 1e-3) / 1e9, "peak": PEAK, "unit": "GB/s",
                   "frac": alg / (ms * 1e-3) / 1e9 / PEAK,
                   "note": ("one scene per launch sequence: 3.5 M triangles cannot fill 256 CUs for long, fixed launch latencies dominate"
                            if B == 1 else "%d scenes per launch sequence; SURVEY 8d byte model: 68 B per vertex + 12 B per triangle + 88 B per pixel" % B)})


def c5():
    cube = sl.Mesh(S.CUBE, physics=False)
    cube.center_bbox()
    cube.scale_to_bbox_diagonal(0.12)
    scene = sl.Scene((640, 480))
    scene.set_camera_intrinsics(*bench.INTRINSICS)
    rng = np.random.default_rng(5)
    base = []
    for i in range(64):
        o = sl.Object(cube)
        pose = np.eye(4, dtype=np.float32)
        pose[:3, :3] = S.random_rotation(rng)
        pose[:3, 3] = [((i % 8) - 3.5) * 0.11, ((i // 8) - 3.5) * 0.085, 1.6 + 0.3 * rng.uniform()]
        o.set_pose(torch.from_numpy(pose))
        scene.add_object(o)
        base.append(torch.from_numpy(pose))
    scene.light_directions = torch.tensor([[0.1, 0.2, 0.9]])
    scene.manual_exposure = 1.0
    grad = torch.from_numpy(np.random.default_rng(1).standard_normal((3, 480, 640)).astype(np.float32)).cuda()
    rp = sl.RenderPass()
    deltas = torch.from_numpy(rng.normal(0, 0.01, (32, 64, 6)).astype(np.float32))

    def hypothesis(h):
        for k, o in enumerate(scene.objects):
            o.set_pose(sl.diff.apply_pose_delta(base[k], deltas[h, k]))
        res = rp.render(scene)
        return sl.diff.backpropagate_gradient_to_poses(scene, res, grad)

    hypothesis(0)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for h in range(32):
        d = hypothesis(h)
    torch.cuda.synchronize()
    total = time.perf_counter() - t0
    # the batch form: all 32 hypotheses in one render launch sequence + 32 backward launches
    hyps = torch.stack([torch.stack([sl.diff.apply_pose_delta(base[k], deltas[h, k]) for k in range(64)]) for h in range(32)])
    db, buf32 = sl.diff.backpropagate_gradient_to_poses_batch(scene, hyps, grad, return_results=True)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(3):
        db = sl.diff.backpropagate_gradient_to_poses_batch(scene, hyps, grad)
    torch.cuda.synchronize()
    total_batch = (time.perf_counter() - t0) / 3
    # the backward half alone, all 32 hypotheses in one launch sequence (3 launches)
    hyp_np = hyps.numpy()
    ms_bb = timed(lambda: sl.diff.pose_backward_batch_on_buffers(scene, buf32, hyp_np, grad), reps=10)
    agree = float((db[31] - d).abs().max() / max(1e-12, float(d.abs().max())))
    res = rp.render(scene)
    ms_b = timed(lambda: sl.diff.backpropagate_gradient_to_poses(scene, res, grad), reps=10)
    alg = 307200 * 35 + 64 * 88
    return emit("C5 sl.diff 64 objects x 32 hypotheses", s_total_32_hypotheses=total, ms_per_hypothesis_render_plus_backward=total / 32 * 1e3,
         s_total_32_hypotheses_batch_api=total_batch, ms_per_hypothesis_batch_api=total_batch / 32 * 1e3,
         batch_vs_loop_max_rel_diff=agree,
         backward_ms_single_hypothesis=ms_b, backward_ms_32_hypotheses_one_sequence=ms_bb, grad_shape=list(d.shape),
         roofline={"bound": "hbm", "kernel": "diff backward, 32 hypotheses in one launch sequence (3 launches)",
                   "algorithmic_bytes": 32 * alg, "achieved": 32 * alg / (ms_bb * 1e-3) / 1e9,
                   "peak": PEAK, "unit": "GB/s", "frac": 32 * alg / (ms_bb * 1e-3) / 1e9 / PEAK,
                   "note": "10.8 MB per hypothesis; one hypothesis alone (4 launches) is launch-latency bound: %.3f ms" % ms_b})


if __name__ == "__main__":
    sl.init_cuda(0)
    which = sys.argv[1:] or ["c1", "c1big", "c3", "c4", "c5"]
    for w in which:
        {"c1": c1, "c1big": lambda: c1(4096), "c3": c3, "c4": c4, "c4x64": lambda: c4(64), "c5": c5}[w]()
