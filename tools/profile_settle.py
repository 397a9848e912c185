#!/usr/bin/env python3
"""Developer tool: cycle breakdown of the settle kernel's phases.  Build the library with
SLHIP_SETTLE_PROFILE=1 (adds wall_clock64 counters), then run this on a GPU box."""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
import stillleben_amd as sl  # noqa: E402
from stillleben_amd import _settle_batch as SB, physics, synthetic  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
FRAMES = int(sys.argv[2]) if len(sys.argv) > 2 else 100
sl.init_cuda(0)
meshes = synthetic.ycb_like_meshes(seed=0, tex_size=64)
scenes = [bench.make_scene(sl, meshes, i) for i in range(B)]
se = physics.settle_engine()
planes = [(physics.prepare_tabletop(s), physics.PLANE_HALF_Z) for s in scenes]
srec, bodies = SB.build_settle_batch(scenes, se.pool, planes)
prm = SB.sizing_hints(SB.default_params(tabletop=True, frames=FRAMES), srec, bodies, se.pool.arrays()[0])
print("hints: bodies %d hull verts %d hulls %d" % (prm["max_bodies_per_scene"], prm["max_hull_verts_per_scene"], prm["max_hulls_per_scene"]))
scr = se.scratch(B, torch.cuda.current_stream().cuda_stream, prm)
scr.zero_()
d = se.eng.upload_records(bodies)
torch.cuda.synchronize()
stream = torch.cuda.current_stream()
if os.environ.get("SLHIP_SETTLE_CUS"):   # confine the launch to a CU range: "first,count"
    import ctypes as C
    from stillleben_amd import _abi
    first, count = [int(v) for v in os.environ["SLHIP_SETTLE_CUS"].split(",")]
    h = C.c_void_p()
    _abi.check(_abi.lib().slhip_stream_create_cu_range(first, count, C.byref(h)), "slhip_stream_create_cu_range")
    stream = torch.cuda.ExternalStream(h.value)
    scr = se.scratch(B, stream.cuda_stream, prm)
    scr.zero_()
    torch.cuda.synchronize()
t = time.perf_counter()
with torch.cuda.stream(stream):
    se.run_device(srec, None, prm, d_bodies=d)
torch.cuda.synchronize()
dt = time.perf_counter() - t
print("B=%d frames=%d: %.1f ms  (%.3f ms per scene-step-batch, %.0f scene-steps/s)" % (B, FRAMES, dt * 1e3, dt * 1e3 / (FRAMES * 4), B * FRAMES * 4 / dt))
sc = scr.cpu().numpy()
if os.environ.get("SLHIP_SETTLE_PROFILE"):
    tot = np.zeros(16)
    cnt = np.zeros(16)
    for b in range(B):
        off = b * 256
        tot += np.frombuffer(sc[off:off + 128].tobytes(), dtype=np.uint64).astype(np.float64)
        cnt += np.frombuffer(sc[off + 128:off + 256].tobytes(), dtype=np.uint64).astype(np.float64)
    steps = FRAMES * 4 * B
    # wall_clock64 ticks at 100 MHz
    for i, n in enumerate(["a load", "b plane", "c broadphase", "d narrow", "d2 ranges+minsep", "wake", "f prep",
                           "g color", "h pos iters", "i integrate", "j vel iters", "k store", "d1 main gjk", "d3 manifold+fill", "d2 tilt runs"]):
        print("  %-18s %8.2f us/step" % (n, tot[i] / steps / 100.0))
    print("  total %.2f us/step" % (tot[:15].sum() / steps / 100.0))
    print("  avg hull pairs %.1f, active contacts %.1f, groups %.1f, colours %.1f" % tuple(cnt[:4] / steps))
if os.environ.get("SLHIP_SETTLE_PROFILE"):
    # per-scene cost against a-priori features (for longest-first launch order)
    per_scene = np.array([np.frombuffer(sc[b * 256:b * 256 + 128].tobytes(), dtype=np.uint64).astype(np.float64)[:15].sum() for b in range(B)])
    hulls_a = se.pool.arrays()[0]
    cnt = hulls_a["vtx_count"].astype(np.int64)
    csum = np.concatenate([[0], np.cumsum(cnt)])
    pbv = csum[bodies["hull_end"]] - csum[bodies["hull_begin"]]
    pbh = bodies["hull_end"].astype(np.int64) - bodies["hull_begin"].astype(np.int64)
    b0, b1 = srec["body_begin"].astype(np.int64), srec["body_end"].astype(np.int64)
    bh = np.concatenate([[0], np.cumsum(pbh)]); bv = np.concatenate([[0], np.cumsum(pbv)])
    sh, sv = (bh[b1] - bh[b0]).astype(np.float64), (bv[b1] - bv[b0]).astype(np.float64)
    sh2 = np.array([np.sum(pbh[a:b].astype(np.float64) ** 2) for a, b in zip(b0, b1)])
    print("  per-scene cost: mean %.1f ms max %.1f ms min %.1f ms" % (per_scene.mean() / 1e5, per_scene.max() / 1e5, per_scene.min() / 1e5))
    for name, f in (("hulls", sh), ("hull verts", sv), ("sum hulls^2", sh2)):
        print("  corr(cost, %s) = %.3f" % (name, np.corrcoef(per_scene, f)[0, 1]))
    order = np.argsort(-sh2)
    top = set(np.argsort(-per_scene)[:B // 8].tolist())
    print("  of the heaviest 1/8 of the scenes, %.0f%% are in the first quarter of the sum-hulls^2 order" % (100.0 * len(top & set(order[:B // 4].tolist())) / len(top)))
if os.environ.get("SLHIP_SETTLE_PROFILE"):
    names = ["a load", "b plane", "c broadphase", "d narrow", "d2 ranges+minsep", "wake", "f prep", "g color", "h pos iters",
             "i integrate", "j vel iters", "k store", "d1 main gjk", "d3 manifold+fill", "d2 tilt runs"]
    allc = np.stack([np.frombuffer(sc[b * 256:b * 256 + 128].tobytes(), dtype=np.uint64).astype(np.float64)[:15] for b in range(B)])
    alln = np.stack([np.frombuffer(sc[b * 256 + 128:b * 256 + 256].tobytes(), dtype=np.uint64).astype(np.float64)[:4] for b in range(B)])
    heavy = np.argsort(-allc.sum(1))[:max(1, B // 20)]
    print("  heaviest 5%% of the scenes (mean %.1f ms):" % (allc[heavy].sum(1).mean() / 1e5))
    for i, n in enumerate(names):
        print("    %-22s %8.2f us/step   (all scenes %8.2f)" % (n, allc[heavy, i].mean() / (FRAMES * 4) / 100.0, allc[:, i].mean() / (FRAMES * 4) / 100.0))
    print("    hull pairs %.1f contacts %.1f groups %.1f colours %.1f" % tuple(alln[heavy].mean(0) / (FRAMES * 4)))
