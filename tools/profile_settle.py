#!/usr/bin/env python3
"""Developer tool: phase breakdown of the settle kernel on the bench workload (C2, SceneBatch staging).
Needs the profile build of the library (wall_clock64 counters around the phases of k_settle):

    python tools/profile_settle.py --build          # here (cross-compile): stillleben_amd/lib/libslhip_prof.so
    SLHIP_LIB=stillleben_amd/lib/libslhip_prof.so python tools/profile_settle.py 2048 100     # on the GPU box
"""
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
PROF_LIB = os.path.join(ROOT, "stillleben_amd", "lib", "libslhip_prof.so")

if "--build" in sys.argv:
    import __graft_entry__ as g

    cmd = [g.HIPCC] + g.HIP_FLAGS + ["-DSLHIP_SETTLE_PROFILE"] + g._sources() + ["-o", PROF_LIB]
    subprocess.run(cmd, check=True)
    print(PROF_LIB)
    sys.exit(0)

import numpy as np  # noqa: E402
import torch  # noqa: E402

import stillleben_amd as sl  # noqa: E402
from stillleben_amd import synthetic  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
FRAMES = int(sys.argv[2]) if len(sys.argv) > 2 else 100
HULLS = sys.argv[3] if len(sys.argv) > 3 else "vhacd"
sl.init_cuda(0)
table = sl.AssetTable(synthetic.ycb_like_meshes(seed=0, tex_size=64, hulls=HULLS))
batch = sl.SceneBatch(table, B, 20, seed=20260929)
batch.set_camera_intrinsics(1066.778, 1067.487, 312.9869, 241.3109)
print("hints: bodies %d hull verts %d hulls %d" % tuple(int(batch.settle_params[k]) for k in
                                                        ("max_bodies_per_scene", "max_hull_verts_per_scene", "max_hulls_per_scene")))
batch.stage()
batch.settle(frames=1)          # allocates the scratch, warms the code
torch.cuda.synchronize()
batch.stage()
scr = batch.se._scratch[torch.cuda.current_stream().cuda_stream]
scr.zero_()
torch.cuda.synchronize()
t = time.perf_counter()
batch.settle(frames=FRAMES)
torch.cuda.synchronize()
dt = time.perf_counter() - t
batch.check_settled()
print("B=%d frames=%d hulls=%s: %.1f ms  (%.0f scene-steps/s)" % (B, FRAMES, HULLS, dt * 1e3, B * FRAMES * 4 / dt))
if "prof" in os.environ.get("SLHIP_LIB", ""):
    STRIDE = 8 + 128 + 128          # ProfScratch: status, cycles[16], counts[16]
    sc = scr.cpu().numpy()
    allc = np.stack([np.frombuffer(sc[b * STRIDE + 8:b * STRIDE + 136].tobytes(), dtype=np.uint64).astype(np.float64)[:15] for b in range(B)])
    alln = np.stack([np.frombuffer(sc[b * STRIDE + 136:b * STRIDE + 264].tobytes(), dtype=np.uint64).astype(np.float64)[:4] for b in range(B)])
    names = ["a load", "b plane", "c broadphase", "d narrow", "d2 ranges+minsep", "wake", "f prep", "g color", "h pos iters",
             "i integrate", "j vel iters", "k store", "d1 main gjk", "d3 manifold+fill", "d2 tilt runs"]
    steps = FRAMES * 4
    heavy = np.argsort(-allc.sum(1))[:max(1, B // 20)]
    tot = allc.mean(0).sum()
    print("  per-scene cost: mean %.1f ms max %.1f ms min %.1f ms   (wall_clock64 ticks at 100 MHz)" % (
        allc.sum(1).mean() / 1e5, allc.sum(1).max() / 1e5, allc.sum(1).min() / 1e5))
    for i, n in enumerate(names):
        print("    %-22s %8.2f us/step %5.1f %%   heaviest 5 %% of the scenes: %8.2f" % (
            n, allc[:, i].mean() / steps / 100.0, 100.0 * allc[:, i].mean() / tot, allc[heavy, i].mean() / steps / 100.0))
    print("    total                  %8.2f us/step" % (tot / steps / 100.0))
    print("    per step: hull pairs %.1f contacts %.1f groups %.1f colours %.1f  (heaviest 5 %%: %.1f %.1f %.1f %.1f)" % (
        tuple(alln.mean(0) / steps) + tuple(alln[heavy].mean(0) / steps)))
