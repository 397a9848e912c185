#!/usr/bin/env python3
"""Developer tool (GPU box): one render chunk of C2 scenes alone, per-phase times from the library's HIP events
(slhip_render_timings).   python tools/time_render.py [scenes=1024] [repeats=3]"""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

import bench  # noqa: E402
import stillleben_amd as sl  # noqa: E402
from stillleben_amd import _abi, synthetic  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
REP = int(sys.argv[2]) if len(sys.argv) > 2 else 3
sl.init_cuda(0)
table = sl.AssetTable(synthetic.ycb_like_meshes(seed=0, tex_size=1024))
batch = sl.SceneBatch(table, B, 20, resolution=bench.RESOLUTION, seed=20260929, render_chunk=B)
batch.set_camera_intrinsics(*bench.INTRINSICS)
L = _abi.lib()
batch.stage()
batch.settle()
batch.check_settled()
batch.place()
buf = batch.render(0, _abi.OUT_GT6, ssao=True)
torch.cuda.synchronize()
names = ["shadow_raster", "shadow_large", "vis_raster", "vis_large", "shade", "ssao", "ssao_apply", "tonemap"]
L.slhip_timing_enable(1)
for r in range(REP):
    buf = batch.render(0, _abi.OUT_GT6, ssao=True, buffers=buf)
    ms = (C.c_float * 8)()
    L.slhip_render_timings(C.byref(ms))
    print("B=%d " % B + "  ".join("%s %.2f" % (n, ms[i]) for i, n in enumerate(names)) + " | sum %.2f" % sum(ms))
L.slhip_timing_enable(0)
