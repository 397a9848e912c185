#!/usr/bin/env python3
"""Developer tool: soak of the device-side scene synthesis and of the whole batched path on the bench workload (C2, V-HACD
hulls): N scenes staged on the GPU vs oracle/synth_ref.c (bit-exact records), settled with BOTH settle implementations
(persistent / lockstep) vs oracle/settle_ref.c (bit-exact bodies, threaded over scenes), placed vs the oracle (bit-exact
scene / draw / chunk records).   usage: soak_synth.py [n_scenes] [seed]"""
import os
import sys
import time
from concurrent.futures import ThreadPoolExecutor

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import oracle  # noqa: E402
import stillleben_amd as sl  # noqa: E402
from stillleben_amd import _settle_batch as SB, synthetic  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 512
SEED = int(sys.argv[2]) if len(sys.argv) > 2 else 424242
sl.init_cuda(0)
oracle.build()
table = sl.AssetTable(synthetic.ycb_like_meshes(seed=0, tex_size=64))


def eq(name, a, b):
    bad = 0
    for f in a.dtype.names:
        x, y = np.ascontiguousarray(a[f]), np.ascontiguousarray(b[f])
        if not np.array_equal(x.view(np.uint8), y.view(np.uint8)):
            bad += 1
            print("MISMATCH %s.%s at %s" % (name, f, np.argwhere(x != y)[:3].tolist()))
    return bad


bad = 0
ref_bodies = None
for impl in ("persistent", "lockstep"):
    os.environ["SLHIP_SETTLE_IMPL"] = impl
    batch = sl.SceneBatch(table, N, 20, seed=SEED, render_chunk=64, scene_id_base=7 * N)
    batch.set_camera_intrinsics(1066.778, 1067.487, 312.9869, 241.3109)
    batch.stage()
    torch.cuda.synchronize()
    rb, rss, robj, rsc = oracle.synth_stage(batch.params, table.records)
    bad += eq("body", batch.host_bodies(), rb) + eq("settle_scene", batch.host_settle_scenes(), rss)
    bad += eq("object", batch.host_objects(), robj) + eq("scene", batch.host_scenes(), rsc)
    t = time.time()
    batch.settle()
    batch.check_settled()
    t_gpu = time.time() - t
    if ref_bodies is None:
        hulls, verts = batch.se.pool.arrays()
        prm = np.array(batch.settle_params)
        t = time.time()

        def one(s):
            b = rb[s * 20:(s + 1) * 20].copy()
            ss = np.zeros(1, dtype=SB.SETTLE_SCENE_DTYPE)
            ss["body_end"], ss["has_plane"], ss["plane_z"] = 20, 1, 0.04
            oracle.settle(ss, b, hulls, verts, prm)
            return b

        with ThreadPoolExecutor(min(32, os.cpu_count() or 1)) as ex:
            ref_bodies = np.concatenate(list(ex.map(one, range(N))))
        print("oracle settle of %d scenes: %.1f s" % (N, time.time() - t))
    got = batch.host_bodies()
    for f in ("pose", "lin_vel", "ang_vel", "separation", "flags", "stuck_counter", "wake_counter"):
        x, y = np.ascontiguousarray(got[f]), np.ascontiguousarray(ref_bodies[f])
        if not np.array_equal(x.view(np.uint8), y.view(np.uint8)):
            bad += 1
            print("MISMATCH settle(%s).%s scenes %s" % (impl, f, np.unique(np.argwhere(x != y)[:, 0] // 20)[:8].tolist()))
    batch.place()
    torch.cuda.synchronize()
    srec, drec, crec = oracle.synth_place(batch.params, table.records, table.templates, got, robj, rsc)
    g_s, g_d, g_c = batch.host_render_records()
    bad += eq("slhip_scene", g_s, srec) + eq("slhip_draw", g_d, drec) + eq("slhip_chunk", g_c, crec)
    print("%s: %d scenes staged, settled (%.2f s on the GPU) and placed: %s" % (impl, N, t_gpu, "bit-exact" if bad == 0 else "%d mismatching fields" % bad))
sys.exit(1 if bad else 0)
