#!/usr/bin/env python3
"""Writes the asset blob examples/slhip_batch_demo.cpp reads: everything a plain C++ caller of the C-ABI needs for a
batch -- mesh pool, hull table, asset table, draw templates, slhip_synth_params, slhip_settle_params -- produced by the
Python host layer from sl.Mesh objects (the reference's loaders are host code as well; the per-scene work is all behind
the C-ABI).  No GPU needed.   usage: export_assets.py <out.bin> [n_scenes n_objects width height seed]"""
import os
import struct
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def export(path, meshes, n_scenes=8, n_objects=4, resolution=(320, 240), seed=11, frames=100):
    import stillleben_amd as sl
    from stillleben_amd import _abi
    from stillleben_amd import _settle_batch as SB
    from stillleben_amd._batch import HostPool

    pool, hulls = HostPool(), SB.HullPool()
    table = sl.AssetTable(meshes, mesh_pool=pool, hull_pool=hulls)
    pos, nrm, uv, col, idx, tex, tan = pool.arrays()
    hull_recs, hull_verts = hulls.arrays()
    proto = sl.Scene(resolution)
    proto.set_camera_intrinsics(533.4, 533.7, resolution[0] / 2 - 3.5, resolution[1] / 2 + 0.6)
    p = np.zeros((), dtype=_abi.SYNTH_PARAMS_DTYPE)
    p["n_scenes"], p["n_objects"], p["n_assets"] = n_scenes, n_objects, len(table)
    p["flags"] = _abi.SYNTH_SAMPLE_DISTINCT | _abi.SYNTH_RANDOM_PBR | _abi.SYNTH_SHADOWS
    p["seed_lo"], p["seed_hi"], p["scene_id_base"], p["render_chunk"] = seed, 0, 0, n_scenes
    p["max_draws_per_scene"] = table.bound(table.n_draws, n_objects, True) + 1
    p["max_chunks_per_scene"] = table.bound(table.n_chunks, n_objects, True) + 1
    p["max_clip_verts_per_scene"] = table.bound(table.n_clip, n_objects, True) + 4
    p["plane_z"] = 0.04
    p["proj"] = proto._projection.reshape(-1)
    p["proj_inv"] = np.linalg.inv(proto._projection.astype(np.float64)).astype(np.float32).reshape(-1)
    p["plane_size"] = (3.0, 3.0)
    p["manual_exposure"] = 1.0
    p["light_color"][:3] = 300.0
    p["ambient"][:3] = 0.05
    sp = SB.default_params(tabletop=True, frames=frames, pair_contact_budget=SB.PAIR_CONTACT_BUDGET)   # as sl.SceneBatch settles
    sp["max_bodies_per_scene"] = n_objects
    sp["max_hulls_per_scene"] = table.bound(table.n_hulls, n_objects, True)
    sp["max_hull_verts_per_scene"] = table.bound(table.n_hull_verts, n_objects, True)
    sections = [pos, nrm, uv, col, tan, idx, tex, hull_recs, hull_verts, table.records, table.templates, np.array(p), np.array(sp)]
    with open(path, "wb") as f:
        f.write(b"SLASSET1")
        f.write(struct.pack("<I", len(sections)))
        for a in sections:
            raw = np.ascontiguousarray(a).tobytes()
            f.write(struct.pack("<Q", len(raw)))
            f.write(raw)
    return table, p, sp


def demo_meshes(sl):
    fx = os.path.join(ROOT, "tests", "fixtures")
    out = []
    for i, d in enumerate((0.12, 0.17, 0.22)):
        m = sl.Mesh(os.path.join(fx, "cube.glb"))
        m.center_bbox()
        m.scale_to_bbox_diagonal(d)
        m.class_index = i + 1
        out.append(m)
    b = sl.Mesh(os.path.join(fx, "stanford_bunny", "scene.gltf"))
    b.center_bbox()
    b.scale_to_bbox_diagonal(0.25)
    b.class_index = 9
    out.append(b)
    return out


if __name__ == "__main__":
    import stillleben_amd as sl

    sl.init()
    a = sys.argv[2:]
    n_scenes, n_objects = (int(a[0]), int(a[1])) if len(a) >= 2 else (8, 4)
    res = (int(a[2]), int(a[3])) if len(a) >= 4 else (320, 240)
    seed = int(a[4]) if len(a) >= 5 else 11
    export(sys.argv[1], demo_meshes(sl), n_scenes, n_objects, res, seed)
    print(sys.argv[1], os.path.getsize(sys.argv[1]), "bytes")
