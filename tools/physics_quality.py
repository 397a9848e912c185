#!/usr/bin/env python3
"""Developer tool (CPU, oracle): how well the settle comes to rest on the C2 workload -- share of bodies
with |v| < 0.05 m/s after the 400 steps, share asleep, redrops per scene, deepest penetration, bodies below the table.
    python tools/physics_quality.py [n_scenes] [first_seed] [n_objects] [threads] [pair_contact_budget = 0: every point, as in PhysX and as sl.SceneBatch runs; 32: the optional reduction]
Scenes are independent: they are settled on `threads` processes (fork)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def stage(n, seed0, n_objects, budget=0):
    import bench
    import oracle
    import stillleben_amd as sl
    from stillleben_amd import _abi, synthetic
    from stillleben_amd import _settle_batch as SB
    from stillleben_amd._batch import HostPool

    sl.init()
    pool, hulls = HostPool(), SB.HullPool()
    table = sl.AssetTable(synthetic.ycb_like_meshes(seed=0, tex_size=64), mesh_pool=pool, hull_pool=hulls)
    hull_recs, hull_verts = hulls.arrays()
    p = np.zeros((), dtype=_abi.SYNTH_PARAMS_DTYPE)
    p["n_scenes"], p["n_objects"], p["n_assets"] = n, n_objects, len(table)
    p["flags"] = _abi.SYNTH_SAMPLE_DISTINCT
    p["seed_lo"], p["render_chunk"] = seed0, n
    p["max_draws_per_scene"] = table.bound(table.n_draws, n_objects, True) + 1
    p["max_chunks_per_scene"] = table.bound(table.n_chunks, n_objects, True) + 1
    p["max_clip_verts_per_scene"] = table.bound(table.n_clip, n_objects, True) + 4
    p["plane_z"] = 0.04
    bodies, ss, objs, scs = oracle.synth_stage(p, table.records)
    return bodies, ss, hull_recs, hull_verts, SB.default_params(tabletop=True, pair_contact_budget=budget)


def settle_range(args):
    import ctypes as C

    import oracle

    bodies, ss, hull_recs, hull_verts, prm, lo, hi = args
    L = oracle.lib()
    frames = int(prm["frames"])
    sub = ss[lo:hi].copy()
    trace = np.zeros((hi - lo, frames, 4), np.float32)
    caps = np.zeros((hi - lo, 7), np.uint32)
    L.slref_settle_set_trace.argtypes = [C.c_void_p]
    L.slref_settle_set_caps.argtypes = [C.c_void_p]
    L.slref_settle_set_trace(C.c_void_p(trace.ctypes.data))
    L.slref_settle_set_caps(C.c_void_p(caps.ctypes.data))
    try:
        oracle.settle(sub, bodies, hull_recs, hull_verts, prm)
    finally:
        L.slref_settle_set_trace(None)
        L.slref_settle_set_caps(None)
    b0, b1 = int(ss[lo]["body_begin"]), int(ss[hi - 1]["body_end"])
    return lo, hi, bodies[b0:b1].copy(), trace, caps


def measure(n=64, seed0=900000, n_objects=20, threads=8, budget=0, quiet=False):
    import multiprocessing as mp

    bodies, ss, hull_recs, hull_verts, prm = stage(n, seed0, n_objects, budget)
    threads = max(1, min(threads, n))
    cuts = np.linspace(0, n, threads + 1).astype(int)
    jobs = [(bodies, ss, hull_recs, hull_verts, prm, int(cuts[i]), int(cuts[i + 1])) for i in range(threads) if cuts[i + 1] > cuts[i]]
    if threads > 1:
        with mp.get_context("fork").Pool(threads) as pool:
            res = pool.map(settle_range, jobs)
    else:
        res = [settle_range(j) for j in jobs]
    frames = int(prm["frames"])
    trace = np.zeros((n, frames, 4), np.float32)
    caps = np.zeros((n, 7), np.uint32)
    for lo, hi, b, t, c in res:
        b0, b1 = int(ss[lo]["body_begin"]), int(ss[hi - 1]["body_end"])
        bodies[b0:b1] = b
        trace[lo:hi] = t
        caps[lo:hi] = c
    from stillleben_amd import _settle_batch as SB

    speed = np.linalg.norm(bodies["lin_vel"][:, :3], axis=1)
    wspeed = np.linalg.norm(bodies["ang_vel"][:, :3], axis=1)
    asleep = (bodies["flags"] & SB.BODY_ASLEEP) != 0
    out = {
        "scenes": n, "objects": n_objects,
        "at_rest": float(np.mean(speed < 0.05)),
        "asleep": float(np.mean(asleep)),
        "redrops_per_scene": float(trace[:, -1, 1].mean()),
        "redrops_last_second": float((trace[:, -1, 1] - trace[:, -26, 1]).mean()),
        "scenes_all_at_rest": float(np.mean((speed.reshape(n, n_objects) < 0.05).all(axis=1))),
        "v_p50": float(np.median(speed)), "v_p95": float(np.quantile(speed, 0.95)), "v_max": float(speed.max()),
        "w_p95": float(np.quantile(wspeed, 0.95)),
        "below_table": int((bodies["pose"][:, 11] < 0.0).sum()),
        "stuck_bodies": int((bodies["stuck_counter"] > 0).sum()),
        "min_separation_p01": float(np.quantile(np.minimum(bodies["separation"], 1.0), 0.01)),
        "active_contacts_last": float(trace[:, -1, 2].mean()),
        "contact_drop_steps": int(caps[:, 0].sum()),      # scene-steps that dropped contacts / hull pairs beyond the default capacities
        "pair_drop_steps": int(caps[:, 1].sum()),
        "max_contacts_offered": int(caps[:, 2].max()), "max_hull_pairs_found": int(caps[:, 3].max()),
        "pair_budget": int(prm["pair_contact_budget"]), "pair_budget_reduced_steps": int(caps[:, 4].sum()),
        "group_drop_steps": int(caps[:, 5].sum()), "contacts_per_scene_step": float(caps[:, 6].sum()) / (n * frames * int(prm["substeps"])),
        "asleep_by_frame": [float(trace[:, f, 0].mean()) for f in (24, 49, 74, 99) if f < frames],
    }
    if not quiet:
        for k, v in out.items():
            print("%-22s %s" % (k, ("%.4f" % v) if isinstance(v, float) else v))
    return out


if __name__ == "__main__":
    a = [int(x) for x in sys.argv[1:]]
    measure(*(a + [64, 900000, 20, 8, 0][len(a):]))
