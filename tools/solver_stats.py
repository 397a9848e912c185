#!/usr/bin/env python3
"""Developer tool (CPU, oracle): the distribution of the contact solver's work per step on the C2 workload -- active
contacts, friction anchors, colours and the sequential chain of one sweep.  These size the LDS tiers of k_w_solve and
explain its lane occupancy.   python tools/solver_stats.py [n_scenes]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
import oracle  # noqa: E402
import stillleben_amd as sl  # noqa: E402
from stillleben_amd import _abi, synthetic  # noqa: E402
from stillleben_amd import _settle_batch as SB  # noqa: E402
from stillleben_amd._batch import HostPool  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 16
sl.init()
pool, hulls = HostPool(), SB.HullPool()
table = sl.AssetTable(synthetic.ycb_like_meshes(seed=0, tex_size=64), mesh_pool=pool, hull_pool=hulls)
hull_recs, hull_verts = hulls.arrays()
p = np.zeros((), dtype=_abi.SYNTH_PARAMS_DTYPE)
p["n_scenes"], p["n_objects"], p["n_assets"] = n, bench.N_OBJECTS, len(table)
p["flags"] = _abi.SYNTH_SAMPLE_DISTINCT
p["seed_lo"], p["render_chunk"] = 900000, n
p["max_draws_per_scene"] = table.bound(table.n_draws, bench.N_OBJECTS, True) + 1
p["max_chunks_per_scene"] = table.bound(table.n_chunks, bench.N_OBJECTS, True) + 1
p["max_clip_verts_per_scene"] = table.bound(table.n_clip, bench.N_OBJECTS, True) + 4
p["plane_z"] = 0.04
bodies, ss, objs, scs = oracle.synth_stage(p, table.records)
st = oracle.settle_stats(ss, bodies, hull_recs, hull_verts, SB.default_params(tabletop=True, pair_contact_budget=SB.PAIR_CONTACT_BUDGET))
gi = st.pop("gjk_iters").astype(np.float64)
print("main GJK runs per step and scene: %.1f; share by iterations:" % (gi.sum() / max(1.0, st["active"].sum())),
      " ".join("%d:%.3f" % (i, gi[i] / gi.sum()) for i in range(len(gi)) if gi[i] / gi.sum() >= 0.002))
ti = st.pop("tilt_iters").astype(np.float64)
print("tilt runs per step and scene: %.1f; share by iterations:" % (ti.sum() / max(1.0, st["active"].sum())),
      " ".join("%d:%.3f" % (i, ti[i] / ti.sum()) for i in range(len(ti)) if ti[i] > 0))
for name, h in st.items():
    h = h.astype(np.float64)
    tot = h.sum()
    cdf = np.cumsum(h) / tot
    mean = (np.arange(256) * h).sum() / tot
    q = {f: int(np.searchsorted(cdf, f)) for f in (0.5, 0.9, 0.99, 0.999, 0.9999)}
    print("%-8s mean %6.1f  median %3d  p90 %3d  p99 %3d  p99.9 %3d  p99.99 %3d  max %3d" % (name, mean, q[0.5], q[0.9], q[0.99], q[0.999], q[0.9999], int(np.nonzero(h)[0].max())))
