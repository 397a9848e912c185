#!/usr/bin/env python3
"""Developer tool (CPU, oracle): how many solver rows a wave walks per sweep when it solves two scenes side by side -- random
partners against neighbours in the cost order (the per-step colour profiles come from slref_settle_set_profile_dump).
usage: python tools/solver_pairing.py [n_scenes]"""
import sys, ctypes as C
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bench, oracle
import stillleben_amd as sl
from stillleben_amd import _abi, synthetic
from stillleben_amd import _settle_batch as SB
from stillleben_amd._batch import HostPool
n=int(sys.argv[1]) if len(sys.argv)>1 else 64
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
L = oracle.lib()
L.slref_settle_set_profile_dump.argtypes=[C.c_char_p]
L.slref_settle_set_profile_dump(b"/tmp/profile.txt")
# stats must be on for step_stats to run
h = np.zeros(1096, np.uint64)
L.slref_settle_set_stats.argtypes=[C.c_void_p]
L.slref_settle_set_stats(h.ctypes.data_as(C.c_void_p))
oracle.settle(ss, bodies, hull_recs, hull_verts, SB.default_params(tabletop=True, pair_contact_budget=SB.PAIR_CONTACT_BUDGET))
L.slref_settle_set_stats(None)
L.slref_settle_set_profile_dump(None)
# file: scene-major (scene 0 steps 0..399, scene 1 ...)
rows=[list(map(int,l.split()))[1:] for l in open('/tmp/profile.txt')]
steps=400
assert len(rows)==n*steps, (len(rows), n*steps)
def vec(r):
    v=np.zeros(16,int); v[:min(16,len(r))]=sorted(r,reverse=True)[:16]; return v   # colours sorted by size: solver could order colours by size too
single=0; paired=0; paired_unsorted_col=0; rnd=0
rng=np.random.default_rng(0)
for t in range(steps):
    P=[rows[s*steps+t] for s in range(n)]
    tot=np.array([sum(r) for r in P])
    single+=tot.sum()
    order=np.argsort(-tot)
    for i in range(0,n,2):
        a,b=P[order[i]],P[order[i+1]]
        m=max(len(a),len(b)); aa=a+[0]*(m-len(a)); bb=b+[0]*(m-len(b))
        paired_unsorted_col+=sum(max(x,y) for x,y in zip(aa,bb))
        paired+=np.maximum(vec(a),vec(b)).sum()
    perm=rng.permutation(n)
    for i in range(0,n,2):
        a,b=P[perm[i]],P[perm[i+1]]
        m=max(len(a),len(b)); aa=a+[0]*(m-len(a)); bb=b+[0]*(m-len(b))
        rnd+=sum(max(x,y) for x,y in zip(aa,bb))
print("rows walked per sweep, summed: one scene per wave %d; two per wave: random pairs %d (%.2f), cost-sorted pairs %d (%.2f), + colours matched by size %d (%.2f)"%(single, rnd, rnd/single, paired_unsorted_col, paired_unsorted_col/single, paired, paired/single))
