"""SURVEY.md 8c k6 on the C2 workload, CPU oracle: after simulate_tabletop_scene on 64 seeded 20-object scenes (the benchmark's
classes and V-HACD hulls) the piles are AT REST and the redrop rule of scene.cpp:742-755 is the exception, as the reference
treats it.  Measured by tools/physics_quality.py (the same numbers DESIGN.md section 2 quotes)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))


def test_k6_c2_piles_come_to_rest(sl, oracle):
    import physics_quality as PQ

    r = PQ.measure(n=64, seed0=900000, n_objects=20, threads=4, quiet=True)
    assert r["below_table"] == 0                                   # every z > -0.5 (in fact above the table)
    assert r["at_rest"] >= 0.95, r                                 # |v| < 0.05 m/s
    assert r["redrops_per_scene"] < 0.5, r
    assert r["asleep"] >= 0.85, r
    assert r["min_separation_p01"] > -0.01, r                      # no body left in a stuck interpenetration


def test_k6_c1_cubes_come_to_rest(sl, oracle):
    import physics_quality as PQ

    r = PQ.measure(n=64, seed0=5000, n_objects=4, threads=4, quiet=True)
    assert r["at_rest"] >= 0.95 and r["redrops_per_scene"] < 0.5 and r["below_table"] == 0, r
