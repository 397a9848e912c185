#!/usr/bin/env python3
"""Developer tool: the exchange step of the N > 1 benchmark for several shard sizes -- `bench.py --gpus N --gather-scenes G` for
G in 64 / 256 / 1024, printing each run's `exchange` object (scenes, bytes per rank, ms, GB/s) and its scenes/s.
    python tools/gather_sweep.py [n_gpus] [steps]
On a 1-GPU box set SLHIP_BENCH_ONE_DEVICE=1 (all ranks on cuda:0, gloo through the host: a functional check, not a measurement)."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
n = int(sys.argv[1]) if len(sys.argv) > 1 else 2
steps = sys.argv[2] if len(sys.argv) > 2 else "2"
for g in (64, 256, 1024):
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(n), "--gather-scenes", str(g), "--steps", steps,
           "--warmup", "1", "--no-cpu-baseline"] + sys.argv[3:]
    r = subprocess.run(cmd, capture_output=True, text=True)
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    if r.returncode != 0 or not lines:
        print(json.dumps({"gather_scenes": g, "error": r.stderr[-400:]}))
        continue
    d = json.loads(lines[-1])        # rank 0's JSON is the last stdout line
    print(json.dumps({"gather_scenes": g, "n_gpus": d["n_gpus"], "scenes_per_s": d["value"], "exchange": d["exchange"]}))
