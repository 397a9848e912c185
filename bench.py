#!/usr/bin/env python
"""Headline benchmark: scenes/sec for {tabletop settle of 20 YCB-like objects + 640x480
6-channel ground-truth render} (BASELINE.json metric, config C2), one process per GPU.

A "step" = one batch of `--batch` freshly seeded scenes per GPU going through the whole hot
path ON THE DEVICE: slhip_synth_stage (object choice, random stack) -> slhip_settle (400 physics
steps per scene) -> slhip_synth_place (camera, light, shadow matrix, draw records) -> slhip_render
(shadow pass, visibility raster, deferred shade, SSAO, tone map) -> for N > 1 an RCCL all-gather of
the step's exchange shard.  Resident in HBM before the timed region: the asset table (mesh pool,
textures, hulls of the 21 classes).  Everything per scene happens inside the timed region.

Prints ONE JSON line (rank 0) -- see the repository brief for the contract."""
import argparse
import ctypes as C
import json
import os
import random
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

RESOLUTION = (640, 480)
INTRINSICS = (1066.778, 1067.487, 312.9869, 241.3109)  # reference examples/ycb.py:32
N_OBJECTS = 20
TARGET_RING = int(os.environ.get("SLHIP_BENCH_TARGET_RING", 4))     # render-target sets kept alive (chunks of --render-chunk scenes): 4 x 1024 scenes = 50 GB


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=None,
                    help="scenes per GPU per step = one settle call.  Every one of a settle's 2 400 launches ends with a tail of idle "
                         "SIMDs, and the more rounds of workgroups a launch has, the smaller the share of that tail: round 2 measured "
                         "4096 -> 4 490, 8192 -> 5 110, 16384 -> 5 550, 24576 -> 5 660 scenes/s; round 5, same box, default lengths: "
                         "16384 -> 9 866, 32768 -> 10 240; 3-step runs: 32768 -> 9 700, 49152 -> 9 790, 65535 -> 9 830 (the settle scratch "
                         "is 2 MB per scene: 65 GB at 32768)")
    ap.add_argument("--render-chunk", type=int, default=None,
                    help="scenes per render launch sequence (fewer, larger sequences: every kernel boundary is a chance for "
                         "queued settle workgroups to take the freed SIMDs); default 512: the same rate as 1024 (round 6, 6-step runs: 9 926 / 9 989 "
                         "against 9 931 / 9 995) with two 30 GB render scratch sets instead of two 60 GB ones -- peak HBM 175 GB instead of 262")
    ap.add_argument("--settle-streams", type=int, default=1,
                    help="settle launches kept in flight on streams of their own (the render of the previous step always "
                         "overlaps the settle of the next one)")
    ap.add_argument("--render-streams", type=int, default=2,
                    help="render streams the chunks of a step alternate between, each with its own scratch (60 GB per 1024-scene "
                         "chunk, shadow maps included).  Round 3 on one MI355X, 6 steps: settle/render streams 2/1 -> 8 310, "
                         "1/1 -> 8 410, 1/2 -> 8 555, 2/2 with 512-scene chunks -> 7 350 scenes/s (2/2 with 1024 does not fit 288 GB)")
    ap.add_argument("--gather-scenes", type=int, default=64,
                    help="N > 1: scenes per rank and step whose ground truth is all-gathered to every rank (BASELINE config C3: "
                         "512 scenes = 64 per GPU x 8); 0 = no exchange")
    ap.add_argument("--gather", default="c3", choices=["c3", "compact", "full"],
                    help="N > 1, what a rank exchanges per step: c3 = --gather-scenes scenes of the first chunk (BASELINE config C3's shape); "
                         "compact = rgb + depth + class + instance (12 B/px) of EVERY scene it renders; full = the whole 6-channel ground "
                         "truth (40 B/px) of every scene -- both streamed in pieces of --gather-piece scenes (double-buffered staging) "
                         "while the next chunks render")
    ap.add_argument("--gather-piece", type=int, default=128, help="scenes per all-gather of the streamed modes")
    ap.add_argument("--config", default="C2", choices=["C1", "C2", "C3", "C4", "C5"],
                    help="BASELINE.json configuration: C2 (default) is the headline metric; C1 4 cubes 320x240 (4096 scenes per step), "
                         "C3 512 C2 scenes through the per-object API, C4 bunny x 50 raster stress, C5 sl.diff 64 objects x 32 "
                         "hypotheses -- one JSON line in the same schema each (single GPU)")
    ap.add_argument("--pair-budget", type=int, default=0,
                    help="slhip_settle_params.pair_contact_budget: 0 (default) = every contact point goes to the solver, as in PhysX; "
                         "N > 0 = the compound manifold reduction (NOT in the reference): a body pair touching through more hull pairs "
                         "keeps the N deepest -- 32 is what rounds 3 and 4 ran (+2 %% scenes/s: 10 090 against 9 880 on one box)")
    ap.add_argument("--hulls", default="vhacd", choices=["vhacd", "native"],
                    help="collision hulls of the 21 classes: vhacd = the decompositions the reference's own V-HACD gave this geometry "
                         "(shipped fixture, the default), native = the in-tree decomposition (stillleben_amd/acd.py over the quick-hull of "
                         "libslhip.so): row S1 of SURVEY 8a feeding the measured number")
    ap.add_argument("--no-ssao", action="store_true")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-scenes", type=int, default=2, help="scenes per host thread of the bounded CPU-baseline sample")
    return ap.parse_args()


def make_scene(sl, meshes, seed):
    """BASELINE config C2: 20 of the 21 YCB-like meshes, YCB intrinsics,
    random metallic/roughness, one directional light (colour 300), 3x3 m background plane."""
    rnd = random.Random(seed)
    scene = sl.Scene(RESOLUTION, seed=seed)
    scene.set_camera_intrinsics(*INTRINSICS)
    for mesh in rnd.sample(meshes, N_OBJECTS):
        obj = sl.Object(mesh)
        obj.metallic = rnd.random()
        obj.roughness = rnd.random()
        scene.add_object(obj)
    scene.background_plane_size = torch.tensor([3.0, 3.0])
    scene.ambient_light = torch.tensor([0.05, 0.05, 0.05])
    return scene


class Pipeline:
    """The whole hot path of one GPU, device-resident: per step ONE slhip_synth_stage (initial stacks of 4096 scenes),
    ONE slhip_settle, ONE slhip_synth_place (cameras, lights, shadow matrices, draw records) on a settle stream, then
    the slhip_render launch sequences of the step's chunks on the render stream; `ring` SceneBatch record sets are
    recycled.  The host only enqueues: nothing is read back, no host thread sits between settle and render."""

    def __init__(self, sl, table, batch, render_chunk, ssao, settle_streams, seed, rank, render_streams=1, pair_contact_budget=0):
        from stillleben_amd import _abi
        from stillleben_amd._context import engine

        self.sl, self._abi = sl, _abi
        self.eng = engine()
        self.batch, self.render_chunk, self.ssao = batch, render_chunk, ssao
        self.mask = _abi.OUT_GT6
        dev = self.eng.device
        # record sets: one more than settle streams, so that a settle never waits for the render before last (SLHIP_BENCH_RING: developer knob)
        self.ring = int(os.environ.get("SLHIP_BENCH_RING", settle_streams + 1))
        self.sets = []
        for _ in range(self.ring):
            b = sl.SceneBatch(table, batch, N_OBJECTS, resolution=RESOLUTION, seed=seed, render_chunk=render_chunk,
                              shadows=not os.environ.get("SLHIP_BENCH_NO_SHADOWS"), pair_contact_budget=pair_contact_budget)
            b.set_camera_intrinsics(*INTRINSICS)
            self.sets.append(b)
        # stream priorities "<settle>,<render>" (-1 = high, 0 = default): the settle is a chain of 2400 short dependent launches, the
        # render a few long ones -- a settle workgroup that waits behind a render kernel's queue delays everything after it
        # (round 5, 10 steps: 9 635 -> 9 700 scenes/s).  SLHIP_BENCH_PRIO overrides (developer knob).
        ps, pr = (int(x) for x in os.environ.get("SLHIP_BENCH_PRIO", "-1,0").split(","))
        self.s_settle = [torch.cuda.Stream(device=dev, priority=ps) for _ in range(settle_streams)]
        self.s_render = torch.cuda.Stream(device=dev, priority=pr)
        n_rs = max(1, int(render_streams))
        self.s_render_all = [self.s_render] + [torch.cuda.Stream(device=dev, priority=pr) for _ in range(n_rs - 1)]
        if os.environ.get("SLHIP_BENCH_SETTLE_CUS"):
            # developer knob: the settle streams confined to the first N compute units, the render streams to the rest
            # (slhip_stream_create_cu_range): no settle workgroup ever waits behind a render kernel's waves
            from stillleben_amd.parallel import cu_partition_streams

            n_cu = int(os.environ["SLHIP_BENCH_SETTLE_CUS"])
            self.s_settle, self.s_render = cu_partition_streams(n_cu, settle_streams, device=dev)
            self.s_render_all = [self.s_render] + [cu_partition_streams(n_cu, 1, device=dev)[1] for _ in range(n_rs - 1)]
        self.free = [None] * self.ring        # event: the set's previous render finished (its records may be rewritten)
        self.buffers = []                     # ring of render-target sets: a chunk's ground truth stays in HBM until TARGET_RING
                                              # further chunks have been rendered (50 GB of the last 4096 scenes at the defaults)
        self.gatherer = None
        self.gather_scenes = 0
        self.gather_mode, self.streamer, self.depth_stage = "c3", None, {}
        self.pending = []
        self.rank = rank
        self.steps_launched = 0

    def launch_step(self, k, scene_id_base):
        """Enqueues step k completely (returns at once): stage + settle + place on a settle stream, the render of
        every chunk on the render stream, for N > 1 the all-gather of the step's exchange shard."""
        b = self.sets[k % self.ring]
        st = self.s_settle[k % len(self.s_settle)]
        rec = {"batch": b}
        with torch.cuda.stream(st):
            if self.free[k % self.ring] is not None:
                st.wait_event(self.free[k % self.ring])
            rec["t_stage0"] = torch.cuda.Event(enable_timing=True)
            rec["t_stage0"].record()
            b.stage(scene_id_base=scene_id_base)
            rec["ev0"] = torch.cuda.Event(enable_timing=True)
            rec["ev0"].record()
            b.settle()
            rec["ev1"] = torch.cuda.Event(enable_timing=True)
            rec["ev1"].record()
            b.place()
            rec["placed"] = torch.cuda.Event(enable_timing=True)
            rec["placed"].record()
        revs = []
        for sr in self.s_render_all:
            sr.wait_event(rec["placed"])
        for ci in range(b.n_render_chunks()):
            # (the chunks alternate between the render streams, every stream with its own scratch; a chunk and its target slot's
            # previous user share a stream because TARGET_RING is a multiple of the stream count)
            with torch.cuda.stream(self.s_render_all[ci % len(self.s_render_all)]):
                slot = ci % TARGET_RING
                if slot == 0:
                    for w in self.pending:
                        w.wait()      # a gather still reads slot 0's targets: order the re-render after it
                    self.pending = []
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                buf = b.render(ci, self.mask, ssao=self.ssao, buffers=self.buffers[slot] if slot < len(self.buffers) else None)
                e1.record()
                if slot >= len(self.buffers):
                    self.buffers.append(buf)
                revs.append((e0, e1))
                if self.streamer is not None:
                    # the rank's whole chunk goes through the exchange, piece by piece, while the next chunk renders
                    if self.gather_mode == "compact":
                        d = self.depth_stage.get(slot)
                        if d is None or d.shape[0] != buf.B:
                            d = self.depth_stage[slot] = torch.empty(buf.coord.shape[:3], dtype=torch.float32, device=buf.coord.device)
                        d.copy_(buf.coord[..., 3])            # depth = the 4th channel of coordDepth, made dense (4 B/px)
                        parts = (buf.rgb, d, buf.cls, buf.instance)
                    else:
                        parts = (buf.rgb, buf.coord, buf.cls, buf.instance, buf.normals)
                    self.pending = self.pending + self.streamer(list(parts))
                elif ci == 0 and self.gatherer is not None and self.gather_scenes > 0:
                    g = min(self.gather_scenes, buf.B)
                    _, works = self.gatherer([t[:g] for t in (buf.rgb, buf.coord, buf.cls, buf.instance, buf.normals)],
                                             async_op=True)
                    self.pending = works
        with torch.cuda.stream(self.s_render):
            for sr in self.s_render_all[1:]:
                ev = torch.cuda.Event()
                ev.record(sr)
                self.s_render.wait_event(ev)
            done = torch.cuda.Event()
            done.record()
            self.free[k % self.ring] = done
        rec["render_events"] = revs
        return rec

    def drain(self):
        for w in self.pending:
            w.wait()
        self.pending = []
        torch.cuda.synchronize()


def _abi_lib():
    from stillleben_amd import _abi

    return _abi.lib()


def _header_define(name):
    """Value of an integer #define of include/slhip.h (the caps the bench line quotes are the library's, not a copy)."""
    import re

    m = re.search(r"^#define %s\s+(\d+)" % name, open(os.path.join(ROOT, "include", "slhip.h")).read(), re.M)
    return int(m.group(1)) if m else None


def host_cpu_facts():
    """What the host lets this process run on: os.cpu_count() is the machine, the affinity mask and the cgroup CPU quota
    (v2 cpu.max, v1 cpu.cfs_quota_us / cpu.cfs_period_us) are the process -- usable_cpus is the smaller of the two."""
    import math

    out = {"os_cpu_count": os.cpu_count(), "affinity": len(os.sched_getaffinity(0)), "cgroup_quota_cpus": None}
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if q != "max":
            out["cgroup_quota_cpus"] = float(q) / float(per)
    except (OSError, ValueError):
        try:
            q = float(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            per = float(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                out["cgroup_quota_cpus"] = q / per
        except (OSError, ValueError):
            pass
    usable = out["affinity"]
    if out["cgroup_quota_cpus"]:
        usable = min(usable, max(1, int(math.floor(out["cgroup_quota_cpus"] + 1e-9))))
    out["usable_cpus"] = max(1, usable)
    return out


def cpu_baseline(table_meshes, scenes_per_thread, ssao, pair_contact_budget=0):
    """The oracle (scalar C restatement) timed on the host cores on a bounded sample of the same workload: every
    thread takes `scenes_per_thread` scenes through {tabletop set-up + 400-step settle + camera / light placement +
    640x480 render}.  Scenes are independent, so the threads mirror the reference's JobQueue workers; the stated figure runs as
    many of them as the host lets this process use at once (host_cpu_facts: affinity mask and cgroup quota, not os.cpu_count()),
    one thread is printed beside it.  The C calls release the GIL and share no state."""
    from concurrent.futures import ThreadPoolExecutor

    import oracle
    import stillleben_amd as sl
    from stillleben_amd import _abi
    from stillleben_amd import _settle_batch as SB
    from stillleben_amd._batch import HostPool

    host = host_cpu_facts()
    ncpu = host["usable_cpus"]
    flags = _abi.OUT_GT6 | _abi.RENDER_SHADOWS | (_abi.RENDER_SSAO if ssao else 0) | _abi.OUT_CAM_COORD
    pool, hulls = HostPool(), SB.HullPool()
    table = sl.AssetTable(table_meshes, mesh_pool=pool, hull_pool=hulls)   # host records only (untimed set-up, as on the GPU)
    pool_arrays = pool.arrays()
    hull_recs, hull_verts = hulls.arrays()
    W, H = RESOLUTION
    proto = sl.Scene(RESOLUTION)
    proto.set_camera_intrinsics(*INTRINSICS)
    sp = SB.default_params(tabletop=True, pair_contact_budget=pair_contact_budget)   # what the timed GPU path runs with

    def params(t):
        p = np.zeros((), dtype=_abi.SYNTH_PARAMS_DTYPE)
        p["n_scenes"], p["n_objects"], p["n_assets"] = scenes_per_thread, N_OBJECTS, len(table)
        p["flags"] = _abi.SYNTH_SAMPLE_DISTINCT | _abi.SYNTH_RANDOM_PBR | _abi.SYNTH_SHADOWS
        p["seed_lo"], p["scene_id_base"], p["render_chunk"] = 900000, t * scenes_per_thread, scenes_per_thread
        p["max_draws_per_scene"] = table.bound(table.n_draws, N_OBJECTS, True) + 1
        p["max_chunks_per_scene"] = table.bound(table.n_chunks, N_OBJECTS, True) + 1
        p["max_clip_verts_per_scene"] = table.bound(table.n_clip, N_OBJECTS, True) + 4
        p["plane_z"] = 0.04
        p["proj"] = proto._projection.reshape(-1)
        p["proj_inv"] = np.linalg.inv(proto._projection.astype(np.float64)).astype(np.float32).reshape(-1)
        p["plane_size"] = (3.0, 3.0)
        p["manual_exposure"] = -1.0
        p["light_color"][:3] = 300.0
        p["ambient"][:3] = 0.05
        return p

    def job(t):
        p = params(t)
        t0 = time.perf_counter()
        bodies, ss, objs, scs = oracle.synth_stage(p, table.records)
        oracle.settle(ss, bodies, hull_recs, hull_verts, sp)
        t1 = time.perf_counter()
        srec, drec, _ = oracle.synth_place(p, table.records, table.templates, bodies, objs, scs)
        md = int(p["max_draws_per_scene"])
        for s in range(scenes_per_thread):       # the oracle renderer takes tightly packed draw lists
            nd = int(srec[s]["draw_end"] - srec[s]["draw_begin"])
            rs, rd = srec[s:s + 1].copy(), drec[s * md:s * md + nd].copy()
            rs["draw_begin"], rs["draw_end"] = 0, nd
            rd["scene"] = 0
            oracle.render(pool_arrays, rs, rd, W, H, flags)
        return t1 - t0, time.perf_counter() - t1

    def leg(threads):
        with ThreadPoolExecutor(threads) as ex:
            w0 = time.perf_counter()
            times = list(ex.map(job, range(threads)))
            wall = time.perf_counter() - w0
        n = threads * scenes_per_thread
        return {"threads": threads, "scenes": n, "wall_s": wall, "scenes_per_s": n / wall,
                "settle_s_per_scene": sum(t[0] for t in times) / n, "render_s_per_scene": sum(t[1] for t in times) / n}

    # The figure beside the GPU line: as many threads as the host lets this process RUN -- the smaller of the affinity mask and the
    # cgroup's CPU quota (os.cpu_count() reports the machine: on the pool's boxes 256 hardware threads behind a quota of 16 CPUs;
    # round 5 ran 128 threads on those 16 and called them cores).  Beside it one thread, and the reference JobQueue's own count --
    # hardware_concurrency() / 2 workers (job_queue.cpp:35-40), which a quota does not lower -- when that is a different number.
    one, full = leg(1), leg(ncpu)
    jq_threads = max(1, (host["os_cpu_count"] or 1) // 2)
    jq = leg(jq_threads) if jq_threads != ncpu and jq_threads <= 8 * ncpu else None
    cpu_model = "?"
    try:
        with open("/proc/cpuinfo") as f:
            cpu_model = next((ln.split(":", 1)[1].strip() for ln in f if ln.startswith("model name")), "?")
    except OSError:
        pass
    notes = []
    for name, lg in (("usable-CPU", full), ("JobQueue-count", jq)):
        if lg is None:
            continue
        slow = (lg["settle_s_per_scene"] + lg["render_s_per_scene"]) / max(1e-9, one["settle_s_per_scene"] + one["render_s_per_scene"])
        lg["per_scene_time_vs_one_thread"] = slow
        if slow > 2.0:
            notes.append("%s leg: a scene takes %.1f x the single-thread time on %d threads -- %s"
                         % (name, slow, lg["threads"],
                            "more threads than usable CPUs (%d): they take turns" % ncpu if lg["threads"] > ncpu else
                            "the threads share memory bandwidth and last-level cache (the oracle's render walks 2048^2 shadow maps per scene)"))
    return {
        "value": full["scenes_per_s"], "unit": "scenes/s", "cores": ncpu, "kind": "port",
        "sample": "%d scenes on %d threads = the CPUs this process may use (host: %s; os.cpu_count %s, affinity %d, cgroup quota %s), %.1f s "
                  "wall; beside it: 1 thread %.2f scenes/s (set-up + settle %.3f s, placement + render %.3f s per scene)%s (oracle/, same C2 "
                  "workload and seeds scheme)"
                  % (full["scenes"], ncpu, cpu_model, host["os_cpu_count"], host["affinity"],
                     ("%.1f CPUs" % host["cgroup_quota_cpus"]) if host["cgroup_quota_cpus"] else "none", full["wall_s"], one["scenes_per_s"],
                     one["settle_s_per_scene"], one["render_s_per_scene"],
                     (", %d threads (the reference JobQueue's hardware_concurrency / 2) %.1f scenes/s" % (jq["threads"], jq["scenes_per_s"])) if jq else ""),
        "host": host, "notes": notes,
        "single_thread": one, "usable_cpu_threads": full, "job_queue_threads": jq,
    }


def respawn_ranks(args):
    """`python bench.py --gpus N` without a launcher: start the N ranks ourselves (one process per GPU,
    exactly what the driver's `python -m torch.distributed.run --nproc-per-node N` does) and pass
    rank 0's JSON line through."""
    import socket
    import subprocess

    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    return subprocess.call(cmd, env=env)


def run_other_config(args):
    """--config C1 | C3 | C4 | C5: the other BASELINE.json configurations (tools/bench_configs.py), one line in the bench schema.
    Single GPU; `steps` / `warmup` are what the configuration's own timing loop ran."""
    import importlib.util

    spec = importlib.util.spec_from_file_location("bench_configs", os.path.join(ROOT, "tools", "bench_configs.py"))
    bc = importlib.util.module_from_spec(spec)
    sys.modules["bench"] = sys.modules[__name__]      # bench_configs imports this module for make_scene / INTRINSICS
    spec.loader.exec_module(bc)
    bc.QUIET = True
    bc.sl.init_cuda(0)
    cfg = args.config
    if cfg == "C1":
        bc.c1(64)          # untimed: the process's one-off set-up (asset upload, scratch allocation, code load)
        r = bc.c1(4096)
        ms = r["settle_s_per_batch"] * 1e3 + r["render_ms_per_batch"]
        line = {"metric": "scenes/sec (settle + 320x240 instance-mask render), 4 cubes", "value": 4096 / (ms * 1e-3), "unit": "scenes/s",
                "steps": 1, "warmup": 0, "ms_per_step": ms,
                "config": {"workload": "C1: 4096 scenes of 4 cubes (tests/fixtures/cube.glb scaled to 0.2 m) through sl.Scene + "
                                       "physics.settle_batch (host glue included) + one 320x240 instance-mask render launch sequence"},
                "roofline": None}
    elif cfg == "C3":
        bc.c1(64)          # untimed: the process's one-off set-up
        bc.c3()            # untimed: one batch (the engine learns the list capacities C2 heaps need: no settle-again afterwards)
        r = bc.c3()
        ms = r["settle_s_incl_host_glue"] * 1e3 + r["render_ms"]
        line = {"metric": "scenes/sec (settle + 640x480 6-ch GT render), 512 C2 scenes through the per-object API", "value": 512 / (ms * 1e-3),
                "unit": "scenes/s", "steps": 1, "warmup": 1, "ms_per_step": ms,
                "config": {"workload": "C3: 512 C2 scenes built as sl.Scene objects, settled in one launch (host glue included) and rendered "
                                       "in 128-scene launch sequences (shadows + SSAO); the 8-rank form shards them 64 per GPU"},
                "roofline": r["roofline"]}
    elif cfg == "C4":
        B4 = args.batch or 1
        r = bc.c4(B4)
        line = {"metric": "renders/sec, stanford bunny x 50 (%d triangles per scene), 640x480, all 8 outputs" % (r["triangles"] // B4),
                "value": B4 * 1e3 / r["render_ms"], "unit": "scenes/s", "steps": 10 if B4 == 1 else 5, "warmup": 1, "ms_per_step": r["render_ms"],
                "config": {"workload": "C4: bunny x 50 raster stress, render only, %d scene(s) per launch sequence (--batch)" % B4,
                           "mtris_per_s": r["mtris_per_s"], "raster_share": r["raster_share"], "raster": r["raster"]},
                "roofline": r["roofline"]}
    else:
        r = bc.c5()
        ms = r["s_total_32_hypotheses_batch_api"] * 1e3
        line = {"metric": "pose hypotheses/sec (render + sl.diff backward), 64 objects x 32 hypotheses", "value": 32 / (ms * 1e-3),
                "unit": "hypotheses/s", "steps": 3, "warmup": 1, "ms_per_step": ms,
                "config": {"workload": "C5: 64 objects in one 640x480 view, 32 pose hypotheses rendered and backpropagated in one launch "
                                       "sequence each (sl.diff.backpropagate_gradient_to_poses_batch)"},
                "roofline": r["roofline"]}
    line.update({"n_gpus": 1, "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
                 "cpu_baseline": None, "detail": r})
    sys.stderr.flush()
    print(json.dumps(line), flush=True)


def main():
    args = parse()
    if args.config != "C2":
        if args.gpus != 1:
            sys.exit("bench.py: --config %s is a single-GPU configuration" % args.config)
        return run_other_config(args)
    if args.batch is None:
        args.batch = 32768
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        sys.exit(respawn_ranks(args))
    if int(os.environ.get("WORLD_SIZE", "1")) != args.gpus:
        sys.exit("bench.py: --gpus %d but the launcher started WORLD_SIZE=%s ranks" % (args.gpus, os.environ.get("WORLD_SIZE")))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    dist = None
    if world > 1:
        import torch.distributed as dist

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if os.environ.get("SLHIP_BENCH_ONE_DEVICE"):
            # developer hook: exercise the N > 1 code path on a 1-GPU box (all ranks on cuda:0, gloo
            # moves the bytes through the host) -- functional check only, never a measurement
            local_rank = 0
            torch.cuda.set_device(0)
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            torch.cuda.set_device(local_rank)
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))

    import stillleben_amd as sl
    from stillleben_amd import synthetic
    from stillleben_amd.parallel import BatchGatherer, SlhipComm

    t_prep = time.perf_counter()
    # (the native decomposition forks workers: before the device is touched)
    hull_sets = synthetic.native_hull_sets(seed=0) if args.hulls == "native" else "vhacd"
    sl.init_cuda(local_rank)
    meshes = synthetic.ycb_like_meshes(seed=0, hulls=hull_sets)
    table = sl.AssetTable(meshes)                 # once per process: the 21 classes' vertices, textures, hulls -> HBM
    if args.render_chunk is None:
        args.render_chunk = 512
    args.render_chunk = min(args.render_chunk, args.batch)
    pipe = Pipeline(sl, table, args.batch, args.render_chunk, not args.no_ssao, max(1, args.settle_streams), seed=20260929, rank=rank,
                    render_streams=min(max(1, args.render_streams), 2), pair_contact_budget=args.pair_budget)
    pipe.eng.L.slhip_timing_enable(1)
    pipe.eng.L.slhip_settle_timing_enable(1)
    table.device()
    pipe.eng.pool_abi()
    torch.cuda.synchronize()
    if rank == 0:
        print("[bench] asset table of %d classes + %d record sets of %d scenes ready in %.1f s (once per process; per-scene "
              "staging is inside the timed region)" % (len(table), pipe.ring, args.batch, time.perf_counter() - t_prep), file=sys.stderr)

    comm = None
    if dist is not None:
        # the exchange step: every rank receives the step's C3 shard (64 scenes) of every other rank -- all-gather per
        # dtype buffer in one RCCL group through the C-ABI (slhip_allgather_group), on its own stream
        pipe.gather_scenes = max(0, min(args.gather_scenes, args.render_chunk))
        if pipe.gather_scenes:
            if os.environ.get("SLHIP_BENCH_ONE_DEVICE") or os.environ.get("SLHIP_BENCH_GATHER") == "torch":
                pipe.gatherer = BatchGatherer(dist, world, depth=2)
            else:
                try:
                    comm = SlhipComm(rank, world, dist=dist)
                    ok = 1
                except Exception as e:      # e.g. librccl not loadable through the C-ABI on this box
                    print("[bench] rank %d: slhip_comm_* unavailable (%s); exchanging through torch.distributed" % (rank, e), file=sys.stderr)
                    comm, ok = None, 0
                flag = torch.tensor([ok], device="cuda", dtype=torch.int32)
                dist.all_reduce(flag, op=dist.ReduceOp.MIN)      # all ranks take the same path
                if int(flag.item()) == 0 and comm is not None:
                    comm.close()
                    comm = None
                pipe.gatherer = BatchGatherer(dist, world, depth=2, comm=comm) if comm is not None else BatchGatherer(dist, world, depth=2)
            if args.gather != "c3":
                from stillleben_amd.parallel import ChunkedGatherer

                pipe.gather_mode = args.gather
                pipe.streamer = ChunkedGatherer(pipe.gatherer, min(args.gather_piece, args.render_chunk))

    def scene_base(k):      # disjoint random streams per rank and step
        return (rank * 4096 + k) * args.batch

    def run(first, count):
        recs = [pipe.launch_step(k, scene_base(k)) for k in range(first, first + count)]
        pipe.drain()
        return recs

    run(0, args.warmup)
    pipe.eng.L.slhip_render_timings(C.byref((C.c_float * 8)()))   # drop warm-up phase timings
    pipe.eng.L.slhip_settle_timings(C.byref((C.c_float * 5)()), C.byref((C.c_uint32 * 5)()))
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    streamed0 = pipe.streamer.bytes_sent if pipe.streamer is not None else 0
    t0 = time.perf_counter()
    recs = run(args.warmup, args.steps)
    torch.cuda.synchronize()
    t_rank = time.perf_counter() - t0             # this rank's own time (before the closing barrier)
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    per_rank_s = [t_rank]
    if dist is not None:
        t = torch.tensor([elapsed], device="cuda", dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        tr = [torch.zeros(1, device="cuda", dtype=torch.float64) for _ in range(world)]
        dist.all_gather(tr, torch.tensor([t_rank], device="cuda", dtype=torch.float64))
        per_rank_s = [float(x.item()) for x in tr]
    streamed = (pipe.streamer.bytes_sent - streamed0) if pipe.streamer is not None else 0
    settle_errors = []
    for r in recs:
        try:
            r["batch"].check_settled()     # a scene the kernel refused (sizing hints) or a list that overflowed must not pass silently ...
        except RuntimeError as e:          # ... nor take the measurement with it: the line reports it (caps, settle_errors)
            settle_errors.append(str(e)[:400])

    # phase timings: the render events of all timed steps accumulate in the library
    ms_all = (C.c_float * 8)()
    pipe.eng.L.slhip_render_timings(C.byref(ms_all))
    st_ms, st_n = (C.c_float * 5)(), (C.c_uint32 * 5)()
    pipe.eng.L.slhip_settle_timings(C.byref(st_ms), C.byref(st_n))     # lockstep kernels: average launch duration inside the timed region
    pipe.eng.L.slhip_settle_timing_enable(0)
    settle_kernels = {n: {"avg_ms_per_launch": float(st_ms[i]), "timed_launches": int(st_n[i])}
                      for i, n in enumerate(("k_w_begin", "k_w_gjk_first+k_w_gjk_rest", "k_w_manifold", "k_w_finish", "k_w_solve"))}
    t_settle = float(np.mean([r["ev0"].elapsed_time(r["ev1"]) for r in recs]))
    t_stage = float(np.mean([r["t_stage0"].elapsed_time(r["ev0"]) for r in recs]))
    t_place = float(np.mean([r["ev1"].elapsed_time(r["placed"]) for r in recs]))
    t_render = float(np.mean([sum(a.elapsed_time(b) for a, b in r["render_events"]) for r in recs]))
    if rank == 0 and os.environ.get("SLHIP_BENCH_TRACE"):
        ref = recs[0]["t_stage0"]
        for k, r in enumerate(recs):
            print("[trace] step %d: stage %.1f, settle %.0f..%.0f ms, render %.0f..%.0f ms" % (
                k, ref.elapsed_time(r["t_stage0"]), ref.elapsed_time(r["ev0"]), ref.elapsed_time(r["ev1"]),
                ref.elapsed_time(r["render_events"][0][0]), ref.elapsed_time(r["render_events"][-1][1])), file=sys.stderr)
    phases = np.array(list(ms_all)) / max(1, args.steps)
    # inside the timed region the render overlaps the next steps' settles, which stretches its event-to-event times;
    # ONE extra non-overlapped render of the last step's chunks (outside the timed region) gives the isolated
    # per-kernel durations the render roofline is priced on
    b_last = recs[-1]["batch"]
    iso_wall = []
    with torch.cuda.stream(pipe.s_render):
        for ci in range(b_last.n_render_chunks()):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            b_last.render(ci, pipe.mask, ssao=pipe.ssao, buffers=pipe.buffers[ci % TARGET_RING])
            e1.record()
            iso_wall.append((e0, e1))
    torch.cuda.synchronize()
    ms_iso = (C.c_float * 8)()
    pipe.eng.L.slhip_render_timings(C.byref(ms_iso))
    iso = np.array(list(ms_iso))
    t_render_iso = float(sum(a.elapsed_time(b) for a, b in iso_wall))
    # ... and ONE settle launch alone on the idle GPU (same shape: the whole step's scenes)
    pipe.eng.L.slhip_settle_timing_enable(1)
    with torch.cuda.stream(pipe.s_settle[0]):
        b_last.stage(scene_id_base=scene_base(args.warmup + args.steps))
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        b_last.settle()
        e1.record()
    torch.cuda.synchronize()
    t_settle_alone = e0.elapsed_time(e1)
    pipe.eng.L.slhip_settle_timings(C.byref(st_ms), C.byref(st_n))     # the same kernels with the GPU to themselves
    pipe.eng.L.slhip_settle_timing_enable(0)
    for i, n in enumerate(settle_kernels):
        settle_kernels[n]["avg_ms_per_launch_alone"] = float(st_ms[i])
    # what the list capacities cost that settle (the reference's PhysX has no caps, scene.cpp:738-739): nothing may be dropped
    cp = b_last.settle_caps()
    scene_steps = args.batch * int(b_last.settle_params["frames"]) * int(b_last.settle_params["substeps"])
    caps = {"scenes": args.batch, "scene_steps": scene_steps,
            "scenes_that_dropped_contacts_or_pairs": cp["scenes_dropped"],
            "contact_drop_steps": cp["contact_drop_steps"], "pair_drop_steps": cp["pair_drop_steps"],
            "scenes_whose_contacts_left_the_lds_part": cp["scenes_spilled"], "share_of_scenes_spilled": cp["scenes_spilled"] / args.batch,
            "spill_step_rate": cp["spill_steps"] / scene_steps,
            "most_contacts_in_a_step": cp["max_contacts"], "most_hull_pairs_in_a_step": cp["max_hull_pairs"],
            "body_pair_drop_steps": cp["group_drop_steps"],
            "contacts_per_scene_step": cp["contact_sum"] / scene_steps,
            "pair_contact_budget": int(b_last.settle_params["pair_contact_budget"]),
            "reduced_steps": cp["reduced_steps"], "reduced_step_rate": cp["reduced_steps"] / scene_steps,
            "solver_wave_lds_bytes": int(_abi_lib().slhip_settle_solver_wave_lds()),
            "contact_capacity": int(b_last.settle_params["max_contacts_per_scene"]) or _header_define("SLHIP_DEFAULT_CONTACTS"),
            "hull_pair_capacity": int(b_last.settle_params["max_hull_pairs_per_scene"]) or _header_define("SLHIP_DEFAULT_HULL_PAIRS"),
            "note": "one settle of the step's scenes after the timed region (slhip_settle_caps): the solver takes every contact a step "
                    "offers -- from LDS as far as the scene's solver wave holds them (waves of 1 / 2 / 4 scenes by need), the rest swept from global memory (`spilled`: nothing lost); a "
                    "drop happens only beyond the capacities the scratch was sized with and must be zero.  pair_contact_budget: the "
                    "compound manifold reduction sl.SceneBatch asks for (NOT in the reference; 0 = every point, as in PhysX): a body pair "
                    "touching through more hull pairs keeps the deepest ones -- `reduced_steps` scene-steps had such a pair"}
    # the exchange step alone: the same shard gathered synchronously after the timed region (inside it the collective runs
    # beside the next chunks' render on its own stream)
    exchange = None
    if pipe.gatherer is not None and pipe.gather_scenes > 0 and pipe.buffers:
        buf = pipe.buffers[0]
        g = min(pipe.gather_scenes, buf.B) if pipe.streamer is None else min(pipe.streamer.piece, buf.B)
        shard = [t[:g] for t in (buf.rgb, buf.coord, buf.cls, buf.instance, buf.normals)]
        nbytes = sum(t.numel() * t.element_size() for t in shard)
        pipe.gatherer(shard)
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        reps = 3
        tx = time.perf_counter()
        for _ in range(reps):
            pipe.gatherer(shard)
        torch.cuda.synchronize()
        ms = (time.perf_counter() - tx) / reps * 1e3
        ranks_seen = dist.get_world_size() if dist is not None else None
        if comm is not None:      # what the communicator behind the C-ABI says about itself
            nr, rk = C.c_int(0), C.c_int(0)
            pipe.eng.L.slhip_comm_info(comm.handle, C.byref(nr), C.byref(rk))
            ranks_seen = int(nr.value)
        per_step = streamed / args.steps if pipe.streamer is not None else nbytes
        step_s = elapsed / args.steps
        exchange = {"mode": args.gather, "scenes": g, "bytes_per_rank": nbytes, "ms": ms, "GBps": (world - 1) * nbytes / (ms * 1e-3) / 1e9,
                    "of_scenes_per_step": args.batch, "ranks_seen": ranks_seen,
                    "scenes_exchanged_per_rank_and_step": args.batch if pipe.streamer is not None else g,
                    "bytes_sent_per_rank_and_step": per_step,
                    # xGMI is point to point: a rank's shard travels to each of its world - 1 peers over the link to that peer
                    "per_link_GBps_in_the_timed_region": per_step / step_s / 1e9,
                    "per_link_budget_GBps": 153.0, "link_share": per_step / step_s / 1e9 / 153.0,
                    "per_rank_scenes_per_s": [args.batch * args.steps / t for t in per_rank_s],
                    "backend": "slhip_allgather_group (RCCL behind the C-ABI)" if comm is not None else "torch.distributed",
                    "note": "ms / GBps: all-gather of one piece (6-channel GT of `scenes` scenes) to every rank, synchronous, %d repetitions "
                            "after the timed region, GBps = bytes received per rank / time.  Inside the timed region the exchange runs on its "
                            "own stream beside the render: c3 = one --gather-scenes piece per step; compact (rgb + depth + class + instance, "
                            "12 B/px) and full (40 B/px) stream EVERY rendered scene in --gather-piece pieces through a double-buffered "
                            "staging set; per_link_GBps = what each of a rank's peer links carries at the measured step rate (budget "
                            "~153 GB/s per xGMI link, MI355X_MICROARCH.md)" % reps}
    out = None
    if rank == 0:
        out = report(args, world, elapsed, table, pipe, t_settle, t_settle_alone, t_stage, t_place, t_render, t_render_iso,
                     phases, iso, settle_kernels, caps)
        out["caps"] = caps
        out["settle_errors"] = settle_errors      # must be empty: refused scenes / overflowed lists of the timed steps' record sets
        out["exchange"] = exchange
        if not args.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline(meshes, args.cpu_scenes, not args.no_ssao, args.pair_budget)
    if comm is not None:
        comm.close()
    if dist is not None:
        dist.destroy_process_group()
    if out is not None:      # rank 0's line is the LAST thing on stdout (library banners come earlier)
        sys.stderr.flush()
        if out.get("settle_errors"):
            # a settle that refused scenes or dropped contacts / pairs did less (and other) work than the metric names: the line stays
            # for diagnosis, marked invalid, and the run fails
            out["valid"] = False
            out["invalid_value"], out["value"] = out["value"], None
        print(json.dumps(out), flush=True)
        if out.get("settle_errors"):
            sys.exit("bench.py: the timed steps' settles reported errors (settle_errors): the line is INVALID")


def kernel_source_sha():
    """Fingerprint of the kernel sources (stillleben_amd/csrc/*.hip|*.inc|*.h, include/slhip.h) -- of their CODE: comments and
    white space do not count (a reworded comment does not make a counter stale).  tools/collect_counters.py files it with the
    counters it takes, and the roofline below only multiplies a live time by a counter taken from THESE sources."""
    import hashlib
    import re

    def code_of(text):
        # string literals first (a "//" inside one is not a comment), then block and line comments, then all white space
        pat = re.compile(r'"(?:\\.|[^"\\])*"|/\*.*?\*/|//[^\n]*', re.S)
        text = pat.sub(lambda m: m.group(0) if m.group(0).startswith('"') else " ", text)
        return re.sub(r"\s+", " ", text).strip()

    h = hashlib.sha256()
    d = os.path.join(ROOT, "stillleben_amd", "csrc")
    for f in sorted(os.listdir(d)) + ["../../include/slhip.h"]:
        if f.endswith((".hip", ".inc", ".h")):
            with open(os.path.join(d, f), "r", encoding="utf-8", errors="replace") as fh:
                h.update(f.encode() + b"\0" + code_of(fh.read()).encode("utf-8", "replace"))
    return h.hexdigest()[:16]


def load_counters():
    """SQ / HBM counters of the dominant kernels, collected by tools/collect_counters.py from separate rocprofv3 --pmc
    passes at the bench shape and committed under profiles/ (the latest round's file wins).  `stale` says that the kernel
    sources changed since the counters were taken: per-instruction figures are then not quoted."""
    rounds = sorted((d for d in os.listdir(os.path.join(ROOT, "profiles")) if d[:1] == "r" and d[1:].isdigit()), reverse=True)
    for rnd in rounds:
        path = os.path.join(ROOT, "profiles", rnd, "counters.json")
        if os.path.exists(path):
            with open(path) as f:
                c = json.load(f)
            c["source"] = "profiles/%s/counters.json" % rnd
            c["stale"] = c.get("kernel_source_sha") != kernel_source_sha()
            return c
    return {}


def report(args, world, elapsed, table, pipe, t_settle, t_settle_alone, t_stage, t_place, t_render, t_render_iso, phases, iso,
           settle_kernels, caps=None):
    W, H = RESOLUTION
    P = W * H
    total_scenes = args.batch * world * args.steps
    n_chunks = (args.batch + args.render_chunk - 1) // args.render_chunk
    names = ["shadow_raster", "shadow_large", "vis_raster", "vis_large", "shade", "ssao", "ssao_apply", "tonemap"]
    cnt = load_counters()
    ck = cnt.get("kernels", {})
    # ---- algorithmic bytes, SURVEY.md 8d (per scene) ----
    verts = float(np.mean(table.n_clip)) * N_OBJECTS + 4            # V_inst: vertices of the drawn instances
    tris = float(np.mean([sum(int(t["n_tris"]) for t in table.templates[int(r["draw_begin"]):int(r["draw_begin"] + r["draw_count"])])
                          for r in table.records])) * N_OBJECTS + 2    # T_inst
    S = 2048
    b_gt6 = verts * 68 + tris * 12 + P * 40                         # scan the geometry once, write the 6-channel GT
    b_shadow = verts * 12 + tris * 12 + S * S * 4                   # one active light
    b_post = P * (16 + 16 + 4 + 16 + 4)                             # SSAO + (blur and tone map in one pass)
    b_scene = b_gt6 + (0 if os.environ.get("SLHIP_BENCH_NO_SHADOWS") else b_shadow) + (b_post if not args.no_ssao else P * 20)
    # per-kernel byte models: what each kernel must move when every byte is touched once
    per_kernel_bytes = {
        "k_shade": P * (8 + 40 + 16 + 16 + 4) + verts * 40,         # key in; GT6 + cam coords + HDR + z plane out; vertex attributes once
        "k_ssao": P * (16 + 16 + 4 + 4),
        "k_ssao_apply": P * (16 + 4 + 4 + 4),                       # HDR, AO and z in; tone-mapped rgb8 out (with SSAO the tone map is
        "k_tonemap": P * (16 + 4),                                  # part of k_ssao_apply and k_tonemap is not launched)
        "k_raster": tris * (12 + 48) + P * 8,
        "k_shadow_raster": tris * (12 + 48) + S * S * 4,
    }
    iso_by_kernel = {"k_shade": iso[4], "k_ssao": iso[5], "k_ssao_apply": iso[6], "k_tonemap": iso[7],
                     "k_raster": iso[2] + iso[3], "k_shadow_raster": iso[0] + iso[1]}
    per_kernel = {}
    for k, bts in per_kernel_bytes.items():
        ms = float(iso_by_kernel[k]) / n_chunks                      # one launch = one render chunk
        if ms > 0.01:
            traffic = ck[k]["hbm_bytes_per_scene"] * args.render_chunk if k in ck else None
            # a model that exceeds what the counters saw move is not a roofline (positions served by the L2 are not HBM traffic):
            # the kernel is priced on min(model, counter bytes)
            priced = bts * args.render_chunk if traffic is None else min(bts * args.render_chunk, traffic)
            # names as in the contract: `algorithmic` = the byte model, `traffic` = what the counters saw; `priced` = the smaller
            per_kernel[k] = {"algorithmic_bytes_per_launch": bts * args.render_chunk, "traffic": traffic,
                             "priced_bytes_per_launch": priced, "priced_on": "traffic" if priced != bts * args.render_chunk else "algorithmic",
                             "ms_per_launch": ms, "achieved_GBps": priced / (ms * 1e-3) / 1e9,
                             "frac": priced / (ms * 1e-3) / 8e12}
    ms_seq = t_render_iso / n_chunks
    roof_render = {
        "bound": "hbm", "kernel": "slhip_render launch sequence (%d scenes: vertex transform, shadow pass, visibility, shade, SSAO, tone map)" % args.render_chunk,
        "achieved": b_scene * args.render_chunk / (ms_seq * 1e-3) / 1e9, "peak": 8000.0, "unit": "GB/s",
        "algorithmic_bytes_per_scene": b_scene, "algorithmic_bytes_per_launch": b_scene * args.render_chunk,
        "ms_per_launch": ms_seq,
        "traffic": sum(v["hbm_bytes_per_scene"] for k, v in ck.items() if k not in ("k_settle", "k_clear_shadow") and not k.startswith(("k_synth", "k_w_"))) * args.render_chunk if ck else None,
        "measured": "HIP events on the render stream around one non-overlapped pass over the last step's chunks",
        "byte_model": "SURVEY.md 8d: V_inst*68 + T_inst*12 + P*40 (GT6) + V_inst*12 + T_inst*12 + 2048^2*4 (one shadow light) + P*56 (SSAO; blur + tone map in one pass)",
        "per_kernel": per_kernel,
    }
    roof_render["frac"] = roof_render["achieved"] / roof_render["peak"]
    # ---- the time-dominant kernel of the whole path: the solver of the settle ----
    lockstep = True
    hulls_per_scene = float(np.mean(table.n_hulls)) * N_OBJECTS
    hverts_per_scene = float(np.mean(table.n_hull_verts)) * N_OBJECTS
    if lockstep:
        kname = "k_w_solve"
        sq = ck.get(kname, {})
        # mean contacts the solver took per scene and step: counted by the kernels themselves (slhip_settle_caps counts[9])
        contacts = caps["contacts_per_scene_step"] if caps else sq.get("contacts_per_scene_step", 45.0)
        # per launch (= one step of every scene): the scene's working bodies (152 B) and prepared contacts (72 B) in, group /
        # colour lists in, the body records (304 B) out -- DESIGN.md section 4
        per_scene = N_OBJECTS * 152 + contacts * 72 + 1500 + N_OBJECTS * 304
        ms_launch = settle_kernels[kname]["avg_ms_per_launch"]
        launches_per_settle = 400
        measured = ("HIP events on the settle streams around every k_w_solve launch of every 8th step of the timed region "
                    "(slhip_settle_timings); %d launches timed" % settle_kernels[kname]["timed_launches"])
    else:
        kname = "k_settle"
        sq = ck.get(kname, {})
        per_scene = N_OBJECTS * 304 * 2 + hulls_per_scene * 64 + hverts_per_scene * 16
        ms_launch = t_settle
        launches_per_settle = 1
        measured = "HIP events on the launch's stream around every slhip_settle of the timed region"
    alg = per_scene * args.batch
    stale = bool(cnt.get("stale", True))
    valu_per_launch = None if stale else sq.get("valu_insts_per_scene_launch")
    # The dominant kernel is bound by VALU issue, not by HBM: `achieved` = wave64 VALU instructions issued per second (SQ_INSTS_VALU
    # of the counters file x scenes per launch / the launch duration measured live), `peak` = one instruction per 2.3 cycles and SIMD
    # -- the rate tools/probes/pk_probe.hip measured for streams of independent v_fma_f32 with several waves per SIMD (a single wave
    # issues every 4.5 cycles).  The HBM figures the schema names stay beside it under "hbm".
    issue_peak = 1024 * 2.4e9 / 2.0 / 1e9     # the guide's figure: a wave64 v_fma_f32 occupies its SIMD for 2 cycles (MI355X_MICROARCH.md)
    hbm = {"achieved": alg / (ms_launch * 1e-3) / 1e9 if ms_launch > 0 else None, "peak": 8000.0, "unit": "GB/s",
           "algorithmic_bytes_per_launch": alg, "traffic": sq["hbm_bytes_per_scene"] * args.batch if "hbm_bytes_per_scene" in sq else None}
    if hbm["achieved"] is not None:
        hbm["frac"] = hbm["achieved"] / hbm["peak"]
    issued = valu_per_launch * args.batch / (ms_launch * 1e-3) / 1e9 if valu_per_launch and ms_launch > 0 else None
    roofline = {
        "bound": "valu-issue", "kernel": kname, "achieved": issued, "peak": issue_peak, "unit": "G wave-instr/s",
        "frac": issued / issue_peak if issued else None,
        "traffic": hbm["traffic"], "hbm": hbm,
        "ms_per_launch": ms_launch, "launches_per_settle": launches_per_settle,
        "measured": measured,
        "note": "Gauss-Seidel sweeps over chains of dependent contact rows: few lanes of a wave are active (active_lanes), so the "
                "kernel is bound by instruction issue and by the latency of a wave's dependent chain, not by bytes.  ms_per_launch is "
                "taken in the timed region, where the kernel shares the GPU with the render streams (the event pairs also bracket its "
                "wait for free CU slots); *_alone: one settle with the GPU to itself after the timed region",
        "valu_insts_per_scene_launch": valu_per_launch,
        "peak_source": "1024 SIMDs x 2.4 GHz / 2 cycles per wave64 VALU instruction (MI355X_MICROARCH.md); tools/probes/pk_probe.hip "
                       "measured 2.3 cycles for streams of independent v_fma_f32 at 16 waves per CU (peak 1068.5: the fraction would read 15 % higher)",
        "counters_stale": stale, "counters_kernel_source_sha": cnt.get("kernel_source_sha"), "kernel_source_sha": kernel_source_sha(),
        "active_lanes": sq.get("active_lanes"),
        # the issue rate rewards instructions, not work: beside it the share of the chip's fp32 LANE issue (x active lanes / 64) and
        # SURVEY 8d's FLOP model of the sweeps (contacts x 8 sweeps x ~120 flop) against the dense fp32 vector peak
        "lane_issue_frac": (issued / issue_peak * sq["active_lanes"] / 64.0) if issued and sq.get("active_lanes") else None,
        "flop_model": {"flops_per_launch": contacts * 8 * 120 * args.batch if lockstep else None,
                       "achieved_TFLOPs": (contacts * 8 * 120 * args.batch / (ms_launch * 1e-3) / 1e12) if lockstep and ms_launch > 0 else None,
                       "peak_TFLOPs": 157.3,
                       "frac": (contacts * 8 * 120 * args.batch / (ms_launch * 1e-3) / 1e12 / 157.3) if lockstep and ms_launch > 0 else None,
                       "model": "SURVEY.md 8d: contacts per scene-step (counted by the kernels) x (4 + 4) sweeps x ~120 flop per row set"},
        "ms_per_launch_alone": settle_kernels[kname].get("avg_ms_per_launch_alone") if lockstep else None,
        "counters_source": cnt.get("source"),
        "settle_kernels_ms_per_launch": {k: v["avg_ms_per_launch"] for k, v in settle_kernels.items()} if lockstep else None,
        "settle_kernels_ms_per_launch_alone": {k: v.get("avg_ms_per_launch_alone") for k, v in settle_kernels.items()} if lockstep else None,
        "settle_ms_per_batch": t_settle, "settle_ms_per_batch_alone": t_settle_alone,
        "steps_scenes_per_s": args.batch * 400 / (t_settle * 1e-3),
    }
    ms_alone = roofline.get("ms_per_launch_alone")
    if ms_alone:
        hbm["achieved_alone"] = alg / (ms_alone * 1e-3) / 1e9
        hbm["frac_alone"] = hbm["achieved_alone"] / hbm["peak"]
        if valu_per_launch:
            roofline["achieved_alone"] = valu_per_launch * args.batch / (ms_alone * 1e-3) / 1e9
            roofline["frac_alone"] = roofline["achieved_alone"] / issue_peak
    k_dom = int(np.argmax(phases)) if phases.sum() > 0 else 4
    return {
        "metric": "scenes/sec (settle + 640x480 6-ch GT render), 20-obj YCB-like",
        "value": total_scenes / elapsed, "unit": "scenes/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f32", "data": "synthetic",
        "config": {
            "workload": "C2: 20 of 21 procedural YCB-like objects (8k verts/16k tris, 1024^2 texture each) per scene; per step and GPU "
                        "%d scenes are staged (random stack), settled (100 frames x 4 substeps), framed (camera, light, shadow "
                        "matrix) and rendered at 640x480 (rgb, coord+depth, class, instance, normals; shadows on, SSAO %s) -- "
                        "all four stages inside the timed region" % (args.batch, "off" if args.no_ssao else "on"),
            "scenes_per_gpu_per_step": args.batch, "render_chunk": args.render_chunk, "resolution": list(RESOLUTION),
            "objects": N_OBJECTS, "settle_streams": len(pipe.s_settle), "render_streams": len(pipe.s_render_all),
            "collision_hulls": ("the reference's V-HACD on this geometry (shipped fixture)" if args.hulls == "vhacd" else
                                "the in-tree decomposition (acd.py + slhip_host_convex_hull)") + ": %d hulls in the 21 classes" % int(sum(table.n_hulls)),
            "pair_contact_budget": caps["pair_contact_budget"] if caps else None,   # (slhip.h; 0 = every contact point, as in PhysX)
            "parallelism": ("scenes sharded by rank, no data-path collective; exchange: RCCL all-gather of a %d-scene C3 shard per "
                            "rank and step (%.0f MB per rank) -- %d of the %d scenes a rank renders per step are exchanged"
                            % (pipe.gather_scenes, pipe.gather_scenes * P * 40 / 1e6, pipe.gather_scenes, args.batch)) if world > 1 else "1 GPU",
        },
        "roofline": roofline,
        "roofline_render": roof_render,
        "breakdown_ms": {
            "stage": t_stage, "settle": t_settle, "settle_alone": t_settle_alone, "place": t_place,
            "render_total_overlapped": t_render, "render_total_isolated": t_render_iso, "render_chunks": n_chunks,
            **{n: float(v) for n, v in zip(names, phases)},
        },
        "breakdown_isolated_ms": {n: float(v) for n, v in zip(names, iso)},
        "dominant_render_phase": names[k_dom],
        # what the rank holds in HBM at its peak (torch's allocator: records, settle scratch, render scratch, the ring of ground truth)
        "hbm_footprint_GB": {"peak_allocated": torch.cuda.max_memory_allocated() / 1e9,
                             "device_total": torch.cuda.get_device_properties(torch.cuda.current_device()).total_memory / 1e9},
    }


if __name__ == "__main__":
    main()
