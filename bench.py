#!/usr/bin/env python
"""Headline benchmark: scenes/sec for {tabletop settle of 20 YCB-like objects + 640x480
6-channel ground-truth render} (BASELINE.json metric, config C2), one process per GPU.

A "step" = one batch of `--batch` freshly seeded scenes per GPU going through the whole hot
path: slhip_settle (400 physics steps per scene) -> camera / light / draw-list assembly ->
slhip_render (shadow pass, visibility raster, deferred shade, SSAO, tone map) -> for N > 1 an
RCCL all-gather of the rendered batches.  Inputs (mesh pool, hull pool, the initial body
states of every timed batch) are resident in HBM before the timed region starts.

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
VALU_INSTS_PER_SCENE = 52762516188 / 2048   # rocprofv3 --pmc SQ_INSTS_VALU, tools/profile_settle.py 2048 100 (profiles/r01/settle_sq_counters_v32.txt)
N_OBJECTS = 20


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=4)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--batch", type=int, default=4096,
                    help="scenes per GPU per step (one settle launch: two rounds of the 2048 resident scenes = 256 CUs x 8; the second round back-fills the tail of the first)")
    ap.add_argument("--render-chunk", type=int, default=None,
                    help="scenes per render launch sequence (fewer, larger sequences: every kernel boundary is a chance for "
                         "queued settle workgroups to take the freed SIMDs); default 512, 256 when the rendered chunks are "
                         "all-gathered (N > 1: the [world, chunk, ...] staging ring grows with it)")
    ap.add_argument("--settle-streams", type=int, default=3,
                    help="settle launches kept in flight: scenes settle in very different times, and a second "
                         "launch on its own stream back-fills the CUs the tail of the first one leaves idle")
    ap.add_argument("--settle-cus", type=int, default=0,
                    help="compute units reserved for the settle streams; the render stream gets the rest (0 = share all CUs)")
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
    def __init__(self, sl, batch, ssao, settle_cus=0, settle_streams=1):
        from stillleben_amd import _abi, physics
        from stillleben_amd._context import engine

        self.sl, self._abi, self.physics = sl, _abi, physics
        self.eng = engine()
        self.se = physics.settle_engine()
        self.batch = batch
        self.ssao = ssao
        self.mask = _abi.OUT_GT6
        self.buffers = []
        if settle_cus > 0:
            # the two halves get disjoint CU ranges: settle workgroups hold their CU slots for ~100 ms, and
            # short render workgroups sharing those CUs fragment both
            from stillleben_amd.parallel import cu_partition_streams

            self.s_settle, self.s_render = cu_partition_streams(settle_cus, settle_streams, self.eng.device)
        else:
            self.s_settle = [torch.cuda.Stream(device=self.eng.device)]
            self.s_render = torch.cuda.Stream(device=self.eng.device)
        self.render_chunk = 128
        self.gatherer = None      # N > 1: BatchGatherer, one asynchronous RCCL all-gather per rendered chunk
        self.pending = {}         # chunk slot -> outstanding collectives reading that slot's render buffers
        self.t_step_host = []
        self.t_settle = []
        self.t_host = []
        self.t_render = []
        self.phase_ms = []

    def prepare(self, scenes, seed):
        """Host-side set-up of one batch BEFORE the timed region: initial stacks, static draw
        records, the random draws of the batch, and the upload of the initial body states."""
        import math

        from stillleben_amd import _fast_batch as FB
        from stillleben_amd import _settle_batch as SB

        planes = [(self.physics.prepare_tabletop(s), self.physics.PLANE_HALF_Z) for s in scenes]
        srec, bodies = SB.build_settle_batch(scenes, self.se.pool, planes)
        chunks = []
        for c0 in range(0, len(scenes), self.render_chunk):
            chunks.append(FB.prepare(scenes[c0:c0 + self.render_chunk], self.eng.pool))
        rng = np.random.default_rng(seed)
        item = {
            "params": SB.sizing_hints(self.params, srec, bodies, self.se.pool.arrays()[0]),
            "scenes": scenes, "srec": srec, "chunks": chunks,
            "d_bodies": self.eng.upload_records(bodies),
            # the random draws of chooseRandomCameraPose / chooseRandomLightDirection
            "az": rng.uniform(-math.pi, math.pi, len(scenes)).astype(np.float32),
            "el": rng.uniform(math.radians(30.0), math.radians(60.0), len(scenes)).astype(np.float32),
            "nrm": rng.standard_normal((len(scenes), 3)).astype(np.float32),
            "plane_pose": np.stack([s._background_plane_pose for s in scenes]),
            "obj_off": np.cumsum([0] + [len(s._objects) for s in scenes]),
        }
        self.se.hulls_dev()
        self.eng.pool_abi()
        return item

    def launch_settle(self, item, slot=0):
        """Asynchronous: slhip_settle on one of the settle streams, then the 288 B/object read-back
        into pinned host memory; returns immediately."""
        while len(self.s_settle) <= slot:
            self.s_settle.append(torch.cuda.Stream(device=self.eng.device))
        with torch.cuda.stream(self.s_settle[slot]):
            item["ev0"] = torch.cuda.Event(enable_timing=True)
            item["ev1"] = torch.cuda.Event(enable_timing=True)
            item["ev0"].record()
            d_bodies = self.se.run_device(item["srec"], None, item["params"], d_bodies=item["d_bodies"])
            item["ev1"].record()
            if "h_bodies" not in item:
                item["h_bodies"] = torch.empty(d_bodies.shape, dtype=d_bodies.dtype, pin_memory=True)
            item["h_bodies"].copy_(d_bodies, non_blocking=True)
            item["ev_copy"] = torch.cuda.Event()
            item["ev_copy"].record()

    def finish(self, item, timed=True):
        """Host assembly (camera, light, draw records) + slhip_render of every chunk on the
        render stream; overlaps with the settle of the next batch."""
        from stillleben_amd import _fast_batch as FB
        from stillleben_amd import _settle_batch as SB

        W, H = RESOLUTION
        item["ev_copy"].synchronize()
        t0 = time.perf_counter()
        bodies = np.frombuffer(item["h_bodies"].numpy().tobytes(), dtype=SB.BODY_DTYPE)
        poses = bodies["pose"].reshape(-1, 4, 4)
        outs, revs = [], []
        with torch.cuda.stream(self.s_render):
            for ci, t in enumerate(item["chunks"]):
                s0 = ci * self.render_chunk
                s1 = s0 + t.n_scenes
                o0, o1 = item["obj_off"][s0], item["obj_off"][s1]
                for w in self.pending.pop(ci, []):
                    w.wait()      # stream-level: this slot's previous gather must finish before it is re-rendered
                th0 = time.perf_counter()
                cam = FB.camera_poses(t, poses[o0:o1], item["az"][s0:s1], item["el"][s0:s1])
                ld = FB.light_directions(cam, item["nrm"][s0:s1])
                srec, drec = FB.update(t, poses[o0:o1], cam, ld, item["plane_pose"][s0:s1])
                th1 = time.perf_counter()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                buf = self.eng.render_records(srec, drec, t.crec, W, H, self.mask, ssao=self.ssao, shadows=not os.environ.get("SLHIP_BENCH_NO_SHADOWS"),
                                              buffers=self.buffers[ci] if ci < len(self.buffers) else None)
                e1.record()
                if self.gatherer is not None:
                    # ordered after the chunk's kernels, runs on RCCL's stream while the next chunks render
                    _, works = self.gatherer([t for t in (buf.rgb, buf.coord, buf.cls, buf.instance, buf.normals)
                                              if t is not None], async_op=True)
                    self.pending[ci] = works
                if ci >= len(self.buffers):
                    self.buffers.append(buf)
                outs.append(buf)
                revs.append((e0, e1))
                if timed:
                    self.t_host.append((th1 - th0) * 1e3)
        item["render_events"] = revs
        item["t_post"] = (time.perf_counter() - t0) * 1e3
        return outs


def cpu_baseline(sl, meshes, scenes_per_thread, ssao, max_threads=32):
    """The oracle (scalar C restatement) timed on the host cores on a bounded sample of the same
    workload: every thread takes `scenes_per_thread` scenes through {400-step settle + 640x480
    render}.  Scenes are independent, so the threads mirror the reference's JobQueue workers
    (job_queue.cpp:35-40); the C calls release the GIL and share no state."""
    from concurrent.futures import ThreadPoolExecutor

    import oracle
    from stillleben_amd import _abi, physics
    from stillleben_amd import _settle_batch as SB
    from stillleben_amd._batch import HostPool, build_batch

    threads = max(1, min(os.cpu_count() or 1, max_threads))
    flags = _abi.OUT_GT6 | _abi.RENDER_SHADOWS | (_abi.RENDER_SSAO if ssao else 0) | _abi.OUT_CAM_COORD
    prm = SB.default_params(tabletop=True)
    jobs = []
    for t in range(threads):   # untimed set-up, as for the GPU path
        scenes = [make_scene(sl, meshes, 900000 + t * scenes_per_thread + i) for i in range(scenes_per_thread)]
        pool_h = SB.HullPool()
        planes = [(physics.prepare_tabletop(s), physics.PLANE_HALF_Z) for s in scenes]
        srec, bodies = SB.build_settle_batch(scenes, pool_h, planes)
        jobs.append((scenes, srec, bodies, pool_h.arrays()))

    def settle(job):
        scenes, srec, bodies, (hulls, verts) = job
        t0 = time.perf_counter()
        oracle.settle(srec, bodies, hulls, verts, prm)
        return time.perf_counter() - t0

    def render(job):
        pool, rs, rd = job
        t0 = time.perf_counter()
        oracle.render(pool.arrays(), rs, rd, RESOLUTION[0], RESOLUTION[1], flags)
        return time.perf_counter() - t0

    with ThreadPoolExecutor(threads) as ex:
        w0 = time.perf_counter()
        t_settle = list(ex.map(settle, jobs))
        w1 = time.perf_counter()
        rjobs = []
        for scenes, srec, bodies, _ in jobs:   # camera / light placement + draw records: untimed host glue
            SB.write_back(scenes, bodies)
            for s in scenes:
                s.choose_random_camera_pose()
                s.choose_random_light_direction()
            pool = HostPool()
            rs, rd, _c = build_batch(scenes, pool, with_shadows=True)
            rjobs.append((pool, rs, rd))
        w2 = time.perf_counter()
        t_render = list(ex.map(render, rjobs))
        w3 = time.perf_counter()
    n = threads * scenes_per_thread
    wall = (w1 - w0) + (w3 - w2)
    cpu_model = "?"
    try:
        with open("/proc/cpuinfo") as f:
            cpu_model = next((ln.split(":", 1)[1].strip() for ln in f if ln.startswith("model name")), "?")
    except OSError:
        pass
    return {
        "value": n / wall, "unit": "scenes/s", "cores": threads, "kind": "port",
        "sample": "%d scenes on %d threads (host: %d x %s): settle %.2f s + render %.2f s wall; per scene on one "
                  "thread: settle %.3f s, render %.3f s, i.e. %.2f scenes/s single-threaded (oracle/, same C2 workload)"
                  % (n, threads, os.cpu_count() or 0, cpu_model, w1 - w0, w3 - w2, sum(t_settle) / n, sum(t_render) / n,
                     n / (sum(t_settle) + sum(t_render))),
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


def main():
    args = parse()
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
    from stillleben_amd import _settle_batch as SB
    from stillleben_amd import synthetic

    sl.init_cuda(local_rank)
    meshes = synthetic.ycb_like_meshes(seed=0)
    pipe = Pipeline(sl, args.batch, not args.no_ssao, args.settle_cus, max(1, args.settle_streams))
    pipe.params = SB.default_params(tabletop=True)
    if args.render_chunk is None:
        args.render_chunk = 512 if world == 1 else 256
    pipe.render_chunk = args.render_chunk
    pipe.eng.L.slhip_timing_enable(1)

    n_items = args.warmup + args.steps
    items = []
    t_prep = time.perf_counter()
    for k in range(n_items):
        base = (rank * n_items + k) * args.batch
        items.append(pipe.prepare([make_scene(sl, meshes, base + i) for i in range(args.batch)], seed=base))
    torch.cuda.synchronize()
    if rank == 0:
        print("[bench] prepared %d batches of %d scenes in %.1f s (untimed set-up)" % (n_items, args.batch, time.perf_counter() - t_prep),
              file=sys.stderr)

    from stillleben_amd.parallel import BatchGatherer

    if dist is not None:
        # every rank receives every rendered chunk: RCCL all-gather per dtype buffer into a ring of
        # [world, chunk, ...] staging buffers, overlapped with the rendering of the following chunks
        pipe.gatherer = BatchGatherer(dist, world, depth=2)

    def run(seq, timed):
        """Software pipeline over a sequence of batches: while the GPU settles batch k+1 (settle
        stream) the host assembles and the GPU renders batch k (render stream).  Every batch's
        settle AND render (and gather) complete inside the call."""
        if not seq:
            return
        ahead = max(1, args.settle_streams)
        for k in range(min(ahead, len(seq))):
            pipe.launch_settle(seq[k], k % ahead)
        for k in range(len(seq)):
            if k + ahead < len(seq):
                pipe.launch_settle(seq[k + ahead], (k + ahead) % ahead)
            pipe.finish(seq[k], timed)
        for works in pipe.pending.values():
            for w in works:
                w.wait()
        pipe.pending.clear()
        torch.cuda.synchronize()

    run(items[:args.warmup], False)
    pipe.eng.L.slhip_render_timings(C.byref((C.c_float * 8)()))   # drop warm-up phase timings
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    run(items[args.warmup:], True)
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([elapsed], device="cuda", dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # phase timings: the render events of all timed batches accumulate in the library
    ms_all = (C.c_float * 8)()
    pipe.eng.L.slhip_render_timings(C.byref(ms_all))
    for it in items[args.warmup:]:
        pipe.t_render.append(sum(a.elapsed_time(b) for a, b in it["render_events"]))
        pipe.t_settle.append(it["ev0"].elapsed_time(it["ev1"]))
        pipe.t_step_host.append(it["t_post"])
    if rank == 0 and os.environ.get("SLHIP_BENCH_TRACE"):
        ref = items[args.warmup]["ev0"]
        for k, it in enumerate(items[args.warmup:]):
            print("[trace] step %d: settle %.0f..%.0f ms, render %.0f..%.0f ms" % (
                k, ref.elapsed_time(it["ev0"]), ref.elapsed_time(it["ev1"]),
                ref.elapsed_time(it["render_events"][0][0]), ref.elapsed_time(it["render_events"][-1][1])), file=sys.stderr)
    pipe.phase_ms.append(np.array(list(ms_all)) / max(1, args.steps))
    # the render kernels overlap with the next batch's settle inside the timed region, which
    # stretches their event-to-event times; one extra NON-overlapped render pass of the last
    # batch (outside the timed region) gives the per-kernel durations the roofline is priced on
    pipe.finish(items[-1], timed=False)
    torch.cuda.synchronize()
    ms_iso = (C.c_float * 8)()
    pipe.eng.L.slhip_render_timings(C.byref(ms_iso))
    iso = np.array(list(ms_iso))
    if rank == 0:
        total_scenes = args.batch * world * args.steps
        value = total_scenes / elapsed
        ms_step = elapsed / args.steps * 1e3
        # ---- roofline of the dominant kernels ----
        W, H = RESOLUTION
        P = W * H
        t_settle = float(np.mean(pipe.t_settle))
        t_render = float(np.mean(pipe.t_render))
        phases = np.mean(np.array(pipe.phase_ms), axis=0) if pipe.phase_ms else np.zeros(8)
        names = ["shadow_raster", "shadow_large", "vis_raster", "vis_large", "shade", "ssao", "ssao_apply", "tonemap"]
        # algorithmic bytes of the deferred shade pass per launch (DESIGN.md "roofline"):
        #   read the 8 B visibility key, write the selected targets + HDR colour, per pixel,
        #   plus the winning triangle's 3 vertices (pos 16 B, normal 16 B, uv 8 B) + 12 B indices
        out_bytes = 16 + 2 + 2 + 16 + 16 + 16  # coord, class, instance, normals, cam_coord(SSAO input), hdr
        n_chunks = (args.batch + args.render_chunk - 1) // args.render_chunk
        shade_bytes = args.batch * P * (8 + out_bytes + 3 * 40 + 12)  # all chunks of one step
        k_dom = int(np.argmax(phases)) if phases.sum() > 0 else 4
        settle_dominant = t_settle > t_render
        roof_render = {
            "bound": "hbm", "kernel": "k_shade",
            "achieved": shade_bytes / (iso[4] * 1e-3) / 1e9 if iso[4] > 0 else None,
            "peak": 8000.0, "unit": "GB/s", "traffic": None,
            "measured": "HIP events, non-overlapped render pass of the last timed batch (inside the timed region "
                        "the render overlaps the next batch's settle: see breakdown_ms)",
            "achieved_overlapped": shade_bytes / (phases[4] * 1e-3) / 1e9 if phases[4] > 0 else None,
        }
        if roof_render["achieved"]:
            roof_render["frac"] = roof_render["achieved"] / roof_render["peak"]
        # HBM traffic from the PMC counters (FETCH_SIZE / WRITE_SIZE, separate rocprofv3 passes on
        # the same workload, committed under profiles/): bytes per scene x scenes per launch
        pmc = os.path.join(ROOT, "profiles", "r01", "pmc_traffic.json")
        pmc_k = {}
        if os.path.exists(pmc):
            with open(pmc) as f:
                pmc_k = json.load(f)["kernels"]
        if "k_shade" in pmc_k:
            roof_render["traffic"] = pmc_k["k_shade"]["hbm_bytes_per_scene"] * args.render_chunk
            roof_render["traffic_source"] = "profiles/r01/pmc_traffic.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE)"
            roof_render["algorithmic_bytes_per_launch"] = shade_bytes / n_chunks
        # settle: hull vertices + body state are read once and written once per scene (HBM),
        # everything else lives in LDS/L2: an HBM fraction is reported for completeness only
        # per scene: body records in + out (288 B each), hull records (64 B) and hull vertices (16 B) once
        hulls_per_scene, verts_per_scene = 68, 1456   # C2 maxima (sizing hints of the batches)
        settle_bytes = args.batch * (N_OBJECTS * 288 * 2 + hulls_per_scene * 64 + verts_per_scene * 16)
        roof_settle = {
            "bound": "hbm", "kernel": "k_settle", "achieved": settle_bytes / (t_settle * 1e-3) / 1e9, "peak": 8000.0,
            "unit": "GB/s", "traffic": pmc_k["k_settle"]["hbm_bytes_per_scene"] * args.batch if "k_settle" in pmc_k else None,
            "algorithmic_bytes_per_launch": settle_bytes,
            "measured": "HIP events around each slhip_settle launch; up to --settle-streams launches share the GPU, "
                        "so a launch's duration is longer than when it runs alone",
            "note": "VALU-issue-bound persistent kernel (400 dependent steps per scene in LDS): neither HBM nor MFMA "
                    "bounds it, the HBM fraction is reported because the schema asks for one -- see valu_frac "
                    "(SQ_INSTS_VALU per scene from profiles/r01/settle_sq_counters_v32.txt x scenes / launch time, "
                    "against 1024 SIMDs x 2.4 GHz / 4 cycles per wave64 instruction) and steps_scenes_per_s",
            "valu_insts_per_scene": VALU_INSTS_PER_SCENE,
            "valu_frac": VALU_INSTS_PER_SCENE * args.batch / (t_settle * 1e-3) / (1024 * 2.4e9 / 4),
            "steps_scenes_per_s": args.batch * 400 / (t_settle * 1e-3),
        }
        roof_settle["frac"] = roof_settle["achieved"] / roof_settle["peak"]
        roofline = dict(roof_settle if settle_dominant else roof_render)
        out = {
            "metric": "scenes/sec (settle + 640x480 6-ch GT render), 20-obj YCB-like",
            "value": value, "unit": "scenes/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms_step, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {
                "workload": "C2: 20 procedural YCB-like objects (8k verts/16k tris, 1024^2 texture each), tabletop "
                            "settle 100 frames x 4 substeps + 640x480 render of rgb/coord+depth/class/instance/normals, "
                            "shadows on, SSAO %s" % ("off" if args.no_ssao else "on"),
                "scenes_per_gpu_per_step": args.batch, "resolution": list(RESOLUTION), "objects": N_OBJECTS,
                "parallelism": "scenes sharded 1 batch/GPU, RCCL all-gather of rendered batches" if world > 1 else "1 GPU",
            },
            "roofline": roofline,
            "roofline_render": roof_render,
            "roofline_settle": roof_settle,
            "breakdown_ms": {
                "settle": t_settle, "host_assembly_per_chunk": float(np.mean(pipe.t_host)),
                "post_settle_wall": float(np.mean(pipe.t_step_host)), "render_total": t_render, "render_chunks": n_chunks,
                **{n: float(v) for n, v in zip(names, phases)},
            },
            "breakdown_isolated_ms": {n: float(v) for n, v in zip(names, iso)},
            "dominant_render_phase": names[k_dom],
        }
        if not args.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline(sl, meshes, args.cpu_scenes, not args.no_ssao)
        print(json.dumps(out))
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
