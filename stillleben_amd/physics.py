"""Python-side drivers of the settle half: Scene.simulate_tabletop_scene / simulate /
check_collisions / find_noncolliding_pose (reference src/scene.cpp:612-759, :903-925,
include/stillleben/scene.h:245-261).  All stepping runs in the HIP kernel behind the C-ABI
(slhip_settle / slhip_overlap_any); this module only assembles inputs and copies results."""
import ctypes as C
import math

import numpy as np
import torch

from . import _abi
from . import _math as M
from . import _settle_batch as SB
from . import pose_sampling
from ._context import engine
from ._math import f32

PLANE_HALF_Z = 0.04  # BOX_HALF_EXTENTS.z (scene.cpp:638)


class SettleEngine:
    """Device state of the settle half (hull pool in HBM, scratch)."""

    def __init__(self, eng):
        self.eng = eng
        self.pool = SB.HullPool()
        self._dev = None
        self._scratch = {}
        self._keep = {}
        # list capacities earlier batches on this engine turned out to need (run): later batches start from them instead of
        # paying the settle-again of the first one (a result never depends on a capacity)
        self._learned = {"max_hull_pairs_per_scene": 0, "max_contacts_per_scene": 0}   # (global-memory lists; the body-pair list sizes LDS)

    def hulls_dev(self):
        if self.pool.dirty or self._dev is None:
            hulls, verts = self.pool.arrays()
            self._dev = (self.eng.upload_records(hulls), torch.from_numpy(verts).to(self.eng.device))
            self.pool.dirty = False
        return self._dev

    def scratch(self, n_scenes, stream=0, params=None):
        """Scratch per stream: launches on different streams may run concurrently.  `params` carries
        the sizing hints of the batch (None: worst-case pair cache)."""
        need = C.c_uint64()
        prm = None if params is None else np.ascontiguousarray(params)
        self.eng.L.slhip_settle_scratch_bytes(n_scenes, C.c_void_p(prm.ctypes.data) if prm is not None else None, C.byref(need))
        cur = self._scratch.get(stream)
        if cur is None or cur.numel() < need.value:
            from ._context import check_free_memory

            self._scratch.pop(stream, None)
            del cur
            check_free_memory(self.eng.device, int(need.value), "the settle scratch of %d scenes (pair cache, contact lists, manifolds)" % n_scenes)
            cur = torch.empty(int(need.value), dtype=torch.uint8, device=self.eng.device)
            self._scratch[stream] = cur
        return cur

    def run(self, srec, bodies, params):
        """Runs slhip_settle on numpy records; returns the updated bodies (numpy).  The reference has no caps on a scene's
        contacts or hull pairs (PhysX allocates as it goes, scene.cpp:738-739): when a step of some scene offered more than the
        scratch's list capacities held (slhip_settle_caps counts what was dropped), the batch is settled again from the same
        start with capacities that hold what was seen -- the result never depends on a capacity."""
        prm = SB.sizing_hints(np.ascontiguousarray(params).copy(), srec, bodies, self.pool.arrays()[0])
        if int(prm["resume"].reshape(-1)[0]) == 0:
            for k, v in self._learned.items():
                if int(prm[k].reshape(-1)[0]) == 0 and v > 0:      # (a capacity the caller names is the caller's)
                    prm[k] = v
        for _ in range(8):
            d_bodies = self.run_device(srec, bodies, prm)
            self.check_status(len(srec))
            caps = self.caps(len(srec))
            self.last_params = prm.copy()      # (with the capacities this run used)
            if caps["scenes_dropped"] == 0:
                break
            if int(prm["resume"].reshape(-1)[0]) != 0:
                raise RuntimeError("slhip_settle: a resumed call ran out of list capacity (%r): size the scene's scratch larger" % (caps,))
            cur_p = int(prm["max_hull_pairs_per_scene"].reshape(-1)[0]) or _abi.DEFAULT_HULL_PAIRS
            cur_c = int(prm["max_contacts_per_scene"].reshape(-1)[0]) or _abi.DEFAULT_CONTACTS
            if caps["pair_drop_steps"]:
                prm["max_hull_pairs_per_scene"] = min(65535, max(2 * cur_p, caps["max_hull_pairs"] * 5 // 4))
            if caps["contact_drop_steps"] or caps["pair_drop_steps"] or caps["group_drop_steps"]:
                prm["max_contacts_per_scene"] = min(65535, max(2 * cur_c if caps["contact_drop_steps"] else cur_c, caps["max_contacts"] * 5 // 4))
            if caps["group_drop_steps"]:
                # the list of touching body pairs (one solver group each): twice what it held, at most every pair of the largest scene
                nb = max(int(r["body_end"]) - int(r["body_begin"]) for r in srec)
                cur_g = int(prm["max_body_pairs_per_scene"].reshape(-1)[0]) or 12 * nb + 64
                if cur_g >= nb * (nb - 1) // 2:
                    raise RuntimeError("slhip_settle: body pairs were dropped although the list holds every pair (%r)" % (caps,))
                prm["max_body_pairs_per_scene"] = min(nb * (nb - 1) // 2, 2 * cur_g)
                prm["max_hull_pairs_per_scene"] = max(int(prm["max_hull_pairs_per_scene"].reshape(-1)[0]) or _abi.DEFAULT_HULL_PAIRS,
                                                      int(prm["max_body_pairs_per_scene"].reshape(-1)[0]))
            if cur_p >= 65535 and cur_c >= 65535 and not caps["group_drop_steps"]:
                raise RuntimeError("slhip_settle: a scene offers more hull pairs / contacts per step than the 16-bit lists hold (%r)" % (caps,))
            for k in self._learned:      # (bounded: a giant heap's lists are not what every later batch should allocate)
                self._learned[k] = max(self._learned[k], min(8192, int(prm[k].reshape(-1)[0])))
        else:
            raise RuntimeError("slhip_settle: list capacities still too small after eight attempts (%r)" % (caps,))
        out = np.frombuffer(d_bodies.cpu().numpy().tobytes(), dtype=SB.BODY_DTYPE).copy()
        return out

    def check_status(self, n_scenes, stream=None, scratch=None):
        """Raises if the last launch on `stream` left scenes untouched (sizing hints too small): the
        kernel refuses such scenes instead of corrupting LDS, and that must not pass silently."""
        eng = self.eng
        if stream is None:
            stream = torch.cuda.current_stream(eng.device).cuda_stream
        if scratch is None:
            scratch = self._scratch[stream]
        bad = C.c_uint32()
        with torch.cuda.device(eng.device):
            st = eng.L.slhip_settle_status(_abi_ptr(scratch), n_scenes, None, C.byref(bad), C.c_void_p(stream))
        _abi.check(st, "slhip_settle")

    def run_with_status(self, srec, bodies, params, **hints):
        """run() with explicit sizing hints; returns (bodies, per-scene status words) instead of raising on refused scenes."""
        d_bodies = self.run_device(srec, bodies, params, hints=hints)
        eng = self.eng
        stream = torch.cuda.current_stream(eng.device).cuda_stream
        status = np.zeros(len(srec), np.uint32)
        bad = C.c_uint32()
        with torch.cuda.device(eng.device):
            eng.L.slhip_settle_status(_abi_ptr(self._scratch[stream]), len(srec), C.c_void_p(status.ctypes.data), C.byref(bad),
                                      C.c_void_p(stream))
        return np.frombuffer(d_bodies.cpu().numpy().tobytes(), dtype=SB.BODY_DTYPE).copy(), status

    def caps(self, n_scenes, stream=None, prm=None, scratch=None):
        """What the list capacities cost since the last cold start on the stream's scratch (slhip_settle_caps), as a dict:
        spill_steps (scene-steps whose contacts went beyond the solver's LDS-resident part: swept from global memory, nothing
        lost), contact_drop_steps / pair_drop_steps (scene-steps that DROPPED contacts / hull pairs beyond the capacities:
        the contract is zero), scenes_dropped, scenes_spilled, max_contacts, max_hull_pairs (the most a step offered),
        reduced_steps (scene-steps in which pair_contact_budget reduced some body pair's points), group_drop_steps (scene-steps
        that dropped body pairs beyond max_body_pairs_per_scene), contact_sum (contacts the solver took over all scene-steps)."""
        eng = self.eng
        if stream is None:
            stream = torch.cuda.current_stream(eng.device).cuda_stream
        out = (C.c_uint64 * 10)()
        if prm is None:
            prm = self._keep[stream][1]
        if scratch is None:
            scratch = self._scratch[stream]
        with torch.cuda.device(eng.device):
            st = eng.L.slhip_settle_caps(_abi_ptr(scratch), n_scenes, C.c_void_p(prm.ctypes.data), C.byref(out), C.c_void_p(stream))
        _abi.check(st, "slhip_settle_caps")
        keys = ("spill_steps", "contact_drop_steps", "pair_drop_steps", "scenes_dropped", "scenes_spilled", "max_contacts", "max_hull_pairs",
                "reduced_steps", "group_drop_steps", "contact_sum")
        return {k: int(v) for k, v in zip(keys, out)}

    def run_with_caps(self, srec, bodies, params):
        """ONE slhip_settle with exactly the given capacities (no growth): (bodies, caps)."""
        d_bodies = self.run_device(srec, bodies, params)
        self.check_status(len(srec))
        out = np.frombuffer(d_bodies.cpu().numpy().tobytes(), dtype=SB.BODY_DTYPE).copy()
        return out, self.caps(len(srec))

    def scratch_bytes(self, n_scenes, prm):
        need = C.c_uint64()
        self.eng.L.slhip_settle_scratch_bytes(n_scenes, C.c_void_p(prm.ctypes.data), C.byref(need))
        return int(need.value)

    def run_device(self, srec, bodies, params, d_bodies=None, hints=None, scratch=None):
        """`scratch`: a caller-owned scratch tensor (a scene that keeps its contact state between calls, SceneState) instead
        of the stream's."""
        eng = self.eng
        d_hulls, d_verts = self.hulls_dev()
        d_s = eng.upload_records(srec)
        if d_bodies is None:
            d_bodies = eng.upload_records(bodies)
        stream = torch.cuda.current_stream(eng.device).cuda_stream
        if bodies is not None:
            params = SB.sizing_hints(params, srec, bodies, self.pool.arrays()[0])
        prm = np.ascontiguousarray(params).copy()
        for k, v in (hints or {}).items():
            prm[k] = v
        if scratch is None:
            scratch = self.scratch(len(srec), stream, prm)
        with torch.cuda.device(eng.device):
            st = eng.L.slhip_settle(_abi_ptr(d_s), len(srec), _abi_ptr(d_bodies), _abi_ptr(d_hulls), _abi_ptr(d_verts),
                                    C.c_void_p(prm.ctypes.data), _abi_ptr(scratch), scratch.numel(), C.c_void_p(stream))
        _abi.check(st, "slhip_settle")
        self._keep[stream] = (d_s, prm)   # alive until the next launch on the same stream
        return d_bodies

    def overlap(self, srec, bodies):
        eng = self.eng
        d_hulls, d_verts = self.hulls_dev()
        d_s, d_b = eng.upload_records(srec), eng.upload_records(bodies)
        flags = torch.zeros(len(bodies), dtype=torch.uint8, device=eng.device)
        stream = torch.cuda.current_stream(eng.device).cuda_stream
        with torch.cuda.device(eng.device):
            st = eng.L.slhip_overlap_any(_abi_ptr(d_s), len(srec), _abi_ptr(d_b), _abi_ptr(d_hulls), _abi_ptr(d_verts),
                                         _abi_ptr(flags), C.c_void_p(stream))
        _abi.check(st, "slhip_overlap_any")
        return flags.cpu().numpy()


def _abi_ptr(t):
    return C.c_void_p(t.data_ptr())


def settle_engine():
    eng = engine()
    if getattr(eng, "_settle", None) is None:
        eng._settle = SettleEngine(eng)
    return eng._settle


def smoke_check(sl, checker):
    """Used by __graft_entry__.smoke() only: a 4-object tabletop settle (25 frames) through the C-ABI,
    compared bit for bit with `checker.settle` (the CPU oracle the caller hands in -- this module never
    imports it)."""
    import os

    cube = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "fixtures", "cube.glb")
    m = sl.Mesh(cube)
    m.center_bbox()
    m.scale_to_bbox_diagonal(0.2)
    scene = sl.Scene((160, 120), seed=11)
    for _ in range(4):
        scene.add_object(sl.Object(m))
    has_plane = prepare_tabletop(scene)
    se = settle_engine()
    srec, bodies = SB.build_settle_batch([scene], se.pool, [(has_plane, PLANE_HALF_Z)])
    prm = SB.default_params(tabletop=True, frames=25)
    gpu = se.run(srec, bodies.copy(), prm)
    hulls, verts = se.pool.arrays()
    ref = bodies.copy()
    checker.settle(srec, ref, hulls, verts, SB.sizing_hints(prm, srec, bodies, hulls))
    for name in ("pose", "lin_vel", "ang_vel", "separation", "flags", "stuck_counter", "stab"):
        a, b = np.ascontiguousarray(gpu[name]), np.ascontiguousarray(ref[name])
        if not np.array_equal(a.view(np.uint8), b.view(np.uint8)):
            raise AssertionError("settle smoke: body field '%s' differs from the oracle" % name)
    if not (gpu["pose"].reshape(-1, 4, 4)[:, 2, 3] > 0.0).all():
        raise AssertionError("settle smoke: an object fell through the table")
    return len(gpu)


# ------------------------------------------------------------------------------------------
def prepare_tabletop_batch(scenes):
    """Host part of simulateTableTopScene before the loop (scene.cpp:612-718) for a list of scenes: plane decision,
    background-plane pose, initial stack of randomly oriented objects -- the arithmetic over all objects at once (numpy), the
    random draws scene by scene from the scene's own stream in the reference's order (plane yaw, then one quaternion per dynamic
    object).  Returns has_plane per scene."""
    has_plane = []
    dyn, z_pos, quats = [], [], []
    bbox_of = {}
    for scene in scenes:
        scene.load_physics()
        d = [o for o in scene._objects if not o._static]
        hp = len(d) == len(scene._objects)
        has_plane.append(hp)
        rng = scene._rng
        z0 = f32(0.4)
        if hp:
            yaw = f32(rng.uniform(-math.pi, math.pi))
            scene._background_plane_pose = (M.rotation_z(yaw) @ M.translation([0.0, 0.0, PLANE_HALF_Z])).astype(np.float32)
            z0 = f32(PLANE_HALF_Z)
        if not d:
            continue
        half = np.empty(2 * len(d) + 1, np.float32)
        half[0] = z0
        for k, o in enumerate(d):
            m = o._mesh
            b = bbox_of.get(id(m))
            if b is None:
                bb = m.bbox
                b = bbox_of[id(m)] = (bb.np_diagonal() / f32(2.0), bb.np_center())
            half[2 * k + 1] = half[2 * k + 2] = b[0]
        z_pos.append(np.cumsum(half, dtype=np.float32)[1::2])          # z + d/2, then + d/2 again for the next one (float32, in order)
        quats.append(np.stack([rng.standard_normal(4) for _ in d]).astype(np.float32))   # one 4-draw per object, in order
        dyn.extend(d)
    if dyn:
        q = np.concatenate(quats)
        q = (q / np.sqrt((q * q).sum(axis=1, dtype=np.float32)).astype(np.float32)[:, None]).astype(np.float32)
        x, y, z, w = (q[:, k].astype(np.float64) for k in range(4))
        n = np.sqrt(x * x + y * y + z * z + w * w)
        x, y, z, w = x / n, y / n, z / n, w / n
        poses = np.zeros((len(dyn), 4, 4), np.float32)
        poses[:, 0, 0] = 1 - 2 * (y * y + z * z); poses[:, 0, 1] = 2 * (x * y - z * w); poses[:, 0, 2] = 2 * (x * z + y * w)
        poses[:, 1, 0] = 2 * (x * y + z * w); poses[:, 1, 1] = 1 - 2 * (x * x + z * z); poses[:, 1, 2] = 2 * (y * z - x * w)
        poses[:, 2, 0] = 2 * (x * z - y * w); poses[:, 2, 1] = 2 * (y * z + x * w); poses[:, 2, 2] = 1 - 2 * (x * x + y * y)
        poses[:, 3, 3] = 1.0
        c = -np.stack([bbox_of[id(o._mesh)][1] for o in dyn]).astype(np.float32)
        R = poses[:, :3, :3]
        t = R[:, :, 0] * c[:, 0:1]                                      # T(q, pos) . T(-bbox centre)
        t = t + R[:, :, 1] * c[:, 1:2]
        t = t + R[:, :, 2] * c[:, 2:3]
        t[:, 2] = t[:, 2] + np.concatenate(z_pos)
        poses[:, :3, 3] = t
        inf = f32(np.inf)
        for k, obj in enumerate(dyn):
            obj._pose = poses[k].copy()
            obj._linear_velocity = np.zeros(3, np.float32)
            obj._angular_velocity = np.zeros(3, np.float32)
            obj._stuck_counter = 0
            obj._separation = inf
    return has_plane


def prepare_tabletop(scene):
    """prepare_tabletop_batch of one scene; returns has_plane."""
    return prepare_tabletop_batch([scene])[0]


def choose_camera_poses_batch(scenes):
    """Scene::chooseRandomCameraPose (scene.cpp:472-610) for a list of scenes: azimuth and elevation from each scene's stream,
    the frustum fit over all scenes at once (_fast_batch.camera_poses)."""
    from . import _fast_batch as FB

    with_objs = [s for s in scenes if s._objects]
    for s in scenes:
        if not s._objects:
            s.choose_random_camera_pose()
    if not with_objs:
        return
    az = np.empty(len(with_objs), np.float32)
    el = np.empty(len(with_objs), np.float32)
    for i, s in enumerate(with_objs):
        az[i] = s._rng.uniform(-math.pi, math.pi)
        el[i] = s._rng.uniform(math.radians(30.0), math.pi / 2.0 - math.radians(30.0))
    t = FB.BatchTemplate()
    objs = [o for s in with_objs for o in s._objects]
    counts = np.array([len(s._objects) for s in with_objs], np.int64)
    t.n_scenes, t.n_obj, t.max_objs = len(with_objs), len(objs), int(counts.max())
    t.obj_scene = np.repeat(np.arange(len(with_objs), dtype=np.int64), counts)
    t.obj_base = np.concatenate([[0], np.cumsum(counts)]).astype(np.int64)
    corners = {}
    cs = []
    for o in objs:
        c = corners.get(id(o._mesh))
        if c is None:
            c = corners[id(o._mesh)] = o._mesh.bbox.corners()
        cs.append(c)
    t.bbox_corners = np.stack(cs).astype(np.float32)
    t.proj = np.stack([s._projection for s in with_objs]).astype(np.float32)
    poses = np.stack([o._pose for o in objs]).astype(np.float32)
    cam = FB.camera_poses(t, poses, az, el)
    for s, c in zip(with_objs, cam):
        s._camera_pose = c.copy()


class SceneState:
    """What the reference's PxScene keeps while the sl.Scene lives (scene.cpp:134-173: one PxScene per Scene, stepped by every
    simulate / ManipulationSim::step / frame of simulateTableTopScene): the contact state of the stepper -- pair cache, persistent
    manifolds and their impulses, table contacts, in the scene's own device scratch --, wake counters and sleep flags of the
    bodies, the number of steps run.  A call continues it (slhip_settle_params.resume) as long as the scene is what the last call
    left: same objects in the same order, same table, poses untouched from outside."""

    def __init__(self):
        self.scratch = None
        self.steps = 0
        self.sig = None
        self.bodies = None      # the records the last call returned
        self.refs = None        # the pose / velocity arrays that call wrote into the objects


def _signature(scene, srec, bodies, prm):
    return (tuple(id(o) for o in scene._objects), bodies["hull_begin"].tobytes(), bodies["hull_end"].tobytes(),
            bodies["inv_mass"].tobytes(), srec.tobytes(), int(prm["max_bodies_per_scene"]), int(prm["max_hull_verts_per_scene"]),
            int(prm["max_hulls_per_scene"]), int(prm["tabletop"]))


def step_scene(scene, plane, **prm_kw):
    """One slhip_settle call on the scene's long-lived state (created cold when the scene has none, or is not what the last
    call left).  Returns the body records."""
    se = settle_engine()
    srec, bodies = SB.build_settle_batch([scene], se.pool, [plane])
    hulls = se.pool.arrays()[0]
    prm = SB.sizing_hints(SB.default_params(**prm_kw), srec, bodies, hulls)
    # A long-lived scene keeps its contact state in ITS scratch, whose layout the capacities fix: they are sized once, from what the
    # scene can offer at most -- every body pair, every hull pair of two different bodies, four points per hull pair and per body
    # against the table -- so that no later step can run out (PhysX allocates as it goes, scene.cpp:738-739).  Scenes beyond the
    # 16-bit lists (hundreds of hulls) get the largest lists there are; a step that still overflows them raises below.
    hb = (bodies["hull_end"].astype(np.int64) - bodies["hull_begin"].astype(np.int64))
    all_pairs = int((hb.sum() ** 2 - (hb ** 2).sum()) // 2)
    nbod = len(bodies)
    prm["max_hull_pairs_per_scene"] = min(65535, max(64, all_pairs))
    prm["max_contacts_per_scene"] = min(65535, max(64, 4 * all_pairs + 4 * nbod))
    # ... the list of touching body pairs as well, as far as the kernels' LDS holds a scene's groups (~26 B each beside 152 B per body:
    # every pair up to 96 bodies); beyond that a body of a pile has a handful of neighbours, not all the others: 12 per body (6 from
    # 256 bodies on, sizing_hints' rule) -- a step that offers more is counted (caps: group drops) and raises below, never dropped silently
    if int(prm["max_body_pairs_per_scene"]) == 0:
        prm["max_body_pairs_per_scene"] = max(1, nbod * (nbod - 1) // 2) if nbod <= 96 else min(nbod * (nbod - 1) // 2, 12 * nbod + 64)
    sig = _signature(scene, srec, bodies, prm)
    st = scene._phys_state
    resume = st is not None and st.sig == sig and len(st.bodies) == len(bodies)
    if resume:
        for i, o in enumerate(scene._objects):
            r = st.refs[i]
            if o._pose is not r[0]:
                resume = False      # set_pose from outside (static bodies too): a teleport -- the contact state is not this arrangement's
                break
    if st is None:
        st = scene._phys_state = SceneState()
    if len(bodies) == 0:            # nothing to step (PxScene::simulate of an empty scene)
        st.steps = (st.steps if resume else 0) + int(prm["frames"]) * int(prm["substeps"])
        st.sig, st.bodies, st.refs = sig, bodies, []
        return bodies
    if resume:
        # what only the stepper knows about a body travels in the records: wake counter, sleep flag
        keep = st.bodies
        for i, o in enumerate(scene._objects):
            r = st.refs[i]
            if o._static:
                continue
            if o._linear_velocity is r[1] and o._angular_velocity is r[2]:
                bodies["wake_counter"][i] = keep["wake_counter"][i]
                bodies["flags"][i] = keep["flags"][i]
                bodies["stab"][i] = keep["stab"][i]        # (the stabilisation's timers, slhip_body.stab)
            # (a velocity set from outside wakes the body, like PxRigidBody::setLinearVelocity's autowake)
        prm["resume"] = st.steps
    else:
        st.steps = 0
        prm["resume"] = 0
    need = se.scratch_bytes(1, np.ascontiguousarray(prm))
    if st.scratch is None or st.scratch.numel() < need or st.scratch.device != se.eng.device:
        if resume:
            raise RuntimeError("settle scratch of a live scene changed size")
        st.scratch = torch.empty(need, dtype=torch.uint8, device=se.eng.device)
    d_bodies = se.run_device(srec, None, prm, d_bodies=se.eng.upload_records(bodies), scratch=st.scratch)
    if not resume:
        se.check_status(1, scratch=st.scratch)
    caps = se.caps(1, prm=np.ascontiguousarray(prm), scratch=st.scratch)
    if caps["scenes_dropped"]:
        scene._phys_state = None
        raise RuntimeError("slhip_settle: the scene offers more hull pairs / contacts per step than its scratch holds (%r)" % (caps,))
    out = np.frombuffer(d_bodies.cpu().numpy().tobytes(), dtype=SB.BODY_DTYPE).copy()
    SB.write_back([scene], out)
    st.steps += int(prm["frames"]) * int(prm["substeps"])
    st.sig = sig
    st.bodies = out
    st.refs = [(o._pose, o._linear_velocity, o._angular_velocity) for o in scene._objects]
    return out


def simulate_tabletop_scene(scene, vis_cb=None):
    has_plane = prepare_tabletop(scene)
    scene._phys_state = None
    if vis_cb is None:
        se = settle_engine()
        srec, bodies = SB.build_settle_batch([scene], se.pool, [(has_plane, PLANE_HALF_Z)])
        bodies = se.run(srec, bodies, SB.default_params(tabletop=True))
        SB.write_back([scene], bodies)
    else:
        # quirk q7: the callback runs BEFORE each frame's sub-steps (scene.cpp:723-724); the frames step ONE scene whose
        # contact state carries over (scene.cpp:720-739) -- the poses are those of the call without a callback, bit for bit
        for i in range(100):
            vis_cb(i)
            step_scene(scene, (has_plane, PLANE_HALF_Z), tabletop=True, frames=1)
        scene._phys_state = None
    scene.choose_random_camera_pose()


def settle_batch(scenes, frames=None):
    """Additive batch API (the GPU counterpart of JobQueue): settles many scenes in one launch."""
    planes = [(hp, PLANE_HALF_Z) for hp in prepare_tabletop_batch(scenes)]
    for sc in scenes:
        sc._phys_state = None
    se = settle_engine()
    srec, bodies = SB.build_settle_batch(scenes, se.pool, planes)
    bodies = se.run(srec, bodies, SB.default_params(tabletop=True, frames=frames))
    SB.write_back(scenes, bodies)
    choose_camera_poses_batch(scenes)


def simulate(scene, dt):
    """One step of dt without a table (scene.cpp:903-912)."""
    scene.load_physics()
    step_scene(scene, (False, 0.0), tabletop=False, dt=dt, frames=1, substeps=1)


def check_collisions(scene):
    """scene.cpp:914-925: separation = -FLT_MAX for colliding objects, else +inf (sic: 0 in
    the reference's isObjectColliding==false branch)."""
    scene.load_physics()
    se = settle_engine()
    srec, bodies = SB.build_settle_batch([scene], se.pool, [(False, 0.0)])
    flags = se.overlap(srec, bodies)
    for obj, f in zip(scene._objects, flags):
        obj._separation = f32(-np.finfo(np.float32).max) if f else f32(0.0)
    return flags


def is_object_colliding(scene, obj):
    scene.load_physics()
    se = settle_engine()
    srec, bodies = SB.build_settle_batch([scene], se.pool, [(False, 0.0)])
    flags = se.overlap(srec, bodies)
    return bool(flags[scene._objects.index(obj)])


def find_noncolliding_pose(scene, obj, sampler="random", max_iterations=10, **kwargs):
    """scene.h:245-261 + python/src/py_scene.cpp:196-245: rejection-sample poses (set directly as
    the object pose, like the reference) until the object does not collide."""
    if obj not in scene._objects:
        raise ValueError("object is not part of the scene")
    diameter = obj._mesh.bbox.np_diagonal()
    pos = pose_sampling.RandomPositionSampler(scene._projection, diameter, kwargs.get("min_size_factor", 0.4))
    if sampler == "random":
        smp = pose_sampling.RandomPoseSampler(pos)
    elif sampler == "viewpoint":
        if "viewpoint" not in kwargs:
            raise ValueError("sampler='viewpoint' needs viewpoint argument")
        vp = kwargs["viewpoint"]
        vp = vp.detach().cpu().numpy() if hasattr(vp, "detach") else np.asarray(vp)
        smp = pose_sampling.ViewPointPoseSampler(pos, vp.astype(np.float32).reshape(3))
    elif sampler == "view_corrected":
        if "orientation" not in kwargs:
            raise ValueError("sampler='view_corrected' needs orientation argument")
        o = kwargs["orientation"]
        o = o.detach().cpu().numpy() if hasattr(o, "detach") else np.asarray(o)
        smp = pose_sampling.ViewCorrectedPoseSampler(pos, o.astype(np.float32))
    else:
        raise ValueError("unknown sampler '%s'" % sampler)
    for _ in range(int(max_iterations)):
        obj._pose = smp(scene._rng)
        if not is_object_colliding(scene, obj):
            return True
    return False
