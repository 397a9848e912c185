"""Host-side assembly of the binary settle description (slhip_body / slhip_hull / hull vertex
pool / slhip_settle_scene) from sl.Scene objects, and write-back of the results."""
import numpy as np

from . import massprops
from ._math import f32

BODY_DTYPE = np.dtype([
    ("pose", np.float32, (16,)),
    ("lin_vel", np.float32, (4,)),
    ("ang_vel", np.float32, (4,)),
    ("com", np.float32, (4,)),
    ("inv_inertia", np.float32, (12,)),
    ("inv_mass", np.float32), ("mu_s", np.float32), ("mu_d", np.float32), ("restitution", np.float32),
    ("bsphere", np.float32, (4,)),
    ("bbox_center", np.float32, (4,)),
    ("max_lin_vel", np.float32), ("separation", np.float32), ("wake_counter", np.float32), ("flags", np.uint32),
    ("hull_begin", np.uint32), ("hull_end", np.uint32), ("stuck_counter", np.int32), ("drive_flags", np.uint32),
    ("drive_target", np.float32, (4,)), ("drive_frame", np.float32, (4,)), ("drive_params", np.float32, (4,)),
    ("stab", np.float32, (4,)),
])
assert BODY_DTYPE.itemsize == 304

HULL_DTYPE = np.dtype([("vtx_begin", np.uint32), ("vtx_count", np.uint32), ("_pad", np.uint32, (2,)),
                       ("sphere", np.float32, (4,)), ("aabb_center", np.float32, (4,)), ("aabb_half", np.float32, (4,))])
assert HULL_DTYPE.itemsize == 64

SETTLE_SCENE_DTYPE = np.dtype([("body_begin", np.uint32), ("body_end", np.uint32), ("has_plane", np.uint32),
                               ("plane_z", np.float32)])

PARAMS_DTYPE = np.dtype([
    ("dt", np.float32), ("substeps", np.uint32), ("frames", np.uint32), ("pos_iters", np.uint32),
    ("vel_iters", np.uint32), ("gravity", np.float32, (3,)), ("contact_offset", np.float32),
    ("rest_offset", np.float32), ("bounce_threshold", np.float32), ("sleep_threshold", np.float32),
    ("wake_time", np.float32), ("angular_damping", np.float32), ("max_angular_velocity", np.float32),
    ("plane_mu_s", np.float32), ("plane_mu_d", np.float32), ("plane_restitution", np.float32),
    ("redrop_z", np.float32), ("stuck_separation", np.float32), ("stuck_frames", np.int32), ("tabletop", np.uint32),
    ("max_bodies_per_scene", np.uint32), ("max_hull_verts_per_scene", np.uint32), ("max_hulls_per_scene", np.uint32),
    ("max_hull_pairs_per_scene", np.uint32), ("max_contacts_per_scene", np.uint32),
    ("pair_contact_budget", np.uint32), ("resume", np.uint32), ("max_body_pairs_per_scene", np.uint32),
    ("stabilization_threshold", np.float32), ("_pad_params", np.uint32),
])
assert PARAMS_DTYPE.itemsize == 128

# slhip_settle_params.pair_contact_budget.  0: every point goes to the solver, as in PhysX -- the default of every entry point, sl.SceneBatch
# and the benchmark included (round 5; rounds 3 and 4 ran the batch path at 32).  > 0: a body pair touching through more hull pairs keeps the
# deepest ones -- nested concave shapes otherwise put several hundred one-point manifolds into ONE Gauss-Seidel chain.  NOT in the
# reference; an option (sl.SceneBatch(..., pair_contact_budget=32), bench.py --pair-budget 32: +2 % scenes/s), stated by settle_caps.
PAIR_CONTACT_BUDGET = 0
PAIR_CONTACT_BUDGET_FAST = 32

BODY_STATIC = 1
BODY_ASLEEP = 2
BODY_FROZEN = 4
MAX_BODIES = 400


def default_params(tabletop=True, dt=None, frames=None, substeps=None, pair_contact_budget=0):
    """Constants of the reference's call sites."""
    p = np.zeros((), dtype=PARAMS_DTYPE)
    p["dt"] = (1.0 / 25.0 / 4.0) if dt is None else dt   # scene.cpp:681-684
    p["substeps"] = 4 if substeps is None else substeps
    p["frames"] = 100 if frames is None else frames        # scene.cpp:720
    p["pos_iters"], p["vel_iters"] = 4, 4                  # object.cpp:209
    p["gravity"] = (0.0, 0.0, -9.81)                       # scene.cpp:157
    p["contact_offset"] = 0.02 * 0.2                       # tolerance length 0.2 (context.cpp:236-238)
    p["rest_offset"] = 0.0015                              # object.cpp:201
    p["bounce_threshold"] = 0.2 * 10.0
    p["sleep_threshold"] = 5e-5 * 10.0 * 10.0
    p["wake_time"] = 0.4
    p["stabilization_threshold"] = 1e-5 * 10.0 * 10.0   # PxSceneFlag::eENABLE_STABILIZATION (scene.cpp:163) with PhysX's default threshold [ext]
    p["angular_damping"] = 0.05
    p["max_angular_velocity"] = 100.0
    p["plane_mu_s"], p["plane_mu_d"], p["plane_restitution"] = 0.5, 0.5, 0.0  # scene.cpp:645
    p["redrop_z"] = -0.5                                   # scene.cpp:746
    p["stuck_separation"] = -0.01                          # scene.cpp:748
    p["stuck_frames"] = 10                                 # 0.4 s * 25 FPS (scene.cpp:750)
    p["tabletop"] = 1 if tabletop else 0
    p["pair_contact_budget"] = pair_contact_budget
    return p


class HullPool:
    """Collision hulls of every mesh in use, object frame, float4 vertices."""

    def __init__(self):
        self.verts = []
        self.hulls = []
        self.n_verts = 0
        self._ranges = {}
        self._mesh_refs = []   # the keyed meshes, kept alive: id() of a collected mesh may be reused by a new one
        self.dirty = True

    def register(self, mesh):
        key = (id(mesh), float(mesh._scale), mesh._pretransform_rigid.tobytes(), mesh._version)
        r = self._ranges.get(key)
        if r is not None:
            return r
        begin = len(self.hulls)
        centers, radii = [], []
        for v, _t in massprops.object_frame_hulls(mesh):
            v = v.astype(np.float32)
            if len(v) > 64:
                raise RuntimeError("collision hull with more than 64 vertices")
            c = ((v.min(axis=0) + v.max(axis=0)) / f32(2.0)).astype(np.float32)
            r_ = f32(np.sqrt(((v - c) ** 2).sum(axis=1).max()))
            h = np.zeros((), dtype=HULL_DTYPE)
            h["vtx_begin"], h["vtx_count"] = self.n_verts, len(v)
            h["sphere"][:3], h["sphere"][3] = c, r_
            h["aabb_center"][:3] = c
            h["aabb_half"][:3] = ((v.max(axis=0) - v.min(axis=0)) / f32(2.0)).astype(np.float32)
            self.hulls.append(h)
            self.verts.append(np.concatenate([v, np.ones((len(v), 1), np.float32)], axis=1))
            self.n_verts += len(v)
            centers.append(c)
            radii.append(r_)
        centers = np.array(centers, np.float32)
        radii = np.array(radii, np.float32)
        bc = ((centers - radii[:, None]).min(axis=0) + (centers + radii[:, None]).max(axis=0)) / f32(2.0)
        br = f32((np.sqrt(((centers - bc) ** 2).sum(axis=1)) + radii).max())
        r = (begin, len(self.hulls), bc.astype(np.float32), br)
        self._ranges[key] = r
        self._mesh_refs.append(mesh)
        self.dirty = True
        return r

    def arrays(self):
        hulls = np.array(self.hulls, dtype=HULL_DTYPE) if self.hulls else np.zeros(0, HULL_DTYPE)
        verts = np.concatenate(self.verts) if self.verts else np.zeros((1, 4), np.float32)
        return hulls, np.ascontiguousarray(verts, dtype=np.float32)


def body_record(obj, pool, rec):
    mesh = obj._mesh
    hb, he, bc, br = pool.register(mesh)
    p = obj._props()
    rec["pose"] = obj._pose.reshape(-1)
    rec["lin_vel"][:3] = obj._linear_velocity
    rec["ang_vel"][:3] = obj._angular_velocity
    rec["com"][:3] = p.com
    ii = np.zeros((3, 4), np.float32)
    ii[:, :3] = p.inv_inertia
    rec["inv_inertia"] = ii.reshape(-1)
    rec["inv_mass"] = 0.0 if obj._static else f32(1.0) / p.mass
    rec["mu_s"], rec["mu_d"], rec["restitution"] = obj._static_friction, obj._dynamic_friction, obj._restitution
    rec["bsphere"][:3], rec["bsphere"][3] = bc, br
    bbox = mesh.bbox
    rec["bbox_center"][:3] = bbox.np_center()
    rec["bbox_center"][3] = bbox.np_diagonal() / f32(2.0)
    lim = float(obj._linear_velocity_limit)
    rec["max_lin_vel"] = lim if lim < 1e15 else 0.0
    rec["separation"] = obj._separation
    rec["wake_counter"] = 0.4
    rec["flags"] = BODY_STATIC if obj._static else 0
    rec["hull_begin"], rec["hull_end"] = hb, he
    rec["stuck_counter"] = obj._stuck_counter
    drv = getattr(obj, "_drive", None)
    if drv is not None:
        rec["drive_flags"] = drv["flags"]
        rec["drive_target"][:3] = drv["target"]
        rec["drive_frame"] = drv["frame"]
        rec["drive_params"][:3] = (drv["stiffness"], drv["damping"], drv["force_limit"])


def _body_template(obj, pool):
    """slhip_body with everything the mesh, the density and the object's material determine -- shared by the objects of one
    mesh with default settings; pose, velocities and per-step state are patched in per use."""
    rec = np.zeros((), dtype=BODY_DTYPE)
    body_record(obj, pool, rec)
    # a template is shared by every object with the same mesh and settings: the spring drive of a ManipulationSim manipulator
    # that happens to be the first object seen for its key must not travel with it
    rec["drive_flags"] = 0
    rec["drive_target"] = 0.0
    rec["drive_frame"] = 0.0
    rec["drive_params"] = 0.0
    return rec


def build_settle_batch(scenes, pool, with_plane):
    """with_plane: list of (has_plane, plane_z) per scene.  Bodies of driven objects (ManipulationSim) are filled one by one;
    all others start from a per-(mesh, density, material) template and get pose, velocities and step state in array sweeps."""
    for scene in scenes:
        if len(scene._objects) > MAX_BODIES:
            raise RuntimeError("at most %d objects per scene are supported by the settle kernel" % MAX_BODIES)
    objs = [o for s in scenes for o in s._objects]
    n = len(objs)
    counts = np.fromiter((len(s._objects) for s in scenes), np.int64, len(scenes))
    srec = np.zeros(len(scenes), dtype=SETTLE_SCENE_DTYPE)
    ends = np.cumsum(counts)
    srec["body_end"] = ends
    srec["body_begin"] = ends - counts
    srec["has_plane"] = np.fromiter((1 if p[0] else 0 for p in with_plane), np.uint32, len(scenes))
    srec["plane_z"] = np.fromiter((p[1] for p in with_plane), np.float32, len(scenes))
    if n == 0:
        return srec, np.zeros(0, dtype=BODY_DTYPE)
    cache = pool.__dict__.setdefault("_body_templates", {})
    rows, tidx = [], np.empty(n, np.int64)
    local = {}
    slow = []
    for k, o in enumerate(objs):
        if getattr(o, "_drive", None) is not None:
            slow.append(k)
        key = (id(o._mesh), o._mesh._version, float(o._mesh._scale), o._mesh._pretransform_rigid.tobytes(), float(o._density),
               bool(o._static), float(o._static_friction), float(o._dynamic_friction), float(o._restitution), float(o._linear_velocity_limit))
        j = local.get(key)
        if j is None:
            t = cache.get(key)
            if t is None:
                if len(cache) > 4096:
                    cache.clear()
                t = cache[key] = (_body_template(o, pool), o._mesh)     # (the mesh is kept alive with its id)
            j = local[key] = len(rows)
            rows.append(t[0])
        tidx[k] = j
    bodies = np.array(rows, dtype=BODY_DTYPE)[tidx]
    bodies["pose"] = np.stack([o._pose for o in objs]).reshape(n, 16)
    bodies["lin_vel"][:, :3] = np.stack([o._linear_velocity for o in objs])
    bodies["ang_vel"][:, :3] = np.stack([o._angular_velocity for o in objs])
    bodies["separation"] = np.fromiter((o._separation for o in objs), np.float32, n)
    bodies["stuck_counter"] = np.fromiter((o._stuck_counter for o in objs), np.int32, n)
    for k in slow:
        body_record(objs[k], pool, bodies[k])
    return srec, bodies


def sizing_hints(params, srec, bodies, hulls):
    """Fills the LDS sizing hints of slhip_settle_params from a built batch."""
    mb, mv, mh = 0, 0, 0
    cnt = hulls["vtx_count"].astype(np.int64)
    csum = np.concatenate([[0], np.cumsum(cnt)])
    per_body_v = csum[bodies["hull_end"]] - csum[bodies["hull_begin"]]
    per_body_h = bodies["hull_end"].astype(np.int64) - bodies["hull_begin"].astype(np.int64)
    bv = np.concatenate([[0], np.cumsum(per_body_v)])
    bh = np.concatenate([[0], np.cumsum(per_body_h)])
    b0, b1 = srec["body_begin"].astype(np.int64), srec["body_end"].astype(np.int64)
    if len(srec):
        mb = int((b1 - b0).max())
        mv = int((bv[b1] - bv[b0]).max())
        mh = int((bh[b1] - bh[b0]).max())
    params = params.copy()
    params["max_bodies_per_scene"], params["max_hull_verts_per_scene"], params["max_hulls_per_scene"] = mb, mv, mh
    if mb > 256 and int(np.asarray(params["max_body_pairs_per_scene"]).reshape(-1)[0]) == 0:
        # hundreds of bodies: the default list of touching body pairs (12 per body) would not fit the kernels' LDS beside the
        # bodies themselves -- six per body do (a heap: ~3); SettleEngine.run grows the list if a scene needs more
        params["max_body_pairs_per_scene"] = 6 * mb
    return params


def write_back(scenes, bodies):
    poses = np.ascontiguousarray(bodies["pose"]).reshape(-1, 4, 4).astype(np.float32)
    lv = np.ascontiguousarray(bodies["lin_vel"][:, :3])
    av = np.ascontiguousarray(bodies["ang_vel"][:, :3])
    sep = bodies["separation"].astype(np.float32)
    stuck = bodies["stuck_counter"].tolist()
    k = 0
    for scene in scenes:
        for obj in scene._objects:
            i = k
            k += 1
            if obj._static:
                continue
            obj._pose = poses[i].copy()
            obj._linear_velocity = lv[i].copy()
            obj._angular_velocity = av[i].copy()
            obj._separation = sep[i]
            obj._stuck_counter = stuck[i]
