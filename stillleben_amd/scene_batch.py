"""Batched scene synthesis: the GPU counterpart of the reference's per-scene host loop
(examples/ycb.py:36-80 -- build a scene from random meshes, `simulate_tabletop_scene`,
`choose_random_light_direction`, `RenderPass.render`) for MANY scenes at once, with every per-scene
step on the device (slhip_synth_stage -> slhip_settle -> slhip_synth_place -> slhip_render,
include/slhip.h).  The host describes the batch once (an asset table built from sl.Mesh objects, the
camera intrinsics, light colour, seed); per scene nothing crosses PCIe.

    table = sl.AssetTable(meshes)                       # once
    batch = sl.SceneBatch(table, n_scenes=4096, n_objects=20, resolution=(640, 480), seed=1)
    batch.set_camera_intrinsics(1066.778, 1067.487, 312.9869, 241.3109)
    batch.stage(); batch.settle(); batch.place()
    for chunk in batch.render_chunks(): ...             # RenderBuffers, [chunk, H, W, C] tensors in HBM
    scene = batch.scene(17)                             # an ordinary sl.Scene rebuilt from the device records
"""
import ctypes as C
import os

import numpy as np
import torch

from . import _abi
from . import _settle_batch as SB
from ._batch import object_draws
from ._context import engine
from ._math import f32

PLANE_HALF_Z = 0.04   # BOX_HALF_EXTENTS.z (scene.cpp:638)


def _dev(arr, device):
    raw = np.frombuffer(np.ascontiguousarray(arr).tobytes(), dtype=np.uint8)
    if raw.size == 0:
        raw = np.zeros(16, np.uint8)
    return torch.from_numpy(raw.copy()).to(device)


class AssetTable:
    """slhip_asset records + sub-mesh draw templates of a list of sl.Mesh objects (built once; the meshes'
    vertices, textures and hulls are registered with the process-wide pools)."""

    def __init__(self, meshes, mesh_pool=None, hull_pool=None):
        """`mesh_pool` / `hull_pool`: host pools to register with instead of the device engine's (the CPU tests
        build the records for the oracle that way); a table built on explicit pools cannot drive a SceneBatch."""
        from .object import Object

        if not meshes:
            raise ValueError("AssetTable needs at least one mesh")
        if len(meshes) > _abi.SYNTH_MAX_ASSETS:
            raise ValueError("at most %d classes per asset table" % _abi.SYNTH_MAX_ASSETS)
        self.meshes = list(meshes)
        if mesh_pool is None or hull_pool is None:
            from . import physics

            self.eng = engine()
            self.se = physics.settle_engine()
            mesh_pool, hull_pool = self.eng.pool, self.se.pool
        else:
            self.eng = self.se = None
        self.mesh_pool, self.hull_pool = mesh_pool, hull_pool
        recs = np.zeros(len(meshes), dtype=_abi.ASSET_DTYPE)
        templates = []
        self.n_hulls, self.n_hull_verts, self.n_draws, self.n_chunks, self.n_clip = [], [], [], [], []
        for i, mesh in enumerate(self.meshes):
            obj = Object(mesh)                               # the defaults of sl.Object (material, flags)
            draws = object_draws(obj, mesh_pool)
            hb, he, bc, br = hull_pool.register(mesh)
            p = obj._props()
            r = recs[i]
            r["mesh_to_object"] = mesh._pretransform.reshape(-1)
            bbox = mesh.bbox
            r["bbox_min"][:3], r["bbox_max"][:3] = bbox._min, bbox._max
            r["com"][:3] = p.com
            ii = np.zeros((3, 4), np.float32)
            ii[:, :3] = p.inv_inertia
            r["inv_inertia"] = ii.reshape(-1)
            r["mass"] = p.mass
            r["mu_s"], r["mu_d"], r["restitution"] = obj._static_friction, obj._dynamic_friction, obj._restitution
            r["bsphere"][:3], r["bsphere"][3] = bc, br
            r["hull_begin"], r["hull_end"] = hb, he
            r["draw_begin"], r["draw_count"] = len(templates), len(draws)
            r["n_verts"] = draws[0]["n_verts"] if draws else 0
            chunks = sum((int(d["n_tris"]) + _abi.CHUNK_TRIS - 1) // _abi.CHUNK_TRIS for d in draws)
            r["n_chunks"] = chunks
            for d in draws:
                d["mesh_to_object"] = 0.0
                d["object_to_world"] = 0.0
                d["normal_to_world"] = 0.0
                d["instance_index"] = 0
                templates.append(d)
            hulls = hull_pool.hulls[hb:he]
            self.n_hulls.append(he - hb)
            self.n_hull_verts.append(int(sum(int(h["vtx_count"]) for h in hulls)))
            self.n_draws.append(len(draws))
            self.n_chunks.append(chunks)
            self.n_clip.append(int(r["n_verts"]) * len(draws))
        self.records = recs
        self.templates = np.array(templates, dtype=_abi.DRAW_DTYPE)
        self._dev = None

    def __len__(self):
        return len(self.meshes)

    def device(self):
        if self.eng is None:
            raise _abi.SlhipError("this AssetTable was built on host pools (test helper); build it without them to use the device")
        if self._dev is None:
            d = self.eng.device
            self._dev = (_dev(self.records, d), _dev(self.templates, d))
        return self._dev

    def bound(self, per_asset, n_objects, distinct):
        """Largest possible per-scene sum of a per-class quantity: the n_objects largest classes when classes are
        drawn without replacement, n_objects times the largest otherwise."""
        v = sorted(per_asset, reverse=True)
        return int(sum(v[:n_objects])) if distinct else int(v[0]) * n_objects


class SceneBatch:
    """n_scenes tabletop scenes of n_objects objects each, resident in HBM.  `asset_ids` ([n_scenes, n_objects]
    class indices into the table) fixes every scene's objects; without it each scene draws n_objects DISTINCT
    classes (examples/ycb.py:60).  `random_pbr`: metallic / roughness ~ U(0,1) per object (examples/ycb.py:63-64)."""

    def __init__(self, table, n_scenes, n_objects, resolution=(640, 480), seed=0, asset_ids=None, random_pbr=True,
                 shadows=True, render_chunk=None, plane_size=(3.0, 3.0), light_color=(300.0, 300.0, 300.0),
                 ambient=(0.05, 0.05, 0.05), manual_exposure=-1.0, scene_id_base=0, pair_contact_budget=SB.PAIR_CONTACT_BUDGET):
        from .scene import Scene

        if not 1 <= n_objects <= 64:      # SLHIP_SYNTH_MAX_OBJECTS: the synthesis kernels map an object to a lane of one wave
            raise ValueError("n_objects must be in [1, 64]")
        self.table, self.eng, self.se = table, table.eng, table.se
        self.n_scenes, self.n_objects = int(n_scenes), int(n_objects)
        self.resolution = tuple(resolution)
        self._proto = Scene(self.resolution)                 # projection bookkeeping of sl.Scene (scene.cpp:222-271)
        distinct = asset_ids is None
        if distinct and len(table) < n_objects:
            raise ValueError("drawing %d distinct classes needs an asset table of at least that size" % n_objects)
        self.asset_ids = None
        if asset_ids is not None:
            ids = np.ascontiguousarray(asset_ids, dtype=np.uint16).reshape(self.n_scenes, self.n_objects)
            if ids.max(initial=0) >= len(table):
                raise ValueError("asset id out of range")
            self.asset_ids = ids
        has_plane = float(plane_size[0]) ** 2 + float(plane_size[1]) ** 2 > 0
        p = np.zeros((), dtype=_abi.SYNTH_PARAMS_DTYPE)
        p["n_scenes"], p["n_objects"], p["n_assets"] = self.n_scenes, self.n_objects, len(table)
        p["flags"] = ((_abi.SYNTH_SAMPLE_DISTINCT if distinct else 0) | (_abi.SYNTH_RANDOM_PBR if random_pbr else 0)
                      | (_abi.SYNTH_SHADOWS if shadows else 0))
        p["seed_lo"], p["seed_hi"] = int(seed) & 0xFFFFFFFF, (int(seed) >> 32) & 0xFFFFFFFF
        p["scene_id_base"] = scene_id_base
        p["render_chunk"] = self.n_scenes if render_chunk is None else int(render_chunk)
        p["max_draws_per_scene"] = table.bound(table.n_draws, n_objects, distinct) + (1 if has_plane else 0)
        p["max_chunks_per_scene"] = table.bound(table.n_chunks, n_objects, distinct) + (1 if has_plane else 0)
        p["max_clip_verts_per_scene"] = table.bound(table.n_clip, n_objects, distinct) + (4 if has_plane else 0)
        p["plane_z"] = PLANE_HALF_Z
        p["plane_size"] = plane_size
        p["manual_exposure"] = manual_exposure
        p["light_color"][:3] = light_color
        p["ambient"][:3] = ambient
        self.params = p
        self.shadows = bool(shadows)
        self._set_projection()
        # settle parameters with the sizing hints of the WORST scene the table can produce (no read-back)
        sp = SB.default_params(tabletop=True, pair_contact_budget=int(pair_contact_budget))   # (0: every point, as in PhysX; > 0: the compound manifold reduction of slhip.h)
        sp["max_bodies_per_scene"] = self.n_objects
        sp["max_hulls_per_scene"] = table.bound(table.n_hulls, n_objects, distinct)
        sp["max_hull_verts_per_scene"] = table.bound(table.n_hull_verts, n_objects, distinct)
        # list capacities of the scratch: no scene may lose a hull pair or a contact (settle_caps() says if one did).  The bound on
        # the pairs is what the table's hull counts allow, capped where 16384 scenes of the 21 YCB-like classes never got (the most
        # ever seen: 3 168 candidate pairs in a step -- mug in bowl on banana; 664 contacts with the default pair_contact_budget)
        h = int(sp["max_hulls_per_scene"])
        sp["max_hull_pairs_per_scene"] = max(64, min(4096, h * h // 2))
        # (32768 C2 scenes: at most 2 191 contacts in a step with every point in the solver, 543 with a pair budget of 32)
        sp["max_contacts_per_scene"] = 4096 if int(pair_contact_budget) == 0 else 1024
        for key, env in (("max_hull_pairs_per_scene", "SLHIP_PAIR_CAP"), ("max_contacts_per_scene", "SLHIP_CONTACT_CAP"),
                         ("pair_contact_budget", "SLHIP_PAIR_BUDGET")):      # developer knobs (tools/probes)
            if os.environ.get(env):
                sp[key] = int(os.environ[env])
        self.settle_params = sp
        dev = self.eng.device
        nb = self.n_scenes * self.n_objects

        def buf(n):
            return torch.empty(max(16, int(n)), dtype=torch.uint8, device=dev)

        self.d_bodies = buf(nb * SB.BODY_DTYPE.itemsize)
        self.d_settle_scenes = buf(self.n_scenes * SB.SETTLE_SCENE_DTYPE.itemsize)
        self.d_objects = buf(nb * _abi.SYNTH_OBJECT_DTYPE.itemsize)
        self.d_scenes = buf(self.n_scenes * _abi.SYNTH_SCENE_DTYPE.itemsize)
        self.d_srec = buf(self.n_scenes * _abi.SCENE_DTYPE.itemsize)
        self.d_drec = buf(self.n_scenes * int(p["max_draws_per_scene"]) * _abi.DRAW_DTYPE.itemsize)
        self.d_crec = buf(self.n_scenes * int(p["max_chunks_per_scene"]) * _abi.CHUNK_DTYPE.itemsize)
        self.d_asset_ids = None if self.asset_ids is None else torch.from_numpy(self.asset_ids.view(np.int16).copy()).to(dev)

    # ---- camera (shared by all scenes of the batch; sl.Scene's setters) --------------------------------------
    def _set_projection(self):
        P = self._proto._projection.astype(np.float32)
        self.params["proj"] = P.reshape(-1)
        self.params["proj_inv"] = np.linalg.inv(P.astype(np.float64)).astype(np.float32).reshape(-1)   # render_pass.cpp:73

    def set_camera_intrinsics(self, fx, fy, cx, cy):
        self._proto.set_camera_intrinsics(fx, fy, cx, cy)
        self._set_projection()

    def set_camera_hfov(self, hfov):
        self._proto.set_camera_hfov(hfov)
        self._set_projection()

    # ---- the four steps -----------------------------------------------------------------------------------------
    def _p(self):
        self._prm = np.array(self.params)       # kept alive: the launch copies it by value
        return C.c_void_p(self._prm.ctypes.data)

    @staticmethod
    def _a(t):
        return C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(0)

    def stage(self, scene_id_base=None):
        """Tabletop set-up of every scene (scene.cpp:612-678) on the current stream."""
        if scene_id_base is not None:
            self.params["scene_id_base"] = scene_id_base
        d_assets, _ = self.table.device()
        self.se.hulls_dev()
        stream = torch.cuda.current_stream(self.eng.device).cuda_stream
        with torch.cuda.device(self.eng.device):
            st = self.eng.L.slhip_synth_stage(self._p(), self._a(d_assets), self._a(self.d_asset_ids), self._a(self.d_bodies),
                                              self._a(self.d_settle_scenes), self._a(self.d_objects), self._a(self.d_scenes),
                                              C.c_void_p(stream))
        _abi.check(st, "slhip_synth_stage")

    def settle(self, frames=None):
        """slhip_settle of the whole batch (scene.cpp:720-756) on the current stream."""
        eng, se = self.eng, self.se
        d_hulls, d_verts = se.hulls_dev()
        prm = self.settle_params if frames is None else self.settle_params.copy()
        if frames is not None:
            prm["frames"] = frames
        prm = np.ascontiguousarray(prm)
        stream = torch.cuda.current_stream(eng.device).cuda_stream
        scratch = se.scratch(self.n_scenes, stream, prm)
        with torch.cuda.device(eng.device):
            st = eng.L.slhip_settle(self._a(self.d_settle_scenes), self.n_scenes, self._a(self.d_bodies), self._a(d_hulls),
                                    self._a(d_verts), C.c_void_p(prm.ctypes.data), self._a(scratch), scratch.numel(),
                                    C.c_void_p(stream))
        _abi.check(st, "slhip_settle")
        self._settle_keep = prm
        self._settle_stream = stream

    def check_settled(self):
        """Synchronises the settle stream; raises if the kernel refused a scene (sizing hints) or if a step of some scene offered
        more hull pairs / contacts / body pairs than the lists of the scratch hold (slhip.h: nothing is ever dropped silently --
        the reference has no caps, scene.cpp:738-739).  A caller that gets the second error settles again with larger
        `settle_params` capacities (the batch is staged from counters, so stage() + settle() reproduce it)."""
        self.se.check_status(self.n_scenes, self._settle_stream)
        caps = self.settle_caps()
        if caps["scenes_dropped"]:
            raise RuntimeError("SceneBatch.settle: %d scene(s) lost hull pairs / contacts / body pairs to the list capacities "
                               "(max_hull_pairs_per_scene %d, max_contacts_per_scene %d; the most a step offered: %d / %d): raise "
                               "them in settle_params and settle again (%r)"
                               % (caps["scenes_dropped"], int(self._settle_keep["max_hull_pairs_per_scene"]),
                                  int(self._settle_keep["max_contacts_per_scene"]), caps["max_hull_pairs"], caps["max_contacts"], caps))

    def settle_caps(self):
        """What the list capacities cost the last settle() (slhip_settle_caps, SettleEngine.caps; synchronises the settle stream)."""
        return self.se.caps(self.n_scenes, self._settle_stream, self._settle_keep)

    def place(self):
        """Camera pose, light direction, shadow matrix and the render records of every scene."""
        d_assets, d_templates = self.table.device()
        stream = torch.cuda.current_stream(self.eng.device).cuda_stream
        with torch.cuda.device(self.eng.device):
            st = self.eng.L.slhip_synth_place(self._p(), self._a(d_assets), self._a(d_templates), self._a(self.d_bodies),
                                              self._a(self.d_objects), self._a(self.d_scenes), self._a(self.d_srec),
                                              self._a(self.d_drec), self._a(self.d_crec), C.c_void_p(stream))
        _abi.check(st, "slhip_synth_place")

    @property
    def render_chunk(self):
        return int(self.params["render_chunk"])

    def n_render_chunks(self):
        return (self.n_scenes + self.render_chunk - 1) // self.render_chunk

    def render(self, chunk=0, mask=_abi.OUT_GT6, ssao=True, buffers=None):
        """slhip_render of render chunk `chunk` (scenes [chunk * render_chunk, ...)) on the current stream."""
        rc = self.render_chunk
        s0 = chunk * rc
        B = min(rc, self.n_scenes - s0)
        if B <= 0:
            raise IndexError("render chunk %d out of range" % chunk)
        md, mk, mv = (int(self.params[k]) for k in ("max_draws_per_scene", "max_chunks_per_scene", "max_clip_verts_per_scene"))
        W, H = self.resolution
        self.eng.pool_abi()
        return self.eng.render_device(
            self.d_srec.data_ptr() + s0 * _abi.SCENE_DTYPE.itemsize,
            self.d_drec.data_ptr() + s0 * md * _abi.DRAW_DTYPE.itemsize,
            self.d_crec.data_ptr() + s0 * mk * _abi.CHUNK_DTYPE.itemsize,
            B, B * md, B * mk, B * mv, W, H, mask, ssao=ssao, shadows=self.shadows, buffers=buffers,
            shadow_lights=1)       # the synthesised scenes have one light (k_synth_place)

    def render_chunks(self, mask=_abi.OUT_GT6, ssao=True):
        for c in range(self.n_render_chunks()):
            yield self.render(c, mask, ssao)

    # ---- host views (tests, inspection, hand-over to the per-scene API) ---------------------------------------
    def _host(self, t, dtype, count):
        return np.frombuffer(t.cpu().numpy().tobytes()[:count * dtype.itemsize], dtype=dtype).copy()

    def host_bodies(self):
        return self._host(self.d_bodies, SB.BODY_DTYPE, self.n_scenes * self.n_objects)

    def host_settle_scenes(self):
        return self._host(self.d_settle_scenes, SB.SETTLE_SCENE_DTYPE, self.n_scenes)

    def host_objects(self):
        return self._host(self.d_objects, _abi.SYNTH_OBJECT_DTYPE, self.n_scenes * self.n_objects)

    def host_scenes(self):
        return self._host(self.d_scenes, _abi.SYNTH_SCENE_DTYPE, self.n_scenes)

    def host_render_records(self):
        md, mk = int(self.params["max_draws_per_scene"]), int(self.params["max_chunks_per_scene"])
        return (self._host(self.d_srec, _abi.SCENE_DTYPE, self.n_scenes),
                self._host(self.d_drec, _abi.DRAW_DTYPE, self.n_scenes * md),
                self._host(self.d_crec, _abi.CHUNK_DTYPE, self.n_scenes * mk))

    def scene(self, index, _cache=None):
        """Scene `index` as an ordinary sl.Scene (objects, poses, velocities, camera, light, plane) rebuilt from the
        device records -- the hand-over to the per-scene API (serialize, render with other settings, ...)."""
        from .object import Object
        from .scene import Scene

        c = _cache or {}
        bodies = c.get("bodies") if "bodies" in c else self.host_bodies()
        objs = c.get("objects") if "objects" in c else self.host_objects()
        scs = c.get("scenes") if "scenes" in c else self.host_scenes()
        srec = c.get("srec") if "srec" in c else self._host(self.d_srec, _abi.SCENE_DTYPE, self.n_scenes)
        scene = Scene(self.resolution)
        scene._projection = self._proto._projection.copy()
        for o in range(self.n_objects):
            k = index * self.n_objects + o
            obj = Object(self.table.meshes[int(objs[k]["asset"])])
            if objs[k]["metallic"] >= 0:
                obj._metallic = f32(objs[k]["metallic"])
            if objs[k]["roughness"] >= 0:
                obj._roughness = f32(objs[k]["roughness"])
            scene.add_object(obj)
            obj._pose = bodies[k]["pose"].reshape(4, 4).copy()
            obj._linear_velocity = bodies[k]["lin_vel"][:3].copy()
            obj._angular_velocity = bodies[k]["ang_vel"][:3].copy()
            obj._separation = f32(bodies[k]["separation"])
        scene._background_plane_pose = scs[index]["plane_pose"].reshape(4, 4).copy()
        scene._background_plane_size = np.asarray(self.params["plane_size"], np.float32).copy()
        scene._camera_pose = scs[index]["camera_pose"].reshape(4, 4).copy()
        scene._light_directions[0] = torch.from_numpy(srec[index]["light_dir"][0][:3].copy())
        scene._light_colors[0] = torch.from_numpy(np.asarray(self.params["light_color"][:3], np.float32).copy())
        scene._ambient_light = np.asarray(self.params["ambient"][:3], np.float32).copy()
        scene._manual_exposure = f32(self.params["manual_exposure"])
        return scene
