"""Scene (de)serialisation -- 'next' row f3 of SURVEY.md 8f (reference src/scene.cpp:761-869,
src/object.cpp:384-452, src/mesh.cpp:1091-1115).  The reference writes a Corrade INI document;
this is the same document structure (groups [object], [mesh], [light]; the same keys) emitted
and parsed with a minimal INI reader, enough for a round trip within this package."""
import numpy as np
import torch


def _fmt(v):
    a = np.asarray(v, dtype=np.float64).reshape(-1)
    return " ".join(repr(float(x)) for x in a)


def serialize(scene):
    L = []
    L.append("viewport=%d %d" % scene._viewport)
    L.append("projection=" + _fmt(scene._projection.T))      # Magnum matrices are column-major
    L.append("cameraPose=" + _fmt(scene._camera_pose.T))
    L.append("ambientLight=" + _fmt(scene._ambient_light))
    L.append("backgroundPlanePose=" + _fmt(scene._background_plane_pose.T))
    L.append("backgroundPlaneSize=" + _fmt(scene._background_plane_size))
    L.append("manualExposure=" + repr(float(scene._manual_exposure)))
    for i in range(scene._light_directions.shape[0]):
        L.append("[light]")
        L.append("direction=" + _fmt(scene._light_directions[i].numpy()))
        L.append("color=" + _fmt(scene._light_colors[i].numpy()))
    for obj in scene._objects:
        m = obj._mesh
        L.append("[object]")
        L.append("pose=" + _fmt(obj._pose.T))
        L.append("instanceIndex=%d" % obj._instance_index)
        L.append("metallic=" + repr(float(obj._metallic)))
        L.append("roughness=" + repr(float(obj._roughness)))
        L.append("static=%s" % ("true" if obj._static else "false"))
        L.append("density=" + repr(float(obj._density)))
        L.append("[object/mesh]")
        L.append("filename=" + m._filename)
        L.append("classIndex=%d" % m._class_index)
        L.append("scale=" + repr(float(m._scale)))
        L.append("rigidPretransform=" + _fmt(m._pretransform_rigid.T))
    return "\n".join(L) + "\n"


def _mat(s):
    return np.array([float(x) for x in s.split()], dtype=np.float32).reshape(4, 4).T.copy()


def deserialize(scene, text, cache=None):
    from .extras import MeshCache
    from .object import Object

    cache = cache or MeshCache()
    scene._objects = []
    section, cur, lights = "", None, []
    objs = []
    for line in text.splitlines():
        line = line.strip()
        if not line:
            continue
        if line.startswith("["):
            section = line.strip("[]")
            if section == "object":
                cur = {"mesh": {}}
                objs.append(cur)
            elif section == "light":
                lights.append({})
            continue
        k, v = line.split("=", 1)
        if section == "":
            if k == "viewport":
                scene._viewport = tuple(int(x) for x in v.split())
            elif k == "projection":
                scene._projection = _mat(v)
            elif k == "cameraPose":
                scene._camera_pose = _mat(v)
            elif k == "ambientLight":
                scene._ambient_light = np.array([float(x) for x in v.split()], np.float32)
            elif k == "backgroundPlanePose":
                scene._background_plane_pose = _mat(v)
            elif k == "backgroundPlaneSize":
                scene._background_plane_size = np.array([float(x) for x in v.split()], np.float32)
            elif k == "manualExposure":
                scene._manual_exposure = np.float32(float(v))
        elif section == "light":
            lights[-1][k] = [float(x) for x in v.split()]
        elif section == "object":
            cur[k] = v
        elif section == "object/mesh":
            cur["mesh"][k] = v
    for i, l in enumerate(lights[:3]):
        scene._light_directions[i] = torch.tensor(l.get("direction", [0, 0, 0]), dtype=torch.float32)
        scene._light_colors[i] = torch.tensor(l.get("color", [0, 0, 0]), dtype=torch.float32)
    for o in objs:
        md = o["mesh"]
        mesh = cache.load(md["filename"])
        if "classIndex" in md:
            mesh.class_index = int(md["classIndex"])
        if "scale" in md:
            mesh._scale = np.float32(float(md["scale"]))
        if "rigidPretransform" in md:
            mesh._pretransform_rigid = _mat(md["rigidPretransform"])
        mesh._update_pretransform()
        obj = Object(mesh)
        obj._pose = _mat(o["pose"])
        obj._instance_index = int(o.get("instanceIndex", 0))
        obj._metallic = np.float32(float(o.get("metallic", -1.0)))
        obj._roughness = np.float32(float(o.get("roughness", -1.0)))
        obj._static = o.get("static", "false") == "true"
        obj._density = np.float32(float(o.get("density", 1000.0)))
        scene._objects.append(obj)
        obj._scene = scene
