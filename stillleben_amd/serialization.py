"""Scene (de)serialisation -- 'next' row f3 of SURVEY.md 8f.

The reference writes a Corrade::Utility::Configuration document (`Scene::serialize`
src/scene.cpp:761-796, `Object::serialize` src/object.cpp:384-406, `Mesh::serialize`
src/mesh.cpp:1091-1097) and reads it back with `deserialize` (scene.cpp:798-869, object.cpp:408-452,
mesh.cpp:1099-1115, `MeshCache::load` src/mesh_cache.cpp:21-37).  This module emits and parses the same
document: the same keys in the same order, values before sub-groups, repeated `[light]` / `[object]`
groups, `[object/mesh]` sub-groups, so a scene settled by either side can be rendered by the other.

Value formats (Corrade / Magnum ConfigurationValue, third-party headers that are not in the reference
tree -- stated here so that the assumption is visible): vectors and quaternions are space-separated
components (quaternion `x y z w`), a Matrix4 is written ROW by ROW (`value[col][row]` with the row
loop outside), Range2D is `min.x min.y max.x max.y`, booleans are `true` / `false`.  Deviation:
Corrade prints floats with 6 significant digits; this writer uses 9 so that poses survive a round
trip bit for bit (the reader accepts either).

Quirks kept: `Object::serialize` writes `linear_velocity_limit` but `deserialize` looks for
`linearVelocityLimit` (object.cpp:405 vs :450) -- the reader accepts both; a document with the legacy
`lightPosition` key gets one light of colour (0, 0.8, 0) (scene.cpp:817-821)."""
import numpy as np
import torch

from . import _math as M


# ---- Corrade-style configuration document ---------------------------------------------------------
class Group:
    def __init__(self):
        self.values = []          # [(key, string)]
        self.groups = []          # [(name, Group)]

    def set(self, key, value):
        self.values.append((key, value))

    def add_group(self, name):
        g = Group()
        self.groups.append((name, g))
        return g

    def has(self, key):
        return any(k == key for k, _ in self.values)

    def value(self, key, default=None):
        for k, v in self.values:
            if k == key:
                return v
        return default

    def group(self, name):
        for n, g in self.groups:
            if n == name:
                return g
        return None

    def groups_named(self, name):
        return [g for n, g in self.groups if n == name]

    def dump(self, path=""):
        out = []
        for k, v in self.values:
            out.append("%s=%s" % (k, _quote(v)))
        for n, g in self.groups:
            full = path + "/" + n if path else n
            out.append("[%s]" % full)
            out.extend(g.dump(full))
        return out


def _quote(v):
    # Corrade quotes values that would otherwise lose leading / trailing whitespace or look like a comment
    if v != v.strip() or v.startswith(('"', "#", ";", "[")) or "\n" in v:
        return '"' + v.replace("\\", "\\\\").replace('"', '\\"') + '"'
    return v


def parse(text):
    root = Group()
    stack = [(0, root)]   # (depth of the path, group)
    cur = root
    open_groups = {(): root}
    for raw in text.splitlines():
        line = raw.strip()
        if not line or line[0] in "#;":
            continue
        if line[0] == "[" and line[-1] == "]":
            path = tuple(line[1:-1].split("/"))
            parent = open_groups.get(path[:-1])
            if parent is None:
                raise ValueError("configuration: group %s has no parent" % line)
            cur = parent.add_group(path[-1])
            open_groups[path] = cur          # later sub-groups attach to the most recent group of that path
            continue
        if "=" not in line:
            raise ValueError("configuration: cannot parse line %r" % raw)
        k, v = line.split("=", 1)
        v = v.strip()
        if len(v) >= 2 and v[0] == '"' and v[-1] == '"':
            v = v[1:-1].replace('\\"', '"').replace("\\\\", "\\")
        cur.set(k.strip(), v)
    del stack
    return root


# ---- value formats ----------------------------------------------------------------------------------
def _f(x):
    return "%.9g" % float(x)


def _vec(v):
    return " ".join(_f(x) for x in np.asarray(v, dtype=np.float64).reshape(-1))


def _mat4(m):
    return _vec(np.asarray(m, dtype=np.float64).reshape(4, 4))   # numpy row-major == row by row


def _floats(s, n=None):
    a = np.array([float(x) for x in s.split()], dtype=np.float32)
    if n is not None and a.size != n:
        raise ValueError("configuration: expected %d numbers, got %r" % (n, s))
    return a


def _bool(s):
    return s.strip().lower() in ("true", "1", "yes", "y")


# ---- writer -------------------------------------------------------------------------------------------
def _serialize_mesh(mesh, g):
    g.set("filename", mesh._filename)
    g.set("classIndex", "%d" % mesh._class_index)
    g.set("scale", _f(mesh._scale))
    g.set("rigidPretransform", _mat4(mesh._pretransform_rigid))


def _serialize_object(obj, g):
    _serialize_mesh(obj._mesh, g.add_group("mesh"))
    g.set("pose", _mat4(obj._pose))
    g.set("instanceIndex", "%d" % obj._instance_index)
    g.set("specularColor", _vec(obj._specular_color))
    g.set("shininess", _f(obj._shininess))
    g.set("roughness", _f(obj._roughness))
    g.set("metallic", _f(obj._metallic))
    g.set("casts_shadows", "true" if obj._casts_shadows else "false")
    sr = obj._sticker_range if obj._sticker_range is not None else np.zeros(4, np.float32)
    g.set("stickerRange", _vec(sr))
    sq = obj._sticker_rotation if obj._sticker_rotation is not None else np.array([0, 0, 0, 1], np.float32)
    g.set("stickerRotation", _vec(sq))
    g.set("static", "true" if obj._static else "false")
    g.set("density", _f(obj._density))
    g.set("linear_velocity_limit", _f(obj._linear_velocity_limit))


def to_document(scene):
    root = Group()
    root.set("viewport", "%d %d" % scene._viewport)
    root.set("projection", _mat4(scene._projection))
    root.set("cameraPosition", _vec(scene._camera_pose[:3, 3]))
    root.set("cameraRotation", _vec(M.matrix_to_quat(scene._camera_pose[:3, :3])))   # x y z w
    for i in range(scene._light_directions.shape[0]):
        lg = root.add_group("light")
        lg.set("direction", _vec(scene._light_directions[i].numpy()))
        lg.set("color", _vec(scene._light_colors[i].numpy()))
    root.set("ambientLight", _vec(scene._ambient_light))
    root.set("numObjects", "%d" % len(scene._objects))
    for obj in scene._objects:
        _serialize_object(obj, root.add_group("object"))
    if scene._light_map is not None:
        root.set("lightMap", str(getattr(scene._light_map, "path", scene._light_map)))
    root.set("backgroundPlanePose", _mat4(scene._background_plane_pose))
    root.set("backgroundPlaneSize", _vec(scene._background_plane_size))
    root.set("manualExposure", _f(scene._manual_exposure))
    return root


def serialize(scene):
    return "\n".join(to_document(scene).dump()) + "\n"


# ---- reader -------------------------------------------------------------------------------------------
def _load_mesh(g, cache):
    """MeshCache::load (mesh_cache.cpp:21-37): the group is only applied when the mesh is first loaded."""
    filename = g.value("filename")
    if filename is None:
        raise RuntimeError("Did not find a filename in the mesh group")
    known = cache._meshes.get(str(filename))
    if known is not None:
        return known
    mesh = cache.load(filename)
    if g.has("classIndex"):
        mesh.class_index = int(g.value("classIndex"))
    if g.has("scale"):
        mesh._scale = np.float32(float(g.value("scale")))
    if g.has("rigidPretransform"):
        mesh._pretransform_rigid = _floats(g.value("rigidPretransform"), 16).reshape(4, 4).copy()
    mesh._update_pretransform()
    return mesh


def _deserialize_object(g, cache):
    from .object import Object

    mg = g.group("mesh")
    if mg is None:
        raise RuntimeError("Did not find mesh subgroup in object")   # object.cpp:411-412
    obj = Object(_load_mesh(mg, cache))
    if g.has("pose"):
        obj._pose = _floats(g.value("pose"), 16).reshape(4, 4).copy()
    if g.has("instanceIndex"):
        obj._instance_index = int(g.value("instanceIndex"))
    if g.has("specularColor"):
        obj._specular_color = _floats(g.value("specularColor"), 4)
    if g.has("shininess"):
        obj._shininess = np.float32(float(g.value("shininess")))
    if g.has("roughness"):
        obj._roughness = np.float32(float(g.value("roughness")))
    if g.has("metallic"):
        obj._metallic = np.float32(float(g.value("metallic")))
    if g.has("casts_shadows"):
        obj._casts_shadows = _bool(g.value("casts_shadows"))
    if g.has("stickerRange"):
        obj._sticker_range = _floats(g.value("stickerRange"), 4)
    if g.has("stickerRotation"):
        obj._sticker_rotation = _floats(g.value("stickerRotation"), 4)
    if g.has("static"):
        obj._static = _bool(g.value("static"))
    if g.has("density"):
        obj._density = np.float32(float(g.value("density")))
    for key in ("linearVelocityLimit", "linear_velocity_limit"):   # reference reads the former, writes the latter
        if g.has(key):
            obj._linear_velocity_limit = np.float32(float(g.value(key)))
    return obj


def from_document(scene, root, cache=None):
    from .extras import MeshCache

    if root.has("viewport"):
        scene._viewport = tuple(int(x) for x in root.value("viewport").split())
    if root.has("projection"):
        scene._projection = _floats(root.value("projection"), 16).reshape(4, 4).copy()
    if root.has("cameraPosition") and root.has("cameraRotation"):
        q = _floats(root.value("cameraRotation"), 4)
        scene._camera_pose = M.from_rt(M.quat_to_matrix(q), _floats(root.value("cameraPosition"), 3))
    if root.has("lightPosition"):
        p = _floats(root.value("lightPosition"), 3)
        scene.light_directions = torch.from_numpy(-M.normalized(p)).reshape(1, 3)
        scene.light_colors = torch.tensor([[0.0, 0.8, 0.0]])
    else:
        lights = root.groups_named("light")
        scene.light_directions = torch.tensor([list(_floats(g.value("direction", "0 0 0"), 3)) for g in lights]).reshape(-1, 3)
        scene.light_colors = torch.tensor([list(_floats(g.value("color", "0 0 0"), 3)) for g in lights]).reshape(-1, 3)
    if root.has("ambientLight"):
        scene._ambient_light = _floats(root.value("ambientLight"), 3)
    if root.has("lightMap"):
        scene.light_map = root.value("lightMap")      # raises: image-based lighting is row f1
    if root.has("backgroundPlanePose"):
        scene._background_plane_pose = _floats(root.value("backgroundPlanePose"), 16).reshape(4, 4).copy()
    if root.has("backgroundPlaneSize"):
        scene._background_plane_size = _floats(root.value("backgroundPlaneSize"), 2)
    if root.has("manualExposure"):
        scene._manual_exposure = np.float32(float(root.value("manualExposure")))
    cache = cache or MeshCache()
    for o in list(scene._objects):
        scene.remove_object(o)
    for g in root.groups_named("object"):
        scene.add_object(_deserialize_object(g, cache))


def deserialize(scene, text, cache=None):
    from_document(scene, parse(text), cache)
