"""sl.Mesh -- host-side mirror of the reference's Mesh (include/stillleben/mesh.h:47-304,
src/mesh.cpp, python/src/py_mesh.cpp:25-67,357-514)."""
import enum
import os
import sys
from concurrent.futures import ThreadPoolExecutor

import numpy as np
import torch

from . import _loaders
from ._math import as_mat4, f32


class Range3D:
    """Axis-aligned box (Magnum::Range3D as exposed by python/src/py_magnum.cpp)."""

    def __init__(self, mn, mx):
        self._min = np.asarray(mn, dtype=np.float32)
        self._max = np.asarray(mx, dtype=np.float32)

    @property
    def min(self):
        return torch.from_numpy(self._min.copy())

    @property
    def max(self):
        return torch.from_numpy(self._max.copy())

    @property
    def center(self):
        return torch.from_numpy(((self._min + self._max) / f32(2.0)).astype(np.float32))

    @property
    def size(self):
        return torch.from_numpy((self._max - self._min).astype(np.float32))

    @property
    def diagonal(self):
        d = (self._max - self._min).astype(np.float32)
        return float(np.sqrt(np.dot(d, d)))

    # numpy-side helpers used inside the package
    def np_center(self):
        return ((self._min + self._max) / f32(2.0)).astype(np.float32)

    def np_size(self):
        return (self._max - self._min).astype(np.float32)

    def np_diagonal(self):
        d = self.np_size()
        return f32(np.sqrt(np.dot(d, d)))

    def corners(self):
        mn, mx = self._min, self._max
        return np.array([[x, y, z] for z in (mn[2], mx[2]) for y in (mn[1], mx[1]) for x in (mn[0], mx[0])],
                        dtype=np.float32)

    def __repr__(self):
        return "Range3D(min=%s, max=%s)" % (self._min.tolist(), self._max.tolist())


class Mesh:
    class Flag(enum.IntFlag):
        NONE = 0
        PHYSICS_FORCE_CONVEX_HULL = 1

    def __init__(self, filename, visual=True, physics=True, flags=Flag.NONE):
        from ._context import require_context

        require_context()
        self._filename = str(filename)  # accepts str or pathlib.Path (py_mesh.cpp:25-34)
        self._flags = Mesh.Flag(int(flags))
        self._data = _loaders.load_any(self._filename)
        self._class_index = 1  # mesh.h:300
        self._scale = f32(1.0)
        self._pretransform_rigid = np.eye(4, dtype=np.float32)
        self._pretransform = np.eye(4, dtype=np.float32)
        self._raw_bbox = None
        self._hulls = None
        self._slot = None       # engine registration (render)
        self._version = 0       # bumped when vertex data changes
        pre = self._filename + ".pretransform"
        self._update_bounding_box()
        if os.path.exists(pre):  # mesh.cpp:888-921
            self.pretransform = np.loadtxt(pre, dtype=np.float32).reshape(4, 4)
        if physics:
            self._load_physics()

    @classmethod
    def from_data(cls, data, hulls=None, filename="memory://mesh"):
        """Additive API: wraps an in-memory consolidated mesh (and optionally its convex hulls);
        used by stillleben_amd.synthetic."""
        from ._context import require_context

        require_context()
        self = cls.__new__(cls)
        self._filename = filename
        self._flags = Mesh.Flag.NONE
        self._data = data
        self._class_index = 1
        self._scale = f32(1.0)
        self._pretransform_rigid = np.eye(4, dtype=np.float32)
        self._pretransform = np.eye(4, dtype=np.float32)
        self._hulls = list(hulls) if hulls is not None else None
        self._slot = None
        self._version = 0
        self._update_bounding_box()
        return self

    @staticmethod
    def load_threaded(filenames, visual=True, physics=True, flags=()):
        flags = list(flags)
        if flags and len(flags) != len(filenames):
            raise ValueError("flags must be empty or have the same length as filenames")

        def job(i):
            try:
                return Mesh(filenames[i], visual, physics, flags[i] if flags else Mesh.Flag.NONE)
            except Exception as e:   # mesh.cpp:972-975: report, keep loading the others, fail at the end
                print("Could not load file %s: %s" % (filenames[i], e), file=sys.stderr)
                return None

        from ._context import require_context

        require_context()
        with ThreadPoolExecutor(max_workers=max(1, (os.cpu_count() or 2))) as ex:
            meshes = list(ex.map(job, range(len(filenames))))
        if any(m is None for m in meshes):
            raise RuntimeError("Could not load one of the meshes")   # mesh.cpp:989-990
        return meshes

    # ---- geometry ------------------------------------------------------------------------
    def _update_bounding_box(self):  # mesh.cpp:1001-1018
        p = self._data.positions
        self._raw_bbox = (p.min(axis=0).astype(np.float32), p.max(axis=0).astype(np.float32))

    def _update_pretransform(self):  # mesh.cpp:1045-1048
        s = np.diag([self._scale, self._scale, self._scale, f32(1.0)]).astype(np.float32)
        self._pretransform = (s @ self._pretransform_rigid).astype(np.float32)

    @property
    def bbox(self):
        # quirk q6: only min and max corners are transformed (mesh.cpp:1075-1081)
        m = self._pretransform
        lo = m[:3, :3] @ self._raw_bbox[0] + m[:3, 3]
        hi = m[:3, :3] @ self._raw_bbox[1] + m[:3, 3]
        return Range3D(lo, hi)

    def center_bbox(self):  # mesh.cpp:1020-1024
        c = (self._raw_bbox[0] + self._raw_bbox[1]) / f32(2.0)
        self._pretransform_rigid[:3, 3] = -(self._pretransform_rigid[:3, :3] @ c)
        self._update_pretransform()

    def scale_to_bbox_diagonal(self, target_diagonal, mode="exact"):  # mesh.cpp:1026-1043
        d = (self._raw_bbox[1] - self._raw_bbox[0]).astype(np.float32)
        diagonal = f32(np.sqrt(np.dot(d, d)))
        scale = f32(target_diagonal) / diagonal
        if mode == "exact":
            self._scale = f32(scale)
        elif mode == "order_of_magnitude":
            self._scale = f32(10.0 ** np.round(np.log10(float(scale))))
        else:
            raise ValueError("invalid value for mode argument")
        self._update_pretransform()

    @property
    def pretransform(self):
        return torch.from_numpy(self._pretransform.copy())

    @pretransform.setter
    def pretransform(self, m):  # mesh.cpp:1050-1073
        m = as_mat4(m)
        u, w, vt = np.linalg.svd(m[:3, :3].astype(np.float64))
        if w.max() - w.min() > 1e-5:
            raise ValueError("Scaling is not uniform")
        self._scale = f32((w.max() + w.min()) / 2.0)
        self._pretransform_rigid = np.eye(4, dtype=np.float32)
        self._pretransform_rigid[:3, :3] = (u @ vt).astype(np.float32)
        self._pretransform_rigid[:3, 3] = (f32(1.0) / self._scale) * m[:3, 3]
        self._update_pretransform()

    @property
    def class_index(self):
        return self._class_index

    @class_index.setter
    def class_index(self, v):
        v = int(v)
        if v < 0 or v > 65535:  # mesh.cpp:1083-1089
            raise ValueError("Mesh::setClassIndex(): out of range")
        self._class_index = v

    @property
    def filename(self):
        return self._filename

    @property
    def points(self):
        return torch.from_numpy(self._data.positions.copy())

    @property
    def normals(self):
        return torch.from_numpy(self._data.normals.copy())

    @property
    def faces(self):
        return torch.from_numpy(self._data.indices.astype(np.int32))

    @property
    def colors(self):
        return torch.from_numpy(self._data.colors.copy())

    # ---- vertex mutation (mesh.cpp:747-886); vertex_indices are the 1-based ids the
    # renderer outputs (mesh.cpp:834 subtracts 1) ---------------------------------------
    def _np(self, t, shape_tail):
        if hasattr(t, "detach"):
            t = t.detach().cpu().numpy()
        a = np.asarray(t)
        if shape_tail is not None:
            a = a.reshape((-1,) + shape_tail)
        return a

    def _touch(self):
        self._version += 1
        self._update_bounding_box()

    def _recompute_normals(self):
        """Mesh::recomputeNormals (mesh.cpp:763-815): per vertex, the normalised sum over its faces of
        (unit face normal x face area), faces visited in index order; a face counts once per corner."""
        d = self._data
        idx = d.indices.astype(np.int64).reshape(-1, 3)
        p = d.positions
        v1, v2, v3 = p[idx[:, 0]], p[idx[:, 1]], p[idx[:, 2]]
        cr = np.cross(v1 - v2, v1 - v3).astype(np.float32)
        area = np.sqrt((cr * cr).sum(axis=1, dtype=np.float32)).astype(np.float32)
        with np.errstate(all="ignore"):
            contrib = ((cr / area[:, None]).astype(np.float32) * area[:, None]).astype(np.float32)
        acc = np.zeros_like(p)
        # np.add.at accumulates in index order (unbuffered): face 0's corners, face 1's, ... -- every vertex sums its faces in
        # face order, the reference's float32 summation order
        np.add.at(acc, idx.reshape(-1), np.repeat(contrib, 3, axis=0))
        ln = np.sqrt((acc * acc).sum(axis=1, dtype=np.float32)).astype(np.float32)
        with np.errstate(all="ignore"):
            d.normals = (acc / ln[:, None]).astype(np.float32)

    @staticmethod
    def _check_update(vertex_indices, update, width, what):
        # argument checks of Mesh_updatePositions / Mesh_updateColors (py_mesh.cpp:69-160)
        vi = vertex_indices if hasattr(vertex_indices, "dim") else torch.as_tensor(np.asarray(vertex_indices))
        up = update if hasattr(update, "dim") else torch.as_tensor(np.asarray(update))
        if vi.dim() != 1:
            raise ValueError("vertex_indices (1st argument) should be one dimensional")
        if up.dim() != 2:
            raise ValueError("%s should be two dimensional" % what)
        if vi.shape[0] != up.shape[0]:
            raise ValueError("vertex_indices and %s should be of same size" % what)
        if up.shape[1] != width:
            raise ValueError("%s should be of shape (N,%d)" % (what, width))
        if vi.device.type != "cpu" or up.device.type != "cpu":
            raise ValueError("vertex_indices and %s should be CPU tensors" % what)
        return vi.detach().numpy().astype(np.int64) - 1, up.detach().numpy().astype(np.float32)

    def _index_range_check(self, vi):
        if len(vi) and (vi.min() < 0 or vi.max() >= len(self._data.positions)):
            raise ValueError("vertex index out of range (ids are 1-based, as the renderer writes them)")

    def update_positions(self, vertex_indices, position_update):
        """Mesh::updateVertexPositions (mesh.cpp:823-841): ADDS the update to the addressed vertices --
        duplicate ids accumulate, in argument order -- then recomputes the normals."""
        vi, upd = self._check_update(vertex_indices, position_update, 3, "position_update")
        self._index_range_check(vi)
        np.add.at(self._data.positions, vi, upd)
        self._recompute_normals()
        self._touch()

    def update_colors(self, vertex_indices, color_update):
        """Mesh::updateVertexColors (mesh.cpp:843-852): colour += update (Nx4)."""
        vi, upd = self._check_update(vertex_indices, color_update, 4, "color_update")
        self._index_range_check(vi)
        np.add.at(self._data.colors, vi, upd)
        self._touch()

    def update_positions_and_colors(self, vertex_indices, position_update, color_update):
        vi, upd = self._check_update(vertex_indices, position_update, 3, "position_update")
        _, cupd = self._check_update(vertex_indices, color_update, 4, "color_update")
        self._index_range_check(vi)
        np.add.at(self._data.positions, vi, upd)
        self._recompute_normals()
        np.add.at(self._data.colors, vi, cupd)
        self._touch()

    def set_new_positions(self, new_positions):
        """Mesh::setVertexPositions (mesh.cpp:857-871)."""
        p = self._np(new_positions, (3,)).astype(np.float32)
        if p.shape != self._data.positions.shape:
            raise ValueError("Number of new vertices should match the existing mesh vertices")
        self._data.positions = p.copy()
        self._recompute_normals()
        self._touch()

    def set_new_colors(self, new_colors):
        """Mesh::setVertexColors (mesh.cpp:873-886)."""
        c = self._np(new_colors, (4,)).astype(np.float32)
        if c.shape != self._data.colors.shape:
            raise ValueError("Number of new vertices should match the existing mesh vertices for vertex color update")
        self._data.colors = c.copy()
        self._touch()

    # ---- physics shapes (mesh.cpp:304-533) -------------------------------------------------
    def _load_physics(self):
        if self._hulls is None:
            from . import hulls

            self._hulls = hulls.hulls_for_mesh(self)
        return self._hulls

    @property
    def physics_mesh_data(self):
        out = []
        for h in self._load_physics():
            out.append({"positions": torch.from_numpy(h.vertices.copy()),
                        "indices": torch.from_numpy(h.triangles.astype(np.int32))})
        return out

    def dump_physics_meshes(self, directory):
        os.makedirs(directory, exist_ok=True)
        for i, h in enumerate(self._load_physics()):
            with open(os.path.join(directory, "physics_%03d.ply" % i), "w") as f:
                f.write("ply\nformat ascii 1.0\nelement vertex %d\nproperty float x\nproperty float y\n"
                        "property float z\nelement face %d\nproperty list uchar int vertex_indices\nend_header\n"
                        % (len(h.vertices), len(h.triangles)))
                for v in h.vertices:
                    f.write("%g %g %g\n" % tuple(v))
                for t in h.triangles:
                    f.write("3 %d %d %d\n" % tuple(t))
