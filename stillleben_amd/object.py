"""sl.Object -- instance of a Mesh in a Scene (reference include/stillleben/object.h:97-296,
src/object.cpp, python/src/py_object.cpp:23-201)."""
import numpy as np
import torch

from ._math import as_mat4, as_vec, f32


class Object:
    def __init__(self, mesh, options=None):
        from ._context import require_context

        require_context()
        options = options or {}
        self._mesh = mesh
        self._pose = np.eye(4, dtype=np.float32)
        self._instance_index = 0
        self._color = options.get("color")
        self._force_color = bool(options.get("force_color", False))
        self._metallic = f32(-1.0)   # object.h:277-278: negative == use the material's
        self._roughness = f32(-1.0)
        self._specular_color = np.ones(4, dtype=np.float32)   # dead parameter (quirk q2)
        self._shininess = f32(80.0)                           # dead parameter (quirk q2)
        self._casts_shadows = True   # object.h:295
        self._static = False
        self._density = f32(1000.0)  # object.h:287
        self._separation = f32(0.0)
        self._stuck_counter = 0
        self._linear_velocity = np.zeros(3, dtype=np.float32)
        self._angular_velocity = np.zeros(3, dtype=np.float32)
        self._linear_velocity_limit = f32(1e16)  # PhysX PX_MAX_F32-ish default: unlimited
        # material: context default (context.cpp:250-252)
        self._static_friction = f32(0.3)
        self._dynamic_friction = f32(0.2)
        self._restitution = f32(0.1)
        self._mass_props = None
        self._scene = None
        # sticker decal (object.h:250-262): range in the sticker frame, rotation of that frame, rectangle texture
        self._sticker_range = None
        self._sticker_rotation = None
        self._sticker_texture = None

    # ---- pose ------------------------------------------------------------------------------
    def pose(self):
        return torch.from_numpy(self._pose.copy())

    def set_pose(self, pose):
        m = as_mat4(pose)
        # with physics the pose must be rigid (object.cpp:349-369)
        det = float(np.linalg.det(m[:3, :3].astype(np.float64)))
        if abs(det - 1.0) > 0.01 and self._mesh._hulls is not None and self._scene is not None \
                and self._scene._physics_loaded:
            raise RuntimeError(
                "You provided a pose which is not a pure rotation / translation:\n%s\n(determinant: %g)\n"
                "This is not supported when using the physics engine." % (m, det))
        self._pose = m

    # ---- simple properties -----------------------------------------------------------------
    @property
    def mesh(self):
        return self._mesh

    @property
    def instance_index(self):
        return self._instance_index

    @instance_index.setter
    def instance_index(self, v):
        v = int(v)
        if v < 0 or v > 65535:  # object.cpp:376-382
            raise ValueError("Object::setInstanceIndex(): out of range")
        self._instance_index = v

    def _float_prop(name):  # noqa: N805
        def g(self):
            return float(getattr(self, name))

        def s(self, v):
            setattr(self, name, f32(v))

        return property(g, s)

    metallic = _float_prop("_metallic")
    roughness = _float_prop("_roughness")
    shininess = _float_prop("_shininess")
    linear_velocity_limit = _float_prop("_linear_velocity_limit")
    static_friction = _float_prop("_static_friction")
    dynamic_friction = _float_prop("_dynamic_friction")
    restitution = _float_prop("_restitution")
    del _float_prop

    @property
    def specular_color(self):
        return torch.from_numpy(self._specular_color.copy())

    @specular_color.setter
    def specular_color(self, v):
        self._specular_color = as_vec(v, 4)

    @property
    def casts_shadows(self):
        return self._casts_shadows

    @casts_shadows.setter
    def casts_shadows(self, v):
        self._casts_shadows = bool(v)

    @property
    def static(self):
        return self._static

    @static.setter
    def static(self, v):
        self._static = bool(v)

    @property
    def separation(self):
        return float(self._separation)

    @property
    def linear_velocity(self):
        return torch.from_numpy(self._linear_velocity.copy())

    @linear_velocity.setter
    def linear_velocity(self, v):
        self._linear_velocity = as_vec(v, 3)

    @property
    def angular_velocity(self):
        return torch.from_numpy(self._angular_velocity.copy())

    @angular_velocity.setter
    def angular_velocity(self, v):
        self._angular_velocity = as_vec(v, 3)

    # ---- sticker decal (object.cpp:480-513; rendered by render_shader.vert:89-94 / frag:248-256) --------
    @property
    def sticker_range(self):
        return self._sticker_range

    @sticker_range.setter
    def sticker_range(self, v):
        """Range2D in the sticker projection frame: (min.x, min.y, max.x, max.y) or a Range2D-like object."""
        if v is None:
            self._sticker_range = None
            return
        if hasattr(v, "min") and hasattr(v, "max"):
            v = list(np.asarray(v.min, dtype=np.float32).reshape(-1)[:2]) + list(np.asarray(v.max, dtype=np.float32).reshape(-1)[:2])
        self._sticker_range = as_vec(v, 4)

    @property
    def sticker_rotation(self):
        return self._sticker_rotation

    @sticker_rotation.setter
    def sticker_rotation(self, q):
        self._sticker_rotation = None if q is None else as_vec(q, 4)   # quaternion x y z w

    @property
    def sticker_texture(self):
        return self._sticker_texture

    @sticker_texture.setter
    def sticker_texture(self, tex):
        self._sticker_texture = tex

    def sticker_view_projection(self):
        """Object::stickerViewProjection (object.cpp:494-513): proj * translate(0,0,1) * rotation."""
        from . import _math as M

        diagonal = self._mesh.bbox.np_diagonal()
        # Magnum's constructor takes COLUMNS: the reference's 4 vectors are the columns of proj
        proj = np.array([[f32(2.0) / diagonal, 0, 0, 0], [0, f32(2.0) / diagonal, 0, 0], [0, 0, 1, 0], [0, 0, 1, 1]], np.float32).T
        trans = np.eye(4, dtype=np.float32)
        trans[2, 3] = 1.0
        rot = np.eye(4, dtype=np.float32)
        q = self._sticker_rotation if self._sticker_rotation is not None else np.array([0, 0, 0, 1], np.float32)
        rot[:3, :3] = M.quat_to_matrix(q)
        return (proj @ trans @ rot).astype(np.float32)

    # ---- mass properties (object.cpp:215-257; PxRigidBodyExt::updateMassAndInertia) --------------
    def _props(self):
        key = self._mass_key()
        if self._mass_props is None or self._mass_props.key != key:
            # the integrals depend on the mesh (and the density) only: objects of one mesh share them
            cache = self._mesh.__dict__.setdefault("_mass_cache", {})
            props = cache.get(key)
            if props is None:
                from . import massprops

                if len(cache) > 16:
                    cache.clear()
                props = cache[key] = massprops.compute(self)
            self._mass_props = props
        return self._mass_props

    def _mass_key(self):
        return (float(self._density), float(self._mesh._scale), self._mesh._pretransform_rigid.tobytes(),
                self._mesh._version)

    @property
    def density(self):
        return float(self._density)

    @density.setter
    def density(self, v):
        self._density = f32(v)

    @property
    def mass(self):
        return float(self._props().mass)

    @mass.setter
    def mass(self, m):  # object.cpp:223-229
        factor = f32(m) / f32(self._props().mass)
        self._density = f32(self._density * factor)

    @property
    def volume(self):
        return float(self._props().mass / self._density)

    @property
    def inertia(self):
        return torch.from_numpy(self._props().inertia_diag.copy())

    @property
    def inertial_frame(self):
        return torch.from_numpy(self._props().inertial_frame.copy())
