"""Import-compatibility shims for the parts of the reference API that are OUTSIDE the hot path
(SURVEY.md section 2 rows 9, 20-23): they keep user scripts importable and fail (or warn)
explicitly when used."""
import warnings

import numpy as np


class Texture2D:
    """2D texture from a file path or an HxWx3/4 uint8 tensor (python/src/py_magnum.cpp:117-197).
    Used for Scene.background_plane_texture."""

    def __init__(self, src):
        if isinstance(src, (str, bytes)) or hasattr(src, "__fspath__"):
            from PIL import Image

            arr = np.asarray(Image.open(str(src)).convert("RGBA"), dtype=np.uint8)
        else:
            if hasattr(src, "detach"):
                src = src.detach().cpu().numpy()
            arr = np.asarray(src, dtype=np.uint8)
            if arr.ndim != 3 or arr.shape[2] not in (3, 4):
                raise ValueError("expected an HxWx3 or HxWx4 uint8 image")
            if arr.shape[2] == 3:
                arr = np.concatenate([arr, np.full(arr.shape[:2] + (1,), 255, np.uint8)], axis=2)
        self._rgba = np.ascontiguousarray(arr)


class Texture(Texture2D):
    """Rectangle texture (background images) -- kept for import compatibility."""


class _OutOfScope:
    _what = "this component"

    def __init__(self, *a, **k):
        raise NotImplementedError("%s is outside the hot-path scope of stillleben_amd (SURVEY.md section 2)" % self._what)


from .light_map import LightMap  # noqa: E402,F401  (image-based lighting, 'next' row f1)


class Viewer(_OutOfScope):
    _what = "the interactive X11 viewer"


class ImageLoader(_OutOfScope):
    _what = "ImageLoader (dataset I/O)"


class ImageSaver(_OutOfScope):
    _what = "ImageSaver (dataset I/O)"


class Animator(_OutOfScope):
    _what = "Animator"


class MeshCache:
    """filename -> Mesh cache used by Scene.deserialize (reference src/mesh_cache.cpp:21-46)."""

    def __init__(self):
        self._meshes = {}

    def add(self, meshes):
        for m in (meshes if isinstance(meshes, (list, tuple)) else [meshes]):
            self._meshes[m.filename] = m

    def load(self, filename):
        from .mesh import Mesh

        m = self._meshes.get(str(filename))
        if m is None:
            m = Mesh(filename)
            self._meshes[str(filename)] = m
        return m


def view(scene):
    """The reference opens an interactive viewer (python/src/py_viewer.cpp:20-57); headless
    here, so that examples/ycb.py:77 keeps running."""
    warnings.warn("sl.view(): no interactive viewer in stillleben_amd (headless); continuing")


def render_debug_image(scene):
    raise NotImplementedError("render_debug_image is outside the hot-path scope")
