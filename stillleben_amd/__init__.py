"""stillleben_amd -- MI355X-native (gfx950, HIP) implementation of stillleben's hot path:
tabletop settling (replaces PhysX) + G-buffer rendering (replaces Magnum/OpenGL) + sl.diff,
behind the reference's Python surface (python/stillleben/__init__.py:15-42).

    import stillleben_amd as sl        # or `import stillleben as sl` via the alias package
"""
import os
import warnings

import torch  # noqa: F401  (device memory + streams)

from ._context import init, init_cuda, _set_install_prefix  # noqa: F401
from ._math import quat_to_matrix as _q2m, matrix_to_quat as _m2q
from .mesh import Mesh, Range3D  # noqa: F401
from .object import Object  # noqa: F401
from .scene import Scene  # noqa: F401
from .render_pass import RenderPass, RenderPassResult  # noqa: F401
from .extras import (Animator, ImageLoader, ImageSaver, LightMap, MeshCache, Texture, Texture2D,  # noqa: F401
                     Viewer, view, render_debug_image)
from .manipulation_sim import ManipulationSim  # noqa: F401
from .job_queue import JobQueue  # noqa: F401
from .scene_batch import AssetTable, SceneBatch  # noqa: F401  (additive: the batch dimension of the GPU path)
from . import camera_model, diff, losses, extension, profiling  # noqa: F401

__all__ = [
    'init', 'init_cuda', 'render_debug_image', 'Animator', 'ImageLoader', 'ImageSaver', 'LightMap',
    'Mesh', 'MeshCache', 'Object', 'Range3D', 'RenderPass', 'RenderPassResult', 'Scene', 'Texture',
    'Texture2D', 'Viewer', 'view', 'ManipulationSim', 'JobQueue', 'AssetTable', 'SceneBatch',
    'camera_model', 'diff', 'extension', 'losses', 'quat_to_matrix', 'matrix_to_quat',
]


def quat_to_matrix(q):
    """[x y z w] -> 3x3 rotation (reference python/src/py_magnum.cpp:83-97)."""
    if hasattr(q, "detach"):
        q = q.detach().cpu().numpy()
    return torch.from_numpy(_q2m(q))


def matrix_to_quat(m):
    if hasattr(m, "detach"):
        m = m.detach().cpu().numpy()
    return torch.from_numpy(_m2q(m))


STILLLEBEN_PATH = os.path.dirname(os.path.abspath(__file__))
