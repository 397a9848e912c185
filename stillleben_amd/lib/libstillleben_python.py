"""`libstillleben_python` -- the reference's MAIN extension module by name (python/src/bridge.cpp:23-42,
`PYBIND11_MODULE(libstillleben_python, m)`), imported by the reference's package as `from .lib.libstillleben_python import *`
and `from .lib.libstillleben_python import _set_install_prefix` (python/stillleben/__init__.py:12-13) and by its diff module
(python/stillleben/diff.py:22).

What answers to the name here is a Python module over the C-ABI of libslhip.so (ctypes, `_abi.py`) and its host C++ record
builders (`csrc/slhip_records.cpp`), not pybind11 C++ -- INTEGRATION.md §B says why.  It exports exactly the names the thirteen
`init(m)` calls of bridge.cpp register, nothing more (the additive batch API -- AssetTable, SceneBatch -- lives in the package,
not in this module):

    py_context.cpp:82-102       init, init_cuda, _set_install_prefix
    py_magnum.cpp:51-157        Range3D, quat_to_matrix, matrix_to_quat, Texture, Texture2D
    py_mesh.cpp:314,516         Mesh, MeshCache
    py_object.cpp:23            Object
    py_light_map.cpp:19         LightMap
    py_scene.cpp:48             Scene
    py_render_pass.cpp:81-282   RenderPassResult, RenderPass, render_debug_image
    py_image_loader.cpp:18      ImageLoader
    py_image_saver.cpp:112      ImageSaver
    py_animator.cpp:20          Animator
    py_viewer.cpp:20,48         Viewer, view
    py_job_queue.cpp:18         JobQueue
    py_manipulation_sim.cpp:20  ManipulationSim
"""
import stillleben_amd as _impl  # noqa: E402

__all__ = [
    'init', 'init_cuda',
    'Range3D', 'quat_to_matrix', 'matrix_to_quat', 'Texture', 'Texture2D',
    'Mesh', 'MeshCache', 'Object', 'LightMap', 'Scene',
    'RenderPassResult', 'RenderPass', 'render_debug_image',
    'ImageLoader', 'ImageSaver', 'Animator', 'Viewer', 'view', 'JobQueue', 'ManipulationSim',
]

for _n in __all__ + ['_set_install_prefix']:
    globals()[_n] = getattr(_impl, _n)
del _n
