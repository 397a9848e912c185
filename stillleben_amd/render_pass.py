"""sl.RenderPass / sl.RenderPassResult (reference include/stillleben/render_pass.h:35-153,
python/src/py_render_pass.cpp:81-279)."""

from . import _abi
from ._context import engine, require_context


class RenderPassResult:
    """Accessors return the dtypes/shapes of the reference (py_render_pass.cpp:20-69):
    rgb u8[H,W,4]; class/instance int16[H,W,1] (R16UI bits reinterpreted); coordinates f32[H,W,3];
    depth f32[H,W]; coordDepth/normals/cam_coordinates f32[H,W,4]; vertex_indices int32[H,W,3];
    barycentric_coeffs f32[H,W,3]."""

    def __init__(self):
        require_context()
        self._buffers = None
        self._index = 0

    def _get(self, name):
        if self._buffers is None:
            raise RuntimeError("RenderPassResult is empty: render something first")
        t = getattr(self._buffers, name)
        if t is None:
            raise RuntimeError("output '%s' was not requested for this render" % name)
        t = t[self._index]
        # fresh tensor per call, like extract() (py_magnum.cpp:17-46)
        return t.clone() if require_context().cuda_outputs else t.cpu()

    def rgb(self):
        return self._get("rgb")

    def class_index(self):
        return self._get("cls")

    def instance_index(self):
        return self._get("instance")

    def coordinates(self):
        return self._get("coord")[:, :, 0:3]

    def depth(self):
        return self._get("coord")[:, :, 3]

    def coordDepth(self):  # noqa: N802 (reference name)
        return self._get("coord")

    def normals(self):
        return self._get("normals")

    def vertex_indices(self):
        return self._get("vertex_idx")[:, :, 0:3]

    def barycentric_coeffs(self):
        return self._get("bary")[:, :, 0:3]

    def cam_coordinates(self):
        return self._get("cam_coord")


class RenderPass:
    def __init__(self, shading="pbr"):
        require_context()
        if shading not in ("pbr", "phong", "flat"):
            raise ValueError("unknown shading type specified")
        self._shading = shading  # stored, never read by the render path (quirk q2)
        self.ssao_enabled = True  # render_pass.h:150
        self._result = RenderPassResult()
        self._buffers = None

    def render(self, scene, result=None, depth_peel=None, predicate=None):
        res = result if result is not None else self._result
        peel = None
        if depth_peel is not None:
            peel = depth_peel._buffers.coord[depth_peel._index:depth_peel._index + 1].contiguous()
        own = res._buffers if (res._buffers is not None and res._buffers.B == 1 and res is not depth_peel) else None
        res._buffers = engine().render([scene], _abi.OUT_ALL, ssao=self.ssao_enabled, shadows=True,
                                       depth_peel=peel, predicate=predicate, buffers=own)
        res._index = 0
        return res

    def render_batch(self, scenes, outputs=_abi.OUT_ALL, shadows=True):
        """Additive API: renders many scenes in one launch sequence; returns the batched
        [B,H,W,C] buffers (device tensors)."""
        return engine().render(list(scenes), outputs, ssao=self.ssao_enabled, shadows=shadows)
