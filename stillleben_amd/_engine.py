"""Device-side state of the process: the mesh pool in HBM, scratch buffers, and the calls into
the C-ABI (lib/libslhip.so).  torch is used for device memory and streams only."""
import ctypes as C

import numpy as np
import torch

from . import _abi
from ._batch import HostPool, build_batch

SHADOW_RES = 2048  # render_pass.cpp:271
QUEUE_ITEMS_PER_SCENE = int(__import__('os').environ.get('SLHIP_QUEUE_ITEMS', 1 << 14))  # (triangle, 8x8 tile) work items; 4800 tiles cover 640x480 once


def _ptr(t):
    return C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(0)


class RenderBuffers:
    """The 8 render targets of a batch, [B,H,W,C], allocated with torch on the HIP device."""

    def __init__(self, device, B, H, W, mask):
        self.B, self.H, self.W, self.mask = B, H, W, mask

        def mk(bit, shape, dtype):
            return torch.empty(shape, dtype=dtype, device=device) if mask & bit else None

        self.rgb = mk(_abi.OUT_RGB, (B, H, W, 4), torch.uint8)
        self.coord = mk(_abi.OUT_COORD, (B, H, W, 4), torch.float32)
        self.cls = mk(_abi.OUT_CLASS, (B, H, W, 1), torch.int16)
        self.instance = mk(_abi.OUT_INSTANCE, (B, H, W, 1), torch.int16)
        self.normals = mk(_abi.OUT_NORMALS, (B, H, W, 4), torch.float32)
        self.vertex_idx = mk(_abi.OUT_VERTEX_IDX, (B, H, W, 4), torch.int32)
        self.bary = mk(_abi.OUT_BARY, (B, H, W, 4), torch.float32)
        self.cam_coord = mk(_abi.OUT_CAM_COORD, (B, H, W, 4), torch.float32)

    def abi(self):
        o = _abi.RenderOut()
        o.d_rgb, o.d_coord, o.d_class, o.d_instance = _ptr(self.rgb), _ptr(self.coord), _ptr(self.cls), _ptr(self.instance)
        o.d_normals, o.d_vertex_idx, o.d_bary, o.d_cam_coord = _ptr(self.normals), _ptr(self.vertex_idx), _ptr(self.bary), _ptr(self.cam_coord)
        return o


class Engine:
    def __init__(self, device_index):
        L = _abi.lib()
        if L.slhip_abi_version() != _abi.ABI_VERSION:
            raise _abi.SlhipError("libslhip.so ABI version mismatch")
        if not torch.cuda.is_available():
            raise _abi.SlhipError(
                "stillleben_amd needs a HIP device (MI355X / gfx950); none is visible and there is no CPU fallback")
        _abi.check(L.slhip_device_init(device_index), "slhip_device_init")
        self.L = L
        self.device = torch.device("cuda", device_index)
        self.pool = HostPool()
        self._pool_dev = None
        self._scratch = {}
        self._light_maps = []        # registered LightMap objects; slot = index
        self._light_maps_dev = None

    # ---- mesh pool -------------------------------------------------------------------------
    def register_mesh(self, mesh):
        return self.pool.register(mesh)

    def register_light_map(self, lm):
        self._light_maps.append(lm)
        self._light_maps_dev = None
        return len(self._light_maps) - 1

    def pool_abi(self):
        if self.pool.dirty or self._pool_dev is None:
            arrs = self.pool.arrays()
            self._pool_dev = [torch.from_numpy(np.ascontiguousarray(a)).to(self.device) for a in arrs]
            self.pool.dirty = False
        p = _abi.MeshPool()
        d = self._pool_dev
        p.d_pos, p.d_nrm, p.d_uv, p.d_col, p.d_idx, p.d_tex = (_ptr(t) for t in d[:6])
        p.d_tan = _ptr(d[6])
        p.n_vertices, p.n_indices, p.n_tex_bytes = self.pool.n_vertices, self.pool.n_indices, self.pool.n_tex_bytes
        if self._light_maps:
            if self._light_maps_dev is None:
                raw = b"".join(bytes(lm.rec) for lm in self._light_maps)
                self._light_maps_dev = torch.from_numpy(np.frombuffer(raw, dtype=np.uint8).copy()).to(self.device)
            p.d_light_maps, p.n_light_maps = _ptr(self._light_maps_dev), len(self._light_maps)
        return p

    # ---- scratch ---------------------------------------------------------------------------
    def scratch(self, B, H, W, want_rgb, ssao, shadows, stream=0, shadow_lights=_abi.NUM_LIGHTS, keep_hdr=False):
        # per stream: launches on different streams may overlap.  `shadow_lights`: shadow maps per scene (a batch whose scenes
        # only use the first lights need not carry 16.8 MB per scene for every unused one)
        key = (B, H, W, want_rgb, ssao, shadows, stream, shadow_lights, keep_hdr)
        s = self._scratch.get(key)
        if s is None:
            sizes = (C.c_uint64 * 7)()
            qcap = max(1 << 20, B * QUEUE_ITEMS_PER_SCENE)
            self.L.slhip_render_scratch_bytes(B, W, H, SHADOW_RES if shadows else 0, qcap, C.byref(sizes))

            from ._context import check_free_memory

            want = (int(sizes[0]) + (int(sizes[1] if keep_hdr else sizes[1] // 2) if want_rgb else 0) + (int(sizes[2]) if ssao else 0)
                    + (int(sizes[3]) // _abi.NUM_LIGHTS * shadow_lights + int(sizes[6]) if shadows else 0) + int(sizes[4]) + int(sizes[5]))
            check_free_memory(self.device, want, "the render scratch of %d scenes at %d x %d (visibility keys, HDR, SSAO planes, shadow maps)" % (B, W, H))

            def buf(n, need=True):
                return torch.empty(max(int(n), 16), dtype=torch.uint8, device=self.device) if need else None

            s = {
                # (the second half of d_hdr is only written under RENDER_KEEP_HDR)
                "vis": buf(sizes[0]), "hdr": buf(sizes[1] if keep_hdr else sizes[1] // 2, want_rgb), "ao": buf(sizes[2], ssao),
                "shadow": buf(sizes[3] // _abi.NUM_LIGHTS * shadow_lights, shadows), "queue": buf(sizes[4]), "lum": buf(sizes[5], want_rgb),
                "tiles": buf(sizes[6], shadows), "qcap": qcap,
                "shadow_ready": False,    # the shadow maps / tile bits are garbage until the first call has reset them
            }
            if len(self._scratch) > 6:
                self._scratch.clear()
            self._scratch[key] = s
        a = _abi.RenderScratch()
        a.d_vis, a.d_hdr, a.d_ao, a.d_shadow, a.d_queue, a.d_lum = (
            _ptr(s["vis"]), _ptr(s["hdr"]), _ptr(s["ao"]), _ptr(s["shadow"]), _ptr(s["queue"]), _ptr(s["lum"]))
        a.d_shadow_tiles = _ptr(s["tiles"])
        a.queue_capacity = s["qcap"]
        a.shadow_res = SHADOW_RES
        a.shadow_lights = shadow_lights
        return a, s

    def ssao_skipped(self, buffers, W, H):
        """(tiles, tiles skipped) of the SSAO pass of the render that filled `buffers` (slhip_render_ssao_skipped; synchronises)."""
        keep = buffers._keepalive[0]
        a = _abi.RenderScratch()
        a.d_ao = _ptr(keep["ao"])
        out = (C.c_uint64 * 2)()
        stream = torch.cuda.current_stream(self.device).cuda_stream
        with torch.cuda.device(self.device):
            _abi.check(self.L.slhip_render_ssao_skipped(C.byref(a), buffers.B, W, H, C.byref(out), C.c_void_p(stream)), "slhip_render_ssao_skipped")
        return int(out[0]), int(out[1])

    # ---- render ----------------------------------------------------------------------------
    def upload_records(self, arr):
        raw = np.frombuffer(arr.tobytes(), dtype=np.uint8) if arr.size else np.zeros(16, np.uint8)
        return torch.from_numpy(raw.copy()).to(self.device)

    def render(self, scenes, mask=_abi.OUT_ALL, ssao=True, shadows=True, depth_peel=None, predicate=None,
               buffers=None, keep_hdr=False):
        W, H = scenes[0]._viewport
        for s in scenes:
            if s._viewport != (W, H):
                raise ValueError("all scenes of a batch must share one viewport")
        want_rgb = bool(mask & _abi.OUT_RGB)
        srec, drec, crec = build_batch(scenes, self.pool, predicate, with_shadows=shadows and want_rgb)
        return self.render_records(srec, drec, crec, W, H, mask, ssao, shadows, depth_peel, buffers, keep_hdr)

    def render_records(self, srec, drec, crec, W, H, mask=_abi.OUT_ALL, ssao=True, shadows=True, depth_peel=None,
                       buffers=None, keep_hdr=False):
        """Renders a batch described by prebuilt slhip_scene / slhip_draw / slhip_chunk records (host arrays)."""
        d_s, d_d, d_c = self.upload_records(srec), self.upload_records(drec), self.upload_records(crec)
        n_clip = int(drec["n_verts"].sum()) if len(drec) else 0
        # shadow maps for the lights the batch uses (a light with zero colour or direction is off: light_active() of the kernels)
        on = (np.abs(srec["light_color"][:, :, :3]).sum(axis=2) > 0) & (np.abs(srec["light_dir"][:, :, :3]).sum(axis=2) > 0)
        lights = max(1, int(np.max(np.nonzero(on.any(axis=0))[0]) + 1)) if on.any() else 1
        buffers = self.render_device(d_s, d_d, d_c, len(srec), len(drec), len(crec), n_clip, W, H, mask, ssao, shadows,
                                     depth_peel, buffers, keep_hdr, shadow_lights=lights)
        buffers._keepalive += (d_s, d_d, d_c)   # alive until the stream has consumed them
        return buffers

    def render_device(self, d_s, d_d, d_c, B, n_draws, n_chunks, n_clip, W, H, mask=_abi.OUT_ALL, ssao=True, shadows=True,
                      depth_peel=None, buffers=None, keep_hdr=False, shadow_lights=_abi.NUM_LIGHTS):
        """slhip_render on records that already live in HBM (device tensors or raw device addresses):
        `n_clip` = clip-position slots the draws' clip_base + n_verts ranges span; `shadow_lights` = shadow maps per scene
        (lights with a higher index cast no shadow)."""
        want_rgb = bool(mask & _abi.OUT_RGB)
        ssao = ssao and want_rgb
        shadows = shadows and want_rgb
        if ssao:
            mask |= _abi.OUT_CAM_COORD | _abi.OUT_NORMALS
        pool = self.pool_abi()
        if buffers is None or (buffers.B, buffers.H, buffers.W, buffers.mask) != (B, H, W, mask):
            buffers = RenderBuffers(self.device, B, H, W, mask)
        out = buffers.abi()
        stream = torch.cuda.current_stream(self.device).cuda_stream
        scratch, keep = self.scratch(B, H, W, want_rgb, ssao, shadows, stream, shadow_lights if shadows else _abi.NUM_LIGHTS, keep_hdr)
        planes = 1 + (_abi.NUM_LIGHTS if shadows else 0)
        # + 80 B per vertex: the post-transform vertex cache of the raster and shading passes (64-byte records = cache lines,
        # then a dense plane of window coordinates)
        voff = (n_clip * planes * 16 + 63) & ~63
        need = max(16, voff + n_clip * 80)
        clips = self.__dict__.setdefault("_clips", {})
        if clips.get(stream) is None or clips[stream].numel() < need:
            clips[stream] = torch.empty(need, dtype=torch.uint8, device=self.device)
        scratch.d_clip = _ptr(clips[stream])
        scratch.n_clip_verts = n_clip
        scratch.d_vattr = C.c_void_p(clips[stream].data_ptr() + voff)
        flags = mask | (_abi.RENDER_SSAO if ssao else 0) | (_abi.RENDER_SHADOWS if shadows else 0)
        if keep_hdr:
            flags |= _abi.RENDER_KEEP_HDR      # (tests: the float image behind the fused SSAO-apply + tone-map pass)
        if shadows and not keep["shadow_ready"]:
            flags |= _abi.RENDER_SHADOW_RESET
        keep["shadow_ready"] = False      # stays False if the call below raises: the next one resets again

        def addr(t):
            return C.c_void_p(t) if isinstance(t, int) else _ptr(t)

        with torch.cuda.device(self.device):
            st = self.L.slhip_render(C.byref(pool), addr(d_s), addr(d_d), addr(d_c), B, n_draws, n_chunks, W, H, flags,
                                     _ptr(depth_peel), C.byref(out), C.byref(scratch), C.c_void_p(stream))
        _abi.check(st, "slhip_render")
        keep["shadow_ready"] = bool(shadows)
        buffers._keepalive = (keep,)
        return buffers
