"""sl.LightMap -- image-based lighting ('next' row f1; reference include/stillleben/light_map.h,
src/light_map.cpp).  Loads an sIBL `.ibl` description (or an equirectangular image directly), builds
the environment / irradiance / prefilter cube maps and the BRDF table on the HIP device through
``slhip_light_map_build`` and carries the directional lights the `.ibl` file names (Sun, Light1,
Light2: light_map.cpp:310-345)."""
import ctypes as C
import math
import os

import numpy as np
import torch

from . import _abi

# reference sizes: light_map.cpp:381 (512^2 faces, full mip chain), :458 (32), :519 + MAX_MIP_LEVELS (128, 5), :577 (512)
DEFAULT_SIZES = dict(env_size=512, env_levels=10, irr_size=32, pre_size=128, pre_levels=5, lut_size=512)


def read_radiance_hdr(path):
    """Radiance RGBE (.hdr / .pic) -> f32 [H,W,3], row 0 = top.  Flat and new-style RLE scanlines."""
    with open(path, "rb") as f:
        data = f.read()
    pos = data.find(b"\n\n")
    if pos < 0 or not data.startswith((b"#?RADIANCE", b"#?RGBE")):
        raise ValueError("%s is not a Radiance HDR file" % path)
    end = data.find(b"\n", pos + 2)
    dims = data[pos + 2:end].split()
    if len(dims) != 4 or dims[0] != b"-Y" or dims[2] != b"+X":
        raise ValueError("%s: unsupported orientation %r" % (path, data[pos + 2:end]))
    H, W = int(dims[1]), int(dims[3])
    buf = np.frombuffer(data, dtype=np.uint8, offset=end + 1)
    out = np.zeros((H, W, 4), np.uint8)
    p = 0
    for y in range(H):
        if W >= 8 and W < 32768 and buf[p] == 2 and buf[p + 1] == 2 and ((int(buf[p + 2]) << 8) | int(buf[p + 3])) == W:
            p += 4
            for c in range(4):
                x = 0
                while x < W:
                    n = int(buf[p]); p += 1
                    if n > 128:
                        n -= 128
                        out[y, x:x + n, c] = buf[p]; p += 1
                    else:
                        out[y, x:x + n, c] = buf[p:p + n]; p += n
                    x += n
        else:
            out[y] = buf[p:p + 4 * W].reshape(W, 4); p += 4 * W
    e = out[:, :, 3].astype(np.int32)
    scale = np.where(e > 0, np.ldexp(1.0, e - 136), 0.0).astype(np.float32)
    return (out[:, :, :3].astype(np.float32) * scale[:, :, None]).astype(np.float32)


def _parse_ibl(path):
    """sIBL files are INI documents: [Reflection] REFfile/REFmap/..., [Sun] SUNcolor/SUNmulti/SUNu/SUNv, [Light1], [Light2]."""
    groups, cur = {}, None
    with open(path, "r", errors="replace") as f:
        for raw in f:
            line = raw.strip()
            if not line or line[0] in ";#":
                continue
            if line.startswith("[") and line.endswith("]"):
                cur = groups.setdefault(line[1:-1].strip(), {})
                continue
            if "=" in line and cur is not None:
                k, v = line.split("=", 1)
                cur[k.strip()] = v.strip().strip('"')
    return groups


def _light_from_spec(g, prefix):
    """LightSpec::load + addLight (light_map.cpp:104-152, :310-323)."""
    multi = float(g.get(prefix + "multi", 1.0))
    color = np.ones(3, np.float32)
    if prefix + "color" in g:
        parts = g[prefix + "color"].split(",")
        if len(parts) != 3:
            raise ValueError("Invalid light spec: %s" % g[prefix + "color"])
        color = np.array([float(x) for x in parts], np.float32) / np.float32(255.0)
    u, v = float(g.get(prefix + "u", 0.0)), float(g.get(prefix + "v", 0.0))
    theta, phi = (u + 0.5) * math.pi * 2.0, v * math.pi
    pos = np.array([math.cos(phi) * math.sin(theta), math.sin(phi) * math.sin(theta), math.cos(theta)], np.float32)
    return -pos, (np.float32(multi) * color).astype(np.float32)


def load_equirect_image(path):
    p = str(path)
    if p.lower().endswith((".hdr", ".pic")):
        return read_radiance_hdr(p)
    if p.lower().endswith(".npy"):
        return np.ascontiguousarray(np.load(p), dtype=np.float32)[:, :, :3]
    from PIL import Image   # LDR images (the reference goes through StbImageImporter)

    return np.asarray(Image.open(p).convert("RGB"), dtype=np.float32) / np.float32(255.0)


class LightMap:
    """LightMap(path) as in the reference; additionally LightMap(array) with an f32 [H,W,3] equirectangular
    radiance map (row 0 = top, +z up), and `sizes=` to override the texture sizes (tests use small ones)."""

    def __init__(self, source, sizes=None):
        from ._context import engine, require_context

        require_context()
        self.path = ""
        self.light_directions = []
        self.light_colors = []
        if isinstance(source, (str, bytes)) or hasattr(source, "__fspath__"):
            self.path = os.fspath(source)
            if not os.path.exists(self.path):
                raise RuntimeError("Could not load light map " + self.path)
            if self.path.endswith(".ibl"):
                groups = _parse_ibl(self.path)
                ref = groups.get("Reflection")
                if ref is None:
                    raise RuntimeError("%s does not contain a Reflection group" % self.path)
                for tag in ("REFfile", "REFmap"):
                    if tag not in ref:
                        raise RuntimeError("IBL file does not contain %s" % tag)
                if int(ref["REFmap"]) != 1:
                    raise RuntimeError("IBL file uses unsupported mapping mode %s" % ref["REFmap"])
                equirect = load_equirect_image(os.path.join(os.path.dirname(self.path), ref["REFfile"]))
                for name, prefix in (("Sun", "SUN"), ("Light1", "LIGHT"), ("Light2", "LIGHT")):
                    if name in groups:
                        d, c = _light_from_spec(groups[name], prefix)
                        self.light_directions.append(d)
                        self.light_colors.append(c)
            else:
                equirect = load_equirect_image(self.path)
        else:
            if hasattr(source, "detach"):
                source = source.detach().cpu().numpy()
            equirect = np.ascontiguousarray(source, dtype=np.float32)
        if equirect.ndim != 3 or equirect.shape[2] < 3:
            raise ValueError("equirectangular map must be [H,W,3]")
        self.equirect = np.ascontiguousarray(equirect[:, :, :3], dtype=np.float32)
        self.sizes = dict(DEFAULT_SIZES)
        self.sizes.update(sizes or {})
        eng = engine()
        L = eng.L
        need = (C.c_uint64 * 4)()
        _abi.check(L.slhip_light_map_floats(*(int(self.sizes[k]) for k in ("env_size", "env_levels", "irr_size", "pre_size",
                                                                          "pre_levels", "lut_size")), C.byref(need)),
                   "slhip_light_map_floats")
        dev = eng.device
        self.env = torch.empty(int(need[0]), dtype=torch.float32, device=dev)
        self.irradiance = torch.empty(int(need[1]), dtype=torch.float32, device=dev)
        self.prefilter = torch.empty(int(need[2]), dtype=torch.float32, device=dev)
        self.brdf_lut = torch.empty(int(need[3]), dtype=torch.float32, device=dev)
        self.rec = _abi.LightMapRec()
        self.rec.d_env, self.rec.d_irradiance = self.env.data_ptr(), self.irradiance.data_ptr()
        self.rec.d_prefilter, self.rec.d_brdf_lut = self.prefilter.data_ptr(), self.brdf_lut.data_ptr()
        for k, v in self.sizes.items():
            setattr(self.rec, k, int(v))
        d_eq = torch.from_numpy(self.equirect).to(dev)
        stream = torch.cuda.current_stream(dev).cuda_stream
        with torch.cuda.device(dev):
            _abi.check(L.slhip_light_map_build(d_eq.data_ptr(), self.equirect.shape[0], self.equirect.shape[1],
                                               C.byref(self.rec), C.c_void_p(stream)), "slhip_light_map_build")
        torch.cuda.current_stream(dev).synchronize()
        self._slot = eng.register_light_map(self)

    # reference accessors (light_map.h:35-60)
    def lightDirections(self):
        return [torch.from_numpy(d.copy()) for d in self.light_directions]

    def lightColors(self):
        return [torch.from_numpy(c.copy()) for c in self.light_colors]
