#!/usr/bin/env python3
"""Generates tests/golden/diff_golden.npz from the REFERENCE itself (run in the build container,
where /root/reference exists):
  - generate_sobel_valid_mask / dilate_object_mask: the reference's CPU loops compiled from
    python/src/bridge_diff.cpp where it lies (oracle/_ref/libstillleben_diff_python.so, built by
    oracle/ref_build/Makefile; loaded RTLD_LAZY because the *Cuda symbols of diff.cu are absent),
  - compute_image_space_gradients / backpropagate_gradient_to_poses / apply_pose_delta: the
    reference's python/stillleben/diff.py imported with stub `Scene` / `RenderPassResult` classes.
Inputs are synthetic analytic G-buffers (spheres with occlusion), seeded."""
import importlib.util
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"


def load_reference_diff():
    so = os.path.join(HERE, "..", "_ref", "libstillleben_diff_python.so")
    flags = sys.getdlopenflags()
    sys.setdlopenflags(os.RTLD_LAZY | os.RTLD_GLOBAL)
    spec = importlib.util.spec_from_file_location("libstillleben_diff_python", so)
    native = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(native)
    sys.setdlopenflags(flags)
    pkg = types.ModuleType("stillleben")
    pkg.__path__ = [os.path.join(REF, "python", "stillleben")]
    lib = types.ModuleType("stillleben.lib")
    lib.__path__ = []
    core = types.ModuleType("stillleben.lib.libstillleben_python")

    class Scene:  # duck-typed stand-ins for the type annotations of diff.py
        pass

    class RenderPassResult:
        pass

    core.Scene, core.RenderPassResult = Scene, RenderPassResult
    sys.modules.update({
        "stillleben": pkg, "stillleben.lib": lib, "stillleben.lib.libstillleben_python": core,
        "stillleben.lib.libstillleben_diff_python": native,
    })
    spec = importlib.util.spec_from_file_location("stillleben.diff", os.path.join(REF, "python", "stillleben", "diff.py"))
    diff = importlib.util.module_from_spec(spec)
    sys.modules["stillleben.diff"] = diff
    spec.loader.exec_module(diff)
    return native, diff


def sphere_gbuffer(H, W, P, spheres, seed):
    """Analytic G-buffer: spheres (centre, radius, instance id, pose) seen by a camera at identity."""
    rng = np.random.default_rng(seed)
    rgb = np.zeros((H, W, 4), np.uint8)
    coord = np.full((H, W, 4), 3000.0, np.float32)
    inst = np.zeros((H, W), np.int16)
    fx, fy, cx, cy = P[0, 0] * W / 2, P[1, 1] * H / 2, (1 - P[0, 2]) * W / 2 if False else None, None
    ys, xs = np.mgrid[0:H, 0:W]
    # ray directions from the projection: ndc = (P row . p) / z
    ndx = (xs + 0.5) / W * 2 - 1
    ndy = (ys + 0.5) / H * 2 - 1
    dx = (ndx - P[0, 2]) / P[0, 0]
    dy = (ndy - P[1, 2]) / P[1, 1]
    d = np.stack([dx, dy, np.ones_like(dx)], axis=-1)
    for (c, r, idx, pose) in spheres:
        c = np.asarray(c, np.float64)
        a = (d * d).sum(-1)
        b = -2 * (d @ c)
        cc = c @ c - r * r
        disc = b * b - 4 * a * cc
        hit = disc > 0
        t = np.where(hit, (-b - np.sqrt(np.maximum(disc, 0))) / (2 * a), np.inf)
        pts = d * t[..., None]
        closer = hit & (pts[..., 2] < coord[..., 3])
        inv = np.linalg.inv(pose)
        obj = pts @ inv[:3, :3].T + inv[:3, 3]
        coord[closer, :3] = obj[closer]
        coord[closer, 3] = pts[closer][:, 2]
        inst[closer] = idx
        n = (pts - c) / r
        shade = np.clip(0.2 + 0.8 * np.maximum(0, -n[..., 2]) + 0.1 * np.sin(40 * obj[..., 0]) * np.cos(31 * obj[..., 1]), 0, 1)
        col = (np.array([0.9, 0.5, 0.3]) if idx % 2 else np.array([0.3, 0.6, 0.9]))[None, None] * shade[..., None]
        rgb[closer, :3] = (col[closer] * 255).astype(np.uint8)
        rgb[closer, 3] = 255
    del rng, fx, fy, cx, cy
    return rgb, coord, inst


class FakeObj:
    def __init__(self, pose, idx):
        self._p = torch.from_numpy(pose.astype(np.float32))
        self.instance_index = idx

    def pose(self):
        return self._p


class FakeScene:
    def __init__(self, P, objects):
        self.objects = objects
        self._P = torch.from_numpy(P.astype(np.float32))

    def projection_matrix(self):
        return self._P


class FakeResult:
    def __init__(self, rgb, coord, inst):
        self._rgb, self._coord, self._inst = torch.from_numpy(rgb), torch.from_numpy(coord), torch.from_numpy(inst)

    def rgb(self):
        return self._rgb

    def coordinates(self):
        return self._coord[:, :, 0:3]

    def depth(self):
        return self._coord[:, :, 3]

    def instance_index(self):
        return self._inst.unsqueeze(-1)


def pattern_grad(H, W):
    """Exactly reproducible pseudo-random image gradient (integer hash -> multiples of 1/8)."""
    c, y, x = np.mgrid[0:3, 0:H, 0:W]
    v = (x * 7 + y * 13 + c * 29 + (x * y) % 11) % 17 - 8
    return (v / 8.0).astype(np.float32)


def projection(W, H, fx, fy, cx, cy):
    f, n = 10.0, 0.1
    L, R = -cx * n / fx, (W - cx) * n / fx
    T, B = -cy * n / fy, (H - cy) * n / fy
    P = np.zeros((4, 4))
    P[0, 0] = 2 * n / (R - L); P[1, 1] = 2 * n / (B - T)
    P[0, 2] = (R + L) / (L - R); P[1, 2] = (T + B) / (T - B)
    P[2, 2] = (f + n) / (f - n); P[3, 2] = 1.0; P[2, 3] = 2 * f * n / (n - f)
    return P


def main():
    native, diff = load_reference_diff()
    out = {}
    cases = [("small", 48, 64, 1), ("occl", 48, 64, 2), ("vga", 480, 640, 3)]
    for name, H, W, seed in cases:
        rng = np.random.default_rng(seed)
        P = projection(W, H, 0.9 * W, 0.95 * W, W / 2 - 3.2, H / 2 + 1.7)
        poses = []
        for k in range(3):
            pose = np.eye(4)
            q = rng.standard_normal(4); q /= np.linalg.norm(q)
            x, y, z, w = q
            pose[:3, :3] = [[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
                            [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                            [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]]
            poses.append(pose)
        centres = [(-0.12, 0.02, 1.0), (0.1, -0.03, 0.8), (0.02, 0.18, 1.3)] if name != "small" else [(0.0, 0.0, 1.0)]
        radii = [0.2, 0.15, 0.25]
        spheres = []
        objs = []
        for k, c in enumerate(centres):
            pose = poses[k].copy()
            pose[:3, 3] = c
            idx = [3, 7, 12][k]
            spheres.append((c, radii[k], idx, pose))
            objs.append(FakeObj(pose, idx))
        rgb, coord, inst = sphere_gbuffer(H, W, P, spheres, seed)
        if name == "occl":  # exercise the image border rules
            inst[0, :10] = 7; coord[0, :10, 3] = 0.5
            inst[:, 0] = 3; coord[:, 0, 3] = 0.7
        grad_img = pattern_grad(H, W)
        scene, res = FakeScene(P, objs), FakeResult(rgb, coord, inst)
        valid = native.generate_sobel_valid_mask(torch.from_numpy(inst), torch.from_numpy(coord[:, :, 3].copy()))
        gx, gy, valid2 = diff.compute_image_space_gradients(scene, res)
        assert torch.equal(valid, valid2)
        dil = []
        for o in objs:
            m, c3 = native.dilate_object_mask(torch.from_numpy(inst == o.instance_index), valid, torch.from_numpy(coord[:, :, :3].copy()))
            dil.append((m.numpy(), c3.numpy()))
        g = diff.backpropagate_gradient_to_poses(scene, res, torch.from_numpy(grad_img))
        out.update({
            name + "_rgb": rgb, name + "_coord": coord, name + "_inst": inst,
            name + "_P": P.astype(np.float32), name + "_poses": np.stack([o.pose().numpy() for o in objs]),
            name + "_obj_inst": np.array([o.instance_index for o in objs], np.int32),
            name + "_valid": np.packbits(valid.numpy()), name + "_grad_x": gx.numpy().astype(np.float16 if name == "vga" else np.float32),
            name + "_grad_y": gy.numpy().astype(np.float16 if name == "vga" else np.float32),
            name + "_dil_mask": np.packbits(np.stack([d[0] for d in dil])),
            name + "_dil_coord_sum": np.stack([d[1].astype(np.float64).sum(axis=(0, 1)) for d in dil]),
            name + "_pose_grad": g.numpy(),
        })
        if name != "vga":
            out[name + "_dil_coord"] = np.stack([d[1] for d in dil])
    # apply_pose_delta
    rng = np.random.default_rng(9)
    pose = np.tile(np.eye(4, dtype=np.float32), (5, 1, 1))
    pose[:, :3, 3] = rng.standard_normal((5, 3))
    delta = (0.05 * rng.standard_normal((5, 6))).astype(np.float32)
    out["apd_pose"], out["apd_delta"] = pose, delta
    out["apd_out_ortho"] = diff.apply_pose_delta(torch.from_numpy(pose), torch.from_numpy(delta), True).numpy()
    out["apd_out_raw"] = diff.apply_pose_delta(torch.from_numpy(pose), torch.from_numpy(delta), False).numpy()
    dst = os.path.join(ROOT, "tests", "golden", "diff_golden.npz")
    np.savez_compressed(dst, **out)
    print("wrote", dst, os.path.getsize(dst) // 1024, "KiB")


if __name__ == "__main__":
    main()
