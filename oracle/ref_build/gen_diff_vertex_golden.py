#!/usr/bin/env python3
"""Generates tests/golden/diff_vertex_golden.npz from the REFERENCE itself (build container only):
`bp_to_vertices_and_colors` of python/stillleben/diff.py:215-352 (row D6) on the synthetic sphere
G-buffers of gen_diff_golden.py, extended with deterministic barycentric / vertex-id targets."""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import gen_diff_golden as G  # noqa: E402


class Result(G.FakeResult):
    def __init__(self, rgb, coord, inst, bary, vidx):
        super().__init__(rgb, coord, inst)
        self._bary, self._vidx = torch.from_numpy(bary), torch.from_numpy(vidx)

    def barycentric_coeffs(self):
        return self._bary

    def vertex_indices(self):
        return self._vidx


def targets(H, W, inst):
    """Barycentrics (positive, sum 1) and 1-based vertex ids from integer hashes of the pixel position."""
    y, x = np.mgrid[0:H, 0:W]
    a = ((x * 5 + y * 3) % 7 + 1).astype(np.float32)
    b = ((x * 2 + y * 11) % 5 + 1).astype(np.float32)
    c = ((x + y * 7) % 3 + 1).astype(np.float32)
    s = a + b + c
    bary = np.stack([a / s, b / s, c / s], axis=-1).astype(np.float32)
    vidx = np.stack([(x * 3 + y) % 97 + 1, (x + y * 5) % 89 + 1, (x * 7 + y * 2) % 83 + 1], axis=-1).astype(np.int32)
    bary[inst == 0] = 0.0
    vidx[inst == 0] = 0
    return bary, vidx


def main():
    native, diff = G.load_reference_diff()
    out = {}
    for name, H, W, seed in [("small", 48, 64, 1), ("occl", 48, 64, 2)]:
        rng = np.random.default_rng(seed)
        P = G.projection(W, H, 0.9 * W, 0.95 * W, W / 2 - 3.2, H / 2 + 1.7)
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
        spheres, objs = [], []
        for k, c in enumerate(centres):
            pose = poses[k].copy()
            pose[:3, 3] = c
            idx = [3, 7, 12][k]
            spheres.append((c, radii[k], idx, pose))
            objs.append(G.FakeObj(pose, idx))
        rgb, coord, inst = G.sphere_gbuffer(H, W, P, spheres, seed)
        bary, vidx = targets(H, W, inst)
        grad_img = G.pattern_grad(H, W)
        scene, res = G.FakeScene(P, objs), Result(rgb, coord, inst, bary, vidx)
        vi, gv, gc = diff.bp_to_vertices_and_colors(scene, res, torch.from_numpy(grad_img))
        assert len(vi) == len(objs)
        out.update({name + "_rgb": rgb, name + "_coord": coord, name + "_inst": inst, name + "_bary": bary, name + "_vidx": vidx,
                    name + "_P": P.astype(np.float32), name + "_poses": np.stack([o.pose().numpy() for o in objs]),
                    name + "_obj_inst": np.array([o.instance_index for o in objs], np.int32), name + "_grad_img": grad_img,
                    name + "_n": np.array([len(v) for v in vi], np.int64),
                    name + "_out_vidx": np.concatenate([v.numpy() for v in vi]),
                    name + "_out_gv": np.concatenate([g.numpy() for g in gv]).astype(np.float32),
                    name + "_out_gc": np.concatenate([g.numpy() for g in gc]).astype(np.float32)})
    dst = os.path.join(G.ROOT, "tests", "golden", "diff_vertex_golden.npz")
    np.savez_compressed(dst, **out)
    print("wrote", dst, os.path.getsize(dst) // 1024, "KiB")


if __name__ == "__main__":
    main()
