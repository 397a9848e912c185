#!/usr/bin/env python3
"""Compares one frame of this repo's renderer with an externally supplied frame of the reference's OpenGL pipeline
(VERDICT r01 item 8): Magnum / EGL are not in this image, so the reference cannot render here -- whoever
has a box with it dumps `RenderPassResult` to an .npz (below) and this harness says how far the two are apart, split into
what the rasterisation rules can explain and what they cannot:

  silhouette pixels   instance ids differ AND the pixel lies within one pixel of an instance boundary in either frame
                      (GL's fill rule / sub-pixel snapping differs from oracle rules R3 / R4 by at most that)
  interior mismatch   instance ids differ elsewhere: a real disagreement (pose, projection, depth order)
  interior values     where the ids agree: max |depth|, max |object coordinate|, max angle between normals, rgb differences

    reference side (python, with the reference built):
        r = renderer.render(scene)
        np.savez("ref.npz", instance=r.instance_index().cpu(), cls=r.class_index().cpu(), coord=r.coordDepth().cpu(),
                 normals=r.normals().cpu(), rgb=r.rgb().cpu())
    this side:    the same scene through stillleben_amd, saved the same way, then
        python tools/compare_gl.py ours.npz ref.npz
Exit code 0 when there is no interior mismatch and the interior values agree within the stated tolerances."""
import argparse
import json
import sys

import numpy as np


def _boundary(inst):
    """Pixels with a 4-neighbour of a different instance id, dilated by one pixel (8-neighbourhood)."""
    i = inst.astype(np.int64)
    b = np.zeros(i.shape, bool)
    b[:, 1:] |= i[:, 1:] != i[:, :-1]
    b[:, :-1] |= i[:, 1:] != i[:, :-1]
    b[1:, :] |= i[1:, :] != i[:-1, :]
    b[:-1, :] |= i[1:, :] != i[:-1, :]
    d = b.copy()
    for dy in (-1, 0, 1):
        for dx in (-1, 0, 1):
            sh = np.zeros_like(b)
            ys = slice(max(dy, 0), b.shape[0] + min(dy, 0)); yd = slice(max(-dy, 0), b.shape[0] + min(-dy, 0))
            xs = slice(max(dx, 0), b.shape[1] + min(dx, 0)); xd = slice(max(-dx, 0), b.shape[1] + min(-dx, 0))
            sh[yd, xd] = b[ys, xs]
            d |= sh
    return d


def _plane(a):
    a = np.asarray(a)
    return a[..., 0] if a.ndim == 3 and a.shape[-1] == 1 else a


def compare(ours, ref, depth_tol=1e-3, coord_tol=1e-3, normal_tol_deg=1.0, rgb_tol=3):
    """`ours`, `ref`: mappings with 'instance' [H,W(,1)] and optionally 'cls', 'coord' [H,W,4] (xyz + depth), 'normals'
    [H,W,3|4], 'rgb' [H,W,3|4].  Returns the report dict; report['ok'] is the verdict."""
    io, ir = _plane(ours["instance"]).astype(np.int64), _plane(ref["instance"]).astype(np.int64)
    if io.shape != ir.shape:
        raise ValueError("frames differ in size: %s vs %s" % (io.shape, ir.shape))
    differ = io != ir
    near_edge = _boundary(io) | _boundary(ir)
    same = ~differ
    obj = same & (io != 0)
    rep = {"pixels": int(io.size), "object_pixels": int(obj.sum()), "silhouette_mismatch": int((differ & near_edge).sum()),
           "interior_mismatch": int((differ & ~near_edge).sum())}
    ok = rep["interior_mismatch"] == 0
    inner = obj & ~near_edge                      # values are compared away from the silhouettes (GL interpolates across them)
    if "cls" in ours and "cls" in ref:
        rep["class_mismatch_interior"] = int((_plane(ours["cls"])[inner] != _plane(ref["cls"])[inner]).sum())
        ok &= rep["class_mismatch_interior"] == 0
    if "coord" in ours and "coord" in ref and inner.any():
        co, cr = np.asarray(ours["coord"], np.float64), np.asarray(ref["coord"], np.float64)
        rep["max_depth_diff"] = float(np.abs(co[..., 3] - cr[..., 3])[inner].max())
        rep["max_coord_diff"] = float(np.abs(co[..., :3] - cr[..., :3])[inner].max())
        ok &= rep["max_depth_diff"] <= depth_tol and rep["max_coord_diff"] <= coord_tol
    if "normals" in ours and "normals" in ref and inner.any():
        no, nr = np.asarray(ours["normals"], np.float64)[..., :3][inner], np.asarray(ref["normals"], np.float64)[..., :3][inner]
        c = (no * nr).sum(-1) / np.maximum(np.linalg.norm(no, axis=-1) * np.linalg.norm(nr, axis=-1), 1e-12)
        rep["max_normal_angle_deg"] = float(np.degrees(np.arccos(np.clip(c, -1.0, 1.0))).max())
        ok &= rep["max_normal_angle_deg"] <= normal_tol_deg
    if "rgb" in ours and "rgb" in ref and inner.any():
        d = np.abs(np.asarray(ours["rgb"], np.int64)[..., :3] - np.asarray(ref["rgb"], np.int64)[..., :3])[inner]
        rep["rgb_max_diff"] = int(d.max())
        rep["rgb_frac_above_tol"] = float((d.max(-1) > rgb_tol).mean())
        ok &= rep["rgb_frac_above_tol"] < 0.01
    rep["ok"] = bool(ok)
    return rep


def main():
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument("ours")
    ap.add_argument("reference")
    ap.add_argument("--depth-tol", type=float, default=1e-3)
    ap.add_argument("--coord-tol", type=float, default=1e-3)
    ap.add_argument("--normal-tol-deg", type=float, default=1.0)
    ap.add_argument("--rgb-tol", type=int, default=3)
    a = ap.parse_args()
    rep = compare(np.load(a.ours), np.load(a.reference), a.depth_tol, a.coord_tol, a.normal_tol_deg, a.rgb_tol)
    print(json.dumps(rep, indent=1))
    sys.exit(0 if rep["ok"] else 1)


if __name__ == "__main__":
    main()
