"""Pins the sl.diff ORACLE (oracle/diff_ref.c, oracle.apply_pose_delta) against golden vectors
produced by the REFERENCE itself (oracle/ref_build/gen_diff_golden.py: bridge_diff.cpp CPU loops
compiled where they lie + diff.py imported)."""
import os

import numpy as np
import pytest

GOLDEN = os.path.join(os.path.dirname(__file__), "golden", "diff_golden.npz")
CASES = ["small", "occl", "vga"]


def pattern_grad(H, W):
    c, y, x = np.mgrid[0:3, 0:H, 0:W]
    v = (x * 7 + y * 13 + c * 29 + (x * y) % 11) % 17 - 8
    return (v / 8.0).astype(np.float32)


@pytest.fixture(scope="module")
def G():
    return np.load(GOLDEN)


@pytest.mark.parametrize("name", CASES)
def test_sobel_valid_mask_bit_exact(oracle, G, name):
    inst, coord = G[name + "_inst"], G[name + "_coord"]
    H, W = inst.shape
    valid = oracle.sobel_valid(inst, coord[:, :, 3])
    ref = np.unpackbits(G[name + "_valid"])[: H * W].reshape(H, W).astype(bool)
    assert np.array_equal(valid, ref)
    assert not valid.all() or name == "small"   # occlusion cases really invalidate pixels


@pytest.mark.parametrize("name", CASES)
def test_dilate_bit_exact(oracle, G, name):
    inst, coord = G[name + "_inst"], G[name + "_coord"]
    H, W = inst.shape
    valid = np.unpackbits(G[name + "_valid"])[: H * W].reshape(H, W).astype(bool)
    obj_inst = G[name + "_obj_inst"]
    ref_masks = np.unpackbits(G[name + "_dil_mask"])[: len(obj_inst) * H * W].reshape(len(obj_inst), H, W).astype(bool)
    for k, idx in enumerate(obj_inst):
        m, c3 = oracle.dilate(inst == idx, valid, coord[:, :, :3])
        assert np.array_equal(m, ref_masks[k])
        assert (m & ~(inst == idx)).sum() > 0           # something was dilated
        assert not m[0].any() and not m[-1].any() and not m[:, 0].any() and not m[:, -1].any()  # border rule
        if name + "_dil_coord" in G:
            assert np.array_equal(c3, G[name + "_dil_coord"][k])
        assert np.allclose(c3.astype(np.float64).sum(axis=(0, 1)), G[name + "_dil_coord_sum"][k], rtol=1e-9, atol=1e-6)


@pytest.mark.parametrize("name", CASES)
def test_image_gradients(oracle, G, name):
    inst = G[name + "_inst"]
    H, W = inst.shape
    valid = np.unpackbits(G[name + "_valid"])[: H * W].reshape(H, W).astype(bool)
    gx, gy = oracle.image_gradients(G[name + "_rgb"], valid)
    tol = 2e-2 if name == "vga" else 1e-5   # the vga golden is stored as float16
    assert np.allclose(gx, G[name + "_grad_x"].astype(np.float32), atol=tol, rtol=2e-3)
    assert np.allclose(gy, G[name + "_grad_y"].astype(np.float32), atol=tol, rtol=2e-3)


@pytest.mark.parametrize("name", CASES)
def test_pose_backward(oracle, G, name):
    inst = G[name + "_inst"]
    H, W = inst.shape
    g = oracle.pose_backward(G[name + "_rgb"], G[name + "_coord"], inst, pattern_grad(H, W), G[name + "_P"],
                             G[name + "_poses"], G[name + "_obj_inst"])
    ref = G[name + "_pose_grad"]
    scale = np.abs(ref).max()
    assert scale > 0
    assert np.abs(g - ref).max() <= 1e-3 * scale + 1e-4   # fp32 torch bmm chain vs fp64 accumulation


def test_apply_pose_delta(oracle, G):
    out = oracle.apply_pose_delta(G["apd_pose"], G["apd_delta"], True)
    assert np.allclose(out, G["apd_out_ortho"], atol=1e-6)
    raw = oracle.apply_pose_delta(G["apd_pose"], G["apd_delta"], False)
    assert np.allclose(raw, G["apd_out_raw"], atol=1e-6)
    R = out[:, :3, :3]
    assert np.allclose(R @ np.transpose(R, (0, 2, 1)), np.eye(3), atol=1e-5)


@pytest.mark.parametrize("name", ["small", "occl"])
def test_d6_vertex_backward_matches_reference(oracle, name):
    """Row D6: oracle/diff_ref.c slref_vertex_backward against the outputs of the REFERENCE's
    bp_to_vertices_and_colors (oracle/ref_build/gen_diff_vertex_golden.py)."""
    G = np.load(os.path.join(os.path.dirname(GOLDEN), "diff_vertex_golden.npz"))
    inst = G[name + "_inst"]
    gv, gc = oracle.vertex_backward(G[name + "_rgb"], G[name + "_coord"], inst, G[name + "_bary"], G[name + "_grad_img"],
                                    G[name + "_P"], G[name + "_poses"], G[name + "_obj_inst"])
    vi_ref, gv_ref, gc_ref = G[name + "_out_vidx"], G[name + "_out_gv"], G[name + "_out_gc"]
    off = 0
    for o, n in zip(G[name + "_obj_inst"], G[name + "_n"]):
        m = inst == o
        assert 3 * int(m.sum()) == int(n)
        assert np.array_equal(G[name + "_vidx"][m].reshape(-1), vi_ref[off:off + n])
        a, b = gv[m].reshape(-1, 3), gv_ref[off:off + n]
        scale = max(1.0, float(np.abs(b).max()))
        assert np.abs(a - b).max() <= 2e-5 * scale, (np.abs(a - b).max(), scale)   # fp32 torch bmm vs the explicit sums
        assert np.abs(gc[m].reshape(-1, 3) - gc_ref[off:off + n]).max() <= 1e-6
        off += int(n)
    assert off == len(vi_ref)
    assert not gv[inst == 0].any() and not gc[inst == 0].any()
