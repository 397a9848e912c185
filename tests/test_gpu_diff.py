"""GPU parity tests of the sl.diff half through the C-ABI: bit-exact masks (D1, D2) and
tolerance-checked gradients (D3, D4) against the reference-generated goldens and the oracle."""
import os

import numpy as np
import pytest
import torch

import scenes as S

pytestmark = pytest.mark.gpu
GOLDEN = os.path.join(os.path.dirname(__file__), "golden", "diff_golden.npz")
CASES = ["small", "occl", "vga"]


def pattern_grad(H, W):
    c, y, x = np.mgrid[0:3, 0:H, 0:W]
    v = (x * 7 + y * 13 + c * 29 + (x * y) % 11) % 17 - 8
    return (v / 8.0).astype(np.float32)


@pytest.fixture(scope="module")
def G():
    return np.load(GOLDEN)


class _Obj:
    def __init__(self, pose, idx):
        self._pose, self.instance_index = torch.from_numpy(pose), int(idx)

    def pose(self):
        return self._pose


class _Scene:
    def __init__(self, P, objs):
        self.objects, self._P = objs, torch.from_numpy(P)

    def projection_matrix(self):
        return self._P


class _Result:
    def __init__(self, rgb, coord, inst):
        self._rgb, self._coord, self._inst = torch.from_numpy(rgb), torch.from_numpy(coord), torch.from_numpy(inst)

    def rgb(self):
        return self._rgb

    def coordinates(self):
        return self._coord[:, :, :3]

    def depth(self):
        return self._coord[:, :, 3]

    def coordDepth(self):  # noqa: N802
        return self._coord

    def instance_index(self):
        return self._inst.unsqueeze(-1)


@pytest.mark.parametrize("name", CASES)
def test_stencils_bit_exact(sl, G, name):
    inst, coord = G[name + "_inst"], G[name + "_coord"]
    H, W = inst.shape
    valid = sl.diff.generate_sobel_valid_mask(torch.from_numpy(inst), torch.from_numpy(coord[:, :, 3].copy()))
    ref = np.unpackbits(G[name + "_valid"])[: H * W].reshape(H, W).astype(bool)
    assert valid.dtype == torch.bool and np.array_equal(valid.numpy(), ref)
    obj_inst = G[name + "_obj_inst"]
    ref_masks = np.unpackbits(G[name + "_dil_mask"])[: len(obj_inst) * H * W].reshape(len(obj_inst), H, W).astype(bool)
    for k, idx in enumerate(obj_inst):
        m, c3 = sl.diff.dilate_object_mask(torch.from_numpy(inst == idx), valid, torch.from_numpy(coord[:, :, :3].copy()))
        assert np.array_equal(m.numpy(), ref_masks[k])
        if name + "_dil_coord" in G:
            assert np.array_equal(c3.numpy(), G[name + "_dil_coord"][k])


@pytest.mark.parametrize("name", CASES)
def test_gradients_and_pose_backward(sl, oracle, G, name):
    inst = G[name + "_inst"]
    H, W = inst.shape
    objs = [_Obj(p, i) for p, i in zip(G[name + "_poses"], G[name + "_obj_inst"])]
    scene, res = _Scene(G[name + "_P"], objs), _Result(G[name + "_rgb"], G[name + "_coord"], inst)
    gx, gy, valid = sl.diff.compute_image_space_gradients(scene, res)
    ogx, ogy = oracle.image_gradients(G[name + "_rgb"], valid.numpy())
    assert np.array_equal(gx.numpy(), ogx) and np.array_equal(gy.numpy(), ogy)   # same fp32 expression
    g = sl.diff.backpropagate_gradient_to_poses(scene, res, torch.from_numpy(pattern_grad(H, W)))
    ref = G[name + "_pose_grad"]
    orc = oracle.pose_backward(G[name + "_rgb"], G[name + "_coord"], inst, pattern_grad(H, W), G[name + "_P"],
                               G[name + "_poses"], G[name + "_obj_inst"])
    scale = np.abs(ref).max()
    assert np.abs(g.numpy() - orc).max() <= 1e-5 * scale      # fp64 on both sides, different summation order
    assert np.abs(g.numpy() - ref).max() <= 1e-3 * scale + 1e-4  # vs the reference's fp32 torch chain


def test_gradient_sign_property(sl):
    # reference tests/test_grad.py:64-153: perturb each pose parameter by +0.01, render, L2 image
    # loss gradient -> backpropagate -> delta[param] > 0.  (Gaussian pyramid replaced by a plain
    # L2 loss on a blurred image: cv2 is not available offline.)
    import torch.nn.functional as F

    mesh = sl.Mesh(S.BUNNY, physics=False)
    mesh.center_bbox()
    mesh.scale_to_bbox_diagonal(0.5, 'order_of_magnitude')
    scene = sl.Scene((640, 480))
    scene.set_camera_intrinsics(1066.778, 1067.487, 312.9869, 241.3109)
    obj = sl.Object(mesh)
    scene.add_object(obj)
    pose = torch.tensor([[0.0596, 0.8315, -0.5523, -0.0651], [0.4715, 0.4642, 0.7498, -0.06036],
                         [0.8798, -0.3051, -0.3644, 0.80551], [0.0, 0.0, 0.0, 1.0]])
    U, _, Vh = torch.linalg.svd(pose[:3, :3])
    pose[:3, :3] = U @ Vh
    obj.set_pose(pose)
    scene.light_directions = torch.tensor([[0.2, 0.3, 0.9]])
    scene.ambient_light = torch.tensor([0.3, 0.3, 0.3])
    scene.manual_exposure = 1.0
    rp = sl.RenderPass()
    gt = rp.render(scene).rgb()[:, :, :3].float() / 255.0
    k = torch.ones(1, 1, 5, 5) / 25.0
    positive = 0
    for param in range(6):
        delta = torch.zeros(6)
        delta[param] = 0.01
        obj.set_pose(sl.diff.apply_pose_delta(pose, delta))
        res = rp.render(scene)
        img = (res.rgb()[:, :, :3].float() / 255.0).permute(2, 0, 1).clone().requires_grad_(True)
        tgt = gt.permute(2, 0, 1)
        loss = ((F.conv2d(img.unsqueeze(1), k, padding=2) - F.conv2d(tgt.unsqueeze(1), k, padding=2)) ** 2).sum()
        loss.backward()
        d = sl.diff.backpropagate_gradient_to_poses(scene, res, img.grad)
        assert torch.isfinite(d).all()
        positive += int(d[0][param] > 0)
        obj.set_pose(pose)
    assert positive >= 5   # the reference asserts > 0 for every parameter with its pyramid loss


def test_c5_many_objects_linearity(sl):
    # BASELINE config 5 scale (64 objects in one 640x480 view): the pose backward is linear in the
    # image gradient and additive over hypotheses -- size-independent properties
    cube = sl.Mesh(S.CUBE, physics=False)
    cube.center_bbox()
    cube.scale_to_bbox_diagonal(0.12)
    scene = sl.Scene((640, 480))
    scene.set_camera_intrinsics(1066.778, 1067.487, 312.9869, 241.3109)
    rng = np.random.default_rng(5)
    for i in range(64):
        o = sl.Object(cube)
        pose = np.eye(4, dtype=np.float32)
        pose[:3, :3] = S.random_rotation(rng)
        pose[:3, 3] = [((i % 8) - 3.5) * 0.11, ((i // 8) - 3.5) * 0.085, 1.6 + 0.3 * rng.uniform()]
        o.set_pose(torch.from_numpy(pose))
        scene.add_object(o)
    scene.light_directions = torch.tensor([[0.1, 0.2, 0.9]])
    scene.manual_exposure = 1.0
    res = sl.RenderPass().render(scene)
    assert len(torch.unique(res.instance_index())) >= 60
    g1 = torch.from_numpy(pattern_grad(480, 640))
    g2 = torch.from_numpy(np.random.default_rng(1).standard_normal((3, 480, 640)).astype(np.float32))
    d1 = sl.diff.backpropagate_gradient_to_poses(scene, res, g1)
    d2 = sl.diff.backpropagate_gradient_to_poses(scene, res, g2)
    d12 = sl.diff.backpropagate_gradient_to_poses(scene, res, 2.0 * g1 - 0.5 * g2)
    assert d1.shape == (64, 6)
    scale = max(float(d1.abs().max()), float(d2.abs().max()))
    assert float((d12 - (2.0 * d1 - 0.5 * d2)).abs().max()) <= 2e-5 * scale
    # an object whose pixels receive zero gradient gets a zero pose gradient
    inst = res.instance_index().squeeze(-1)
    g0 = g2.clone()
    near = torch.nn.functional.max_pool2d((inst == 1).float()[None, None], 5, 1, 2)[0, 0] > 0
    g0[:, near] = 0.0
    d0 = sl.diff.backpropagate_gradient_to_poses(scene, res, g0)
    assert float(d0[0].abs().max()) == 0.0 and float(d0[1:].abs().max()) > 0.0


class _ResultVB(_Result):
    def __init__(self, rgb, coord, inst, bary, vidx):
        super().__init__(rgb, coord, inst)
        self._bary, self._vidx = torch.from_numpy(bary), torch.from_numpy(vidx)

    def barycentric_coeffs(self):
        return self._bary

    def vertex_indices(self):
        return self._vidx


@pytest.mark.parametrize("name", ["small", "occl"])
def test_d6_bp_to_vertices_and_colors(sl, oracle, name):
    """Row D6 through the public API: identical to the oracle (same float32 operation order) and within float
    tolerance of the reference's own outputs (tests/golden/diff_vertex_golden.npz)."""
    V = np.load(os.path.join(os.path.dirname(GOLDEN), "diff_vertex_golden.npz"))
    objs = [_Obj(p, i) for p, i in zip(V[name + "_poses"], V[name + "_obj_inst"])]
    scene = _Scene(V[name + "_P"], objs)
    res = _ResultVB(V[name + "_rgb"], V[name + "_coord"], V[name + "_inst"], V[name + "_bary"], V[name + "_vidx"])
    vi, gv, gc = sl.diff.bp_to_vertices_and_colors(scene, res, torch.from_numpy(V[name + "_grad_img"]))
    assert len(vi) == len(objs)
    vi_c, gv_c, gc_c = (torch.cat(x).numpy() for x in (vi, gv, gc))
    assert np.array_equal(vi_c, V[name + "_out_vidx"])
    ogv, ogc = oracle.vertex_backward(V[name + "_rgb"], V[name + "_coord"], V[name + "_inst"], V[name + "_bary"], V[name + "_grad_img"],
                                      V[name + "_P"], V[name + "_poses"], V[name + "_obj_inst"])
    ref_gv = np.concatenate([ogv[V[name + "_inst"] == o].reshape(-1, 3) for o in V[name + "_obj_inst"]])
    ref_gc = np.concatenate([ogc[V[name + "_inst"] == o].reshape(-1, 3) for o in V[name + "_obj_inst"]])
    assert np.array_equal(gv_c.view(np.uint32), ref_gv.view(np.uint32)) and np.array_equal(gc_c.view(np.uint32), ref_gc.view(np.uint32))
    scale = max(1.0, float(np.abs(V[name + "_out_gv"]).max()))
    assert np.abs(gv_c - V[name + "_out_gv"]).max() <= 2e-5 * scale and np.abs(gc_c - V[name + "_out_gc"]).max() <= 1e-6


def test_d6_on_a_rendered_scene_and_soft_forward(sl):
    """The vertex path end to end on real render targets: ids / barycentrics come from the renderer, the result
    feeds Mesh.update_positions, and soft_forward blends two depth-peel layers."""
    scene = S.clutter_scene(sl, 3, n_objects=3, size=(160, 120))
    scene.manual_exposure = 1.0
    rp = sl.RenderPass()
    res = rp.render(scene)
    g = torch.from_numpy(pattern_grad(120, 160))
    vi, gv, gc = sl.diff.bp_to_vertices_and_colors(scene, res, g)
    inst = res.instance_index().squeeze(-1)
    seen = [o for o in scene.objects if bool((inst == o.instance_index).any())]
    assert len(vi) == len(seen) >= 2
    for o, v, a, c in zip(seen, vi, gv, gc):
        n = int((inst == o.instance_index).sum())
        assert v.shape == (3 * n,) and a.shape == (3 * n, 3) and c.shape == (3 * n, 3)
        assert int(v.min()) >= 1 and int(v.max()) <= o.mesh.points.shape[0]
        assert torch.isfinite(a).all() and float(a.abs().max()) > 0
    # linear in the image gradient
    vi2, gv2, _ = sl.diff.bp_to_vertices_and_colors(scene, res, 2.0 * g)
    assert torch.equal(vi2[0], vi[0]) and torch.allclose(gv2[0], 2.0 * gv[0], rtol=1e-5, atol=1e-7)
    # the gradients are in the units Mesh.update_positions expects
    m = seen[0].mesh
    before = m.points.clone()
    # (the reference ADDS the update, mesh.cpp:836: a gradient step is update_positions(ids, lr * grad))
    m.update_positions(vi[0][:3].contiguous(), (1e-4 * gv[0][:3]).contiguous())
    assert not torch.equal(m.points, before)
    moved = (m.points - before).abs().max()
    assert 0 < float(moved) <= 1e-4 * float(gv[0][:3].abs().sum()) + 1e-7
    peel = rp.render(scene, depth_peel=res)

    def loss_fn(pred, obs):
        d = (pred - obs) ** 2
        return d.mean(), d

    obs = torch.rand(3, 120, 160)
    soft, layers, loss_img, loss, svi, sgv, sgc = sl.diff.soft_forward(scene, [res, peel], obs, loss_fn)
    assert soft.shape == (3, 120, 160) and len(layers) == 2 and loss > 0 and len(svi) == len(sgv) == len(sgc) >= len(seen)


def test_pose_hypothesis_batch_equals_the_loop(sl):
    """sl.diff.backpropagate_gradient_to_poses_batch (BASELINE config C5's shape: K pose hypotheses of one scene in one
    render launch sequence) == set_pose + RenderPass().render + backpropagate_gradient_to_poses hypothesis by hypothesis."""
    scene = S.clutter_scene(sl, 5, n_objects=5, size=(160, 120))
    scene.manual_exposure = 1.0
    rng = np.random.default_rng(2)
    base = [o.pose() for o in scene.objects]
    K = 4
    hyps = torch.stack([torch.stack([sl.diff.apply_pose_delta(b, torch.from_numpy(rng.normal(0, 0.02, 6).astype(np.float32))) for b in base])
                        for _ in range(K)])
    g = torch.from_numpy(pattern_grad(120, 160))
    got, buf = sl.diff.backpropagate_gradient_to_poses_batch(scene, hyps, g, return_results=True)
    assert tuple(got.shape) == (K, 5, 6)
    rp = sl.RenderPass()
    for k in range(K):
        for o, p in zip(scene.objects, hyps[k]):
            o.set_pose(p)
        res = rp.render(scene)
        assert torch.equal(res.instance_index().squeeze(-1).cpu(), buf.instance[k].squeeze(-1).cpu())
        assert torch.equal(res.coordDepth().cpu(), buf.coord[k].cpu())
        ref = sl.diff.backpropagate_gradient_to_poses(scene, res, g)
        scale = max(1e-9, float(ref.abs().max()))
        assert float((got[k] - ref).abs().max()) <= 1e-4 * scale
    with pytest.raises(ValueError):
        sl.diff.backpropagate_gradient_to_poses_batch(scene, hyps[:, :3], g)
    # a second light: the batch form has to give the per-hypothesis answer as well (its replicated records carry light 0 only)
    ld = scene.light_directions.clone()
    ld[1] = torch.tensor([0.3, -0.2, -1.0])
    scene.light_directions = ld
    lc = scene.light_colors.clone()
    lc[1] = torch.tensor([120.0, 100.0, 80.0])
    scene.light_colors = lc
    got2 = sl.diff.backpropagate_gradient_to_poses_batch(scene, hyps, g)
    for k in range(K):
        for o, p in zip(scene.objects, hyps[k]):
            o.set_pose(p)
        ref = sl.diff.backpropagate_gradient_to_poses(scene, rp.render(scene), g)
        scale = max(1e-9, float(ref.abs().max()))
        assert float((got2[k] - ref).abs().max()) <= 1e-4 * scale


def test_c5_at_its_real_shape_64_objects_32_hypotheses(sl):
    """BASELINE config C5 as BASELINE.json states it: 64 objects x 32 pose hypotheses, 640x480, ONE
    slhip_diff_pose_backward_batch behind sl.diff.backpropagate_gradient_to_poses_batch.  Held against (1) the per-hypothesis
    path (set_pose + render + backpropagate_gradient_to_poses) on four sampled hypotheses, (2) additivity / linearity in the image
    gradient on ALL 32 x 64 pose gradients, (3) a repeated hypothesis gives the same row."""
    cube = sl.Mesh(S.CUBE, physics=False)
    cube.center_bbox()
    cube.scale_to_bbox_diagonal(0.12)
    scene = sl.Scene((640, 480))
    scene.set_camera_intrinsics(1066.778, 1067.487, 312.9869, 241.3109)
    rng = np.random.default_rng(55)
    for i in range(64):
        o = sl.Object(cube)
        pose = np.eye(4, dtype=np.float32)
        pose[:3, :3] = S.random_rotation(rng)
        pose[:3, 3] = [((i % 8) - 3.5) * 0.11, ((i // 8) - 3.5) * 0.085, 1.6 + 0.3 * rng.uniform()]
        o.set_pose(torch.from_numpy(pose))
        scene.add_object(o)
    scene.light_directions = torch.tensor([[0.1, 0.2, 0.9]])
    scene.manual_exposure = 1.0
    base = [o.pose() for o in scene.objects]
    K = 32
    hyps = torch.stack([torch.stack([sl.diff.apply_pose_delta(b, torch.from_numpy(rng.normal(0, 0.01, 6).astype(np.float32))) for b in base])
                        for _ in range(K)])
    hyps[K - 1] = hyps[3]                                               # (3)
    g1 = torch.from_numpy(pattern_grad(480, 640))
    g2 = torch.from_numpy(np.random.default_rng(1).standard_normal((3, 480, 640)).astype(np.float32))
    d1, buf = sl.diff.backpropagate_gradient_to_poses_batch(scene, hyps, g1, return_results=True)
    assert tuple(d1.shape) == (K, 64, 6) and torch.isfinite(d1).all()
    assert tuple(buf.instance.shape[:3]) == (K, 480, 640)
    seen = [len(torch.unique(buf.instance[k])) for k in range(K)]
    assert min(seen) >= 60                                             # every hypothesis shows (nearly) all 64 objects
    assert float(d1.abs().amax(dim=(1, 2)).min()) > 0                  # ... and has a gradient
    assert torch.equal(d1[K - 1], d1[3]) and not torch.equal(d1[0], d1[1])
    # (1) the per-hypothesis path
    rp = sl.RenderPass()
    for k in (0, 11, 22, 31):
        for o, p in zip(scene.objects, hyps[k]):
            o.set_pose(p)
        res = rp.render(scene)
        assert torch.equal(res.instance_index().squeeze(-1).cpu(), buf.instance[k].squeeze(-1).cpu())
        ref = sl.diff.backpropagate_gradient_to_poses(scene, res, g1)
        scale = max(1e-9, float(ref.abs().max()))
        assert float((d1[k] - ref).abs().max()) <= 1e-4 * scale
    for o, p in zip(scene.objects, base):
        o.set_pose(p)
    # (2) linear and additive on all of them
    d2 = sl.diff.backpropagate_gradient_to_poses_batch(scene, hyps, g2)
    d12 = sl.diff.backpropagate_gradient_to_poses_batch(scene, hyps, 2.0 * g1 - 0.5 * g2)
    scale = max(float(d1.abs().max()), float(d2.abs().max()))
    assert float((d12 - (2.0 * d1 - 0.5 * d2)).abs().max()) <= 2e-5 * scale
