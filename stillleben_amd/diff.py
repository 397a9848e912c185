"""stillleben.diff -- render-and-compare pose gradients (reference python/stillleben/diff.py,
python/src/bridge_diff.cpp, python/src/diff.cu) on the HIP device.

Same public surface as the reference: ``compute_image_space_gradients``,
``backpropagate_gradient_to_poses``, ``apply_pose_delta`` plus the two native stencils
``generate_sobel_valid_mask`` / ``dilate_object_mask``.  All pixel work runs in
``lib/libslhip.so`` (no CPU implementation: CPU tensors are moved to the HIP device and the
results moved back, so "the device follows the first argument" as in bridge_diff.cpp:27-28)."""
import ctypes as C

import numpy as np
import torch

from . import _abi
from ._context import engine
from .profiling import Timer

__all__ = [
    'compute_image_space_gradients', 'backpropagate_gradient_to_poses', 'apply_pose_delta',
    'generate_sobel_valid_mask', 'dilate_object_mask', 'bp_to_vertices_and_colors', 'soft_forward',
    'backpropagate_gradient_to_poses_batch',
]

DIFF_AVAILABLE = True


def _p(t):
    return C.c_void_p(t.data_ptr())


def _stream(eng):
    return C.c_void_p(torch.cuda.current_stream(eng.device).cuda_stream)


def _native():
    """libstillleben_diff_python: the reference's extension module for these two functions (bridge_diff.cpp:160-180), here
    pybind11 host C++ over the C-ABI (csrc/py/diff_module.cpp).  No fallback: a missing build is an error."""
    try:
        from .lib import libstillleben_diff_python as m
    except ImportError as e:
        raise NotImplementedError("stillleben_amd.lib.libstillleben_diff_python is not built (run __graft_entry__.build()): %s" % e)
    return m


def generate_sobel_valid_mask(instance_indices, depth_image):
    """bridge_diff.cpp:13-69: int16[H,W], float[H,W] -> bool[H,W]."""
    engine()                       # the HIP context of the process (sl.init_cuda): raises when there is none
    return _native().generate_sobel_valid_mask(instance_indices, depth_image)


def dilate_object_mask(object_mask, sobel_valid_mask, coordinates):
    """bridge_diff.cpp:71-157: bool[H,W], bool[H,W], float[H,W,3] -> (bool[H,W], float[H,W,3])."""
    engine()
    return _native().dilate_object_mask(object_mask, sobel_valid_mask, coordinates)


def compute_image_space_gradients(scene, render_result):
    """diff.py:73-127 -> (grad_x [3,H,W], grad_y [3,H,W], sobel_valid_mask bool[H,W])."""
    eng = engine()
    rgb = render_result.rgb()
    out_dev = rgb.device
    H, W = rgb.shape[:2]
    with Timer('sobel_valid_mask'):
        inst = render_result.instance_index().squeeze(-1).to(eng.device).contiguous()
        depth = render_result.depth().to(eng.device).contiguous()
        valid = torch.empty((H, W), dtype=torch.uint8, device=eng.device)
        with torch.cuda.device(eng.device):
            _abi.check(eng.L.slhip_diff_sobel_valid(_p(inst), _p(depth), 1, H, W, _p(valid), _stream(eng)),
                       "slhip_diff_sobel_valid")
    with Timer('sobel'):
        rgb_d = rgb.to(eng.device).contiguous()
        gx = torch.empty((3, H, W), dtype=torch.float32, device=eng.device)
        gy = torch.empty((3, H, W), dtype=torch.float32, device=eng.device)
        with torch.cuda.device(eng.device):
            _abi.check(eng.L.slhip_diff_image_gradients(_p(rgb_d), _p(valid), H, W, _p(gx), _p(gy), _stream(eng)),
                       "slhip_diff_image_gradients")
    return gx.to(out_dev), gy.to(out_dev), valid.bool().to(out_dev)


def backpropagate_gradient_to_poses(scene, render_result, grad_objective_wrt_rnd_img, visualize_grad=False):
    r"""diff.py:355-523: gradient of the objective w.r.t. the locally linearised pose parameters
    (alpha, beta, gamma, a, b, c) of every object -> float[N, 6] (CPU tensor, as the reference)."""
    eng = engine()
    objs = scene.objects
    n = len(objs)
    out = torch.zeros(n, 6)
    if n == 0:
        return out
    rgb = render_result.rgb().to(eng.device).contiguous()
    H, W = rgb.shape[:2]
    coord = render_result.coordDepth().to(eng.device).contiguous()
    inst = render_result.instance_index().squeeze(-1).to(eng.device).contiguous()
    g = grad_objective_wrt_rnd_img.to(eng.device, torch.float32).contiguous()
    if tuple(g.shape) != (3, H, W):
        raise ValueError("grad_objective_wrt_rnd_img must be 3xHxW")
    poses = torch.stack([o.pose() for o in objs]).to(eng.device, torch.float32).contiguous()
    ids = torch.tensor([o.instance_index for o in objs], dtype=torch.int32, device=eng.device)
    P = np.ascontiguousarray(scene.projection_matrix().numpy(), dtype=np.float32)
    valid = torch.empty((H, W), dtype=torch.uint8, device=eng.device)
    acc = torch.empty(6 * n, dtype=torch.float64, device=eng.device)
    res = torch.empty((n, 6), dtype=torch.float32, device=eng.device)
    with torch.cuda.device(eng.device):
        st = eng.L.slhip_diff_pose_backward(_p(rgb), _p(coord), _p(inst), _p(g), C.c_void_p(P.ctypes.data), _p(poses),
                                            _p(ids), n, H, W, _p(valid), _p(acc), _p(res), _stream(eng))
    _abi.check(st, "slhip_diff_pose_backward")
    return res.cpu()


def backpropagate_gradient_to_poses_batch(scene, pose_hypotheses, grad_objective_wrt_rnd_img, ssao=True, return_results=False):
    r"""Additive batch form of render-and-compare (BASELINE config C5: 64 objects x 32 pose hypotheses): renders the scene
    under every pose hypothesis -- `pose_hypotheses` float[K, N, 4, 4], object poses in `scene.objects` order -- in ONE
    slhip_render launch sequence (K scenes in a batch) and backpropagates `grad_objective_wrt_rnd_img` (float[3,H,W], or
    float[K,3,H,W] for one gradient image per hypothesis) through each of them.  Returns float[K, N, 6]: row k equals
    `backpropagate_gradient_to_poses` after `set_pose(pose_hypotheses[k, i])` on every object and `RenderPass().render`.
    The reference has no batch form: it loops hypothesis by hypothesis through the GL renderer (diff.py:355-523)."""
    eng = engine()
    objs = scene.objects
    n = len(objs)
    hyp = pose_hypotheses.detach().cpu().numpy() if hasattr(pose_hypotheses, "detach") else np.asarray(pose_hypotheses)
    hyp = np.ascontiguousarray(hyp, dtype=np.float32)
    if hyp.ndim != 4 or hyp.shape[1:] != (n, 4, 4):
        raise ValueError("pose_hypotheses must be K x %d x 4 x 4" % n)
    K = hyp.shape[0]
    W, H = scene.viewport
    g = grad_objective_wrt_rnd_img.to(eng.device, torch.float32).contiguous()
    per_hyp_grad = g.dim() == 4
    if tuple(g.shape[-3:]) != (3, H, W) or (per_hyp_grad and g.shape[0] != K):
        raise ValueError("grad_objective_wrt_rnd_img must be 3xHxW or Kx3xHxW")
    if K == 0 or n == 0:
        return torch.zeros(K, n, 6)
    # K copies of the scene's descriptors with the hypotheses' poses: the C++ host layer assembles the records of all of them
    # (shadow matrices of every active light included) in one call
    from . import _host_records as HR

    if not HR.eligible([scene], None):
        # sticker decals / a background image: hypothesis by hypothesis through RenderPass
        from .render_pass import RenderPass

        rp = RenderPass()
        rp.ssao_enabled = ssao
        saved = [o.pose() for o in objs]
        out = torch.zeros(K, n, 6)
        try:
            for k in range(K):
                for i, o in enumerate(objs):
                    o.set_pose(torch.from_numpy(hyp[k, i]))
                out[k] = backpropagate_gradient_to_poses(scene, rp.render(scene), (g[k] if per_hyp_grad else g))
        finally:
            for o, p in zip(objs, saved):
                o.set_pose(p)
        return (out, None) if return_results else out
    hs, ho, tmpl = HR.describe([scene], eng.pool)
    hsK = np.repeat(hs, K)
    hsK["obj_begin"] = np.arange(K, dtype=np.uint32) * n
    hsK["obj_end"] = hsK["obj_begin"] + n
    hoK = np.tile(ho, K)
    hoK["pose"] = hyp.reshape(K * n, 16)
    lit = any(bool(np.any(hs[0]["light_dir"][i])) and bool(np.any(hs[0]["light_color"][i])) for i in range(_abi.NUM_LIGHTS))
    srec, drec, crec = HR.build_from(hsK, hoK, tmpl, with_shadows=lit)
    mask = _abi.OUT_RGB | _abi.OUT_COORD | _abi.OUT_INSTANCE
    buf = eng.render_records(srec, drec, crec, W, H, mask, ssao=ssao, shadows=lit)
    res = pose_backward_batch_on_buffers(scene, buf, hyp, g)
    out = res.cpu()
    return (out, buf) if return_results else out


def pose_backward_batch_on_buffers(scene, buf, hyp, g):
    """The backward half of backpropagate_gradient_to_poses_batch on render targets that are already there: `buf` the [K,H,W,..]
    buffers of the K hypotheses' renders, `hyp` float32[K,N,4,4] (numpy), `g` the gradient image(s) on the device.  ONE launch
    sequence (slhip_diff_pose_backward_batch: the hypothesis is a grid dimension).  Returns float32[K,N,6] on the device."""
    eng = engine()
    objs = scene.objects
    n, K = len(objs), hyp.shape[0]
    W, H = scene.viewport
    per_hyp_grad = g.dim() == 4
    d_poses = torch.from_numpy(np.ascontiguousarray(hyp, dtype=np.float32)).to(eng.device)
    ids = torch.tensor([o.instance_index for o in objs], dtype=torch.int32, device=eng.device)
    P = np.ascontiguousarray(scene.projection_matrix().numpy(), dtype=np.float32)
    valid = torch.empty((K, H, W), dtype=torch.uint8, device=eng.device)
    acc = torch.empty(K * 6 * n, dtype=torch.float64, device=eng.device)
    res = torch.empty((K, n, 6), dtype=torch.float32, device=eng.device)
    with torch.cuda.device(eng.device):
        st = eng.L.slhip_diff_pose_backward_batch(_p(buf.rgb), _p(buf.coord), _p(buf.instance), _p(g), 3 * H * W if per_hyp_grad else 0,
                                                  C.c_void_p(P.ctypes.data), _p(d_poses), _p(ids), n, K, H, W, _p(valid), _p(acc),
                                                  _p(res), _stream(eng))
    _abi.check(st, "slhip_diff_pose_backward_batch")
    return res


def bp_to_vertices_and_colors(scene, render_result, grad_objective_wrt_rnd_img, visualize_grad=False):
    r"""diff.py:215-352 (row D6): backpropagates an image-space gradient to the vertices (and vertex colours)
    of the meshes.  Returns three lists with one entry per rendered object: vertex indices int32[3P] (the
    1-based ids of the vertex-index target, pixel by pixel in row-major order), -d objective / d vertex
    f32[3P,3] and -d objective / d colour f32[3P,3] (barycentric-weighted, ready for Mesh.update_*)."""
    eng = engine()
    objs = scene.objects
    rgb = render_result.rgb().to(eng.device).contiguous()
    H, W = rgb.shape[:2]
    out_dev = render_result.rgb().device
    g = grad_objective_wrt_rnd_img.to(eng.device, torch.float32).contiguous()
    if tuple(g.shape) != (3, H, W):
        raise ValueError("grad_objective_wrt_rnd_img must be 3xHxW")
    vertex_index_2_bp, grad_vertices_2_bp, grad_colors_2_bp = [], [], []
    if not objs:
        return vertex_index_2_bp, grad_vertices_2_bp, grad_colors_2_bp
    coord = render_result.coordDepth().to(eng.device).contiguous()
    inst = render_result.instance_index().squeeze(-1).to(eng.device).contiguous()
    bary = render_result.barycentric_coeffs().to(eng.device, torch.float32)
    if bary.shape[-1] == 3:
        bary = torch.cat([bary, torch.zeros_like(bary[..., :1])], dim=-1)
    bary = bary.contiguous()
    vidx = render_result.vertex_indices().to(eng.device)[..., :3].reshape(-1, 3)
    poses = torch.stack([o.pose() for o in objs]).to(eng.device, torch.float32).contiguous()
    ids = torch.tensor([o.instance_index for o in objs], dtype=torch.int32, device=eng.device)
    P = np.ascontiguousarray(scene.projection_matrix().numpy(), dtype=np.float32)
    valid = torch.empty((H, W), dtype=torch.uint8, device=eng.device)
    gv = torch.empty((H * W, 3, 3), dtype=torch.float32, device=eng.device)
    gc = torch.empty((H * W, 3, 3), dtype=torch.float32, device=eng.device)
    with torch.cuda.device(eng.device):
        st = eng.L.slhip_diff_vertex_backward(_p(rgb), _p(coord), _p(inst), _p(bary), _p(g), C.c_void_p(P.ctypes.data), _p(poses),
                                              _p(ids), len(objs), H, W, _p(valid), _p(gv), _p(gc), _stream(eng))
    _abi.check(st, "slhip_diff_vertex_backward")
    flat_inst = inst.reshape(-1)
    for obj in objs:
        m = flat_inst == obj.instance_index          # the reference's boolean indexing: pixels in row-major order
        if not bool(m.any()):
            print('instance_index image for the current object is empty')
            print('object not rendered as a part of the scene')
            continue
        vertex_index_2_bp.append(vidx[m].reshape(-1).to(out_dev))
        grad_vertices_2_bp.append(gv[m].reshape(-1, 3).to(out_dev))
        grad_colors_2_bp.append(gc[m].reshape(-1, 3).to(out_dev))
    return vertex_index_2_bp, grad_vertices_2_bp, grad_colors_2_bp


def soft_forward(scene, render_result, obs_rgb, loss_fn):
    """diff.py:130-213: blends the depth-peeled renders (weights 0.7, 0.3, 0.1, 0.1, 0.05), blurs with the 11x11
    Gaussian, evaluates `loss_fn` against the observation and backpropagates the image gradient of every peel layer
    to vertices and colours.  `render_result` is the list of results in peeling order.  (The reference additionally
    demands an `is_extracted` attribute that nothing in its tree sets, diff.py:153 -- not required here.)"""
    import types

    if not isinstance(render_result, (list, tuple)):
        raise ValueError("render_result should be a list or tuple")
    if obs_rgb.dim() != 3:
        raise ValueError("Observed RGB should have 3 dimension CxHxW")
    if obs_rgb.shape[0] != 3:
        raise ValueError("Observed RGB should of format CxHxW with C=3")
    if obs_rgb.dtype != torch.float32:
        raise ValueError("Observed RGB should be of type torch.float32")
    if obs_rgb.max().item() > 1:
        raise ValueError("Observed RGB should have range [0,1]")
    if not isinstance(loss_fn, types.FunctionType):
        raise ValueError("loss_fn should be a callable function")
    for rr in render_result:
        if rr.rgb().shape != render_result[0].rgb().shape:
            raise ValueError("render_results should correspond to the same scene")
    device = render_result[0].rgb().device
    rgbs = torch.stack([rr.rgb()[:, :, :3].permute(2, 0, 1).float() / 255.0 for rr in render_result]).detach()
    rgbs.requires_grad = True
    weights = torch.tensor([0.7, 0.3, 0.1, 0.1, 0.05], device=device)
    soft_rgb = (rgbs * weights[: rgbs.shape[0], None, None, None]).sum(dim=0)
    ks, sig = 11, 1.0                                                  # diff.py:61-71
    ax = torch.arange(ks, dtype=torch.float32, device=device) - (ks - 1) / 2.0
    k1 = torch.exp(-0.5 * (ax / sig) ** 2)
    kernel = (k1[:, None] * k1[None, :])
    kernel = (kernel / kernel.sum()).view(1, 1, ks, ks)
    soft_gauss = torch.nn.functional.conv2d(soft_rgb.unsqueeze(1), kernel, padding=ks // 2).squeeze(1)
    loss, loss_img = loss_fn(soft_gauss.unsqueeze(0), obs_rgb.to(device).unsqueeze(0))
    loss.backward()
    grads = rgbs.grad.clone()
    vertex_index_2_bp, grad_vertices_2_bp, grad_colors_2_bp = [], [], []
    for ir, rr in enumerate(render_result):
        vi, gv, gc = bp_to_vertices_and_colors(scene, rr, grads[ir])
        vertex_index_2_bp += vi
        grad_vertices_2_bp += gv
        grad_colors_2_bp += gc
    return (soft_rgb.detach().clone(), [r.clone().detach() for r in rgbs], loss_img, loss.item(), vertex_index_2_bp,
            grad_vertices_2_bp, grad_colors_2_bp)


def apply_pose_delta(pose, delta, orthonormalize=True):
    r"""diff.py:525-590 (host-side in the reference too): pose . [I + skew(alpha,beta,gamma) | abc]."""
    if pose.dim() == 3:
        assert delta.dim() == 2
        batched = True
    else:
        assert delta.dim() == 1
        batched = False
        pose = pose.unsqueeze(0)
        delta = delta.unsqueeze(0)
    device = pose.device
    pose = pose.cpu().float()
    delta = delta.cpu().float()
    B = pose.size(0)
    dm = torch.zeros(B, 4, 4)
    dm[:, 0, 0] = 1.0; dm[:, 0, 1] = -delta[:, 2]; dm[:, 0, 2] = delta[:, 1]
    dm[:, 1, 0] = delta[:, 2]; dm[:, 1, 1] = 1.0; dm[:, 1, 2] = -delta[:, 0]
    dm[:, 2, 0] = -delta[:, 1]; dm[:, 2, 1] = delta[:, 0]; dm[:, 2, 2] = 1.0
    dm[:, :3, 3] = delta[:, 3:]
    dm[:, 3, 3] = 1.0
    new_poses = torch.matmul(pose, dm)
    if orthonormalize:
        for b in range(B):
            U, S, Vh = torch.linalg.svd(new_poses[b, :3, :3])
            new_poses[b, :3, :3] = U @ Vh
    if not batched:
        new_poses = new_poses[0]
    return new_poses.to(device)
