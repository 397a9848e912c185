"""Camera noise model -- 'next' row f4 of SURVEY.md 8f (reference python/stillleben/camera_model.py).

``process_deterministic`` / ``process_image`` run the whole pipeline (chromatic aberration -> blur ->
exposure -> Poissonian-Gaussian noise -> clamp -> hue jitter -> post blur -> clamp) as two fused HIP
kernels through ``slhip_camera_model`` (include/slhip.h); there is no CPU fallback.  The per-stage
helpers of the reference (``chromatic_aberration``, ``blur``, ``exposure``, ``noise``,
``color_jitter``) are kept for API compatibility as thin PyTorch compositions on the tensor's device."""
import ctypes as C
import math
import random

import numpy as np
import torch

from . import _abi, profiling

__all__ = ["chromatic_aberration", "blur", "exposure", "noise", "color_jitter", "process_deterministic",
           "process_image", "process_batch"]


def _gaussian(sigma):
    """5x5 kernel of the reference (camera_model.py:77-104), same float32 operation sequence."""
    ax = torch.arange(5).float()
    x_grid = ax.repeat(5).view(5, 5)
    xy = torch.stack([x_grid, x_grid.t()], dim=-1)
    variance = sigma ** 2.0
    g = (1.0 / (2.0 * math.pi * variance)) * torch.exp(-torch.sum((xy - 2.0) ** 2.0, dim=-1) / (2 * variance))
    g = g / torch.sum(g)
    return g.view(1, 1, 5, 5)


def make_params(chromatic_translation, chromatic_scaling, blur_sigma, exposure_deltaS, do_noise, noise_a, noise_b,
                hue_shift, seed=None):
    """One slhip_camera_params record (numpy) from the arguments of process_deterministic."""
    p = np.zeros((), _abi.CAMERA_DTYPE)
    p["translation"] = torch.as_tensor(chromatic_translation, dtype=torch.float32).reshape(6).numpy()
    p["scaling"] = torch.as_tensor(chromatic_scaling, dtype=torch.float32).reshape(3).numpy()
    p["blur_enabled"] = 1 if blur_sigma > 0.0 else 0
    if blur_sigma > 0.0:
        p["blur_kernel"] = _gaussian(blur_sigma).reshape(25).numpy()
    p["post_kernel"] = _gaussian(0.4).reshape(25).numpy()
    p["exposure_gain"] = np.float32(math.exp(exposure_deltaS))
    p["noise_enabled"] = 1 if do_noise else 0
    p["noise_a"], p["noise_b"] = np.float32(noise_a), np.float32(noise_b)
    p["hue_shift"] = np.float32(hue_shift)
    if seed is None:
        seed = int(torch.randint(0, 2 ** 62, (1,)).item())   # follows torch.manual_seed
    p["seed_lo"], p["seed_hi"] = seed & 0xFFFFFFFF, (seed >> 32) & 0xFFFFFFFF
    return p


def process_batch(rgb, params):
    """Additive batch API: rgb f32[B,3,H,W] on a HIP device, params = list of make_params records."""
    if rgb.dim() != 4 or rgb.size(1) != 3:
        raise ValueError("input tensor has invalid size {}".format(tuple(rgb.size())))
    if not rgb.is_cuda:
        raise _abi.SlhipError("camera_model runs on the HIP device: pass a cuda tensor (there is no CPU path)")
    L = _abi.lib()
    src = rgb.contiguous().float()
    B, _, H, W = src.shape
    if len(params) != B:
        raise ValueError("one parameter record per image")
    rec = np.stack([np.asarray(p, dtype=_abi.CAMERA_DTYPE) for p in params])
    d_params = torch.from_numpy(np.frombuffer(rec.tobytes(), dtype=np.uint8).copy()).to(src.device)
    out, tmp = torch.empty_like(src), torch.empty_like(src)
    stream = torch.cuda.current_stream(src.device).cuda_stream
    with torch.cuda.device(src.device):
        st = L.slhip_camera_model(src.data_ptr(), out.data_ptr(), tmp.data_ptr(), B, H, W, d_params.data_ptr(),
                                  C.c_void_p(stream))
    _abi.check(st, "slhip_camera_model")
    out._keepalive = (d_params, tmp, src)
    return out


def process_deterministic(rgb, chromatic_translation, chromatic_scaling, blur_sigma, exposure_deltaS, do_noise,
                          noise_a, noise_b, hue_shift):
    """Process image with given noise model parameters (camera_model.py:222-263)."""
    assert rgb.dim() == 3
    assert rgb.size(0) == 3
    p = make_params(chromatic_translation, chromatic_scaling, blur_sigma, exposure_deltaS, do_noise, noise_a, noise_b,
                    hue_shift)
    return process_batch(rgb.unsqueeze(0), [p])[0]


@profiling.Timer("camera_model.process_image")
def process_image(rgb):
    """Process image with random noise parameters (camera_model.py:265-286)."""
    assert rgb.dim() == 3
    assert rgb.size(0) == 3
    hue_jitter = 0.05
    return process_deterministic(
        rgb,
        chromatic_translation=torch.empty(3, 2).uniform_(-0.002, 0.002),
        chromatic_scaling=torch.empty(3).uniform_(0.998, 1.002),
        blur_sigma=random.uniform(0.0, 3.0) if random.random() > 0.3 else 0.0,
        exposure_deltaS=random.uniform(-2, 1.2),
        do_noise=random.random() > 0.3,
        noise_a=random.random() * 0.04,
        noise_b=random.random() * 0.02,
        hue_shift=random.uniform(-hue_jitter, hue_jitter),
    )


# ---- per-stage helpers of the reference API (PyTorch compositions, any device) ---------------------
def chromatic_aberration(rgb, translations, scaling):
    assert rgb.dim() == 3 and rgb.size(0) == 3, "input tensor has invalid size {}".format(rgb.size())
    theta = torch.zeros(3, 2, 3)
    theta[:, 0, 0] = scaling
    theta[:, 1, 1] = scaling
    theta[:, 0:2, 2] = translations
    grid = torch.nn.functional.affine_grid(theta.to(rgb.device), (3, 1, rgb.size(1), rgb.size(2)), align_corners=False)
    return torch.nn.functional.grid_sample(rgb.unsqueeze(1), grid, mode="bilinear", padding_mode="reflection",
                                           align_corners=False)[:, 0]


def blur(rgb, sigma):
    return torch.nn.functional.conv2d(rgb.unsqueeze(1), _gaussian(sigma).to(rgb.device), padding=2)[:, 0]


def exposure(rgb, deltaS):
    return 1.0 / (1.0 + math.exp(deltaS) * (1.0 / (rgb + 0.0001) - 1.0))


def noise(rgb, a, b):
    poisson_part = torch.poisson((1.0 / a) * rgb) * a if a > 0.0 else rgb
    gaussian_part = torch.empty_like(rgb).normal_(std=b) if b > 0.0 else torch.zeros_like(rgb)
    return (poisson_part + gaussian_part).clamp_(0.0, 1.0)


def color_jitter(tensor_img, hue_shift):
    """Hue shift through the fused kernel with every other stage neutral is not exact (the resampling
    stage is never a bit-exact identity), so this helper evaluates the HSV round trip with torch."""
    assert tensor_img.size(0) == 3
    M, Mi = tensor_img.max(dim=0)
    m = tensor_img.min(dim=0)[0]
    Cc = M - m
    R, G, B = tensor_img[0], tensor_img[1], tensor_img[2]
    Hh = torch.where(Mi == 0, (G - B) / Cc, torch.where(Mi == 1, (B - R) / Cc + 2.0, (R - G) / Cc + 4.0))
    Hh = torch.where(Cc == 0, torch.zeros_like(Hh), Hh) * 60.0
    Hh = torch.where(Hh < 0, Hh + 360.0, Hh) + hue_shift * 360.0
    Hh = torch.where(Hh < 0, Hh + 360.0, Hh)
    Hh = torch.where(Hh > 360.0, Hh - 360.0, Hh) / 60.0
    X = Cc * (1.0 - (Hh.fmod(2.0) - 1).abs())
    oc = Hh.long().clamp_(0, 5)
    order = torch.tensor([[0, 1, 2], [1, 0, 2], [2, 0, 1], [2, 1, 0], [1, 2, 0], [0, 2, 1]], device=tensor_img.device)
    sel = order[oc.view(-1)].view(tensor_img.size(1), tensor_img.size(2), 3).permute(2, 0, 1)
    return torch.stack((Cc, X, torch.zeros_like(Cc))).gather(0, sel) + m.unsqueeze(0)
