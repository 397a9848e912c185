"""GPU parity tests of the camera model (slhip_camera_model through the C-ABI): bit-exact against the
oracle (same operation order), within the documented float32 tolerances against the golden vectors of
the REFERENCE's camera_model.py, noise stage by moments."""
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def G():
    return np.load(os.path.join(GOLDEN, "camera_model_golden.npz"))


def det_params(p, seed=0):
    from stillleben_amd import camera_model as cm

    return cm.make_params(p[0:6].reshape(3, 2), p[6:9], float(p[9]), float(p[10]), False, 0.0, 0.0, float(p[11]), seed=seed)


@pytest.mark.parametrize("case", [0, 1, 2, 3])
def test_golden_cases(sl, oracle, G, case):
    from stillleben_amd import camera_model as cm

    img = G["c%d_in" % case]
    p = det_params(G["c%d_params" % case])
    out = cm.process_batch(torch.from_numpy(img)[None].cuda(), [p])[0].cpu().numpy()
    ref = oracle.camera_model(img[None], [p])[0]
    assert np.array_equal(out.view(np.uint32), ref.view(np.uint32)), "HIP != oracle (max %g)" % np.abs(out - ref).max()
    d = np.abs(out - G["c%d_out" % case])
    assert (d > 2e-4).mean() <= 2e-3, "vs reference golden: max %g" % d.max()   # tolerances: tests/test_oracle_camera.py


def test_full_size_batch_bit_exact(sl, oracle):
    """640x480 (the benchmark's resolution), 3 images with different parameters in one launch."""
    from stillleben_amd import camera_model as cm

    rng = np.random.default_rng(5)
    img = rng.random((3, 3, 480, 640), dtype=np.float32)
    img[1] = np.round(img[1] * 4) / 4          # flat regions: ties in max/min, C == 0
    ps = [cm.make_params(rng.uniform(-0.002, 0.002, (3, 2)), rng.uniform(0.998, 1.002, 3), s, d, False, 0, 0, h, seed=0)
          for s, d, h in ((2.1, -2.0, 0.05), (0.0, 1.2, -0.05), (0.9, 0.3, 0.0))]
    out = cm.process_batch(torch.from_numpy(img).cuda(), ps).cpu().numpy()
    ref = oracle.camera_model(img, ps)
    assert np.array_equal(out.view(np.uint32), ref.view(np.uint32))
    assert out.min() >= 0.0 and out.max() <= 1.0


def test_noise_moments(sl):
    """Poissonian-Gaussian stage: y ~ Poisson(x / a) * a + N(0, b^2).  With a flat image, no aberration,
    deltaS = 0 and no hue shift the output is the post blur of the noisy image: mean x, variance
    (a x + b^2) * sum(k^2)."""
    from stillleben_amd import camera_model as cm

    x, a, b = 0.4, 0.02, 0.015
    img = torch.full((1, 3, 256, 256), x).cuda()
    p = cm.make_params(np.zeros((3, 2)), np.ones(3), 0.0, 0.0, True, a, b, 0.0, seed=1234)
    out = cm.process_batch(img, [p])[0].cpu().numpy()[:, 4:-4, 4:-4]
    k = cm._gaussian(0.4).numpy().reshape(-1)
    # R = G = B carries independent noise; the hue round trip is the identity up to rounding for hue_shift = 0
    assert abs(out.mean() - x) < 2e-3
    expect = (a * x + b * b) * float((k * k).sum())
    assert abs(out.var() / expect - 1.0) < 0.08
    # large-rate branch of the Poisson sampler (PTRS) and the small-rate branch (multiplication)
    for xx, aa in ((0.9, 0.001), (0.05, 0.04)):
        q = cm.make_params(np.zeros((3, 2)), np.ones(3), 0.0, 0.0, True, aa, 0.0, 0.0, seed=99)
        o = cm.process_batch(torch.full((1, 3, 256, 256), xx).cuda(), [q])[0].cpu().numpy()[:, 4:-4, 4:-4]
        assert abs(o.mean() - xx) < 3e-3
        assert abs(o.var() / (aa * xx * float((k * k).sum())) - 1.0) < 0.1
    # different seeds give different noise, the same seed the same image
    o1 = cm.process_batch(img, [p])[0]
    assert torch.equal(o1.cpu(), cm.process_batch(img, [p])[0].cpu())
    p2 = cm.make_params(np.zeros((3, 2)), np.ones(3), 0.0, 0.0, True, a, b, 0.0, seed=1235)
    assert not torch.equal(o1.cpu(), cm.process_batch(img, [p2])[0].cpu())


def test_public_api(sl):
    from stillleben_amd import camera_model as cm

    torch.manual_seed(0)
    img = torch.rand(3, 120, 160).cuda()
    out = sl.camera_model.process_image(img)
    assert out.shape == img.shape and out.is_cuda
    assert float(out.min()) >= 0.0 and float(out.max()) <= 1.0
    with pytest.raises(Exception):
        cm.process_deterministic(img.cpu(), torch.zeros(3, 2), torch.ones(3), 0.0, 0.0, False, 0.0, 0.0, 0.0)
