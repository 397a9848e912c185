"""Camera model ('next' row f4): the oracle (oracle/camera_ref.c) against golden vectors produced by the
REFERENCE's own python/stillleben/camera_model.py (oracle/ref_build/gen_camera_golden.py), stage by
stage, and the host-side Gaussian kernels against the reference's `_gaussian`."""
import os

import numpy as np
import pytest

from conftest import GOLDEN


@pytest.fixture(scope="module")
def G():
    return np.load(os.path.join(GOLDEN, "camera_model_golden.npz"))


def params_of(G, i, stage_blur=True):
    from stillleben_amd import camera_model as cm

    p = G["c%d_params" % i]
    return cm.make_params(p[0:6].reshape(3, 2), p[6:9], float(p[9]), float(p[10]), False, 0.0, 0.0, float(p[11]), seed=0)


def test_gaussian_kernels_bit_exact(G):
    from stillleben_amd import camera_model as cm

    for j in range(4):
        g, sigma = G["gauss_%d" % j], float(G["gauss_sigma_%d" % j])
        mine = cm._gaussian(sigma).reshape(-1).numpy()
        assert np.array_equal(mine.view(np.uint32), g.view(np.uint32)), "sigma %g" % sigma


# Tolerances: the resampling and the two convolutions are float32 sums whose order torch does not
# document (grid_sample / conv2d CPU kernels); everything else is elementwise.  The hue stage is
# discontinuous (arg-max / sector selection), so an input that differs in the last bit can flip a
# sector for isolated pixels: a small outlier fraction is allowed there and after it.
# Re-exposure has slope up to 1 / e^deltaS near black (7.4 for deltaS = -2), which scales the 2e-6.
# The sampling position itself is a float32 of magnitude ~W (ulp 4e-6 .. 8e-6 px) whose last bit depends
# on how affine_grid's matmul rounds: times the local image gradient that is up to ~1e-5 in value.
STAGES = [("chromatic", 0, 1e-5, 0.0), ("blur", 1, 1e-5, 0.0), ("exposure", 2, 8e-5, 0.0), ("jitter", 3, 2e-4, 2e-3),
          ("out", 4, 2e-4, 2e-3)]


@pytest.mark.parametrize("case", [0, 1, 2, 3])
def test_oracle_matches_reference_stages(oracle, G, case):
    img = G["c%d_in" % case]
    p = params_of(G, case)
    for name, stage, tol, outliers in STAGES:
        ref = G["c%d_%s" % (case, name)]
        got = oracle.camera_model(img[None], [p], stage=stage)[0]
        d = np.abs(got - ref)
        bad = (d > tol).mean()
        assert bad <= outliers, "case %d stage %s: %.2e of the pixels differ by more than %g (max %g)" % (
            case, name, bad, tol, d.max())


def test_identity_parameters_stay_close(oracle):
    """No aberration, no blur, deltaS = 0, no hue shift: only the 0.4-sigma post blur acts."""
    from stillleben_amd import camera_model as cm

    rng = np.random.default_rng(0)
    img = rng.random((1, 3, 24, 31), dtype=np.float32)
    p = cm.make_params(np.zeros((3, 2)), np.ones(3), 0.0, 0.0, False, 0.0, 0.0, 0.0, seed=0)
    mid = oracle.camera_model(img, [p], stage=3)
    assert np.abs(mid - img).max() < 2e-4          # exposure adds/removes its 1e-4 epsilon
    out = oracle.camera_model(img, [p], stage=4)
    k = cm._gaussian(0.4).reshape(5, 5).numpy()
    assert abs(float(k.sum()) - 1.0) < 1e-6 and k[2, 2] > 0.8
    assert np.abs(out[:, :, 2:-2, 2:-2] - img[:, :, 2:-2, 2:-2]).max() < 0.25


def test_noise_stage_is_not_restated(oracle):
    from stillleben_amd import camera_model as cm

    p = cm.make_params(np.zeros((3, 2)), np.ones(3), 0.0, 0.0, True, 0.01, 0.01, 0.0, seed=1)
    with pytest.raises(ValueError):
        oracle.camera_model(np.zeros((1, 3, 4, 4), np.float32), [p])
