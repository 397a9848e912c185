#!/usr/bin/env python3
"""Generates tests/golden/camera_model_golden.npz from the REFERENCE itself (run in the build
container, where /root/reference exists): python/stillleben/camera_model.py is imported by path
(with a stub `stillleben.profiling`) and its deterministic stages are evaluated on CPU torch for
seeded inputs.  The random stage (`noise`) is only pinned through its moments in the tests."""
import importlib.util
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"


def load_reference_camera_model():
    pkg = types.ModuleType("stillleben")
    pkg.__path__ = [os.path.join(REF, "python", "stillleben")]
    prof = types.ModuleType("stillleben.profiling")

    def Timer(name):
        def deco(f):
            return f
        return deco

    prof.Timer = Timer
    sys.modules.update({"stillleben": pkg, "stillleben.profiling": prof})
    spec = importlib.util.spec_from_file_location("stillleben.camera_model",
                                                  os.path.join(REF, "python", "stillleben", "camera_model.py"))
    cm = importlib.util.module_from_spec(spec)
    sys.modules["stillleben.camera_model"] = cm
    spec.loader.exec_module(cm)
    return cm


def test_image(H, W, seed):
    """Smooth colour blobs + texture + a few saturated / black / grey regions (hue edge cases)."""
    rng = np.random.default_rng(seed)
    y, x = np.mgrid[0:H, 0:W].astype(np.float32)
    img = np.zeros((3, H, W), np.float32)
    for c in range(3):
        fx, fy, ph = rng.uniform(0.02, 0.3, 3)
        img[c] = 0.5 + 0.35 * np.sin(fx * x + ph) * np.cos(fy * y - ph) + 0.15 * rng.random((H, W), dtype=np.float32)
    img = np.clip(img, 0.0, 1.0)
    img[:, : H // 6, : W // 6] = 0.0                    # black: M == 0
    img[:, -H // 6:, -W // 6:] = 1.0                    # white: C == 0
    img[:, : H // 6, -W // 6:] = 0.5                    # grey
    img[0, H // 3: H // 2, : W // 5] = 1.0              # pure-ish red
    img[1:, H // 3: H // 2, : W // 5] = 0.0
    return img


CASES = [
    # H, W, seed, chromatic translation (3x2), scaling (3), blur sigma, deltaS, hue shift
    dict(H=37, W=53, seed=1, tr=[[0.002, -0.001], [0.0, 0.0], [-0.0015, 0.002]], sc=[1.002, 1.0, 0.998], sigma=1.3, dS=0.4, hue=0.03),
    dict(H=48, W=64, seed=2, tr=[[0.0, 0.0]] * 3, sc=[1.0, 1.0, 1.0], sigma=0.0, dS=-1.7, hue=-0.05),
    dict(H=64, W=40, seed=3, tr=[[-0.002, 0.002], [0.001, 0.001], [0.002, -0.002]], sc=[0.998, 1.001, 1.002], sigma=2.9, dS=1.2, hue=0.5),
    dict(H=21, W=19, seed=4, tr=[[0.05, -0.08], [0.0, 0.1], [-0.12, 0.03]], sc=[1.1, 0.9, 1.05], sigma=0.7, dS=0.0, hue=-0.33),
]


def main():
    cm = load_reference_camera_model()
    out = {"n_cases": np.int32(len(CASES))}
    for i, c in enumerate(CASES):
        img = test_image(c["H"], c["W"], c["seed"])
        rgb = torch.from_numpy(img)
        tr = torch.tensor(c["tr"], dtype=torch.float32)
        sc = torch.tensor(c["sc"], dtype=torch.float32)
        stages = {}
        stages["chromatic"] = cm.chromatic_aberration(rgb, tr, sc)
        stages["blur"] = cm.blur(stages["chromatic"], c["sigma"]) if c["sigma"] > 0 else stages["chromatic"]
        stages["exposure"] = cm.exposure(stages["blur"], c["dS"])
        stages["jitter"] = cm.color_jitter(stages["exposure"].clamp(0.0, 1.0), c["hue"])
        final = cm.process_deterministic(rgb.clone(), tr, sc, c["sigma"], c["dS"], False, 0.0, 0.0, c["hue"])
        out["c%d_in" % i] = img
        out["c%d_params" % i] = np.array(sum(c["tr"], []) + c["sc"] + [c["sigma"], c["dS"], c["hue"]], np.float64)
        for k, v in stages.items():
            out["c%d_%s" % (i, k)] = v.numpy().astype(np.float32)
        out["c%d_out" % i] = final.numpy().astype(np.float32)
    # Gaussian kernels of the reference for a few sigmas (host code must reproduce them exactly)
    for j, s in enumerate([0.4, 0.7, 1.3, 2.9]):
        out["gauss_%d" % j] = cm._gaussian(s).numpy().reshape(-1).astype(np.float32)
        out["gauss_sigma_%d" % j] = np.float64(s)
    dst = os.path.join(ROOT, "tests", "golden", "camera_model_golden.npz")
    np.savez_compressed(dst, **out)
    print("wrote", dst, os.path.getsize(dst), "bytes")


if __name__ == "__main__":
    main()
