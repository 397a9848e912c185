"""Camera noise model -- OUT OF SCOPE for the hot path ('next' row
f4): the reference's stillleben/camera_model.py is pure torch post-processing that runs on
PyTorch-ROCm unchanged.  Only the deterministic entry point used by examples is provided."""
import torch


def process_deterministic(rgb):
    """Identity camera model (no noise, no blur): float CHW image in [0,1] -> same."""
    return rgb.clamp(0.0, 1.0)


def process_image(rgb):
    """Reference: chromatic aberration -> blur -> exposure -> noise -> blur (camera_model.py:222-286).
    Not part of the accelerated path; returns the deterministic image."""
    return process_deterministic(rgb)
