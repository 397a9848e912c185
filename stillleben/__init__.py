"""Alias package: `import stillleben as sl` resolves to the MI355X-native implementation
(stillleben_amd), so that scripts written against the reference (e.g. its examples/ycb.py)
run unchanged."""
import sys as _sys

import stillleben_amd as _impl
from stillleben_amd import *  # noqa: F401,F403
from stillleben_amd import _set_install_prefix  # noqa: F401
from stillleben_amd import camera_model, diff, extension, losses, profiling  # noqa: F401

__all__ = _impl.__all__
for _name in ("camera_model", "diff", "extension", "losses", "profiling"):
    _sys.modules[__name__ + "." + _name] = getattr(_impl, _name)
