"""Alias package: `import stillleben as sl` resolves to the MI355X-native implementation
(stillleben_amd), so that scripts written against the reference (e.g. its examples/ycb.py)
run unchanged.  The package is assembled the way the reference's python/stillleben/__init__.py:6-13 assembles it: the
sub-modules, then everything the main extension module `stillleben.lib.libstillleben_python` exports."""
import sys as _sys

import stillleben_amd as _impl
from stillleben_amd import camera_model, diff, extension, losses, profiling  # noqa: F401
from .lib.libstillleben_python import *  # noqa: F401,F403
from .lib.libstillleben_python import _set_install_prefix  # noqa: F401
from stillleben_amd import AssetTable, SceneBatch  # noqa: F401  (additive: the batch dimension of the GPU path)

__all__ = _impl.__all__
for _name in ("camera_model", "diff", "extension", "losses", "profiling"):
    _sys.modules[__name__ + "." + _name] = getattr(_impl, _name)
