"""`stillleben.lib` as the reference lays it out (python/stillleben/__init__.py:12-13, diff.py:22-30): the extension modules
live in stillleben_amd/lib and are aliased here."""
import importlib as _il
import sys as _sys

libstillleben_diff_python = _il.import_module("stillleben_amd.lib.libstillleben_diff_python")
_sys.modules[__name__ + ".libstillleben_diff_python"] = libstillleben_diff_python

# the main module (python/src/bridge.cpp:23-42) under the name the reference imports it by
libstillleben_python = _il.import_module("stillleben_amd.lib.libstillleben_python")
_sys.modules[__name__ + ".libstillleben_python"] = libstillleben_python
