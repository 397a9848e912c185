"""Helper for building user extensions against this package (counterpart of the reference's
stillleben/extension.py, which wraps torch.utils.cpp_extension)."""
import os


def include_dirs():
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    return [os.path.join(root, "include")]


def load(name, sources, **kwargs):
    from torch.utils import cpp_extension

    kwargs.setdefault("extra_include_paths", []).extend(include_dirs())
    return cpp_extension.load(name=name, sources=sources, **kwargs)
