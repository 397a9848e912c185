"""Native libraries of the package, built in-tree by __graft_entry__.build(): libslhip.so (the C-ABI: HIP kernels + host C++,
loaded through ctypes by _abi.py) and libstillleben_diff_python (pybind11 host C++ over the C-ABI -- the reference's second
extension module, python/src/bridge_diff.cpp:160-180; `from stillleben.lib import libstillleben_diff_python` as in the
reference's python/stillleben/diff.py:22-30)."""
