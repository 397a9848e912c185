"""Process-global context: sl.init() / sl.init_cuda() (reference python/src/py_context.cpp:17-52,
src/context.cpp:392-560).  There is no EGL/GL context any more; the context records which HIP
device renders/simulates and where result tensors live."""
import warnings

_CTX = None


class Context:
    def __init__(self, device_index, cuda_outputs):
        self.device_index = device_index  # HIP device that does all the work
        self.cuda_outputs = cuda_outputs  # True: tensors stay on cuda:<idx> (init_cuda)
        self.install_prefix = None
        self.engine = None                # created lazily on first device use


def init():
    """Results are returned as CPU tensors (the compute still runs on HIP device 0)."""
    global _CTX
    if _CTX is not None:
        if _CTX.cuda_outputs:
            warnings.warn("stillleben context was already created with different settings")
        return
    _CTX = Context(0, False)


def init_cuda(device_index=0, use_cuda=True):
    """Results stay on ``cuda:<device_index>`` (zero-copy: kernels write into torch storage)."""
    global _CTX
    if _CTX is not None:
        if _CTX.device_index != device_index or _CTX.cuda_outputs != bool(use_cuda):
            warnings.warn("stillleben context was already created with different settings")
        return
    _CTX = Context(int(device_index), bool(use_cuda))


def _set_install_prefix(path):
    if _CTX is not None:
        _CTX.install_prefix = str(path)


def require_context():
    if _CTX is None:
        raise RuntimeError("Call sl::init() first")  # py_context.cpp:69-75
    return _CTX


def engine():
    ctx = require_context()
    if ctx.engine is None:
        from ._engine import Engine

        ctx.engine = Engine(ctx.device_index)
    return ctx.engine


def _reset_for_tests():
    global _CTX
    _CTX = None


def check_free_memory(device, need_bytes, what):
    """Raises a RuntimeError that says what is being sized, how much it needs and how much the device has left (free memory of
    the device + what torch's caching allocator holds unused) BEFORE a large allocation runs into torch's out-of-memory error --
    scratch sizes follow from the batch shape (scenes per settle launch, scenes per render chunk, list capacities), so the
    message names the knobs.  Small requests are not checked."""
    import torch

    if need_bytes < (256 << 20) or not torch.cuda.is_available():
        return
    free, total = torch.cuda.mem_get_info(device)
    cached = torch.cuda.memory_reserved(device) - torch.cuda.memory_allocated(device)
    if need_bytes > free + cached:
        raise RuntimeError("%s needs %.1f GB of device memory, %.1f GB are free (of %.1f GB; another process may hold the rest): "
                           "use fewer scenes per batch / per render chunk or smaller list capacities"
                           % (what, need_bytes / 1e9, (free + cached) / 1e9, total / 1e9))
