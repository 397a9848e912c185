"""Import-compatibility shims for the parts of the reference API that are OUTSIDE the hot path
 plus the host utilities of row f4 (ImageSaver, ImageLoader, Animator):
the shims keep user scripts importable and fail (or warn)
explicitly when used."""
import warnings

import numpy as np


class Texture2D:
    """2D texture from a file path or an HxWx3/4 uint8 tensor (python/src/py_magnum.cpp:117-197).
    Used for Scene.background_plane_texture."""

    def __init__(self, src):
        if isinstance(src, (str, bytes)) or hasattr(src, "__fspath__"):
            from PIL import Image

            arr = np.asarray(Image.open(str(src)).convert("RGBA"), dtype=np.uint8)
        else:
            if hasattr(src, "detach"):
                src = src.detach().cpu().numpy()
            arr = np.asarray(src, dtype=np.uint8)
            if arr.ndim != 3 or arr.shape[2] not in (3, 4):
                raise ValueError("expected an HxWx3 or HxWx4 uint8 image")
            if arr.shape[2] == 3:
                arr = np.concatenate([arr, np.full(arr.shape[:2] + (1,), 255, np.uint8)], axis=2)
        self._rgba = np.ascontiguousarray(arr)


class Texture(Texture2D):
    """Rectangle texture (background images) -- kept for import compatibility."""


class _OutOfScope:
    _what = "this component"

    def __init__(self, *a, **k):
        raise NotImplementedError("%s is outside the hot-path scope of stillleben_amd (SURVEY.md section 2)" % self._what)


from .light_map import LightMap  # noqa: E402,F401  (image-based lighting, 'next' row f1)


class Viewer(_OutOfScope):
    _what = "the interactive X11 viewer"


class ImageSaver:
    """Multi-threaded asynchronous image writer ('next' row f4; reference src/image_saver.cpp:20-110,
    python/src/py_image_saver.cpp:24-108).  ``save`` returns once the job is queued (it blocks while
    2 x n_threads jobs are pending, image_saver.cpp:97-108); every image is on disk when ``__exit__``
    returns.  Accepts uint8 HxWx3 / HxWx4 / HxW and int16 HxW (written as 16-bit grayscale) tensors; device
    tensors are brought to the host with a non-blocking copy on a side stream so the render stream
    keeps running."""

    def __init__(self):
        self._pool = None

    def __enter__(self):
        import os
        import queue
        import threading

        n = max(1, min(os.cpu_count() or 1, 16))
        self._queue = queue.Queue(maxsize=2 * n)
        self._errors = []
        self._pool = [threading.Thread(target=self._worker, daemon=True) for _ in range(n)]
        for t in self._pool:
            t.start()
        return self

    def _worker(self):
        from PIL import Image

        while True:
            job = self._queue.get()
            if job is None:
                return
            tensor, event, path = job
            try:
                if event is not None:
                    event.synchronize()
                arr = tensor.numpy()
                if arr.dtype == np.int16:
                    img = Image.fromarray(arr.view(np.uint16))
                else:
                    img = Image.fromarray(arr)
                img.save(path)
            except Exception as e:  # reported from __exit__, like the reference's thread exception
                self._errors.append("Could not write image %s: %s" % (path, e))
            finally:
                self._queue.task_done()

    def save(self, image, path):
        import torch

        if self._pool is None:
            raise RuntimeError("Call __enter__() first")
        if image.dim() == 3:
            if image.size(2) not in (3, 4):
                raise ValueError("Color images need to have shape HxWx3 or HxWx4")
            if image.dtype != torch.uint8:
                raise ValueError("Color images need to have type uint8")
        elif image.dim() == 2:
            if image.dtype not in (torch.uint8, torch.int16):
                raise ValueError("Grayscale images need to be byte or short type")
        else:
            raise ValueError("Images need to have shape HxW, HxWx3 or HxWx4")
        event = None
        if image.is_cuda:
            host = torch.empty(image.shape, dtype=image.dtype, pin_memory=True)
            side = self._side_stream(image.device)
            side.wait_stream(torch.cuda.current_stream(image.device))
            with torch.cuda.stream(side):
                host.copy_(image.contiguous(), non_blocking=True)
                image.record_stream(side)
                event = torch.cuda.Event()
                event.record(side)
            image = host
        else:
            image = image.detach().contiguous().clone()
        self._queue.put((image, event, str(path)))

    def _side_stream(self, device):
        import torch

        if not hasattr(self, "_streams"):
            self._streams = {}
        if device not in self._streams:
            self._streams[device] = torch.cuda.Stream(device)
        return self._streams[device]

    def __exit__(self, *exc):
        self._queue.join()
        for _ in self._pool:
            self._queue.put(None)
        for t in self._pool:
            t.join()
        self._pool = None
        if self._errors and exc[0] is None:
            raise RuntimeError(self._errors[0])
        return False


class ImageLoader:
    """Multi-threaded random image loader (reference src/image_loader.cpp:27-235,
    python/src/py_image_loader.cpp:18-52): worker threads decode randomly drawn files of ``path`` ahead
    of the consumer; ``next()`` returns a rectangle ``Texture`` (background images, stickers),
    ``next_texture2d()`` a ``Texture2D``.  Unreadable files and images that are not 8-bit RGB/RGBA are skipped."""

    def __init__(self, path, seed=None):
        import os
        import queue
        import random
        import threading

        path = str(path)
        names = sorted(n for n in os.listdir(path) if not os.path.isdir(os.path.join(path, n)))
        self._paths = [os.path.join(path, n) for n in names]
        if not self._paths:
            raise RuntimeError("Could not find any images in '%s'" % path)
        self._rng = random.Random(seed)
        self._in, self._out = queue.Queue(), queue.Queue()
        n = max(1, min(os.cpu_count() or 1, 16))
        self._threads = [threading.Thread(target=self._worker, daemon=True) for _ in range(n)]
        for t in self._threads:
            t.start()
            self._enqueue()

    def _enqueue(self):
        self._in.put(self._paths[self._rng.randrange(len(self._paths))])

    def _worker(self):
        from PIL import Image

        while True:
            p = self._in.get()
            if p is None:
                return
            try:
                with Image.open(p) as im:
                    if im.mode not in ("RGB", "RGBA"):
                        self._out.put(None)      # image_loader.cpp:190-197: only RGB8 / RGBA8 are taken
                        continue
                    self._out.put(np.asarray(im, dtype=np.uint8).copy())
            except Exception:
                self._out.put(None)

    def _next_array(self):
        import time

        errors = 0
        while True:
            if errors >= 10:
                warnings.warn("ImageLoader: 10 errors in a row, probably something is wrong with your images")
                time.sleep(0.5)
                errors = 0
            self._enqueue()
            arr = self._out.get()
            if arr is None:
                errors += 1
                continue
            return arr

    def next(self):
        """Return next image (randomly sampled).  This is the same as next_rectangle_texture()."""
        return Texture(self._next_array())

    next_rectangle_texture = next

    def next_texture2d(self):
        return Texture2D(self._next_array())

    def close(self):
        for _ in self._threads:
            self._in.put(None)
        for t in self._threads:
            t.join()
        self._threads = []

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Animator:
    """Interpolates between 4x4 poses over ``ticks`` steps (reference src/animator.cpp:17-55,
    python/src/py_animator.cpp:43-63): keyframe i sits at tick ``i * ticks // (n - 1)``, positions are
    interpolated linearly and orientations by normalised linear quaternion interpolation."""

    def __init__(self, poses, ticks):
        from ._math import matrix_to_quat

        if len(poses) < 2:
            raise ValueError("Need at least two poses to animate...")
        n = len(poses)
        self._ticks = int(ticks)
        self._times = [i * self._ticks // (n - 1) for i in range(n)]
        mats = [np.asarray(p.detach().cpu().numpy() if hasattr(p, "detach") else p, dtype=np.float32) for p in poses]
        self._pos = [m[:3, 3].copy() for m in mats]
        self._quat = [np.asarray(matrix_to_quat(m[:3, :3]), dtype=np.float32).reshape(4) for m in mats]
        self._index = 0

    def __iter__(self):
        return self

    def __len__(self):
        return self._ticks

    def _segment(self, t):
        times = self._times
        if t <= times[0]:
            return 0, 0, 0.0
        if t >= times[-1]:
            return len(times) - 1, len(times) - 1, 0.0
        i = 0
        while times[i + 1] <= t:
            i += 1
        return i, i + 1, float(t - times[i]) / float(times[i + 1] - times[i])

    def __next__(self):
        import torch

        from ._math import quat_to_matrix

        if self._index >= self._ticks:
            raise StopIteration
        a, b, f = self._segment(self._index)
        self._index += 1
        pos = (1.0 - f) * self._pos[a] + f * self._pos[b]
        q = (1.0 - f) * self._quat[a] + f * self._quat[b]
        q = q / np.linalg.norm(q)
        out = torch.eye(4)
        out[:3, :3] = torch.as_tensor(np.asarray(quat_to_matrix(q), dtype=np.float32))
        out[:3, 3] = torch.from_numpy(pos.astype(np.float32))
        return out


class MeshCache:
    """filename -> Mesh cache used by Scene.deserialize (reference src/mesh_cache.cpp:21-46)."""

    def __init__(self):
        self._meshes = {}

    def add(self, meshes):
        for m in (meshes if isinstance(meshes, (list, tuple)) else [meshes]):
            self._meshes[m.filename] = m

    def load(self, filename):
        from .mesh import Mesh

        m = self._meshes.get(str(filename))
        if m is None:
            m = Mesh(filename)
            self._meshes[str(filename)] = m
        return m


def view(scene):
    """The reference opens an interactive viewer (python/src/py_viewer.cpp:20-57); headless
    here, so that examples/ycb.py:77 keeps running."""
    warnings.warn("sl.view(): no interactive viewer in stillleben_amd (headless); continuing")


def render_debug_image(scene):
    raise NotImplementedError("render_debug_image is outside the hot-path scope")
