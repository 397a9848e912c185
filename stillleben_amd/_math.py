"""Small fp32 linear-algebra helpers mirroring the Magnum calls of the reference (SURVEY.md
Appendix C).  All matrices are row-major 4x4 float32 numpy arrays (== the tensors of the
reference's Python API)."""
import numpy as np

f32 = np.float32


def as_mat4(x):
    """Accepts torch tensors / numpy / nested lists (any device, any dtype) like the
    reference's converters do (python/src/py_magnum.h:55-69)."""
    if hasattr(x, "detach"):
        x = x.detach().to("cpu").float().numpy()
    m = np.array(x, dtype=np.float32)
    if m.shape != (4, 4):
        raise ValueError("expected a 4x4 matrix, got shape %s" % (m.shape,))
    return m


def as_vec(x, n):
    if hasattr(x, "detach"):
        x = x.detach().to("cpu").float().numpy()
    v = np.array(x, dtype=np.float32).reshape(-1)
    if v.shape[0] != n:
        raise ValueError("expected a vector of length %d, got %d" % (n, v.shape[0]))
    return v


def inverted_rigid(m):
    """Matrix4::invertedRigid: [R^T | -R^T t]."""
    r = m[:3, :3].astype(np.float32)
    t = m[:3, 3].astype(np.float32)
    out = np.eye(4, dtype=np.float32)
    out[:3, :3] = r.T
    # k-ordered products and sums (csrc/slhip_records.cpp inverted_rigid): numpy's matmul leaves the order to its BLAS
    s = r.T[:, 0] * t[0]
    s = s + r.T[:, 1] * t[1]
    s = s + r.T[:, 2] * t[2]
    out[:3, 3] = -s
    return out


def normal_matrix(m):
    """Matrix4::normalMatrix(): inverse-transpose of the upper-left 3x3 -- cofactors over the determinant in float64, in the
    operation order of normal_matrix() in csrc/slhip_records.cpp (the batch path): both give the same bits."""
    a = m[:3, :3].astype(np.float64)
    m00, m01, m02 = float(a[0, 0]), float(a[0, 1]), float(a[0, 2])
    m10, m11, m12 = float(a[1, 0]), float(a[1, 1]), float(a[1, 2])
    m20, m21, m22 = float(a[2, 0]), float(a[2, 1]), float(a[2, 2])
    c = [m11 * m22 - m12 * m21, m12 * m20 - m10 * m22, m10 * m21 - m11 * m20,
         m02 * m21 - m01 * m22, m00 * m22 - m02 * m20, m01 * m20 - m00 * m21,
         m01 * m12 - m02 * m11, m02 * m10 - m00 * m12, m00 * m11 - m01 * m10]
    det = m00 * c[0] + m01 * c[1] + m02 * c[2]
    with np.errstate(all="ignore"):
        return (np.array(c, np.float64) / det).astype(np.float32).reshape(3, 3)


def mul44(a, b):
    """4x4 product in float32 with the k-ordered sums of mul() in csrc/slhip_records.cpp (numpy's matmul leaves the order to
    its BLAS): the per-scene and the batch path of the record assembly multiply pose by pretransform the same way."""
    a = a.astype(np.float32)
    b = b.astype(np.float32)
    s = a[:, 0:1] * b[0:1, :]
    s = s + a[:, 1:2] * b[1:2, :]
    s = s + a[:, 2:3] * b[2:3, :]
    s = s + a[:, 3:4] * b[3:4, :]
    return s.astype(np.float32)


def transform_point(m, p):
    p = np.asarray(p, dtype=np.float32)
    q = m[:3, :3] @ p + m[:3, 3]
    w = m[3, :3] @ p + m[3, 3]
    return (q / w).astype(np.float32)


def transform_vector(m, v):
    return (m[:3, :3] @ np.asarray(v, dtype=np.float32)).astype(np.float32)


def normalized(v):
    v = np.asarray(v, dtype=np.float32)
    return (v / f32(np.sqrt(np.dot(v, v)))).astype(np.float32)


def rotation_z(a):
    c, s = f32(np.cos(a)), f32(np.sin(a))
    m = np.eye(4, dtype=np.float32)
    m[0, 0] = c; m[0, 1] = -s; m[1, 0] = s; m[1, 1] = c
    return m


def rotation_y(a):
    c, s = f32(np.cos(a)), f32(np.sin(a))
    m = np.eye(4, dtype=np.float32)
    m[0, 0] = c; m[0, 2] = s; m[2, 0] = -s; m[2, 2] = c
    return m


def translation(t):
    m = np.eye(4, dtype=np.float32)
    m[:3, 3] = np.asarray(t, dtype=np.float32)
    return m


def from_rt(r, t):
    m = np.eye(4, dtype=np.float32)
    m[:3, :3] = r
    m[:3, 3] = t
    return m


def quat_to_matrix(q):
    """q = [x y z w] (reference python/src/py_magnum.cpp:83-113)."""
    x, y, z, w = [float(v) for v in q]
    n = np.sqrt(x * x + y * y + z * z + w * w)
    x, y, z, w = x / n, y / n, z / n, w / n
    return np.array([
        [1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
        [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
        [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)],
    ], dtype=np.float32)


def matrix_to_quat(m):
    m = np.asarray(m, dtype=np.float64)
    t = np.trace(m)
    if t > 0:
        s = np.sqrt(t + 1.0) * 2
        w = 0.25 * s
        x = (m[2, 1] - m[1, 2]) / s
        y = (m[0, 2] - m[2, 0]) / s
        z = (m[1, 0] - m[0, 1]) / s
    elif m[0, 0] > m[1, 1] and m[0, 0] > m[2, 2]:
        s = np.sqrt(1.0 + m[0, 0] - m[1, 1] - m[2, 2]) * 2
        w = (m[2, 1] - m[1, 2]) / s
        x = 0.25 * s
        y = (m[0, 1] + m[1, 0]) / s
        z = (m[0, 2] + m[2, 0]) / s
    elif m[1, 1] > m[2, 2]:
        s = np.sqrt(1.0 + m[1, 1] - m[0, 0] - m[2, 2]) * 2
        w = (m[0, 2] - m[2, 0]) / s
        x = (m[0, 1] + m[1, 0]) / s
        y = 0.25 * s
        z = (m[1, 2] + m[2, 1]) / s
    else:
        s = np.sqrt(1.0 + m[2, 2] - m[0, 0] - m[1, 1]) * 2
        w = (m[1, 0] - m[0, 1]) / s
        x = (m[0, 2] + m[2, 0]) / s
        y = (m[1, 2] + m[2, 1]) / s
        z = 0.25 * s
    return np.array([x, y, z, w], dtype=np.float32)
