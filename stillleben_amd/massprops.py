"""Mass, centre of mass and inertia of an object's collision hulls -- the job of
PxRigidBodyExt::updateMassAndInertia(body, density) in the reference (src/object.cpp:215-221):
per-shape polyhedral integrals, summed over shapes (overlapping hulls are double counted,
as in PhysX), expressed in the OBJECT frame (pretransform incl. scale applied)."""
import numpy as np


class MassProps:
    pass


def hull_integrals(v, t):
    """Volume, first and second moments of a closed triangle mesh via signed tetrahedra."""
    v = v.astype(np.float64)
    a, b, c = v[t[:, 0]], v[t[:, 1]], v[t[:, 2]]
    det = np.einsum("ij,ij->i", a, np.cross(b, c))
    vol = det.sum() / 6.0
    centroid = ((a + b + c) * det[:, None]).sum(axis=0) / 24.0  # = integral of x dV
    # covariance integral C = int x x^T dV for each tet (0,a,b,c): det/120 * (sum_i sum_j (1+d_ij) p_i p_j^T)
    s = a + b + c
    C = (np.einsum("i,ij,ik->jk", det, s, s) + np.einsum("i,ij,ik->jk", det, a, a)
         + np.einsum("i,ij,ik->jk", det, b, b) + np.einsum("i,ij,ik->jk", det, c, c)) / 120.0
    if vol < 0:
        vol, centroid, C = -vol, -centroid, -C
    return vol, centroid, C


def object_frame_hulls(mesh):
    """Hull vertices with the mesh pretransform applied (object frame)."""
    M = mesh._pretransform.astype(np.float64)
    out = []
    for h in mesh._load_physics():
        v = h.vertices.astype(np.float64) @ M[:3, :3].T + M[:3, 3]
        out.append((v.astype(np.float32), h.triangles))
    return out


def compute(obj):
    mesh = obj._mesh
    vol = 0.0
    first = np.zeros(3)
    C = np.zeros((3, 3))
    for v, t in object_frame_hulls(mesh):
        vv, ff, cc = hull_integrals(v, t)
        vol += vv
        first += ff
        C += cc
    p = MassProps()
    p.key = obj._mass_key()
    density = float(obj._density)
    if vol <= 0:
        raise RuntimeError("collision hulls have no volume")
    com = first / vol
    # inertia about the origin: I = tr(C) 1 - C ; shift to the COM
    I0 = (np.trace(C) * np.eye(3) - C) * density
    m = density * vol
    I = I0 - m * (np.dot(com, com) * np.eye(3) - np.outer(com, com))
    p.mass = np.float32(m)
    p.volume = np.float32(vol)
    p.com = com.astype(np.float32)
    p.inertia = I.astype(np.float32)
    p.inv_inertia = np.linalg.inv(I).astype(np.float32)
    w, V = np.linalg.eigh(I)
    if np.linalg.det(V) < 0:
        V[:, 2] = -V[:, 2]
    p.inertia_diag = w.astype(np.float32)  # massSpaceInertiaTensor (object.cpp:245-250)
    frame = np.eye(4, dtype=np.float32)
    frame[:3, :3] = V.astype(np.float32)
    frame[:3, 3] = p.com
    p.inertial_frame = frame                # cMassLocalPose (object.cpp:252-257)
    return p
