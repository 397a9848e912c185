"""Scene::chooseRandomCameraPose (reference src/scene.cpp:472-610): pick a view direction
(azimuth, elevation), then push the four side planes of the frustum until they touch the
bounding-box corners of every object and put the camera on the rear-most intersection line."""
import numpy as np

from . import _math as M
from ._math import f32


def camera_rotation(azimuth, elevation):
    # rotation into the image coordinate system: columns (-y, -z, x)  (scene.cpp:489-493)
    cam_rot = np.array([[0, 0, 1], [-1, 0, 0], [0, -1, 0]], dtype=np.float32)
    return (M.rotation_z(azimuth) @ M.rotation_y(elevation) @ M.from_rt(cam_rot, np.zeros(3, np.float32))).astype(np.float32)


def choose_camera_pose(scene, azimuth, elevation):
    camera_rot = camera_rotation(azimuth, elevation)
    if not scene._objects:
        return (M.translation([0.0, 0.0, -1.0]) @ camera_rot).astype(np.float32)
    to_work = M.inverted_rigid(camera_rot)
    pts = []
    for obj in scene._objects:
        trans = (to_work @ obj._pose).astype(np.float32)
        c = obj._mesh.bbox.corners()
        pts.append(c @ trans[:3, :3].T + trans[:3, 3])
    pts = np.concatenate(pts).astype(np.float32)
    P = scene._projection
    frustum = np.stack([P[3] + P[0], P[3] - P[0], P[3] + P[1], P[3] - P[1]]).astype(np.float32)
    for k in range(4):
        frustum[k] = frustum[k] / f32(np.sqrt(np.dot(frustum[k, :3], frustum[k, :3])))
        frustum[k, 3] = -f32((pts @ frustum[k, :3]).min())

    def intersect(a, b, ia):
        la = np.array([a[ia], a[2], a[3]], np.float32)
        lb = np.array([b[ia], b[2], b[3]], np.float32)
        x = np.cross(la, lb).astype(np.float32)
        if abs(x[2]) < 1e-3:
            x = np.array([0.0, 0.0, 1.0], np.float32)
        return x[0] / x[2], x[1] / x[2]

    lr_x, lr_z = intersect(frustum[0], frustum[1], 0)
    tb_y, tb_z = intersect(frustum[2], frustum[3], 1)
    cam_position = np.array([lr_x, tb_y, min(lr_z, tb_z)], dtype=np.float32)
    return (camera_rot @ M.translation(cam_position)).astype(np.float32)
