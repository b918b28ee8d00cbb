"""Camera-sweep pose generators of the reference's datasets, host side (numpy float64 -> float32):
the target extrinsics a novel-view sweep renders (world->camera 4x4, OpenCV convention).

* ``dtu_cam_sweep_extrinsics``        ``DTUDataSet.get_cam_sweep_extrinsics``        src/data/dtu.py:246-340
* ``facescape_cam_sweep_extrinsics``  ``FacescapeDataSet.get_cam_sweep_extrinsics``  src/data/facescape.py:365-423

Pinned by ``tests/golden/wire.npz`` (outputs of the reference's own methods, ``oracle/gen_golden.py --wire-only``); the
reference evaluates in float32 (and scipy's double Slerp), so agreement is to ~1e-5, stated in tests/test_poses.py.
"""
from __future__ import annotations

import numpy as np


def _closest_points(o1, d1, o2, d2):
    """Points on ray 1 and ray 2 where the rays are closest (least squares, src/util/cam_geometry.py:129-146)."""
    A = np.stack([d1, -d2], axis=-1)                 # [3,2]
    t, *_ = np.linalg.lstsq(A, (o2 - o1)[:, None], rcond=None)
    return o1 + d1 * t[0, 0], o2 + d2 * t[1, 0]


def _quat_from_matrix(R):
    """unit quaternion (x, y, z, w) of a rotation matrix"""
    m = R
    t = np.trace(m)
    if t > 0:
        s = np.sqrt(t + 1.0) * 2
        q = np.array([(m[2, 1] - m[1, 2]) / s, (m[0, 2] - m[2, 0]) / s, (m[1, 0] - m[0, 1]) / s, 0.25 * s])
    else:
        i = int(np.argmax(np.diag(m)))
        j, k = (i + 1) % 3, (i + 2) % 3
        s = np.sqrt(1.0 + m[i, i] - m[j, j] - m[k, k]) * 2
        q = np.zeros(4)
        q[i] = 0.25 * s
        q[j] = (m[j, i] + m[i, j]) / s
        q[k] = (m[k, i] + m[i, k]) / s
        q[3] = (m[k, j] - m[j, k]) / s
    return q / np.linalg.norm(q)


def _matrix_from_quat(q):
    x, y, z, w = q
    return np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
                     [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                     [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]])


def _slerp(q0, q1, u):
    d = float(np.dot(q0, q1))
    if d < 0:                      # shortest arc (what scipy's Slerp does through the relative rotation vector)
        q1, d = -q1, -d
    th = np.arccos(np.clip(d, -1.0, 1.0))
    if th < 1e-12:
        return q0
    return (np.sin((1 - u) * th) * q0 + np.sin(u * th) * q1) / np.sin(th)


def dtu_cam_sweep_extrinsics(cam_extrinsics, nframes: int) -> np.ndarray:
    """Sweep left (camera 11) -> centre (24) -> right (18) of the DTU rig (dtu.py:255-257): camera centres on great-circle arcs
    around the point where the three optical axes (nearly) meet, rotations by quaternion slerp with knots at t = 0, 0.5, 1.
    :param cam_extrinsics: [>=25,4,4] world->camera matrices of the rig
    :return: [nframes,4,4] float32 world->camera"""
    E = np.asarray(cam_extrinsics, dtype=np.float64)
    poses = [np.linalg.inv(E[i]) for i in (11, 24, 18)]                      # left, centre, right (camera->world)
    rays = [(p[:3, 3], p[:3, 2]) for p in poses]                             # origin, optical axis
    pts = []
    for a, b in ((0, 1), (1, 2), (0, 2)):                                     # :269-272
        pts += list(_closest_points(*rays[a], *rays[b]))
    origin = np.mean(pts, axis=0)
    radius = np.mean([np.linalg.norm(origin - p[:3, 3]) for p in poses])     # :275-277
    t = np.linspace(0.0, 1.0, nframes, dtype=np.float32).astype(np.float64)  # torch.linspace in float32
    x = [(p[:3, 3] - origin) / np.linalg.norm(p[:3, 3] - origin) for p in poses]
    th1 = np.arccos(np.clip(np.dot(x[0], x[1]), -1, 1))
    th2 = np.arccos(np.clip(np.dot(x[1], x[2]), -1, 1))
    q = [_quat_from_matrix(p[:3, :3]) for p in poses]
    out = np.zeros((nframes, 4, 4))
    for i, ti in enumerate(t):
        if ti < 0.5:                                                          # :294-300
            u, th, a, b, qa, qb = ti * 2, th1, x[0], x[1], q[0], q[1]
        else:
            u, th, a, b, qa, qb = ti * 2 - 1, th2, x[1], x[2], q[1], q[2]
        c = np.sin((1 - u) * th) / np.sin(th) * a + np.sin(u * th) / np.sin(th) * b
        pose = np.eye(4)
        pose[:3, :3] = _matrix_from_quat(_slerp(qa, qb, u))                   # :305-309
        pose[:3, 3] = c * radius + origin
        out[i] = np.linalg.inv(pose)
    return out.astype(np.float32)


def facescape_cam_sweep_extrinsics(src_extrinsics, nframes: int, radius: float = 1.8, sweep_range: float = 45.0) -> np.ndarray:
    """Horizontal sweep around the world z axis through the mean direction of the source cameras (facescape.py:365-399):
    base camera at ``radius`` along that direction looking at the origin with the image y axis along world -z, rotated by
    ``nframes`` angles in [-sweep_range, +sweep_range] degrees.
    :param src_extrinsics: [N,4,4] world->camera of the source views
    :return: [nframes,4,4] float32 world->camera"""
    E = np.asarray(src_extrinsics, dtype=np.float64)
    centers = -np.einsum("nji,nj->ni", E[:, :3, :3], E[:, :3, 3])            # -R^T t
    dirs = centers / np.linalg.norm(centers, axis=-1, keepdims=True)
    mean_dir = dirs.sum(0)
    mean_dir /= np.linalg.norm(mean_dir)
    center = mean_dir * radius
    z_ax = -center / np.linalg.norm(center)
    y_ax = np.array([0.0, 0.0, -1.0])
    x_ax = np.cross(y_ax, z_ax)
    x_ax /= np.linalg.norm(x_ax)
    base = np.eye(4)
    base[:3, 0], base[:3, 1], base[:3, 2], base[:3, 3] = x_ax, y_ax, z_ax, center
    out = np.zeros((nframes, 4, 4))
    for i, al in enumerate(np.linspace(-sweep_range / 180 * np.pi, sweep_range / 180 * np.pi, nframes)):
        rot = np.array([[np.cos(al), -np.sin(al), 0, 0], [np.sin(al), np.cos(al), 0, 0], [0, 0, 1, 0], [0, 0, 0, 1.0]])
        out[i] = np.linalg.inv(rot @ base)
    return out.astype(np.float32)
