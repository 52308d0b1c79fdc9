"""Reference-trajectory projection used when building the 3-D X_GOAL table (host side, NumPy).

Behaviour follows the reference's ``math_and_models/transformations.py``: orthogonal
``projection_matrix`` (:98-105) and ``transform_trajectory`` (:107-123), including its quirk that
the homogeneous coordinate 1 is appended to the *velocities* as well, so the plane offset
``(point . n) n`` is added to them too.  ``rot_xyz`` is the numeric twin of ``csRotXYZ``
(:125-182): R = Rz(psi) Ry(theta) Rx(phi).
"""
import numpy as np


def projection_matrix(point, normal):
    """4x4 homogeneous orthogonal projection onto the plane through ``point`` with ``normal``."""
    n = np.asarray(normal, dtype=np.float64)[:3]
    n = n / np.sqrt(n @ n)
    p = np.asarray(point, dtype=np.float64)[:3]
    M = np.eye(4)
    M[:3, :3] = np.eye(3) - np.outer(n, n)
    M[:3, 3] = (p @ n) * n
    return M


def transform_trajectory(pos, vel, trans_info):
    """(T,3) planar position / velocity references -> their projections, both affine."""
    M = projection_matrix(trans_info['point'], trans_info['normal'])
    A, b = M[:3, :3], M[:3, 3]
    return pos @ A.T + b, vel @ A.T + b


def rot_xyz(phi, theta, psi):
    """Body-to-world rotation for roll phi, pitch theta, yaw psi (SDFormat / PyBullet convention).
    Accepts scalars or arrays; returns (..., 3, 3)."""
    phi, theta, psi = np.broadcast_arrays(np.asarray(phi, float), np.asarray(theta, float), np.asarray(psi, float))
    cx, sx, cy, sy, cz, sz = np.cos(phi), np.sin(phi), np.cos(theta), np.sin(theta), np.cos(psi), np.sin(psi)
    R = np.empty(phi.shape + (3, 3))
    R[..., 0, 0] = cz * cy
    R[..., 0, 1] = cz * sy * sx - sz * cx
    R[..., 0, 2] = cz * sy * cx + sz * sx
    R[..., 1, 0] = sz * cy
    R[..., 1, 1] = sz * sy * sx + cz * cx
    R[..., 1, 2] = sz * sy * cx - cz * sx
    R[..., 2, 0] = -sy
    R[..., 2, 1] = cy * sx
    R[..., 2, 2] = cy * cx
    return R
