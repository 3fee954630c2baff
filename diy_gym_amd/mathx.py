"""Small fp64 rigid-transform helpers used on the host (numpy).

Conventions follow pybullet's, because the reference's configs and addons are
written against them (reference: diy_gym/model.py:53, respawn.py:19-27):
quaternions are ``[x, y, z, w]``; ``rpy`` are fixed-axis XYZ angles, i.e.
``R = Rz(yaw) @ Ry(pitch) @ Rx(roll)`` -- the URDF convention.
"""
import math

import numpy as np


def quat_from_euler(rpy):
    r, p, y = (float(v) for v in rpy)
    cr, sr = math.cos(r * 0.5), math.sin(r * 0.5)
    cp, sp = math.cos(p * 0.5), math.sin(p * 0.5)
    cy, sy = math.cos(y * 0.5), math.sin(y * 0.5)
    return np.array([
        sr * cp * cy - cr * sp * sy,
        cr * sp * cy + sr * cp * sy,
        cr * cp * sy - sr * sp * cy,
        cr * cp * cy + sr * sp * sy,
    ])


def euler_from_quat(q):
    """Inverse of :func:`quat_from_euler` (same branch choices as Bullet's
    ``getEulerZYX``-based ``getEulerFromQuaternion``: pitch in [-pi/2, pi/2])."""
    x, y, z, w = (float(v) for v in q)
    sarg = -2.0 * (x * z - w * y)
    if sarg <= -0.99999:
        return np.array([0.0, -0.5 * math.pi, 2.0 * math.atan2(x, -y)])
    if sarg >= 0.99999:
        return np.array([0.0, 0.5 * math.pi, 2.0 * math.atan2(-x, y)])
    sqx, sqy, sqz, sqw = x * x, y * y, z * z, w * w
    roll = math.atan2(2.0 * (y * z + w * x), sqw - sqx - sqy + sqz)
    pitch = math.asin(sarg)
    yaw = math.atan2(2.0 * (x * y + w * z), sqw + sqx - sqy - sqz)
    return np.array([roll, pitch, yaw])


def quat_mul(a, b):
    """Hamilton product a (x) b, xyzw."""
    ax, ay, az, aw = a
    bx, by, bz, bw = b
    return np.array([
        aw * bx + ax * bw + ay * bz - az * by,
        aw * by - ax * bz + ay * bw + az * bx,
        aw * bz + ax * by - ay * bx + az * bw,
        aw * bw - ax * bx - ay * by - az * bz,
    ])


def quat_conj(q):
    return np.array([-q[0], -q[1], -q[2], q[3]])


def quat_normalize(q):
    q = np.asarray(q, dtype=np.float64)
    return q / np.linalg.norm(q)


def mat_from_quat(q):
    x, y, z, w = (float(v) for v in q)
    n = x * x + y * y + z * z + w * w
    s = 2.0 / n if n > 0 else 0.0
    xs, ys, zs = x * s, y * s, z * s
    wx, wy, wz = w * xs, w * ys, w * zs
    xx, xy, xz = x * xs, x * ys, x * zs
    yy, yz, zz = y * ys, y * zs, z * zs
    return np.array([[1.0 - (yy + zz), xy - wz, xz + wy], [xy + wz, 1.0 - (xx + zz), yz - wx],
                     [xz - wy, yz + wx, 1.0 - (xx + yy)]])


def quat_from_mat(R):
    R = np.asarray(R, dtype=np.float64)
    tr = R[0, 0] + R[1, 1] + R[2, 2]
    if tr > 0:
        s = math.sqrt(tr + 1.0) * 2.0
        q = [(R[2, 1] - R[1, 2]) / s, (R[0, 2] - R[2, 0]) / s, (R[1, 0] - R[0, 1]) / s, 0.25 * s]
    elif R[0, 0] > R[1, 1] and R[0, 0] > R[2, 2]:
        s = math.sqrt(1.0 + R[0, 0] - R[1, 1] - R[2, 2]) * 2.0
        q = [0.25 * s, (R[0, 1] + R[1, 0]) / s, (R[0, 2] + R[2, 0]) / s, (R[2, 1] - R[1, 2]) / s]
    elif R[1, 1] > R[2, 2]:
        s = math.sqrt(1.0 + R[1, 1] - R[0, 0] - R[2, 2]) * 2.0
        q = [(R[0, 1] + R[1, 0]) / s, 0.25 * s, (R[1, 2] + R[2, 1]) / s, (R[0, 2] - R[2, 0]) / s]
    else:
        s = math.sqrt(1.0 + R[2, 2] - R[0, 0] - R[1, 1]) * 2.0
        q = [(R[0, 2] + R[2, 0]) / s, (R[1, 2] + R[2, 1]) / s, 0.25 * s, (R[1, 0] - R[0, 1]) / s]
    return quat_normalize(q)


def mat_from_euler(rpy):
    return mat_from_quat(quat_from_euler(rpy))


class Transform:
    """Rigid transform ``x_parent = R @ x_child + p``."""
    __slots__ = ('R', 'p')

    def __init__(self, R=None, p=None):
        self.R = np.eye(3) if R is None else np.asarray(R, dtype=np.float64)
        self.p = np.zeros(3) if p is None else np.asarray(p, dtype=np.float64)

    @classmethod
    def from_xyz_rpy(cls, xyz, rpy):
        return cls(mat_from_euler(rpy), np.asarray(xyz, dtype=np.float64))

    @classmethod
    def from_xyz_quat(cls, xyz, quat):
        return cls(mat_from_quat(quat), np.asarray(xyz, dtype=np.float64))

    def __mul__(self, other):
        return Transform(self.R @ other.R, self.R @ other.p + self.p)

    def inverse(self):
        return Transform(self.R.T, -self.R.T @ self.p)

    def apply(self, x):
        return self.R @ np.asarray(x, dtype=np.float64) + self.p

    @property
    def quat(self):
        return quat_from_mat(self.R)

    def matrix(self):
        T = np.eye(4)
        T[:3, :3] = self.R
        T[:3, 3] = self.p
        return T
