"""Independent numpy kinematics/dynamics helpers for the known-answer tests.
They work from the parsed URDF tree directly (not from FlatBody, not from the
oracle): forward kinematics by chaining joint transforms, geometric Jacobians,
and the joint-space mass matrix / gravity vector as sums over the URDF links."""
import numpy as np

from diy_gym_amd.mathx import Transform


def rot(axis, q):
    a = np.asarray(axis, dtype=float)
    K = np.array([[0, -a[2], a[1]], [a[2], 0, -a[0]], [-a[1], a[0], 0]])
    return np.eye(3) + np.sin(q) * K + (1 - np.cos(q)) * K @ K


def link_frames(robot, q, T_base=None):
    """World transform of every URDF link frame, plus (origin, axis, type) of every movable joint."""
    T = {robot.root: T_base if T_base is not None else Transform()}
    joints = []
    for j in robot.joints:
        Tj = T[j.parent] * j.origin
        if j.movable:
            if j.type == 'prismatic':
                Tj = Tj * Transform(np.eye(3), j.axis * q[j.q_index])
            else:
                Tj = Tj * Transform(rot(j.axis, q[j.q_index]), np.zeros(3))
            joints.append((j, Tj.p.copy(), Tj.R @ j.axis))
        T[j.child] = Tj
    return T, joints


def ancestors(robot, link_name):
    out = set()
    while robot.links[link_name].parent_joint is not None:
        j = robot.links[link_name].parent_joint
        if j.movable:
            out.add(j.q_index)
        link_name = j.parent
    return out


def mass_matrix_and_gravity(robot, q, g=(0, 0, -9.81), T_base=None):
    n = robot.num_dofs
    T, joints = link_frames(robot, q, T_base)
    M = np.zeros((n, n))
    G = np.zeros(n)
    for name, link in robot.links.items():
        if link.mass <= 0 or name == robot.root:
            continue
        Tc = T[name] * link.inertial_origin
        Iw = Tc.R @ link.inertia @ Tc.R.T
        anc = ancestors(robot, name)
        Jv = np.zeros((3, n))
        Jw = np.zeros((3, n))
        for j, o, a in joints:
            if j.q_index in anc:
                if j.type == 'prismatic':
                    Jv[:, j.q_index] = a
                else:
                    Jv[:, j.q_index] = np.cross(a, Tc.p - o)
                    Jw[:, j.q_index] = a
        M += link.mass * Jv.T @ Jv + Jw.T @ Iw @ Jw
        G += Jv.T @ (link.mass * np.asarray(g, dtype=float))
    return M, G
