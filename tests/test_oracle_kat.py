"""Known-answer tests that pin the CPU oracle (oracle/dgsim_oracle.c).

The reference holds no golden vectors and pybullet cannot be run
here, so the oracle is pinned by closed-form / independently computed answers:
free fall, resting contact, pendulum period and energy, double-pendulum
accelerations (Lagrangian closed form incl. Coriolis terms), mass matrix and
gravity vector of the UR5 computed independently in numpy, forward kinematics
of the UR5 at the reference's rest pose, IK fixed point and progress, and the
restated pybullet behaviours (default velocity motors, force clamp).
"""
import os

import numpy as np
import pytest
import torch

from diy_gym_amd import DIYGym
from diy_gym_amd.mathx import Transform, mat_from_quat
from diy_gym_amd.scene import K
from diy_gym_amd.urdf import UrdfRobot
from nphelpers import link_frames, mass_matrix_and_gravity
from oracle_backend import OracleBackend

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
G = os.path.join(ROOT, 'tests', 'golden')
NODAMP = dict(linear_damping=0.0, angular_damping=0.0)
H = 1.0 / 480.0


def make(cfg, B=1, **engine):
    return DIYGym(os.path.join(G, cfg) if not os.path.isabs(cfg) else cfg, num_envs=B, backend_factory=OracleBackend, engine=engine)


def link_q(env, body, dof):
    return env.layout.link_state_off[env.layout.body_first_link[body] + dof]


def test_free_fall_matches_semi_implicit_euler_closed_form(tmp_path):
    cfg = tmp_path / 'fall.yaml'
    cfg.write_text('hot_start: 0\nball: {model: sphere2.urdf, xyz: [0, 0, 10.0]}\n')
    env = make(str(cfg), **NODAMP)
    n = 50
    for _ in range(n):
        env.sim.step(0)
    st = env.sim.get_state()[0]
    so = env.layout.body_state_off[0]
    k = 2 * n  # substeps
    assert abs(st[so + 2] - (10.0 - 9.81 * H * H * k * (k + 1) / 2)) < 1e-10
    assert abs(st[so + K.BS_LINVEL + 2] - (-9.81 * H * k)) < 1e-10
    assert st[K.ST_STEP] == n


def test_marble_rests_on_the_plane():
    env = make('basic_env_nocam.yaml')
    for _ in range(240):
        env.sim.step(0)
    st = env.sim.get_state()[0]
    for b in (1, 2, 3):
        so = env.layout.body_state_off[b]
        assert abs(st[so + 2] - 0.5) < 2e-4
        assert np.abs(st[so + K.BS_LINVEL:so + K.BS_LINVEL + 6]).max() < 1e-3
    # 3 marble-plane contacts + the red and green marbles, which spawn exactly touching (centres 1.0 apart, r = 0.5)
    assert env.sim.contacts(0) == 4


def test_pendulum_period_and_energy():
    env = make('pendulum.yaml', **NODAMP)
    env.sim.set_motor_cfg(np.array([[0.0, 1.0, 0.0]]))  # free joint: motor force 0 (pybullet idiom)
    qo = link_q(env, 0, 0)
    st = env.sim.get_state(); st[0, qo] = 0.05; env.sim.set_state(st)
    m, L, g = 1.0, 0.5, 9.81
    qs, es = [], []
    for _ in range(720):
        env.sim.step(0)
        s = env.sim.get_state()[0]
        qs.append(s[qo]); es.append(0.5 * (m * L * L + 1e-6) * s[qo + 1]**2 + m * g * L * (1 - np.cos(s[qo])))
    qs = np.array(qs)
    ups = [i for i in range(1, len(qs)) if qs[i - 1] < 0 <= qs[i]]
    t = [(i - 1 + (-qs[i - 1]) / (qs[i] - qs[i - 1])) / 240.0 for i in ups]
    period = np.mean(np.diff(t))
    assert abs(period - 2 * np.pi * np.sqrt(L / g)) / period < 3e-3
    assert (max(es) - min(es)) / max(es) < 2e-2  # symplectic Euler: bounded energy oscillation, no drift


def test_default_velocity_motor_holds_a_joint_and_force_clamp():
    # [R] every joint gets a velocity motor (target 0) at load: a pendulum released at 1 rad does not swing
    env = make('pendulum.yaml', **NODAMP)
    qo = link_q(env, 0, 0)
    st = env.sim.get_state(); st[0, qo] = 1.0; env.sim.set_state(st)
    for _ in range(100):
        env.sim.step(0)
    s = env.sim.get_state()[0]
    assert abs(s[qo] - 1.0) < 1e-3 and abs(s[qo + 1]) < 1e-3
    # reported effort = gravity torque it resists: m g L sin(q)
    assert abs(abs(s[qo + K.LS_APPLIED]) - 1.0 * 9.81 * 0.5 * np.sin(1.0)) < 2e-2
    # a motor with max force below the gravity torque saturates at exactly that force
    env.sim.set_motor_cfg(np.array([[0.0, 1.0, 2.0]]))
    env.sim.step(0)
    assert abs(abs(env.sim.get_state()[0, qo + K.LS_APPLIED]) - 2.0) < 1e-9


def test_double_pendulum_accelerations_match_lagrangian_closed_form():
    env = make('double_pendulum.yaml', **NODAMP)
    m1, m2, l1, l2, g = 1.5, 0.7, 0.4, 0.3, 9.81
    rng = np.random.default_rng(3)
    for _ in range(10):
        q1, q2, w1, w2 = rng.uniform(-1.5, 1.5, 4)
        st = env.sim.get_state()
        st[0, link_q(env, 0, 0):link_q(env, 0, 0) + 2] = (q1, w1)
        st[0, link_q(env, 0, 1):link_q(env, 0, 1) + 2] = (q2, w2)
        env.sim.set_state(st)
        qdd, _ = env.sim.forward_dynamics(0, 0, 2)
        t1, t2, o1, o2 = q1, q1 + q2, w1, w1 + w2
        den = 2 * m1 + m2 - m2 * np.cos(2 * t1 - 2 * t2)
        a1 = (-g * (2 * m1 + m2) * np.sin(t1) - m2 * g * np.sin(t1 - 2 * t2) - 2 * np.sin(t1 - t2) * m2 * (o2**2 * l2 + o1**2 * l1 * np.cos(t1 - t2))) / (l1 * den)
        a2 = (2 * np.sin(t1 - t2) * (o1**2 * l1 * (m1 + m2) + g * (m1 + m2) * np.cos(t1) + o2**2 * l2 * m2 * np.cos(t1 - t2))) / (l2 * den)
        assert np.allclose(qdd, [a1, a2 - a1], rtol=1e-6, atol=1e-6)


def ur5():
    return UrdfRobot(os.path.join(ROOT, 'diy_gym_amd', 'data', 'ur5', 'ur5_robot.urdf'))


def test_ur5_mass_matrix_inverse_and_gravity_against_numpy():
    env = make(os.path.join(ROOT, 'examples', 'ur_high_5', 'ur_high_5_joint.yaml'), **NODAMP)
    robot = ur5()
    rng = np.random.default_rng(5)
    Tb = Transform.from_xyz_rpy([-0.55, 0.4, 0.0], [0, 0, -1.57])
    for _ in range(4):
        q = rng.uniform(-2, 2, 6)
        st = env.sim.get_state()
        for i in range(6):
            st[0, link_q(env, 0, i)] = q[i]; st[0, link_q(env, 0, i) + 1] = 0.0
        env.sim.set_state(st)
        M, Gv = mass_matrix_and_gravity(robot, q, T_base=Tb)
        Minv = np.stack([env.sim.unit_response(0, 0, j, 6)[6:] for j in range(6)], axis=1)
        assert np.allclose(Minv, Minv.T, atol=1e-10)
        assert np.allclose(M @ Minv, np.eye(6), atol=1e-8)
        qdd, _ = env.sim.forward_dynamics(0, 0, 6)
        assert np.allclose(M @ qdd, Gv, atol=1e-8)


def test_ur5_forward_kinematics_at_reference_rest_pose():
    env = make(os.path.join(ROOT, 'examples', 'ur_high_5', 'ur_high_5.yaml'))
    robot = ur5()
    rest = np.array([-0.17, -0.73, -1.93, -0.36, -0.03, -0.06])  # ur_high_5.yaml:18
    st = env.sim.get_state()
    for b in (0, 1):
        for i in range(6):
            st[0, link_q(env, b, i)] = rest[i]
    env.sim.set_state(st)
    for b, (xyz, yaw) in enumerate([([-0.55, 0.4, 0.0], -1.57), ([0.55, 0.4, 0.0], 1.57)]):
        T, _ = link_frames(robot, rest, Transform.from_xyz_rpy(xyz, [0, 0, yaw]))
        got = env.sim.frame_state64(b, 7, com=False)[0]
        assert np.allclose(got[:3], T['ee_link'].p, atol=1e-12)
        q = T['ee_link'].quat
        assert min(np.abs(got[3:7] - q).max(), np.abs(got[3:7] + q).max()) < 1e-9
        # wrist_3 link frame (joint index 6)
        assert np.allclose(env.sim.frame_state64(b, 6)[0][:3], T['wrist_3_link'].p, atol=1e-12)


def _ik_env(**engine):
    env = make(os.path.join(ROOT, 'examples', 'ur_high_5', 'ur_high_5.yaml'), **engine)
    op = [i for i, o in enumerate(env.builder.ops) if o[0][K.OI_CODE] == K.OP_IK_CONTROL][0]
    return env, op


def test_ik_single_iteration_equals_numpy_dls_with_nullspace():
    env, op = _ik_env(ik_iterations=1)
    robot = ur5()
    Tb = Transform.from_xyz_rpy([-0.55, 0.4, 0.0], [0, 0, -1.57])
    rest = np.array([-0.17, -0.73, -1.93, -0.36, -0.03, -0.06])
    rng = np.random.default_rng(11)
    for trial in range(4):
        q0 = rest + (rng.uniform(-0.4, 0.4, 6) if trial else 0.0)
        st = env.sim.get_state()
        for i in range(6):
            st[0, link_q(env, 0, i)] = q0[i]
        env.sim.set_state(st)
        lin, rot = rng.uniform(-0.01, 0.01, 3), rng.uniform(-0.01, 0.01, 3)
        q = env.sim.ik(0, op, np.concatenate([lin, rot]), 6)
        T, joints = link_frames(robot, q0, Tb)
        pe = T['ee_link'].p
        J = np.zeros((6, 6))
        for j, o, a in joints:
            J[:3, j.q_index] = np.cross(a, pe - o)
            J[3:, j.q_index] = a
        # target orientation = q_ee (x) quat(rpy): for small angles the world-frame error vector is R_ee @ (rotation vector)
        from diy_gym_amd.mathx import mat_from_euler
        Rerr = T['ee_link'].R @ mat_from_euler(rot) @ T['ee_link'].R.T
        ang = np.arccos(np.clip((np.trace(Rerr) - 1) / 2, -1, 1))
        axis = np.array([Rerr[2, 1] - Rerr[1, 2], Rerr[0, 2] - Rerr[2, 0], Rerr[1, 0] - Rerr[0, 1]]) / (2 * np.sin(ang))
        e = np.concatenate([lin, axis * ang])
        U = J @ J.T + 0.36 * np.eye(6)
        v0 = 0.001 * (rest - q0)
        dq = J.T @ np.linalg.solve(U, e) + v0 - J.T @ np.linalg.solve(U, J @ v0)
        assert np.allclose(q - q0, dq, atol=1e-9), trial


def test_ik_fixed_point_and_progress(tmp_path):
    env, op = _ik_env()
    q0 = np.array([env.sim.get_state()[0, link_q(env, 0, i)] for i in range(6)])
    # zero action at the rest pose (where the null-space bias vanishes) is a fixed point
    q = env.sim.ik(0, op, np.zeros(6), 6)
    assert np.allclose(q, q0, atol=1e-5)
    # With use_orientation the rest pose of ur_high_5.yaml is next to a wrist singularity (smallest
    # singular value of the 6-D Jacobian 0.003), so a pure translation barely moves -- as DLS intends.
    d = np.array([0.01, 0.0, 0.0])
    p0 = env.sim.frame_state64(0, 7, com=True)[0][:3]
    q = env.sim.ik(0, op, np.concatenate([d, np.zeros(3)]), 6)
    assert np.abs(q - q0).max() < 0.02
    # position-only IK (use_orientation: no), 5 iterations: identical to the same recursion written in numpy,
    # and the error shrinks monotonically
    import yaml
    tree = yaml.safe_load(open(os.path.join(ROOT, 'examples', 'ur_high_5', 'ur_high_5.yaml')))
    for arm in ('ur5_l', 'ur5_r'):
        tree[arm]['controller']['use_orientation'] = False
        tree[arm]['model'] = os.path.join(ROOT, 'diy_gym_amd', 'data', 'ur5', 'ur5_robot.urdf')
    cfg = tmp_path / 'ur_pos_only.yaml'
    cfg.write_text(yaml.dump(tree))
    env2 = make(str(cfg), ik_iterations=5)
    op2 = [i for i, o in enumerate(env2.builder.ops) if o[0][K.OI_CODE] == K.OP_IK_CONTROL][0]
    q = env2.sim.ik(0, op2, d, 6)
    robot, Tb = ur5(), Transform.from_xyz_rpy([-0.55, 0.4, 0.0], [0, 0, -1.57])
    rest = np.array([-0.17, -0.73, -1.93, -0.36, -0.03, -0.06])
    qn, errs = q0.copy(), []
    T, _ = link_frames(robot, qn, Tb)
    tp = T['ee_link'].p + d
    for _ in range(5):
        T, joints = link_frames(robot, qn, Tb)
        pe = T['ee_link'].p
        J = np.zeros((3, 6))
        for j, o, a in joints:
            J[:, j.q_index] = np.cross(a, pe - o)
        U = J @ J.T + 0.36 * np.eye(3)
        v0 = 0.001 * (rest - qn)
        qn = qn + J.T @ np.linalg.solve(U, tp - pe) + v0 - J.T @ np.linalg.solve(U, J @ v0)
        errs.append(np.linalg.norm(tp - link_frames(robot, qn, Tb)[0]['ee_link'].p))
    assert np.allclose(q, qn, atol=1e-9)
    assert all(b < a for a, b in zip([np.linalg.norm(d)] + errs, errs))


def test_position_motors_hold_the_ur5_against_gravity():
    env = make(os.path.join(ROOT, 'examples', 'ur_high_5', 'ur_high_5_joint.yaml'), B=2)
    rest = torch.tensor([-0.17, -0.73, -1.93, -0.36, -0.03, -0.06] * 2).repeat(2, 1)
    for _ in range(240):
        env.sim.step(env._all_slots, rest)
    q = env.sim.obs[:, 0:6]
    assert float((q - rest[:, :6]).abs().max()) < 5e-3


def test_drone_needs_more_than_hover_thrust_to_lift_8kg():
    import diy_gym_amd.examples  # noqa: F401
    env = make(os.path.join(ROOT, 'examples', 'drone_pilot', 'drone_pilot.yaml'))
    z = {}
    for level in (0.9, 1.0):
        env.sim.reset(None)
        for _ in range(300):
            env.sim.step(env._all_slots, torch.full((1, 4), level))
        z[level] = env.sim.get_state()[0, env.layout.body_state_off[2] + 2]
    # 4 rotors x 20 N x 0.9 = 72 N < 8 kg x 9.81 = 78.5 N < 80 N
    assert z[0.9] < 0.3 and z[1.0] > z[0.9] + 0.05


def test_admittance_gravity_compensation_holds_and_wrench_maps_through_the_jacobian(tmp_path):
    """admittance_controller.py:36-55: zero wrench + exact gravity compensation => the arm does not move;
    a force F produces joint torques J^T F (checked against a finite-difference Jacobian of the
    end-effector point and the mass-matrix inverse from the impulse response)."""
    import yaml
    cfg = yaml.safe_load(open(os.path.join(ROOT, 'tests', 'golden', 'ur_admittance.yaml')))
    cfg['hot_start'] = 0
    cfg['arm']['wrench'].update(p_gain=0.0, d_gain=0.0, offset_admittance_point=[0.02, -0.01, 0.03])
    path = tmp_path / 'adm.yaml'
    yaml.safe_dump(cfg, open(path, 'w'))
    env = make(str(path), **NODAMP)
    rest = np.array(cfg['arm']['wrench']['rest_position'])
    zero = {'arm': {'wrench': {'force': torch.zeros(1, 3), 'torque': torch.zeros(1, 3)}}}
    for _ in range(240):
        o, _, _, _ = env.step(zero)
    assert np.allclose(o['arm']['joints']['position'][0].numpy(), rest, atol=1e-9)
    assert np.allclose(o['arm']['joints']['velocity'][0].numpy(), 0.0, atol=1e-9)
    assert np.all(env.sim.motor_cfg()[:6, 2] == 0.0)      # velocity motors off (:34)

    # finite-difference Jacobian of the admittance point (COM frame of the end-effector link + offset)
    ee = env.models['arm'].get_frame_id('ee_fixed_joint')
    off = np.array([0.02, -0.01, 0.03])

    def point_and_rot(q):
        st = env.sim.get_state()
        for i in range(6):
            st[0, link_q(env, 0, i)] = q[i]; st[0, link_q(env, 0, i) + 1] = 0.0
        env.sim.set_state(st)
        f = env.sim.frame_state64(0, ee, com=True)[0]
        R = mat_from_quat(f[3:7])
        return f[:3] + R @ off, R

    eps = 1e-6
    p0, R0 = point_and_rot(rest)
    Jl, Ja = np.zeros((3, 6)), np.zeros((3, 6))
    for j in range(6):
        dq = rest.copy(); dq[j] += eps
        p1, R1 = point_and_rot(dq)
        Jl[:, j] = (p1 - p0) / eps
        W = (R1 @ R0.T - np.eye(3)) / eps
        Ja[:, j] = [W[2, 1], W[0, 2], W[1, 0]]
    point_and_rot(rest)
    Minv = np.stack([env.sim.unit_response(0, 0, j, 6)[6:] for j in range(6)], axis=1)
    F, T = np.array([1.5, -2.0, 3.0]), np.array([0.3, 0.2, -0.5])
    o, _, _, _ = env.step({'arm': {'wrench': {'force': torch.tensor(F[None], dtype=torch.float32),
                                             'torque': torch.tensor(T[None], dtype=torch.float32)}}})
    want = Minv @ (Jl.T @ F + Ja.T @ T) * (1.0 / 240.0)
    got = o['arm']['joints']['velocity'][0].numpy()
    # one step = two substeps with the pose moving in between: first-order agreement
    assert np.allclose(got, want, rtol=2e-2, atol=2e-3 * np.abs(want).max()), (got, want)


def test_marble_pushed_sideways_rolls_without_slipping(tmp_path):
    """A constant horizontal force through the centre of a sphere resting on a frictional plane: the contact's
    friction rows must supply exactly the torque that keeps v = omega x r, so a = F / (m + I / r^2) (here 10 kg,
    I = 1, r = 0.5: F / 14), far from the frictionless F / m.  Checks the normal + two friction rows, the friction
    cone (mu m g = 98 N >> the 2.9 N needed) and the coupling of linear and angular rows in one Gauss-Seidel solve."""
    import yaml
    cfg = {'render': False, 'plane': {'model': 'grass/plane.urdf'},
           'ball': {'model': 'sphere2.urdf', 'xyz': [0.0, 0.0, 0.5],
                    'push': {'addon': 'external_force', 'xyz': [0.0, 0.0, 0.5]},   # the force acts at this WORLD point
                    'state': {'addon': 'object_state_sensor', 'include_rotation': True, 'include_velocity': True}}}
    path = tmp_path / 'roll.yaml'
    yaml.safe_dump(cfg, open(path, 'w'))
    env = make(str(path), **NODAMP)
    for _ in range(120):                       # settle
        env.step({'ball': {'push': torch.zeros(1, 3)}})
    F, steps = 10.0, 60                        # 0.25 s: the ball moves 2 cm, the push point stays (almost) at its centre
    for _ in range(steps):
        o, _, _, _ = env.step({'ball': {'push': torch.tensor([[F, 0.0, 0.0]])}})
    v = o['ball']['state']['velocity'][0].numpy(); w = o['ball']['state']['angular_velocity'][0].numpy()
    a_roll = F / (10.0 + 1.0 / 0.25)
    assert abs(v[0] - a_roll * steps / 240.0) < 0.02 * a_roll * steps / 240.0      # 2 %: rolling, not sliding (F / m would be 40 % more)
    assert abs(w[1] - v[0] / 0.5) < 0.02 * v[0] / 0.5 and abs(v[1]) < 1e-6 and abs(w[0]) < 1e-6


def test_joint_limit_stops_a_falling_link():
    """cart_tree's pole joints carry limits: driven hard against one, the joint must stop within the solver's slop
    of the limit and stay there (velocity-level limit row with ERP, never an impulse that pulls)."""
    env = make(os.path.join(ROOT, 'tests', 'golden', 'cart_tree.yaml'))
    robot = env.models['cart'].robot
    lim = [(j.q_index, j.lower, j.upper) for j in robot.joints if j.q_index > -1 and j.lower <= j.upper]
    assert lim, 'fixture has no limited joint'
    hi = env.action_space
    from diy_gym_amd.utils import flatten, get_bounds_for_space
    top = torch.as_tensor(flatten(get_bounds_for_space(hi, False)), dtype=torch.float32)[None]
    for _ in range(300):
        env.sim.step(env._all_slots, top)
    st = env.sim.get_state()[0]
    for qi, lo, up in lim:
        q = st[link_q(env, list(env.models).index('cart'), qi)]
        assert lo - 2e-3 <= q <= up + 2e-3, (qi, q, lo, up)


def test_head_on_collision_is_inelastic_and_conserves_momentum(tmp_path):
    """No gravity, no plane: a 10 kg sphere pushed to 1 m/s runs into an identical sphere at rest.  The contact row
    removes the approach velocity and never pulls (restitution 0, impulse >= 0): afterwards both move at v/2 and the
    total momentum is what the push put in."""
    import yaml
    cfg = {'render': False, 'gravity': [0.0, 0.0, 0.0],
           'a': {'model': 'sphere2.urdf', 'xyz': [-1.0, 0.0, 0.0], 'push': {'addon': 'external_force', 'xyz': [-1.0, 0.0, 0.0]},
                 'state': {'addon': 'object_state_sensor', 'include_velocity': True}},
           'b': {'model': 'sphere2.urdf', 'xyz': [0.6, 0.0, 0.0],
                 'state': {'addon': 'object_state_sensor', 'include_velocity': True}}}
    path = tmp_path / 'hit.yaml'
    yaml.safe_dump(cfg, open(path, 'w'))
    env = make(str(path), **NODAMP)
    # 24 steps of 100 N on 10 kg: 1 m/s (the push point is a world point: the ball has moved only 5 cm, and a force
    # along x through a point on the x axis exerts no torque anyway)
    for _ in range(24):
        o, _, _, _ = env.step({'a': {'push': torch.tensor([[100.0, 0.0, 0.0]])}})
    va = float(o['a']['state']['velocity'][0, 0]); assert abs(va - 1.0) < 1e-6
    for _ in range(240):
        o, _, _, _ = env.step({'a': {'push': torch.zeros(1, 3)}})
    va, vb = float(o['a']['state']['velocity'][0, 0]), float(o['b']['state']['velocity'][0, 0])
    assert abs((va + vb) - 1.0) < 1e-6                      # momentum (equal masses)
    assert abs(va - 0.5) < 2e-3 and abs(vb - 0.5) < 2e-3    # perfectly inelastic: common velocity
    assert vb >= va - 1e-6                                  # separating or together, never interpenetrating further


def test_fixed_constraint_rows_make_two_free_bodies_move_as_one(tmp_path):
    """``attach: constraint`` (reference model.py:74-75, createConstraint(JOINT_FIXED) as six solver rows), no gravity, no
    plane: a 10 kg sphere carries a second one on a 0.5 m offset.  Pushed through the pair's common centre of mass with
    100 N for 24 steps, the two end up at F t / (m_a + m_b) = 0.5 m/s, without rotation, the offset kept; pushed through the
    PARENT's own centre the pair also turns -- angular momentum about the common centre = lever x impulse, with the pair's
    inertia 2 (2/5 m r^2) + 2 m (d/2)^2 -- and the child stays 0.5 m from the parent."""
    import yaml
    def build(push_at):
        cfg = {'render': False, 'gravity': [0.0, 0.0, 0.0],
               'a': {'model': 'sphere2.urdf', 'xyz': [0.0, 0.0, 0.0], 'push': {'addon': 'external_force', 'xyz': push_at},
                     'state': {'addon': 'object_state_sensor', 'include_velocity': True},
                     'b': {'model': 'sphere2.urdf', 'attach': 'constraint', 'xyz': [0.0, 0.5, 0.0], 'constraint_max_force': 1e5}}}
        path = tmp_path / ('pair_%s.yaml' % abs(hash(str(push_at))))
        yaml.safe_dump(cfg, open(path, 'w'))
        return make(str(path), **NODAMP)
    env = build([0.0, 0.25, 0.0])          # through the common centre of mass
    assert env.layout.n_bodies == 2
    for _ in range(24):
        o, _, _, _ = env.step({'a': {'push': torch.tensor([[100.0, 0.0, 0.0]])}})
    for _ in range(60):
        o, _, _, _ = env.step({'a': {'push': torch.zeros(1, 3)}})
    b = env.sim.frame_state64(env.models['a'].models['b'].uid, -1, com=True)[0]; a = env.sim.frame_state64(env.models['a'].uid, -1, com=True)[0]
    assert abs(a[7] - 0.5) < 1e-4 and abs(b[7] - 0.5) < 1e-4 and np.abs(a[10:13]).max() < 1e-4      # 100 N x 0.1 s / 20 kg, no spin
    assert np.allclose(b[:3] - a[:3], [0.0, 0.5, 0.0], atol=1e-5)
    env = build([0.0, 0.0, 0.0])           # through the parent's centre: 0.25 m beside the common one
    for _ in range(24):
        o, _, _, _ = env.step({'a': {'push': torch.tensor([[100.0, 0.0, 0.0]])}})
    for _ in range(60):
        o, _, _, _ = env.step({'a': {'push': torch.zeros(1, 3)}})
    b = env.sim.frame_state64(env.models['a'].models['b'].uid, -1, com=True)[0]; a = env.sim.frame_state64(env.models['a'].uid, -1, com=True)[0]
    m, r = 10.0, 0.5                        # sphere2.urdf: 10 kg, radius 0.5 (inertia as in the file)
    I_pair = 2 * float(env.builder.bodies[0][0].base_inertia[2, 2]) + 2 * m * 0.25 ** 2
    assert abs(0.5 * (a[7] + b[7]) - 0.5) < 1e-3                                   # linear momentum / 20 kg
    assert abs(np.linalg.norm(b[:3] - a[:3]) - 0.5) < 1e-4                           # rigid
    wz = a[12]
    assert abs(wz - b[12]) < 1e-4 and abs(abs(wz) - 0.25 * 10.0 / I_pair) < 0.03 * 0.25 * 10.0 / I_pair    # |L| = lever x impulse (the lever turns with the pair: 3 %)


# ---------------------------------------------------------------- dynamics_randomizer (reference dynamics_randomizer.py:24-32)
def _pendulum_with_randomizer(tmp_path, body, B=1, seed=0):
    cfg = tmp_path / 'pend_rand.yaml'
    cfg.write_text('render: no\npend:\n  model: %s\n  xyz: [0, 0, 0]\n  rand: {addon: dynamics_randomizer, %s}\n' %
                   (os.path.join(G, 'urdf', 'pendulum.urdf'), body))
    return DIYGym(str(cfg), num_envs=B, seed=seed, backend_factory=OracleBackend, engine=NODAMP)


def _held_effort(env):
    """Torque the default velocity motor applies to hold the bob at 1 rad (= m g L sin q while it holds)."""
    qo = link_q(env, 0, 0)
    st = env.sim.get_state(); st[:, qo] = 1.0; st[:, qo + 1] = 0.0; env.sim.set_state(st)
    env.sim.set_motor_cfg(np.array([[0.0, 1.0, 1e4]]))  # a velocity motor strong enough for any mass drawn here
    for _ in range(20):
        env.sim.step(0)
    return np.abs(env.sim.get_state()[:, qo + K.LS_APPLIED])


def test_dynamics_randomizer_compounds_masses_like_the_reference(tmp_path):
    # log(U(e^2, e^2)) = 2: the reference multiplies the CURRENT mass by it at construction, again in the constructor's
    # reset() and again at every later reset -> 4 m, 8 m, 16 m.  The holding torque m g L sin(q) shows the mass.
    e2 = float(np.exp(2.0))
    env = _pendulum_with_randomizer(tmp_path, 'mass_range: [%r, %r], damping_range: [%r, %r], mass_scale_limits: [1.0e-3, 1.0e3]' % (e2, e2, np.e, np.e))
    rand = env.models['pend'].addons['rand']
    base = 1.0 * 9.81 * 0.5 * np.sin(1.0)
    for want in (4.0, 8.0, 16.0):
        assert abs(float(rand.mass_scales()[0, 0]) - want) < 1e-9
        assert abs(_held_effort(env)[0] - want * base) < 2e-2 * want
        env.reset()
    # angular damping = log(U(e, e)) * URDF joint damping = 1 * 0 for this URDF: Bullet's default 0.04 is switched off
    assert float(rand.angular_damping()[0]) == 0.0


def test_dynamics_randomizer_guards_and_streams(tmp_path):
    # U = 0.5 -> log U = -0.693: the reference would set a NEGATIVE mass; the documented guard takes |log U|
    env = _pendulum_with_randomizer(tmp_path, 'mass_range: [0.5, 0.5]')
    rand = env.models['pend'].addons['rand']
    assert abs(float(rand.mass_scales()[0, 0]) - np.log(2.0)**2) < 1e-12
    # the accumulated scale is clamped (compounding would otherwise run away): 0.48^k hits the lower limit
    env = _pendulum_with_randomizer(tmp_path, 'mass_range: [0.5, 0.5], mass_scale_limits: [0.2, 5.0]')
    rand = env.models['pend'].addons['rand']
    for _ in range(4):
        env.reset()
    assert float(rand.mass_scales()[0, 0]) == 0.2
    # default ranges: every env draws its own factors, a masked reset only re-draws the masked envs, and a shard
    # with env_index_base reproduces the same envs
    env = _pendulum_with_randomizer(tmp_path, 'mass_range: [0.25, 4.0]', B=6, seed=9)
    rand = env.models['pend'].addons['rand']
    s0 = rand.mass_scales()[:, 0].clone()
    assert len({round(float(v), 9) for v in s0}) == 6 and float(s0.min()) > 0
    mask = torch.tensor([1, 0, 0, 1, 0, 0], dtype=torch.uint8)
    env.reset(mask)
    s1 = rand.mass_scales()[:, 0]
    assert all((float(s1[i]) != float(s0[i])) == bool(mask[i]) for i in range(6))
    cfg = tmp_path / 'pend_rand.yaml'
    shard = DIYGym(str(cfg), num_envs=3, seed=9, env_index_base=3, backend_factory=OracleBackend, engine=NODAMP)
    assert np.array_equal(shard.models['pend'].addons['rand'].mass_scales()[:, 0].numpy(), s0[3:].numpy())


def test_dynamics_randomizer_needs_a_movable_joint(tmp_path):
    cfg = tmp_path / 'm.yaml'
    cfg.write_text('ball:\n  model: sphere2.urdf\n  rand: {addon: dynamics_randomizer}\n')
    with pytest.raises(ValueError, match='no movable joint'):
        DIYGym(str(cfg), num_envs=1, backend_factory=OracleBackend)


# ---------------------------------------------------------------- force_torque_sensor (reference force_torque_sensor.py:14-23)
def _pendulum_tool(tmp_path, extra='', B=1):
    cfg = tmp_path / 'pend_tool.yaml'
    cfg.write_text('render: no\n%spend:\n  model: %s\n  xyz: [0, 0, 0]\n  wrist: {addon: force_torque_sensor, frame: mount}\n'
                   '  shoulder: {addon: force_torque_sensor, frame: hinge}\n' % (extra, os.path.join(G, 'urdf', 'pendulum_tool.urdf')))
    return DIYGym(str(cfg), num_envs=B, backend_factory=OracleBackend, engine=NODAMP)


def test_force_torque_sensor_static_load_is_weight_at_the_right_lever_arm(tmp_path):
    env = _pendulum_tool(tmp_path)
    qo, q = link_q(env, 0, 0), 0.8
    st = env.sim.get_state(); st[0, qo] = q; env.sim.set_state(st)
    env.sim.set_motor_cfg(np.array([[0.0, 1.0, 1e4]]))  # a strong velocity motor holds the joint
    for _ in range(30):
        env.sim.step(0)
    obs = env.observe()['pend']
    g, c, s_ = 9.81, np.cos(q), np.sin(q)
    Rt = np.array([[c, 0, -s_], [0, 1, 0], [s_, 0, c]])  # world -> link axes (rotation by q about y, transposed)
    # across the FIXED joint: the parent holds the tool's weight; about the tool's own COM that force has no lever arm
    assert np.allclose(obs['wrist']['force'], Rt @ [0, 0, 0.3 * g], atol=2e-3)
    assert np.allclose(obs['wrist']['torque'], 0.0, atol=2e-4)
    # across the hinge: rod + tool (1.3 kg); torque about the ROD's COM = tool weight x 0.6 m lever, about the y axis
    assert np.allclose(obs['shoulder']['force'], Rt @ [0, 0, 1.3 * g], atol=5e-3)
    assert np.allclose(obs['shoulder']['torque'], [0, 0.6 * s_ * 0.3 * g, 0], atol=2e-3)
    # ... whose component along the hinge axis, taken about the hinge, is what the motor reports as its effort
    lever = 0.5 * s_ * 1.0 * g + 1.1 * s_ * 0.3 * g
    assert abs(abs(env.sim.get_state()[0, qo + K.LS_APPLIED]) - lever) < 2e-2


def test_force_torque_sensor_swinging_pendulum_matches_newton_euler_closed_form(tmp_path):
    env = _pendulum_tool(tmp_path)
    qo = link_q(env, 0, 0)
    st = env.sim.get_state(); st[0, qo] = 1.2; env.sim.set_state(st)
    env.sim.set_motor_cfg(np.array([[0.0, 1.0, 0.0]]))  # motor off: free swing
    for _ in range(60):
        env.sim.step(0)
    obs = env.observe(_refresh=False)['pend']
    q, qd = env.sim.get_state()[0, qo:qo + 2]
    g = 9.81
    # hinge-axis inertia and gravity torque of rod + tool about the hinge
    I_h = (0.02 + 1.0 * 0.5**2) + (0.001 + 0.3 * 1.1**2)
    qdd = -(1.0 * 0.5 + 0.3 * 1.1) * g * np.sin(q) / I_h
    def part(m, L):  # force needed to move a point mass on the rod at distance L (world x, z), minus gravity
        t = np.array([-np.cos(q), 0, np.sin(q)]) * L   # d(position)/dq for position = (-L sin q, 0, 1 - L cos q)
        n = np.array([np.sin(q), 0, np.cos(q)]) * L    # -d2(position)/dq2 ... centripetal direction
        return m * (t * qdd + n * qd * qd + np.array([0, 0, g]))
    F_world = part(1.0, 0.5) + part(0.3, 1.1)
    c, s_ = np.cos(q), np.sin(q)
    Rt = np.array([[c, 0, -s_], [0, 1, 0], [s_, 0, c]])
    assert np.allclose(obs['shoulder']['force'], Rt @ F_world, rtol=0, atol=0.02 * np.linalg.norm(F_world))
    assert np.allclose(obs['wrist']['force'], Rt @ part(0.3, 1.1), rtol=0, atol=0.02 * np.linalg.norm(F_world))


def test_force_torque_sensor_sees_contact_forces_on_the_child_side(tmp_path):
    # the pendulum leans on a box top with its tool sphere (motor off): the wrist then carries the tool's weight MINUS the support
    box = ('prop:\n  model: %s\n  use_fixed_base: yes\n  xyz: [-0.7, 0, 0.0]\n' % os.path.join(G, 'urdf', 'ft_prop.urdf'))
    (tmp_path / 'x').mkdir()
    env = _pendulum_tool(tmp_path, extra=box)
    qo = link_q(env, 1, 0)
    st = env.sim.get_state(); st[0, qo] = 0.95; env.sim.set_state(st)
    env.sim.set_motor_cfg(np.array([[0.0, 1.0, 0.0]]))
    for _ in range(400):
        env.sim.step(0)
    s = env.sim.get_state()[0]
    q, qd = s[qo], s[qo + 1]
    assert abs(qd) < 2e-3 and env.sim.contacts(0) == 1          # at rest on the prop
    obs = env.observe(_refresh=False)['pend']
    g = 9.81
    # moment balance about the hinge: N x_contact = (m_rod x_rod + m_tool x_tool) g with a vertical support force
    # (the sphere touches the horizontal top face; friction keeps it from sliding but carries no load at rest)
    x_rod, x_tool = 0.5 * np.sin(q), 1.1 * np.sin(q)
    N = (1.0 * x_rod + 0.3 * x_tool) * g / x_tool
    c, s_ = np.cos(q), np.sin(q)
    Rt = np.array([[c, 0, -s_], [0, 1, 0], [s_, 0, c]])
    F = Rt.T @ np.asarray(obs['wrist']['force'], dtype=np.float64)[0]
    assert abs(F[2] - (0.3 * g - N)) < 0.03 * N and abs(F[0]) < 0.03 * N
    Fh = Rt.T @ np.asarray(obs['shoulder']['force'], dtype=np.float64)[0]
    assert abs(Fh[2] - (1.3 * g - N)) < 0.03 * N


def test_fp32_build_of_the_oracle_and_what_it_says_about_tolerances():
    """oracle/libdgsim_oracle_f32.so is the same source with ``real = float`` (bench.py's cpu_baseline times it).  Besides
    being a baseline it calibrates the GPU tolerances: what separates the fp32 and fp64 builds of ONE implementation is
    rounding alone.  (i) ur_high_5 (motors, no contacts): 1e-4 after 30 steps.  (ii) r2d2_maze with the wheels commanded to
    different speeds: the two builds drift apart by more than 5e-2 within 60 steps -- the contact problem is chaotic at
    fp32 resolution, which is why the GPU rollout test of that scene is loose.  (iii) Single steps from a COMMON state
    are heavy-tailed too (which four hull points of a wheel are deepest, whether a sweep leaves at the residual threshold:
    both flip on the last bit): median 2e-5, 90th percentile 2.5e-3, worst > 1e-2 over (env, step) pairs in contact --
    so tests/test_parity_gpu.py::test_r2d2_maze_single_steps_from_the_oracle_state asserts the median and the 90th
    percentile of the HIP kernel's single steps, not their maximum."""
    import oracle_backend as ob
    from diy_gym_amd import DIYGym
    f32 = ob.flavour('f32')
    assert f32 is not None and ob.lib(f32.lib_path).dgo_real_bytes() == 4 and ob.lib().dgo_real_bytes() == 8
    ur = os.path.join(ROOT, 'examples', 'ur_high_5', 'ur_high_5.yaml')
    a, b = DIYGym(ur, num_envs=4, seed=3, backend_factory=ob.OracleBackend), DIYGym(ur, num_envs=4, seed=3, backend_factory=f32)
    gen = torch.Generator().manual_seed(0)
    for _ in range(30):
        act = (torch.rand((4, 12), generator=gen) * 2 - 1) * 0.01
        a.sim.step(a._all_slots, act); b.sim.step(b._all_slots, act)
    assert float((a.sim.obs - b.sim.obs).abs().max()) < 1e-4
    maze = os.path.join(ROOT, 'examples', 'r2d2_maze', 'r2d2_maze.yaml')
    a, b = DIYGym(maze, num_envs=8, seed=5, backend_factory=ob.OracleBackend), DIYGym(maze, num_envs=8, seed=5, backend_factory=f32)
    L = a.layout
    eff = [o + 5 for o in L.link_state_off]; kin = [k for k in range(L.physical_dim) if k not in eff]   # (not the cached contact impulses)
    drift = 0.0
    for _ in range(60):
        act = (torch.rand((8, 4), generator=gen) * 2 - 1) * 10.0
        a.sim.step(a._all_slots, act); b.sim.step(b._all_slots, act)
        drift = max(drift, float(np.abs(a.sim.get_state() - b.sim.get_state())[:, kin].max()))
    assert drift > 5e-2, drift
    single = []
    for _ in range(40):
        b.sim.set_state(a.sim.get_state())
        act = (torch.rand((8, 4), generator=gen) * 2 - 1) * 10.0
        a.sim.step(a._all_slots, act); b.sim.step(b._all_slots, act)
        single += list(np.abs(a.sim.get_state() - b.sim.get_state())[:, kin].max(1))
    assert np.median(single) < 1e-4 and np.quantile(single, 0.9) < 1e-2 and np.max(single) > 1e-2, (np.median(single), np.quantile(single, 0.9), np.max(single))


# ---- limit_guess: a motor pushing into an active joint limit (DG_HF_LIMIT_GUESS) -----------------------------------------------
def _limited_pendulum(tmp_path, rest, iterations=150, **engine):
    import yaml
    cfg = {'render': False, 'solver_iterations': iterations,
           'pend': {'model': os.path.join(G, 'urdf', 'pendulum_limited.urdf'), 'xyz': [0, 0, 0],
                    'drive': {'addon': 'joint_controller', 'control_mode': 'position', 'rest_position': [rest]},
                    'joints': {'addon': 'joint_state_sensor', 'include_effort': True}}}
    path = tmp_path / ('limited_%s_%d.yaml' % ('_'.join('%s%s' % kv for kv in sorted(engine.items())), iterations))
    yaml.safe_dump(cfg, open(path, 'w'), sort_keys=False)
    return make(str(path), **NODAMP, **engine)


@pytest.mark.parametrize('side', [1.0, -1.0])
def test_motor_pushed_into_a_joint_limit_starts_at_its_fixed_point(tmp_path, side):
    """A 1 kg bob on a 0.5 m arm rests ON its joint limit (+-0.5 rad); the position target lies 1e-4 rad beyond it.  The two
    rows of that joint share their Jacobian: the fixed point is the motor saturated into the limit (+-300 N m, the URDF's
    effort) with the limit row holding whatever gravity (m g L sin q = 2.35 N m, pulling back towards 0) leaves of it.  Left to the sweeps the two rows ramp
    up against each other by (b_motor - b_limit) / diag per sweep -- ~1 700 sweeps here -- so at pybullet's 150 iterations
    the reported effort is a few per cent of the true one.  With limit_guess the joint enters the motor guess as one
    unknown and the sweeps start AT that fixed point.  Asserted: effort = +-300 to 1e-6 after <= 3 sweeps; the joint does
    not move; without the guess the cap is hit and the effort is far off; and with the cap lifted the ungessed sweeps
    reach the same state (same fixed point)."""
    q0 = 0.5 * side
    act = torch.tensor([[q0 + 1e-4 * side]])
    on = _limited_pendulum(tmp_path, q0)
    on.sim.step(on._all_slots, act)
    qo = link_q(on, 0, 0)
    s = on.sim.get_state()[0]
    assert on.sim.iterations(0) <= 3, on.sim.iterations(0)
    assert abs(s[qo + K.LS_APPLIED] - 300.0 * side) < 1e-6 and abs(s[qo] - q0) < 1e-6 and abs(s[qo + 1]) < 1e-6
    off = _limited_pendulum(tmp_path, q0, limit_guess=0.0)
    off.sim.step(off._all_slots, act)
    so = off.sim.get_state()[0]
    assert off.sim.iterations(0) == 150 and abs(so[qo + K.LS_APPLIED]) < 60.0   # the ramp: 150 of ~1 700 sweeps
    assert abs(so[qo] - q0) < 1e-5                                               # (the joint itself is held either way)
    full = _limited_pendulum(tmp_path, q0, iterations=20000, limit_guess=0.0)
    full.sim.step(full._all_slots, act)
    sf = full.sim.get_state()[0]
    assert 150 < full.sim.iterations(0) < 20000
    assert abs(sf[qo + K.LS_APPLIED] - 300.0 * side) < 1e-2 and abs(sf[qo] - s[qo]) < 1e-7 and abs(sf[qo + 1] - s[qo + 1]) < 1e-6


def test_limit_guess_leaves_a_motor_alone_that_cannot_reach_the_limit(tmp_path):
    """The other branch: the limit row is a candidate (the joint is within 0.25 rad of the limit) and the target lies beyond it,
    but the motor (2 N m) is weaker than gravity (2.1 N m at 0.45 rad): it cannot even hold the joint, let alone reach the
    limit velocity.  The pinned unknown then exceeds the motor's bound, the joint is held at that bound like any saturated
    row and the limit row starts at zero -- the state after 20 steps is the one without limit_guess, bit for bit, and the
    bob sinks."""
    act = torch.tensor([[0.6]])
    a = _limited_pendulum(tmp_path, 0.45); b = _limited_pendulum(tmp_path, 0.45, limit_guess=0.0)
    for e in (a, b):
        e.sim.set_motor_cfg(np.array([[0.03, 1.0, 2.0]]))
    for _ in range(20):
        for e in (a, b):
            e.sim.step(0)   # (mask 0: the controller op does not rewrite the motor table; targets stay at the reset's rest position)
    qo = link_q(a, 0, 0)
    sa, sb = a.sim.get_state()[0], b.sim.get_state()[0]
    assert np.array_equal(sa, sb) and sa[qo] < 0.45 - 1e-5 and abs(abs(sa[qo + K.LS_APPLIED]) - 2.0) < 1e-9


def test_motor_impulse_time_base_substep_or_full_step(tmp_path):
    """``motor_impulse_timebase``: the impulse bound of a motor row is max force x the SUBSTEP by default; 'step' makes it max
    force x the full fixedTimeStep -- the other reading of pybullet's maxAppliedImpulse [R] -- i.e. a saturated motor is
    numSubSteps (2) times as strong.  A 1.5 N m velocity motor against 1.91 N m of gravity at 0.4 rad: saturated and sinking
    in the first reading; in the second its bound is 3.0 N m, it holds the joint and reports the gravity torque."""
    q0 = 0.4
    sub = _limited_pendulum(tmp_path, q0); full = _limited_pendulum(tmp_path, q0, motor_impulse_timebase='step')
    for e in (sub, full):
        e.sim.set_motor_cfg(np.array([[0.0, 1.0, 1.5]]))   # velocity motor, target 0, 1.5 N m
        for _ in range(10):
            e.sim.step(0)
    qo = link_q(sub, 0, 0)
    ss, sf = sub.sim.get_state()[0], full.sim.get_state()[0]
    grav = 1.0 * 9.81 * 0.5 * np.sin(q0)   # 1.91 N m
    assert abs(abs(ss[qo + K.LS_APPLIED]) - 1.5) < 1e-9 and ss[qo] < q0 - 1e-5         # 1.5 < 1.91: saturated, sinks
    assert abs(abs(sf[qo + K.LS_APPLIED]) - grav) < 2e-2 and abs(sf[qo] - q0) < 1e-4   # bound 3.0 > 1.91: holds, effort = gravity


def test_link_frame_of_an_external_force_is_the_inertial_frame(tmp_path):
    """p.applyExternalForce(uid, link, F, pos, LINK_FRAME): pybullet resolves force and position against the link's INERTIAL
    frame (centre of mass, btMultiBody's cached world transform [R]), not the joint frame.  The pendulum's bob has its centre
    of mass 0.5 m below the hinge: a unit force along x applied at pos = 0 in LINK_FRAME acts AT the bob, lever 0.5 m about
    the hinge (y axis) -> qdd = tau / (m L^2) = 0.5 / 0.25 = 2 rad/s^2; the same force given in WORLD_FRAME at the world
    position of the hinge (0, 0, 1) has no lever at all.  No gravity, motor off."""
    import yaml
    cfg = {'render': False, 'gravity': [0.0, 0.0, 0.0], 'pend': {'model': os.path.join(G, 'urdf', 'pendulum.urdf'), 'xyz': [0, 0, 0]}}
    path = tmp_path / 'pend0g.yaml'
    yaml.safe_dump(cfg, open(path, 'w'), sort_keys=False)
    for flags, pos, expect in ((OracleBackend.LINK_FRAME, [0.0, 0.0, 0.0], 2.0), (OracleBackend.WORLD_FRAME, [0.0, 0.0, 1.0], 0.0), (OracleBackend.WORLD_FRAME, [0.0, 0.0, 0.5], 2.0)):
        env = make(str(path), **NODAMP)
        env.sim.set_motor_cfg(np.array([[0.0, 1.0, 0.0]]))   # no motor
        uid = env.models['pend'].uid
        env.sim.apply_external_force(uid, env.models['pend'].get_frame_id('hinge'), [1.0, 0.0, 0.0], pos, flags)
        env.sim.step(0)
        qd = env.sim.get_state()[0, link_q(env, 0, 0) + 1]
        assert abs(abs(qd) - expect * 0.25 / (0.25 + 1e-6) / 240.0) < 1e-9, (flags, pos, qd * 240.0)   # (the bob's own 1e-6 kg m^2 added to m L^2)
