"""Scene-level host tests on the oracle: the stand-in assets, the maze generator, frozen bodies,
broad-phase groups and the contact budget."""
import os
import subprocess
import sys

import numpy as np
import torch

import diy_gym_amd.examples  # noqa: F401
from diy_gym_amd import DIYGym
from diy_gym_amd.scene import K, as_box, body_bound, box_corners
from diy_gym_amd.mathx import Transform
from oracle_backend import OracleBackend

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_r2d2_stand_in_has_the_joint_names_the_reference_pins():
    from diy_gym_amd.urdf import UrdfRobot
    r = UrdfRobot(os.path.join(ROOT, 'diy_gym_amd', 'data', 'pybullet_data', 'r2d2.urdf'))
    # generate_maze.py:30-31,49 and from_the_readme.yaml:18
    for name in ('left_front_wheel_joint', 'left_back_wheel_joint', 'right_front_wheel_joint', 'right_back_wheel_joint',
                 'left_gripper_joint', 'left_tip_joint'):
        assert name in r.joint_names
    assert r.num_dofs == 8 and len(r.joints) == 15


def test_maze_generator_is_seeded_and_scene_compiles(tmp_path):
    out = tmp_path / 'maze.yaml'
    subprocess.check_call([sys.executable, os.path.join(ROOT, 'tools', 'generate_maze.py'), '--maze_size', '4', '--seed', '3', '--out', str(out)])
    a = out.read_text()
    subprocess.check_call([sys.executable, os.path.join(ROOT, 'tools', 'generate_maze.py'), '--maze_size', '4', '--seed', '3', '--out', str(out)])
    assert out.read_text() == a
    env = DIYGym(str(out), num_envs=2, backend_factory=OracleBackend)
    I = env.layout.I
    n_walls = sum(1 for k in env.models if k.startswith('wall_'))
    assert n_walls >= 16
    # walls (box meshes) became analytic boxes of the frozen static world; they carry no per-env state
    SI = I[I[K.H_OFF_SHAPE_I]:I[K.H_OFF_SHAPE_I] + I[K.H_N_SHAPES] * K.SI_STRIDE].reshape(-1, K.SI_STRIDE)
    wall_uid = env.models['wall_0'].uid
    row = SI[SI[:, K.SI_BODY] == wall_uid][0]
    assert row[K.SI_TYPE] == K.SHAPE_BOX and row[K.SI_FLAGS] & K.SHAPE_WORLD
    assert env.layout.body_state_off[wall_uid] == -1 and env.layout.physical_dim < 100
    # one broad-phase group per (R2D2, static shape)
    assert I[K.H_N_GROUPS] == n_walls + 1 and I[K.H_N_PAIRS] == 16 * (n_walls + 1)
    assert I[K.H_MAX_CONTACTS] == 12  # the generator's max_contacts key
    # drives forward on its wheels
    x0 = env.sim.get_state()[0, env.layout.body_state_off[env.models['r2d2'].uid]]
    for _ in range(60):
        env.step({'r2d2': {'wheel_driver': torch.full((2, 4), 5.0)}})
    x1 = env.sim.get_state()[0, env.layout.body_state_off[env.models['r2d2'].uid]]
    assert 0.02 < x1 - x0 < 0.06  # 60 steps at 5 rad/s x 0.035 m wheels, minus the initial drop and spin-up


def test_box_detection_and_bounds():
    T = Transform.from_xyz_rpy([1, 2, 3], [0.3, -0.2, 0.5])
    half = np.array([0.5, 0.05, 0.5])
    got = as_box(box_corners(T, half))
    assert got is not None and np.allclose(sorted(got[1]), sorted(half))
    assert as_box(np.random.default_rng(0).normal(size=(8, 3))) is None
    from diy_gym_amd import mesh
    from diy_gym_amd.urdf import FlatBody, UrdfRobot
    f = FlatBody(UrdfRobot(os.path.join(ROOT, 'diy_gym_amd', 'data', 'ur5', 'ur5_robot.urdf')), mesh_loader=mesh.load_convex)
    assert 0.9 < body_bound(f) < 1.6  # UR5 reach is 0.85 m; the bound adds link offsets conservatively


def test_from_the_readme_scene_runs():
    env = DIYGym(os.path.join(ROOT, 'examples', 'from_the_readme', 'from_the_readme.yaml'), backend_factory=OracleBackend)
    ctl = env.models['robot'].addons['controller']
    assert ctl.end_effector_joint_id == 8 and len(ctl.joint_ids) == 7
    obs = env.reset()
    assert obs['r2d2']['arm_camera']['rgb'].shape == (200, 200, 3)
    assert set(env.action_space['robot']['controller'].spaces) == {'linear'}
    _, rew, term, _ = env.step({'robot': {'controller': {'linear': [0.01, 0, 0]}}})
    assert set(rew['from_the_readme']) == {'grab_r2d2'} and set(rew['robot']) == {'lazy_robot'}
    assert set(term['from_the_readme']) == {'grab_r2d2', 'episode_timer'}
