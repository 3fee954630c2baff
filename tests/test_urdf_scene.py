"""URDF flattening and the scene blob (host side of model.py / loadURDF)."""
import os
import re

import numpy as np

from diy_gym_amd import mesh
from diy_gym_amd.mathx import Transform, euler_from_quat, mat_from_euler, quat_from_euler, quat_from_mat
from diy_gym_amd.scene import K, SceneBuilder
from diy_gym_amd.urdf import FlatBody, UrdfRobot

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DATA = os.path.join(ROOT, 'diy_gym_amd', 'data')


def test_ur5_joint_numbering_follows_pybullet_dfs():
    r = UrdfRobot(os.path.join(DATA, 'ur5', 'ur5_robot.urdf'))
    assert r.root == 'world' and r.num_dofs == 6
    # depth-first from the root, children in file order, fixed joints included
    assert r.joint_names == ['world_joint', 'shoulder_pan_joint', 'shoulder_lift_joint', 'elbow_joint', 'wrist_1_joint',
                             'wrist_2_joint', 'wrist_3_joint', 'ee_fixed_joint', 'wrist_3_link-tool0_fixed_joint',
                             'base_link-base_fixed_joint']
    assert r.joint_names.index('ee_fixed_joint') == 7
    assert [j.q_index for j in r.joints] == [-1, 0, 1, 2, 3, 4, 5, -1, -1, -1]
    info = r.joint_info(4)
    assert info['max_force'] == 28.0 and abs(info['upper'] - 2 * np.pi) < 1e-9 and info['max_velocity'] == 3.2


def test_ur5_flatten_lumps_fixed_links_and_world_root_is_fixed():
    r = UrdfRobot(os.path.join(DATA, 'ur5', 'ur5_robot.urdf'))
    f = FlatBody(r, mesh_loader=mesh.load_convex)
    assert f.fixed_base and f.num_dofs == 6 and f.base_mass == 0.0
    assert [l.parent for l in f.links] == [-1, 0, 1, 2, 3, 4]
    # wrist_3 carries ee_link and tool0, which have no <inertial>: Bullet gives each mass 1, inertia 1 [R]
    assert abs(f.links[5].mass - (0.1879 + 2.0)) < 1e-12
    assert abs(f.links[0].mass - 3.7) < 1e-12
    # named frames: index == pybullet link index, anchored to the right moving link
    assert f.frames[7].name == 'ee_fixed_joint' and f.frames[7].link == 5
    assert np.allclose(f.frames[7].T.p, [0.0, 0.0823, 0.0])
    assert f.frame_id('ee_fixed_joint') == 7 and f.frame_id('nope') == -1
    assert len(f.shapes) == 8  # 7 convex meshes + the ee box


def test_quadrotor_mass_override_and_inertialess_links():
    r = UrdfRobot(os.path.join(DATA, 'hector_quadrotor', 'quadrotor.urdf'))
    f = FlatBody(r, mass_override=4.0, mesh_loader=mesh.load_convex)
    assert not f.fixed_base and f.num_dofs == 0
    # base 4.0 (override, reference drone_pilot.yaml:25) + 4 inertia-less motor links at mass 1 each [R]
    assert abs(f.base_mass - 8.0) < 1e-12
    assert np.allclose(f.base_com, [0, 0, 4 * 0.033 / 8.0])
    assert f.frame_id('motor3_joint') == 2 and np.allclose(f.frames[2].T.p, [0.27, 0.0, 0.033])


def test_scale_applies_to_geometry():
    r = UrdfRobot(os.path.join(DATA, 'pybullet_data', 'sphere2.urdf'))
    f = FlatBody(r, scale=0.4)
    assert abs(f.shapes[0].params[0] - 0.2) < 1e-12 and f.base_mass == 10.0


def test_euler_quaternion_round_trip_matches_urdf_convention():
    rng = np.random.default_rng(0)
    for _ in range(50):
        rpy = rng.uniform(-1.4, 1.4, 3)
        q = quat_from_euler(rpy)
        assert np.allclose(euler_from_quat(q), rpy, atol=1e-9)
        R = mat_from_euler(rpy)
        cr, sr, cp, sp, cy, sy = np.cos(rpy[0]), np.sin(rpy[0]), np.cos(rpy[1]), np.sin(rpy[1]), np.cos(rpy[2]), np.sin(rpy[2])
        Rz = np.array([[cy, -sy, 0], [sy, cy, 0], [0, 0, 1]])
        Ry = np.array([[cp, 0, sp], [0, 1, 0], [-sp, 0, cp]])
        Rx = np.array([[1, 0, 0], [0, cr, -sr], [0, sr, cr]])
        assert np.allclose(R, Rz @ Ry @ Rx, atol=1e-12)
        assert np.allclose(np.abs(quat_from_mat(R)), np.abs(q), atol=1e-9)


def test_header_constants_parse_and_blob_layout():
    assert K.MAGIC == 0x44475953 and K.LS_STRIDE == 6 and K.OP_IK_CONTROL == 2 and K.OP_TERM_TIMER == 66
    text = open(os.path.join(ROOT, 'include', 'diygym_scene.h')).read()
    assert int(re.search(r'#define DG_VERSION (\d+)', text).group(1)) == K.VERSION
    b = SceneBuilder(max_episode_steps=10)
    f = FlatBody(UrdfRobot(os.path.join(DATA, 'ur5', 'ur5_robot.urdf')), mesh_loader=mesh.load_convex)
    b.add_body(f, [0, 0, 0], [0, 0, 0, 1])
    h = b.add_op(K.OP_OBS_JOINT_STATE, 'obs', body=0, flags=K.JS_VELOCITY, ilist=list(range(6)), io_dim=12)
    lay = b.finalize()
    I, F = lay.I, lay.F
    assert I[K.H_MAGIC] == K.MAGIC and I[K.H_N_LINKS] == 6 and I[K.H_N_BODIES] == 1 and I[K.H_OBS_DIM] == 12 and h.io_off == 0
    assert I[K.H_STATE_DIM] == K.ST_PREFIX + K.BS_FIXED_END + K.EXT_STRIDE + 6 * K.LS_STRIDE
    assert abs(F[K.HF_DT] - 1 / 480.0) < 1e-15 and F[K.HF_GRAV_Z] == -9.81
    LF = F[I[K.H_OFF_LINK_F]:I[K.H_OFF_LINK_F] + 6 * K.LF_STRIDE].reshape(6, K.LF_STRIDE)
    assert np.allclose(LF[0, K.LF_POS:K.LF_POS + 3], [0, 0, 0.089159]) and LF[3, K.LF_MAX_FORCE] == 28.0
    assert I.dtype == np.int32 and F.dtype == np.float64 and I.flags['C_CONTIGUOUS']


def test_mesh_hull_thinning_is_inscribed_and_close():
    rng = np.random.default_rng(1)
    pts = rng.normal(size=(400, 3)) * [0.05, 0.05, 0.2]
    full = mesh.convex_points(pts, 10**9)
    thin = mesh.convex_points(pts, 24)
    assert len(thin) <= 24 and mesh.hull_error(full, thin) < 0.03
    assert all(any(np.allclose(t, f) for f in full) for t in thin)
