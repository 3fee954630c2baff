"""Host-side behaviour of DIYGym, restating the reference's
diy_gym/tests/test_environment.py:12-40 and test_utils.py:13-20.  The backend is
the CPU oracle (injected from tests/), so these run without a GPU; the same
assertions run on the HIP path in test_environment_gpu.py."""
import os
from collections import OrderedDict

import numpy as np
import pytest
import torch

import diy_gym_amd.examples  # noqa: F401
from diy_gym_amd import Addon, AddonFactory, DIYGym
from diy_gym_amd.utils import flatten, unflatten
from oracle_backend import OracleBackend

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BASIC = os.path.join(ROOT, 'tests', 'golden', 'basic_env_nocam.yaml')


@pytest.fixture
def env():
    return DIYGym(BASIC, backend_factory=OracleBackend)


def test_load_environment(env):
    for name in ('plane', 'red_marble', 'green_marble', 'blue_marble'):
        assert name in env.models
    assert list(env.receptors) == ['basic_env_nocam', 'blue_marble', 'green_marble', 'plane', 'red_marble']
    assert [env.models[n].uid for n in ('plane', 'red_marble', 'green_marble', 'blue_marble')] == [0, 1, 2, 3]  # YAML order


def test_spaces(env):
    assert 'force' in env.action_space['blue_marble'].spaces
    assert 'pose' in env.observation_space['green_marble'].spaces
    assert 'respawn' not in env.action_space['blue_marble'].spaces


def test_episode(env):
    """Reference test_environment.py:23-40, numpy in / numpy out, one env."""
    observation = env.reset()
    initial_position = observation['green_marble']['pose']['position']
    assert isinstance(initial_position, np.ndarray) and initial_position.shape == (3, )
    for _ in range(500):
        observation, _, _, info = env.step({'blue_marble': {'force': [0, -100, 0]}})
    final_position = observation['green_marble']['pose']['position']
    assert abs(np.linalg.norm(initial_position) - np.linalg.norm(final_position)) > 0.5
    assert info == {}
    observation = env.reset()
    reset_position = observation['green_marble']['pose']['position']
    assert abs(np.linalg.norm(initial_position) - np.linalg.norm(reset_position)) < 0.05


def test_flatten_unflatten(env):
    action = env.action_space.sample()
    back = unflatten(flatten(action), env.action_space)
    assert np.all(action['red_marble']['force'] == back['red_marble']['force'])
    assert np.all(action['blue_marble']['force'] == back['blue_marble']['force'])


def test_only_named_addons_are_updated():
    env = DIYGym(BASIC, num_envs=2, backend_factory=OracleBackend)
    env.step({'red_marble': {'force': torch.tensor([[5.0, 0, 0], [0, 5.0, 0]])}})
    assert env.sim.act[:, 3:6].tolist() == [[5.0, 0, 0], [0, 5.0, 0]]  # columns: blue(0:3), red(3:6)
    assert env._mask == 0b10
    env.step({'blue_marble': {'force': [1.0, 2.0, 3.0]}})
    assert env._mask == 0b01 and env.sim.act[1, 0:3].tolist() == [1.0, 2.0, 3.0]


def test_batched_outputs_are_views_and_collapse_modes():
    cfg = os.path.join(ROOT, 'examples', 'drone_pilot', 'drone_pilot.yaml')
    env = DIYGym(cfg, num_envs=4, backend_factory=OracleBackend)
    assert env.collapse_terminals_func is any
    act = {'drone': {'motor%d' % i: torch.full((4, 1), 0.5) for i in (1, 2, 3, 4)}}
    obs, rew, term, _ = env.step(act)
    assert obs['drone']['pose']['position'].shape == (4, 3) and obs['drone']['motor1'].shape == (4, 1)
    assert list(obs['drone']['pose'].keys()) == ['position', 'velocity', 'rotation', 'angular_velocity']
    off = env.models['drone'].addons['pose'].op.io_off  # motor1..4 come first (addons are name-sorted)
    assert off == 4 and obs['drone']['pose']['position'].data_ptr() == env.sim.obs[:, off:off + 3].data_ptr()
    assert rew['drone_pilot']['reach_goal'].shape == (4, ) and term.shape == (4, ) and term.dtype == torch.bool
    assert float(obs['drone']['motor1'][0, 0]) == pytest.approx(0.05)  # first-order spool-up, drone_pilot.py:34
    assert 'episode_timer' not in str(type(term))


def test_terminal_dict_has_episode_timer_and_terminal_if_all():
    cfg = os.path.join(ROOT, 'tests', 'golden', 'cart_tree.yaml')
    env = DIYGym(cfg, num_envs=2, backend_factory=OracleBackend)
    act = {'cart': {'drive': torch.zeros(2, 3)}}
    for i in range(39):
        _, _, term, _ = env.step(act)
    # terminal_if_all: needs every receptor that owns terminals to fire; only 'cart_tree' owns any here
    env.collapse_terminals_func = None
    d = env.is_terminal()
    assert list(d['cart_tree'].keys()) == ['near', 'episode_timer']
    assert d['cart_tree']['episode_timer'].tolist() == [False, False]
    env.step(act)
    assert env.is_terminal()['cart_tree']['episode_timer'].tolist() == [True, True]
    assert env.sim.term_flag.tolist() == [1, 1]


def test_flat_paths_are_zero_copy(tmp_path):
    import yaml
    tree = yaml.safe_load(open(os.path.join(ROOT, 'examples', 'ur_high_5', 'ur_high_5.yaml')))
    tree.update(flatten_observations=True, flatten_actions=True, sum_rewards=True)
    for arm in ('ur5_l', 'ur5_r'):
        tree[arm]['model'] = os.path.join(ROOT, 'diy_gym_amd', 'data', 'ur5', 'ur5_robot.urdf')
    cfg = tmp_path / 'flat.yaml'
    cfg.write_text(yaml.dump(tree))
    env = DIYGym(str(cfg), num_envs=3, backend_factory=OracleBackend)
    assert env.action_space.shape == (12, ) and env.observation_space.shape == (27, )
    obs, rew, term, _ = env.step(torch.zeros(3, 12))
    assert obs.shape == (3, 27) and obs.data_ptr() == env.sim.obs.data_ptr()
    assert rew.shape == (3, ) and term.shape == (3, )
    # flat layout == flatten() of the nested dict
    env.flatten_observations = False
    nested = env.observe(_refresh=False)
    assert torch.equal(flatten(nested, batch_dims=1), obs)


def test_custom_python_addon_still_works():
    """User addons without compile() keep the reference's hook contract (addon.py:91-186)."""
    class Counter(Addon):
        def __init__(self, parent, config):
            super().__init__(parent, config)
            self.n = 0

        def update(self, action):
            self.n += 1

        def reward(self):
            return torch.full((self.env.num_envs, ), float(self.n))

    AddonFactory.register_addon('counter', Counter)
    from diy_gym_amd.config import Configuration
    import yaml
    tree = yaml.safe_load(open(BASIC))
    tree['green_marble']['count'] = {'addon': 'counter'}
    env = DIYGym(Configuration.from_dict('custom', tree), num_envs=2, backend_factory=OracleBackend)
    _, rew, _, _ = env.step({'green_marble': {'count': None}})
    assert rew['green_marble']['count'].tolist() == [1.0, 1.0]


def test_error_conventions():
    from diy_gym_amd.config import Configuration
    with pytest.raises(KeyError):
        DIYGym(Configuration.from_dict('e', {'x': {'addon': 'no_such_addon'}}), backend_factory=OracleBackend)
    with pytest.raises(ValueError, match='Could not find URDF'):
        DIYGym(Configuration.from_dict('e', {'x': {'model': 'nope.urdf'}}), backend_factory=OracleBackend)
    with pytest.raises(NotImplementedError):
        DIYGym(Configuration.from_dict('e', {'gui': {'addon': 'draw_coords'}}), backend_factory=OracleBackend)


def test_spawn_multiple_clones_into_parent_models():
    from diy_gym_amd.config import Configuration
    tree = {'plane': {'model': 'grass/plane.urdf'},
            'crowd': {'addon': 'spawn_multiple', 'num_models': 3,
                      'ball': {'model': 'sphere2.urdf', 'scale': 0.2, 'xyz': [0, 0, 1.0],
                               'pose': {'addon': 'object_state_sensor'}}}}
    env = DIYGym(Configuration.from_dict('crowd_env', tree), num_envs=2, backend_factory=OracleBackend)
    assert [k for k in env.models] == ['plane', 'ball_0', 'ball_1', 'ball_2']  # appended after the sorted models, like the reference
    assert list(env.receptors) == ['ball_0', 'ball_1', 'ball_2', 'crowd_env', 'plane']
    assert sorted(env.models[k].uid for k in ('ball_0', 'ball_1', 'ball_2')) == [1, 2, 3]
    obs = env.reset()
    assert obs['ball_1']['pose']['position'].shape == (2, 3)


def test_child_model_is_bolted_to_the_parent_frame():
    """Child models (reference model.py:69-77): physically present, attached at ``parent_frame`` with the child's
    ``xyz`` / ``rpy`` as the pivot, not a receptor of their own (diy_gym.py:92 only lists top-level models)."""
    from diy_gym_amd.mathx import Transform, mat_from_quat
    env = DIYGym(os.path.join(ROOT, 'tests', 'golden', 'ur5_child_gripper.yaml'), num_envs=2, backend_factory=OracleBackend)
    arm = env.models['arm']; grip = arm.models['gripper']
    assert list(env.receptors) == ['arm', 'ur5_child_gripper'] or 'gripper' not in env.receptors
    assert env.layout.n_bodies == 1 and env.layout.n_links == 6 + 6          # merged into the arm's body
    assert grip.uid >= env.builder.ALIAS_BASE and env.builder.resolve(grip.uid)[0] == arm.uid
    assert env.action_space['arm']['controller'].shape == (6, )                # the parent's addons see the arm only
    for _ in range(25):
        env.step(env.action_space.sample())
    body, _, _, basef = env.builder.resolve(grip.uid)
    pe = env.sim.frame_state64(arm.uid, arm.get_frame_id('ee_fixed_joint'), com=True)
    pg = env.sim.frame_state64(body, basef, com=True)
    for e in range(2):
        Te = Transform(mat_from_quat(pe[e][3:7]), pe[e][:3]); Tg = Transform(mat_from_quat(pg[e][3:7]), pg[e][:3])
        rel = Te.inverse() * Tg
        assert np.allclose(rel.p, [0.0, 0.0, 0.02], atol=1e-9)
        assert np.allclose(rel.R, [[0, 0, 1], [0, 1, 0], [-1, 0, 0]], atol=1e-5)   # rpy = (0, pi/2, 0)
    xyz, quat = grip.get_transform()
    assert xyz.shape == (2, 3) and torch.isfinite(xyz).all() and torch.isfinite(quat).all()


def test_child_model_held_by_a_fixed_constraint(tmp_path):
    """``attach: constraint`` -- the reference's own arrangement (model.py:69-77): the child stays a body of its own and
    createConstraint(JOINT_FIXED) becomes six solver rows.  Holding the rest pose, the gripper is pulled onto the pivot
    (xyz / rpy in the inertial frame of the parent link) and ends up where the rigidly merged child sits."""
    from diy_gym_amd.mathx import Transform, mat_from_quat
    from diy_gym_amd.scene import K
    env = DIYGym(os.path.join(ROOT, 'tests', 'golden', 'ur5_constrained_gripper.yaml'), num_envs=2, backend_factory=OracleBackend)
    ref = DIYGym(os.path.join(ROOT, 'tests', 'golden', 'ur5_child_gripper.yaml'), num_envs=2, backend_factory=OracleBackend)
    arm = env.models['arm']; grip = arm.models['gripper']
    assert env.layout.n_bodies == 2 and env.layout.n_links == 6 + 6 and grip.uid == 1     # a body of its own
    assert int(env.layout.I[K.H_N_CONSTRAINTS]) == 1 and int(env.layout.I[K.H_N_PAIRS]) == 0   # the two do not collide with each other
    assert env.action_space['arm']['controller'].shape == (6, )
    hold = {'arm': {'controller': torch.tensor([[0.3, -1.2, 1.4, -0.6, 0.4, 0.1]] * 2)}}
    for _ in range(160):
        env.step(hold); ref.step(hold)
    pe = env.sim.frame_state64(arm.uid, arm.get_frame_id('ee_fixed_joint'), com=True)
    pg = env.sim.frame_state64(grip.uid, -1, com=True)
    rb, _, _, basef = ref.builder.resolve(ref.models['arm'].models['gripper'].uid)
    pr = ref.sim.frame_state64(rb, basef, com=True)
    for e in range(2):
        Te = Transform(mat_from_quat(pe[e][3:7]), pe[e][:3]); Tg = Transform(mat_from_quat(pg[e][3:7]), pg[e][:3])
        rel = Te.inverse() * Tg
        assert np.allclose(rel.p, [0.0, 0.0, 0.02], atol=1e-6)
        assert np.allclose(rel.R, [[0, 0, 1], [0, 1, 0], [-1, 0, 0]], atol=1e-5)
        assert np.allclose(pg[e][:3], pr[e][:3], atol=2e-5) and abs(abs(np.dot(pg[e][3:7], pr[e][3:7])) - 1.0) < 1e-9 and np.abs(pg[e][7:]).max() < 1e-5
    import yaml
    cfg = yaml.safe_load(open(os.path.join(ROOT, 'tests', 'golden', 'ur5_constrained_gripper.yaml'))); cfg['arm']['gripper']['attach'] = 'glue'
    yaml.safe_dump(cfg, open(tmp_path / 'glue.yaml', 'w'))
    with pytest.raises(ValueError):
        DIYGym(str(tmp_path / 'glue.yaml'), num_envs=1, backend_factory=OracleBackend)


def test_child_frame_attaches_the_child_by_that_link(tmp_path):
    """``child_frame`` (reference model.py:71-77): the CHILD LINK of that joint is what gets pinned to the parent frame;
    the rest of the child -- including its URDF root -- hangs from it (the child is re-rooted at that link)."""
    import yaml
    from diy_gym_amd.mathx import Transform, mat_from_quat
    cfg = yaml.safe_load(open(os.path.join(ROOT, 'tests', 'golden', 'ur5_child_gripper.yaml')))
    cfg['arm']['gripper']['child_frame'] = 'left_inner_finger_joint'
    path = tmp_path / 'c.yaml'
    yaml.safe_dump(cfg, open(path, 'w'))
    env = DIYGym(str(path), num_envs=2, backend_factory=OracleBackend)
    arm = env.models['arm']; grip = arm.models['gripper']
    assert env.layout.n_bodies == 1 and env.layout.n_links == 6 + 6
    assert grip.robot.root == 'left_inner_finger' and grip.robot.joint_names[0] != 'finger_joint'   # numbering follows the new tree
    for _ in range(10):
        env.step(env.action_space.sample())
    body, _, _, basef = env.builder.resolve(grip.uid)
    pe = env.sim.frame_state64(arm.uid, arm.get_frame_id('ee_fixed_joint'), com=True)
    pg = env.sim.frame_state64(body, basef, com=True)      # inertial frame of the pinned link
    for e in range(2):
        Te = Transform(mat_from_quat(pe[e][3:7]), pe[e][:3]); Tg = Transform(mat_from_quat(pg[e][3:7]), pg[e][:3])
        rel = Te.inverse() * Tg
        assert np.allclose(rel.p, [0.0, 0.0, 0.02], atol=1e-9)
        assert np.allclose(rel.R, [[0, 0, 1], [0, 1, 0], [-1, 0, 0]], atol=1e-5)
    # same total mass as the base-attached variant: nothing of the child is lost by re-rooting
    ref = DIYGym(os.path.join(ROOT, 'tests', 'golden', 'ur5_child_gripper.yaml'), num_envs=1, backend_factory=OracleBackend)
    assert abs(sum(fl.mass for fl in env.builder.bodies[0][0].links) - sum(fl.mass for fl in ref.builder.bodies[0][0].links)) < 1e-9


def test_rerooted_urdf_is_the_same_mechanism():
    """UrdfRobot.rerooted: for random joint values every link's inertial frame sits, relative to the new root, exactly where the
    original description puts it (serial arm, gripper tree, tree with a prismatic joint)."""
    from diy_gym_amd.mathx import Transform
    from diy_gym_amd.urdf import UrdfRobot

    def rot(axis, q):
        a = axis / np.linalg.norm(axis); Kx = np.array([[0, -a[2], a[1]], [a[2], 0, -a[0]], [-a[1], a[0], 0]])
        return np.eye(3) + np.sin(q) * Kx + (1 - np.cos(q)) * Kx @ Kx

    def fk(robot, qs):
        T = {robot.root: Transform()}
        for j in robot.joints:
            m = Transform(rot(j.axis, qs[j.name])) if j.type in ('revolute', 'continuous') else Transform(p=j.axis * qs[j.name]) if j.type == 'prismatic' else Transform()
            T[j.child] = T[j.parent] * j.origin * m
        return {n: T[n] * robot.links[n].inertial_origin for n in T}

    rng = np.random.default_rng(0)
    data = os.path.join(ROOT, 'diy_gym_amd', 'data')
    for path, roots in ((os.path.join(data, 'ur5', 'ur5_robot.urdf'), ['wrist_2_link', 'ee_link']),
                        (os.path.join(data, 'robotiq_2f', 'gripper.urdf'), ['right_inner_finger', 'left_inner_knuckle']),
                        (os.path.join(ROOT, 'tests', 'golden', 'urdf', 'cart_tree.urdf'), ['arm_b', 'tip'])):
        r = UrdfRobot(path)
        for nr in roots:
            rr = r.rerooted(nr)
            assert sorted(rr.links) == sorted(r.links) and len(rr.joints) == len(r.joints) and rr.num_dofs == r.num_dofs and rr.root == nr
            assert r.root != nr and UrdfRobot(path).root == r.root            # the original is untouched
            for _ in range(4):
                qs = {j.name: rng.uniform(-1, 1) if j.movable else 0.0 for j in r.joints}
                A, B = fk(r, qs), fk(rr, qs)
                ia, ib = A[nr].inverse(), B[nr].inverse()
                assert max(np.abs((ia * A[n]).matrix() - (ib * B[n]).matrix()).max() for n in A) < 1e-12
            lim = {j.name: (j.lower, j.upper, j.effort) for j in r.joints}
            assert {j.name: (j.lower, j.upper, j.effort) for j in rr.joints} == lim


def test_child_model_on_a_floating_parent_merges_inertia(tmp_path):
    """Two 10 kg spheres bolted one metre apart fall as ONE body of 20 kg with its centre of mass half way."""
    import yaml
    from diy_gym_amd.scene import K
    cfg = {'render': False,
           'a': {'model': 'sphere2.urdf', 'xyz': [0.0, 0.0, 5.0],
                 'state': {'addon': 'object_state_sensor', 'include_velocity': True},
                 'b': {'model': 'sphere2.urdf', 'xyz': [1.0, 0.0, 0.0]}}}
    path = tmp_path / 'pair.yaml'
    yaml.safe_dump(cfg, open(path, 'w'))
    env = DIYGym(str(path), num_envs=1, backend_factory=OracleBackend, engine=dict(linear_damping=0.0, angular_damping=0.0))
    L = env.layout
    bf = L.F[L.I[K.H_OFF_BODY_F]:L.I[K.H_OFF_BODY_F] + K.BF_STRIDE]
    assert L.n_bodies == 1 and abs(bf[K.BF_MASS] - 20.0) < 1e-12
    assert np.allclose(bf[K.BF_COM:K.BF_COM + 3], [0.5, 0.0, 0.0], atol=1e-12)
    # inertia about the common centre: 2 x (1 about own centre) + 2 x 10 kg x (0.5 m)^2 about y and z
    assert np.allclose([bf[K.BF_INERTIA], bf[K.BF_INERTIA + 3], bf[K.BF_INERTIA + 5]], [2.0, 7.0, 7.0], atol=1e-12)
    for _ in range(24):
        o, _, _, _ = env.step({})
    v = o['a']['state']['velocity'][0].numpy()
    assert abs(v[2] + 9.81 * 25 / 240.0) < 1e-6 and abs(v[0]) < 1e-9 and abs(v[1]) < 1e-9   # free fall (24 + 1 hot-start step), no spin-up


def test_three_finger_gripper_asset_loads_and_steps():
    import yaml, tempfile
    cfg = {'render': False, 'arm': {'model': 'ur5/ur5_3f.urdf', 'use_fixed_base': True,
                                    'joints': {'addon': 'joint_state_sensor'}}}
    with tempfile.TemporaryDirectory() as d:
        path = os.path.join(d, 'g3.yaml')
        yaml.safe_dump(cfg, open(path, 'w'))
        env = DIYGym(path, num_envs=1, backend_factory=OracleBackend)
        assert env.layout.n_links == 17
        for _ in range(10):
            o, _, _, _ = env.step({})
        assert torch.isfinite(o['arm']['joints']['position']).all()


def test_python_hook_addon_pushes_on_the_world_like_the_compiled_op():
    """The batched ``sim.apply_external_force`` API for user addons (reference addon.py:80-186 plugin contract, external_force.py:9-24):
    a plain Python hook addon equals the compiled ``external_force`` addon on the oracle backend (same arithmetic, fp64)."""
    import copy
    import yaml
    from diy_gym_amd.addons.addon import AddonFactory
    from diy_gym_amd.config import Configuration
    from user_addons import PyExternalForce
    AddonFactory.register_addon('py_external_force', PyExternalForce)
    tree = yaml.safe_load(open(os.path.join(ROOT, 'tests', 'golden', 'basic_env_nocam.yaml')))
    forces = [(r, k) for r, v in tree.items() if isinstance(v, dict) for k, a in v.items() if isinstance(a, dict) and a.get('addon') == 'external_force']
    assert forces
    hooked_tree = copy.deepcopy(tree)
    for r, k in forces:
        hooked_tree[r][k]['addon'] = 'py_external_force'
    a = DIYGym(Configuration.from_dict('basic_env', tree), num_envs=3, backend_factory=OracleBackend)
    b = DIYGym(Configuration.from_dict('basic_env', hooked_tree), num_envs=3, backend_factory=OracleBackend)
    assert len(b._hook_addons) == len(forces) and not a._hook_addons
    gen = torch.Generator().manual_seed(0)
    for _ in range(50):
        act = {r: {k: (torch.rand((3, 3), generator=gen) * 2 - 1) * 10.0} for r, k in forces}
        a.step(act); b.step(act)
    n = a.layout.addon_off
    assert np.allclose(a.sim.get_state()[:, :n], b.sim.get_state()[:, :n], rtol=0, atol=1e-12)
    assert np.abs(a.sim.get_state()[:, :n]).max() > 1.0


def test_wrench_on_an_attached_child_model_resolves_its_alias_uid():
    """A user addon on a CHILD model follows the README pattern -- ``self.uid = parent.uid`` and then
    ``sim.apply_external_force(self.uid, frame_id, ...)``.  A child attached with the default ``attach: merge`` owns no body:
    its uid is an alias (>= ALIAS_BASE) into the parent's body, its frames sit behind an offset there and its base is one of
    the parent's frames.  Asserted: the call is accepted and equals the same wrench given through the parent's body and the
    resolved frame, for a frame of the child and for its base (-1)."""
    cfg = os.path.join(ROOT, 'tests', 'golden', 'ur5_child_gripper.yaml')
    for child_frame in (-1, 2):
        a = DIYGym(cfg, num_envs=2, backend_factory=OracleBackend); b = DIYGym(cfg, num_envs=2, backend_factory=OracleBackend)
        grip = a.models['arm'].models['gripper']
        assert grip.uid >= a.builder.ALIAS_BASE
        body, frame = a.layout.resolve_frame(grip.uid, child_frame)
        assert body == a.models['arm'].uid and frame >= 0
        f = torch.tensor([[0.0, 3.0, -2.0], [1.0, 0.0, 0.5]])
        a.sim.apply_external_force(grip.uid, child_frame, f, [0.01, 0.0, 0.02], a.sim.LINK_FRAME)
        b.sim.apply_external_force(body, frame, f, [0.01, 0.0, 0.02], b.sim.LINK_FRAME)
        zero = torch.zeros((2, a.layout.act_dim))
        a.sim.step(0, zero); b.sim.step(0, zero)
        sa, sb = np.asarray(a.sim.get_state()), np.asarray(b.sim.get_state())
        assert np.array_equal(sa, sb)
        c = DIYGym(cfg, num_envs=2, backend_factory=OracleBackend); c.sim.step(0, zero)
        assert np.abs(sa - np.asarray(c.sim.get_state())).max() > 1e-6   # and it did push
