"""Every engine parameter of diy_gym_amd/scene.py::DEFAULTS -- each one a pybullet default restated from RECOLLECTION or a
deliberate deviation of this build (the ledger) -- is LIVE: overriding it reaches the scene blob and changes what
the oracle computes on a scene where it matters.  This is what makes the day someone runs pybullet beside this build a
calibration of named parameters rather than an archaeology project; it pins no value.  The GPU twin
(tests/test_parity_gpu.py::test_every_engine_parameter_overridden_at_once) runs the kernels against the oracle with all of them
moved off their defaults."""
import os

import numpy as np
import pytest
import torch

from diy_gym_amd import DIYGym
from diy_gym_amd.scene import DEFAULTS, K
from oracle_backend import OracleBackend

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
G = os.path.join(ROOT, 'tests', 'golden')
SCENES = {'marbles': os.path.join(G, 'basic_env_nocam.yaml'), 'ur_ik': os.path.join(ROOT, 'examples', 'ur_high_5', 'ur_high_5.yaml'),
          'maze': os.path.join(ROOT, 'examples', 'r2d2_maze', 'r2d2_maze.yaml'), 'cart_tree': os.path.join(G, 'cart_tree.yaml'),
          'touching': os.path.join(G, 'ur_arms_touching.yaml'), 'pendulum': os.path.join(G, 'pendulum.yaml'),
          'drone': os.path.join(ROOT, 'examples', 'drone_pilot', 'drone_pilot.yaml')}
# parameter: (another value, scene on which it must show, steps, action scale, 'top' = the top of the action range every step)
OVERRIDES = {
    'residual_threshold': (1e-13, 'touching', 10, 0.3, False),   # (a contact-free arm's sweeps only confirm the motor guess: nothing to see on ur_ik)
    'contact_erp': (0.3, 'marbles', 40, 1.0, False),
    'limit_erp': (0.5, 'cart_tree', 3, 1.0, True),   # (acts once a joint is BEYOND its limit: the rollout below starts it there)
    'linear_slop': (2e-3, 'marbles', 40, 1.0, False),
    'linear_damping': (0.0, 'drone', 20, 1.0, False),
    'angular_damping': (0.0, 'drone', 20, 1.0, False),
    'max_coordinate_velocity': (2.0, 'maze', 10, 20.0, False),
    'default_motor_impulse': (1e-3, 'pendulum', 20, 1.0, False),   # (the velocity motor every joint gets at load: holds a pendulum released at 1 rad, or does not)
    'ik_iterations': (5, 'ur_ik', 3, 1.0, False),
    'ik_lambda_sq': (0.01, 'ur_ik', 3, 1.0, False),
    'ik_joint_damping': (0.5, 'ur_ik', 3, 1.0, False),   # (joint-space variant only: from_the_readme's Jaco; shown on the blob alone here)
    'ik_residual': (1e-2, 'ur_ik', 3, 1.0, False),
    'ik_max_angle': (1e-4, 'ur_ik', 3, 1.0, False),
    'ik_null_rest_gain': (0.1, 'ur_ik', 3, 1.0, False),
    'ik_null_limit_gain': (100.0, 'ur_ik', 3, 1.0, False),   # (acts on joints beyond their limits only: shown on the blob alone here)
    'contact_margin': (0.1, 'touching', 10, 0.3, False),
    'warmstart': (0.85, 'touching', 10, 0.3, False),
    'warmstart_friction': (0.5, 'touching', 10, 0.3, False),
    'motor_guess': (0.0, 'ur_ik', 5, 1.0, False),
    'limit_guess': (0.0, 'cart_tree', 60, 1.0, True),
    'motor_impulse_timebase': ('step', 'touching', 10, 0.3, False),
    'hull_contacts': (0.0, 'touching', 10, 0.3, False),   # (the capsule fitted to each hull instead of the hull)
    'hull_margin': (0.004, 'touching', 10, 0.3, False),
}
HF_SLOT = {'residual_threshold': 'HF_RESIDUAL_THRESHOLD', 'contact_erp': 'HF_CONTACT_ERP', 'limit_erp': 'HF_LIMIT_ERP', 'linear_slop': 'HF_LINEAR_SLOP',
           'linear_damping': 'HF_LIN_DAMPING', 'angular_damping': 'HF_ANG_DAMPING', 'max_coordinate_velocity': 'HF_MAX_COORD_VEL',
           'default_motor_impulse': 'HF_DEFAULT_MOTOR_IMPULSE', 'ik_lambda_sq': 'HF_IK_LAMBDA_SQ', 'ik_joint_damping': 'HF_IK_JOINT_DAMPING',
           'ik_residual': 'HF_IK_RESIDUAL', 'ik_max_angle': 'HF_IK_MAX_ANGLE', 'ik_null_rest_gain': 'HF_IK_NULL_REST_GAIN', 'ik_null_limit_gain': 'HF_IK_NULL_LIMIT_GAIN',
           'contact_margin': 'HF_CONTACT_MARGIN', 'warmstart': 'HF_WARMSTART', 'warmstart_friction': 'HF_WARMSTART_FRICTION', 'motor_guess': 'HF_MOTOR_GUESS',
           'limit_guess': 'HF_LIMIT_GUESS', 'motor_impulse_timebase': 'HF_MOTOR_IMPULSE_SCALE', 'hull_contacts': 'HF_HULL_CONTACTS',
           'hull_margin': 'HF_HULL_MARGIN'}
BLOB_ONLY = ('ik_joint_damping', 'ik_null_limit_gain')


def test_the_table_covers_every_parameter():
    assert set(OVERRIDES) == set(DEFAULTS)


def rollout(scene, steps, scale, top, **engine):
    import diy_gym_amd.examples  # noqa: F401
    from diy_gym_amd.utils import flatten, get_bounds_for_space
    env = DIYGym(SCENES[scene], num_envs=3, seed=5, backend_factory=OracleBackend, engine=engine)
    if scene == 'pendulum':
        st = np.array(env.sim.get_state()); st[:, env.layout.link_state_off[0]] = 1.0; env.sim.set_state(st)
    if scene == 'cart_tree' and steps <= 3:   # every limited joint 0.05 beyond its upper limit
        st = np.array(env.sim.get_state()); body = list(env.models).index('cart')
        for j in env.models['cart'].robot.joints:
            if j.q_index > -1 and j.lower <= j.upper:
                st[:, env.layout.link_state_off[env.layout.body_first_link[body] + j.q_index]] = j.upper + 0.05
        env.sim.set_state(st)
    if not env.layout.act_dim:   # (a scene without a controller addon)
        for _ in range(steps):
            env.sim.step(0)
        return env, np.asarray(env.sim.get_state(), dtype=np.float64).copy()
    lo = torch.as_tensor(flatten(get_bounds_for_space(env.action_space, True)), dtype=torch.float32)
    hi = torch.as_tensor(flatten(get_bounds_for_space(env.action_space, False)), dtype=torch.float32)
    gen = torch.Generator().manual_seed(3)
    for _ in range(steps):
        act = hi[None].repeat(3, 1) if top else (lo + (hi - lo) * torch.rand((3, lo.numel()), generator=gen)) * scale
        env.sim.step(env._all_slots, act)
    return env, np.asarray(env.sim.get_state(), dtype=np.float64).copy()


@pytest.mark.parametrize('name', sorted(OVERRIDES))
def test_engine_parameter_is_live(name):
    value, scene, steps, scale, top = OVERRIDES[name]
    base_env, base = rollout(scene, steps, scale, top)
    env, moved = rollout(scene, steps, scale, top, **{name: value})
    if name == 'ik_iterations':
        assert int(env.layout.I[K.H_IK_ITERS]) == value
    else:
        slot = getattr(K, HF_SLOT[name])
        assert env.layout.F[slot] != base_env.layout.F[slot]
    if name in BLOB_ONLY:
        return
    n = min(base.shape[1], moved.shape[1])   # (warmstart = 0 drops the impulse cache from the state)
    assert np.abs(base[:, :n] - moved[:, :n]).max() > 1e-9, name


def test_unknown_engine_parameter_is_refused():
    with pytest.raises(KeyError):
        DIYGym(SCENES['pendulum'], num_envs=1, backend_factory=OracleBackend, engine={'warm_start': 1.0})
