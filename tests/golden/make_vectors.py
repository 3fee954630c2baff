#!/usr/bin/env python3
"""Generates tests/golden/vectors.npz: seeded action sequences and the resulting observations / rewards /
terminals / states of every scene, computed by the fp64 CPU oracle.

Why the oracle and not the reference: the reference's arithmetic lives in the pybullet wheel, which cannot be
installed here, and the reference ships no numeric fixtures.  These vectors therefore pin (a) the
oracle against accidental change and (b) the HIP path against a committed answer that does not depend on the
oracle being rebuilt on the GPU box.  Re-run after any intended change of the algorithm:  python tests/golden/make_vectors.py
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))

SCENES = {  # name: (config, envs, steps, action scale)
    'marbles': ('tests/golden/basic_env_nocam.yaml', 3, 40, 1.0),
    'drone': ('examples/drone_pilot/drone_pilot.yaml', 3, 30, 1.0),
    'ur_ik': ('examples/ur_high_5/ur_high_5.yaml', 3, 30, 1.0),
    'ur_joint': ('examples/ur_high_5/ur_high_5_joint.yaml', 3, 30, 1.0),
    'cart_tree': ('tests/golden/cart_tree.yaml', 3, 12, 1.0),
    'maze': ('examples/r2d2_maze/r2d2_maze.yaml', 2, 12, 10.0),
    'admittance': ('tests/golden/ur_admittance.yaml', 3, 30, 1.0),
    'readme': ('examples/from_the_readme/from_the_readme.yaml', 2, 30, 0.2),   # R2D2 lands on the table: 25 contacts
    # crossed forearms driven THROUGH each other: contacts between two arms in every form of the sweeps, with the narrow phase that
    # is smooth in the poses (the capsule fitted to each hull; tests/test_parity_gpu.py SMOOTH_CONTACTS says why)
    'touching': ('tests/golden/ur_arms_touching.yaml', 3, 30, 0.3, dict(hull_contacts=0.0)),
    # the same arms holding that pose and turning one shoulder by 0.06 / 0.10 / 0.14 rad: the forearms meet at ~0.2 rad/s and stay
    # pressed against each other -- the hulls colliding as hulls (the default narrow phase), resting contact under load
    'pressing': ('tests/golden/ur_arms_touching.yaml', 3, 60, 'press'),
    'gripper': ('tests/golden/ur5_gripper.yaml', 3, 30, 0.5),                  # UR5 + two-finger gripper, 12-DoF tree
}


CROSSED = [1.35, -1.08, 1.03, -0.01, 0.09, 0.86]   # rest_position of ur_arms_touching.yaml


def press_actions(env, steps):
    act = torch.tensor(CROSSED * 2, dtype=torch.float32)[None].repeat(env.num_envs, 1)
    act[:, 0] += 0.06 + 0.08 * torch.arange(env.num_envs) / max(env.num_envs - 1, 1)
    return act[None].repeat(steps, 1, 1)


def actions_for(env, steps, scale, seed=123):
    if scale == 'press':
        return press_actions(env, steps)
    from diy_gym_amd.utils import flatten, get_bounds_for_space
    lo = torch.as_tensor(flatten(get_bounds_for_space(env.action_space, True)), dtype=torch.float32)
    hi = torch.as_tensor(flatten(get_bounds_for_space(env.action_space, False)), dtype=torch.float32)
    gen = torch.Generator().manual_seed(seed)
    return (lo + (hi - lo) * torch.rand((steps, env.num_envs, lo.numel()), generator=gen)) * scale


# Second set ("ref/<scene>/..."): the solver settings the reference is believed to run with (diy_gym/diy_gym.py:76-82 sets
# nothing but the iteration count and the substeps, so Bullet's own defaults apply [R]): motor rows started from ZERO and
# contact normal rows warm started with Bullet's factor 0.85 -- instead of this build's production defaults (motor_guess 1,
# warmstart 1).  Scenes with motors and / or contacts.
REFERENCE_SETTINGS = dict(motor_guess=0.0, warmstart=0.85)
REF_SCENES = ('ur_ik', 'ur_joint', 'cart_tree', 'maze', 'readme', 'touching', 'pressing', 'gripper', 'marbles')


def run(name, backend_factory=None, device=None, engine=None):
    import diy_gym_amd.examples  # noqa: F401
    from diy_gym_amd import DIYGym
    cfg, B, steps, scale = SCENES[name][:4]
    kw = dict(backend_factory=backend_factory) if backend_factory else dict(device=device)
    engine = dict(SCENES[name][4] if len(SCENES[name]) > 4 else {}, **(engine or {}))
    if engine:
        kw['engine'] = engine
    env = DIYGym(os.path.join(ROOT, cfg), num_envs=B, seed=11, **kw)
    acts = actions_for(env, steps, scale)
    obs = []
    for s in range(steps):
        env.sim.step(env._all_slots, acts[s].to(env.device))
        obs.append(env.sim.obs.detach().cpu().numpy().copy())
    return dict(actions=acts.numpy(), obs=np.stack(obs), rew=env.sim.rew.cpu().numpy(), term=env.sim.term.cpu().numpy(),
                state=np.asarray(env.sim.get_state(), dtype=np.float64))


def main():
    from oracle_backend import OracleBackend
    out = {}
    for name in SCENES:
        r = run(name, backend_factory=OracleBackend)
        for k, v in r.items():
            out['%s/%s' % (name, k)] = v.astype(np.float32) if v.dtype == np.float64 and k != 'state' else v
        print(name, 'obs', r['obs'].shape, 'state', r['state'].shape)
    for name in REF_SCENES:
        r = run(name, backend_factory=OracleBackend, engine=REFERENCE_SETTINGS)
        for k, v in r.items():
            if k != 'actions':   # (the same seeded actions as the first set)
                out['ref/%s/%s' % (name, k)] = v.astype(np.float32) if v.dtype == np.float64 and k != 'state' else v
        print('ref/' + name, 'obs', r['obs'].shape, 'max |obs - production settings| = %.3g' % float(np.abs(r['obs'] - out[name + '/obs']).max()) if r['obs'].size else '')
    np.savez_compressed(os.path.join(HERE, 'vectors.npz'), **out)
    print('wrote', os.path.join(HERE, 'vectors.npz'), os.path.getsize(os.path.join(HERE, 'vectors.npz')), 'bytes')


if __name__ == '__main__':
    main()
