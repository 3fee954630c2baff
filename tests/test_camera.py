"""Camera addon (reference diy_gym/addons/sensors/camera.py:26-98): known answers on the oracle (CPU) and
HIP-vs-oracle parity (GPU).  The reference's own fixture basic_env.yaml is used unmodified."""
import os

import numpy as np
import pytest
import torch

from diy_gym_amd import DIYGym
from oracle_backend import OracleBackend

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BASIC = os.path.join(ROOT, 'tests', 'golden', 'basic_env.yaml')


def test_reference_fixture_loads_with_its_camera():
    env = DIYGym(BASIC, backend_factory=OracleBackend)
    # reference test_environment.py:18-21
    assert 'force' in env.action_space['blue_marble'].spaces
    assert 'camera' in env.observation_space['basic_env'].spaces
    assert 'pose' in env.observation_space['green_marble'].spaces
    cam = env.observation_space['basic_env']['camera']
    assert cam['rgb'].shape == (50, 50, 3) and cam['depth'].shape == (50, 50) and 'segmentation_mask' not in cam.spaces


def test_top_down_depth_known_answers():
    env = DIYGym(BASIC, backend_factory=OracleBackend)
    obs = env.reset()['basic_env']['camera']
    depth, rgb = obs['depth'], obs['rgb']
    # camera at z = 3 looking straight down (eye -z = world -z): the plane (top face z = 0) is at eye z = -3 ...
    assert depth.shape == (50, 50) and abs(depth.min() + 3.0) < 1e-5
    # ... and the highest point of a r = 0.5 marble resting on it (centre z = 0.5) is 2 m away
    assert abs(depth.max() + 2.0) < 2e-3
    # silhouettes: three discs of radius 0.5 at 3 - 0.5 = 2.5 m below a 70 degree camera over 50 px
    px_per_m = 25.0 / (np.tan(np.radians(35.0)) * 2.5)
    area = (depth > -2.99).sum() / 3.0
    assert abs(np.sqrt(area / np.pi) / px_per_m - 0.5) < 0.04
    assert rgb.shape == (50, 50, 3) and rgb.min() >= 0.0 and rgb.max() <= 1.0
    # marbles carry their configured colours (basic_env.yaml:17,29,41): red, green, blue blobs exist
    for ch in range(3):
        other = [c for c in range(3) if c != ch]
        assert ((rgb[..., ch] > 0.5) & (rgb[..., other[0]] < 0.1) & (rgb[..., other[1]] < 0.1)).sum() > 20


def test_segmentation_and_background(tmp_path):
    import yaml
    tree = yaml.safe_load(open(BASIC))
    tree['camera']['use_segmentation_mask'] = True
    tree['sky'] = {'addon': 'camera', 'xyz': [0, 0, 1], 'rpy': [3.14159265, 0, 0], 'resolution': [8, 8], 'clipping_boundaries': [0.1, 50]}
    cfg = tmp_path / 'seg.yaml'
    cfg.write_text(yaml.dump(tree))
    env = DIYGym(str(cfg), backend_factory=OracleBackend)
    obs = env.reset()['seg']
    seg = obs['camera']['segmentation_mask']
    uids = {env.models[n].uid for n in ('plane', 'red_marble', 'green_marble', 'blue_marble')}
    assert set(np.unique(seg)) == uids  # base links: uid + ((−1 + 1) << 24) = uid
    # a camera looking up sees nothing: depth = -far everywhere
    assert np.allclose(obs['sky']['depth'], -50.0)


def test_camera_follows_its_parent_frame(tmp_path):
    cfg = tmp_path / 'follow.yaml'
    cfg.write_text('plane: {model: grass/plane.urdf}\n'
                   'ball:\n  model: sphere2.urdf\n  xyz: [0, 0, 0.5]\n  scale: 0.2\n'
                   '  eye: {addon: camera, xyz: [0, 0, 2.0], resolution: [9, 9], use_segmentation_mask: yes}\n')
    env = DIYGym(str(cfg), backend_factory=OracleBackend)
    d0 = env.reset()['ball']['eye']['depth']
    # the camera rides 2 m above the ball's base frame looking down: centre pixel sees the ball's top (r = 0.1)
    st = env.sim.get_state()
    z = st[0, env.layout.body_state_off[1] + 2]
    assert abs(d0[4, 4] + (2.0 - 0.1)) < 1e-3 and abs(d0[0, 0] + (2.0 + z)) < 1e-3


@pytest.mark.gpu
def test_hip_render_matches_oracle():
    import yaml
    from diy_gym_amd.config import Configuration
    tree = yaml.safe_load(open(BASIC))
    tree['camera']['use_segmentation_mask'] = True
    tree['camera']['resolution'] = [64, 64]
    tree['green_marble']['eye'] = {'addon': 'camera', 'xyz': [0, -2.0, 0.5], 'rpy': [1.2, 0, 0], 'resolution': [40, 40],
                                   'use_segmentation_mask': True}
    B = 5
    gpu = DIYGym(Configuration.from_dict('basic_env', tree), num_envs=B, device='cuda:0', seed=2)
    import copy
    cpu = DIYGym(Configuration.from_dict('basic_env', copy.deepcopy(tree)), num_envs=B, seed=2, backend_factory=OracleBackend)  # same key order = same uids
    gen = torch.Generator().manual_seed(0)
    for _ in range(40):
        act = (torch.rand((B, 6), generator=gen) * 20 - 10)
        gpu.sim.step(gpu._all_slots, act.to('cuda:0')); cpu.sim.step(cpu._all_slots, act)
    gpu._tick += 1; cpu._tick += 1
    for rec, name in (('basic_env', 'camera'), ('green_marble', 'eye')):
        g = gpu.receptors[rec].addons[name].observe(); c = cpu.receptors[rec].addons[name].observe()
        dg, dc = g['depth'].cpu(), c['depth']
        sg, sc = g['segmentation_mask'].cpu(), c['segmentation_mask']
        same = sg == sc
        # silhouette-edge pixels may flip between fp32 and fp64; everything else agrees to 1e-3 m
        assert same.float().mean() > 0.995, name
        assert float((dg - dc).abs()[same].max()) < 1e-3, name
        assert float((g['rgb'].cpu() - c['rgb']).abs()[same].max()) < 1e-3, name
