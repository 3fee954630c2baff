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


@pytest.mark.gpu
def test_gripper_camera_at_the_baseline_batch_1024_envs_200x200():
    """BASELINE config 5 at its size: the 200 x 200 gripper camera of from_the_readme over 1 024 envs (655 MB of rgb +
    depth per render).  Size-independent properties: bitwise repeatability, finiteness, depth in [-far, -near],
    segmentation ids in range, rgb in [0, 1]; a sample of envs agrees with the oracle's renderer; odd image sizes
    (rows that do not fill cache lines, a last band that is cut short) give the same picture as the oracle too."""
    import diy_gym_amd.examples  # noqa: F401
    import yaml
    from diy_gym_amd.config import Configuration
    cfg = os.path.join(ROOT, 'examples', 'from_the_readme', 'from_the_readme.yaml')
    tree = yaml.safe_load(open(cfg))
    tree['r2d2']['arm_camera']['use_segmentation_mask'] = True
    # a second camera that actually looks at the scene (the gripper camera mostly sees the sky once R2D2 has settled)
    tree['overview'] = {'addon': 'camera', 'xyz': [1.2, -0.9, 1.4], 'rpy': [0.9, 0.0, 0.9], 'resolution': [200, 200], 'use_segmentation_mask': True}
    B = 1024
    env = DIYGym(Configuration.from_dict('from_the_readme', tree), num_envs=B, device='cuda:0', seed=4)
    lo = torch.full((B, env.layout.act_dim), -0.01, device='cuda:0')
    gen = torch.Generator(device='cuda:0').manual_seed(0)
    for _ in range(12):
        env.sim.step(env._all_slots, lo + 0.02 * torch.rand(lo.shape, generator=gen, device='cuda:0'))
    n_bodies = env.layout.n_bodies
    for rec, name in (('r2d2', 'arm_camera'), ('from_the_readme', 'overview')):
        cam = env.receptors[rec].addons[name]
        env._tick += 1
        first = {k: v.clone() for k, v in cam.observe().items()}
        env._tick += 1
        again = cam.observe()
        for k in first:
            assert torch.equal(first[k], again[k]), (name, k)   # bitwise repeatable
        assert first['rgb'].shape == (B, 200, 200, 3) and first['depth'].shape == (B, 200, 200)
        assert bool(torch.isfinite(first['rgb']).all()) and bool(torch.isfinite(first['depth']).all())
        assert float(first['depth'].max()) <= -0.01 and float(first['depth'].min()) >= -100.0
        assert float(first['rgb'].min()) >= 0.0 and float(first['rgb'].max()) <= 1.0
        seg = first['segmentation_mask']
        assert int(seg.min()) >= -1 and int((seg[seg >= 0] & 0xFFFFFF).max() if (seg >= 0).any() else 0) < n_bodies
    # the cone culling must not change a single pixel: render again with every shape tested for every pixel group
    try:
        for rec, name in (('r2d2', 'arm_camera'), ('from_the_readme', 'overview')):
            cam = env.receptors[rec].addons[name]
            culled = {k: v.clone() for k, v in cam.observe().items()}
            env._tick += 1
            env.sim.set_render_diag(1)
            brute = cam.observe()
            env.sim.set_render_diag(0); env._tick += 1
            for k in culled:
                assert torch.equal(culled[k], brute[k]), (name, k)
    finally:
        env.sim.set_render_diag(0)
    env._tick += 1
    seen = (env.receptors['from_the_readme'].addons['overview'].observe()['segmentation_mask'] >= 0).float().mean()
    assert float(seen) > 0.3   # the overview camera does see the table, the arm and R2D2
    # parity on a sample: the same state in the oracle (state copied over), envs 0, 511, 1023
    pick = [0, 511, 1023]
    import copy
    cpu = DIYGym(Configuration.from_dict('from_the_readme', copy.deepcopy(tree)), num_envs=len(pick), seed=4, backend_factory=OracleBackend)   # same key order = same uids
    cpu.sim.set_state(env.sim.get_state()[pick])
    cpu._tick += 1
    for rec, name in (('r2d2', 'arm_camera'), ('from_the_readme', 'overview')):
        g = env.receptors[rec].addons[name].observe(); c = cpu.receptors[rec].addons[name].observe()
        sg, sc = g['segmentation_mask'][pick].cpu(), c['segmentation_mask']
        same = sg == sc
        # the gripper camera sits a centimetre from R2D2's own fingers (surfaces at the near plane): fp32 / fp64 flips there
        assert same.float().mean() > (0.97 if name == 'arm_camera' else 0.995), name
        assert float((g['depth'][pick].cpu() - c['depth']).abs()[same].max()) < 2e-3, name
        fg = sc >= 0   # (98 % of the gripper camera's picture is sky: the overall fraction above is blind to a missing foreground)
        if int(fg.sum()) > 100:   # (none at all in this state for the gripper camera; tests/test_parity_gpu.py holds the case where it sees the arm's hand)
            assert same[fg].float().mean() > 0.9, name


@pytest.mark.gpu
def test_more_shapes_in_view_than_the_render_list_holds():
    """r2d2_maze seen from above: 119 walls + R2D2's shapes + the ground pass the picture's cone -- more than the 96 entries of the
    render kernel's per-band list, so the band is rendered by the general path (every pixel against every shape, from the
    tables).  Same picture as the oracle's renderer; a camera that sees only a corner of the maze (the list holds what it sees)
    agrees with it too."""
    import copy
    import yaml
    from diy_gym_amd.config import Configuration
    tree = yaml.safe_load(open(os.path.join(ROOT, 'examples', 'r2d2_maze', 'r2d2_maze.yaml')))
    tree['above'] = {'addon': 'camera', 'xyz': [0.0, 0.0, 14.0], 'rpy': [0.0, 0.0, 0.0], 'resolution': [72, 60], 'use_segmentation_mask': True}
    tree['corner'] = {'addon': 'camera', 'xyz': [3.0, 3.0, 2.5], 'rpy': [0.0, 0.0, 0.3], 'resolution': [72, 60], 'use_segmentation_mask': True}
    B = 3
    gpu = DIYGym(Configuration.from_dict('r2d2_maze', tree), num_envs=B, device='cuda:0', seed=2)
    cpu = DIYGym(Configuration.from_dict('r2d2_maze', copy.deepcopy(tree)), num_envs=B, seed=2, backend_factory=OracleBackend)
    assert gpu.layout.n_bodies > 120
    gpu.sim.set_render_diag(512)    # one band per picture, as at the batch sizes that matter (a band of 8 rows sees fewer shapes)
    gpu._tick += 1; cpu._tick += 1
    for name in ('above', 'corner'):
        g = gpu.receptors['r2d2_maze'].addons[name].observe(); c = cpu.receptors['r2d2_maze'].addons[name].observe()
        sg, sc = g['segmentation_mask'].cpu(), c['segmentation_mask']
        same = sg == sc
        assert same.float().mean() > 0.99, (name, float(same.float().mean()))
        assert float((g['depth'].cpu() - c['depth']).abs()[same].max()) < 2e-3, name
        assert float((g['rgb'].cpu() - c['rgb']).abs()[same].max()) < 2e-3, name
        if name == 'above':
            assert len(torch.unique(sc)) > 3 and float((c['depth'] > -13.9).float().mean()) > 0.05     # walls (and R2D2) are in the picture, not just the ground


@pytest.mark.gpu
@pytest.mark.parametrize('res', [[50, 50], [33, 33], [7, 7], [130, 130]])
def test_hip_render_odd_sizes_match_oracle(res):
    # rows that are not a whole number of cache lines, bands cut short by the image edge, images smaller than a pixel group
    import copy
    import yaml
    from diy_gym_amd.config import Configuration
    tree = yaml.safe_load(open(BASIC))
    tree['camera']['use_segmentation_mask'] = True
    tree['camera']['resolution'] = res
    gpu = DIYGym(Configuration.from_dict('basic_env', tree), num_envs=3, device='cuda:0', seed=2)
    cpu = DIYGym(Configuration.from_dict('basic_env', copy.deepcopy(tree)), num_envs=3, seed=2, backend_factory=OracleBackend)
    g = gpu.receptors['basic_env'].addons['camera'].observe(); c = cpu.receptors['basic_env'].addons['camera'].observe()
    same = g['segmentation_mask'].cpu() == c['segmentation_mask']
    assert same.float().mean() > 0.99
    assert float((g['depth'].cpu() - c['depth']).abs()[same].max()) < 1e-3


def test_visual_randomizer_retextures_the_model_per_env_and_episode(tmp_path):
    """visual_randomizer (reference visual_randomizer.py:14-46) with procedural textures instead of its image data set: two
    colours, a frequency and a pattern per env and episode (DG_TX_* / DG_TEX_*), seen by the camera's rgb only.  Every
    pixel of the marble is a blend of the two drawn colours times the shading factor; depth is untouched; a masked reset
    re-draws for the reset envs only."""
    import yaml
    tree = yaml.safe_load(open(BASIC))
    tree['red_marble']['look'] = {'addon': 'visual_randomizer'}
    from diy_gym_amd.config import Configuration
    env = DIYGym(Configuration.from_dict('basic_env', tree), num_envs=4, seed=3, backend_factory=OracleBackend)
    look = env.models['red_marble'].addons['look']
    t0 = look.textures().clone()
    assert t0.shape == (4, 8) and float(t0[:, :6].min()) >= 0.0 and float(t0[:, :6].max()) <= 1.0 and len({tuple(r.tolist()) for r in t0}) == 4
    assert float(t0[:, 6].min()) >= 2.0 and float(t0[:, 6].max()) <= 16.0 and set(t0[:, 7].tolist()) <= {1.0, 2.0, 3.0}
    obs = env.observe()['basic_env']['camera']
    rgb, depth = np.asarray(obs['rgb']), np.asarray(obs['depth'])
    ref = DIYGym(BASIC, num_envs=4, seed=3, backend_factory=OracleBackend).observe()['basic_env']['camera']
    assert np.array_equal(depth, np.asarray(ref['depth']))                      # geometry untouched
    changed = np.abs(rgb - np.asarray(ref['rgb'])).max(-1) > 1e-6
    assert changed.any() and changed.mean() < 0.2                                 # only the marble's pixels
    for e in range(4):   # a marble pixel = (A + (B - A) t) x shade with t in [0, 1]: it lies on the segment between the two colours, scaled
        A, Bc = t0[e, 0:3].numpy().astype(np.float64), t0[e, 3:6].numpy().astype(np.float64)
        px = rgb[e][changed[e]].astype(np.float64)
        # solve px = a A + b B (least squares over the three channels): both weights non-negative, residual ~ 0
        M = np.stack([A, Bc], 1); w, *_ = np.linalg.lstsq(M, px.T, rcond=None)
        assert np.abs(M @ w - px.T).max() < 1e-4 and w.min() > -1e-4
        if int(t0[e, 7]) in (1, 2):   # checker / stripes: every pixel is pure A or pure B
            assert (np.minimum(np.abs(w[0]), np.abs(w[1])) < 1e-4).all()
    env.reset(torch.tensor([1, 0, 0, 1], dtype=torch.uint8))
    t1 = look.textures()
    assert [bool((t1[i] != t0[i]).any()) for i in range(4)] == [True, False, False, True]


def test_link_materials_and_the_yaml_colour_reach_the_camera():
    """N4, first slice: the colour of a pixel is the URDF <material><color> of the link that was hit, except that the YAML
    ``color`` key (reference model.py:82-83, p.changeVisualShape(uid, -1, rgbaColor)) overrides the BASE link's.  Scene:
    R2D2 (blue body, white legs and head, black wheels: its URDF materials) on the grass plane, a fixed camera."""
    from diy_gym_amd.config import Configuration
    tree = {'plane': {'model': 'grass/plane.urdf', 'color': [0.2, 0.6, 0.2, 1.0]},
            'r2d2': {'model': 'r2d2.urdf', 'xyz': [0, 0, 0.47]},
            'camera': {'addon': 'camera', 'xyz': [1.8, 0.0, 0.5], 'rpy': [1.5708, 0.0, 1.5708], 'resolution': [64, 64], 'use_segmentation_mask': True}}   # looks along -x, z up
    env = DIYGym(Configuration.from_dict('look', tree), num_envs=1, backend_factory=OracleBackend)
    o = env.observe()['look']['camera']
    rgb, seg = np.asarray(o['rgb'])[0], np.asarray(o['segmentation_mask'])[0]
    uid = {k: m.uid for k, m in env.models.items()}
    def hue(mask):   # colour with the shading divided out (largest channel = 1)
        px = rgb[mask].reshape(-1, 3).astype(np.float64); px = px[px.max(1) > 1e-3]
        return np.unique(np.round(px / px.max(1, keepdims=True), 2), axis=0)
    plane = hue((seg & 0xFFFFFF) == uid['plane']) if (seg >= 0).any() else None
    assert plane is not None and len(plane) == 1 and np.allclose(plane[0], [1 / 3, 1.0, 1 / 3], atol=0.02)       # the YAML colour, not the file's white
    r2 = (seg >= 0) & ((seg & 0xFFFFFF) == uid['r2d2'])
    assert r2.mean() > 0.02
    body = rgb[r2 & ((seg >> 24) == 0)]; assert len(body) and np.allclose(body / body.max(1, keepdims=True), [0.0, 0.0, 1.0], atol=0.02)   # base link: blue
    others = rgb[r2 & ((seg >> 24) > 0)].reshape(-1, 3)
    white = others[(others.min(1) > 0.2)]; black = others[(others.max(1) < 1e-3)]
    assert len(white) > 0 and np.allclose(white / white.max(1, keepdims=True), 1.0, atol=0.02) and len(black) > 0                              # legs / head white, wheels black


@pytest.mark.gpu
def test_visual_randomizer_hip_matches_oracle(tmp_path):
    import copy
    import yaml
    from diy_gym_amd.config import Configuration
    tree = yaml.safe_load(open(BASIC))
    tree['red_marble']['look'] = {'addon': 'visual_randomizer'}
    tree['plane']['look'] = {'addon': 'visual_randomizer'}
    gpu = DIYGym(Configuration.from_dict('basic_env', tree), num_envs=6, device='cuda:0', seed=2)
    cpu = DIYGym(Configuration.from_dict('basic_env', copy.deepcopy(tree)), num_envs=6, seed=2, backend_factory=OracleBackend)
    for _ in range(2):
        g = gpu.receptors['basic_env'].addons['camera'].observe(); c = cpu.receptors['basic_env'].addons['camera'].observe()
        assert np.allclose(gpu.models['plane'].addons['look'].textures().numpy(), cpu.models['plane'].addons['look'].textures().numpy(), rtol=1e-6, atol=1e-6)
        same = (g['depth'].cpu() - c['depth']).abs() < 1e-3
        # (a pixel on the border of a texture cell may fall on the other side in fp32: the picture agrees in > 98 % of the pixels)
        rgb_same = (g['rgb'].cpu() - c['rgb']).abs().max(-1).values < 2e-3
        assert same.float().mean() > 0.99 and float((rgb_same & same).float().mean()) > 0.98
        mask = torch.tensor([1, 0, 1, 0, 1, 1], dtype=torch.uint8)
        gpu.reset(mask.to('cuda:0')); cpu.reset(mask)


@pytest.mark.gpu
def test_link_materials_hip_matches_oracle():
    """The rgb output of the scene of test_link_materials_and_the_yaml_colour_reach_the_camera (URDF materials per link, the
    YAML colour on the base link) and of basic_env.yaml (the reference's own fixture: three coloured marbles on the plane),
    HIP against the oracle: the same colour wherever the same surface is seen."""
    import copy
    import yaml
    from diy_gym_amd.config import Configuration
    scenes = [{'plane': {'model': 'grass/plane.urdf', 'color': [0.2, 0.6, 0.2, 1.0]}, 'r2d2': {'model': 'r2d2.urdf', 'xyz': [0, 0, 0.47]},
               'camera': {'addon': 'camera', 'xyz': [1.8, 0.0, 0.5], 'rpy': [1.5708, 0.0, 1.5708], 'resolution': [64, 64], 'use_segmentation_mask': True}},
              yaml.safe_load(open(BASIC))]
    for tree in scenes:
        gpu = DIYGym(Configuration.from_dict('look', copy.deepcopy(tree)), num_envs=3, device='cuda:0', seed=2)
        cpu = DIYGym(Configuration.from_dict('look', copy.deepcopy(tree)), num_envs=3, seed=2, backend_factory=OracleBackend)
        g = gpu.receptors['look'].addons['camera'].observe(); c = cpu.receptors['look'].addons['camera'].observe()
        same = (g['depth'].cpu() - c['depth']).abs() < 1e-3
        assert same.float().mean() > 0.99
        assert float((g['rgb'].cpu() - c['rgb']).abs()[same].max()) < 2e-3
        assert len(torch.unique((c['rgb'] * 50).round(), dim=0)) >= 1 and float(c['rgb'].std()) > 0.05   # a picture with several colours in it
