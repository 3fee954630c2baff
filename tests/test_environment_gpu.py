"""The reference's own behavioural tests (diy_gym/tests/test_environment.py:12-40,
test_utils.py:13-20) on the HIP path, plus full-size properties that do not
need the oracle: determinism, shard == whole, finiteness at BASELINE sizes."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def phys_state(env):
    from test_parity_gpu import phys_state as f
    return f(env)
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BASIC = os.path.join(ROOT, 'tests', 'golden', 'basic_env_nocam.yaml')
UR = os.path.join(ROOT, 'examples', 'ur_high_5', 'ur_high_5.yaml')
DRONE = os.path.join(ROOT, 'examples', 'drone_pilot', 'drone_pilot.yaml')


def test_reference_episode_on_the_gpu():
    from diy_gym_amd import DIYGym
    env = DIYGym(BASIC)  # one env, numpy in / numpy out, like the reference
    for name in ('plane', 'red_marble', 'green_marble', 'blue_marble'):
        assert name in env.models
    assert 'force' in env.action_space['blue_marble'].spaces and 'pose' in env.observation_space['green_marble'].spaces
    observation = env.reset()
    initial_position = observation['green_marble']['pose']['position']
    for _ in range(500):
        observation, _, _, _ = env.step({'blue_marble': {'force': [0, -100, 0]}})
    final_position = observation['green_marble']['pose']['position']
    assert abs(np.linalg.norm(initial_position) - np.linalg.norm(final_position)) > 0.5
    reset_position = env.reset()['green_marble']['pose']['position']
    assert abs(np.linalg.norm(initial_position) - np.linalg.norm(reset_position)) < 0.05


def run(cfg, B, steps, base=0, total=None, seed=3):
    import diy_gym_amd.examples  # noqa: F401
    from diy_gym_amd import DIYGym
    from test_parity_gpu import action_bounds
    env = DIYGym(cfg, num_envs=B, device='cuda:0', seed=seed, env_index_base=base)
    lo, hi = action_bounds(env)
    gen = torch.Generator().manual_seed(9)
    total = total or B
    for _ in range(steps):
        act = lo + (hi - lo) * torch.rand((total, lo.numel()), generator=gen)
        env.sim.step(env._all_slots, act[base:base + B].to('cuda:0').contiguous())
        env.sim.reset(env.sim.term_flag)
    torch.cuda.synchronize()
    return env


def test_full_size_ur_high_5_is_deterministic_and_finite():
    a = run(UR, 16384, 20)
    b = run(UR, 16384, 20)
    assert torch.equal(a.sim.state, b.sim.state) and torch.equal(a.sim.obs, b.sim.obs)
    assert bool(torch.isfinite(a.sim.state).all()) and bool(torch.isfinite(a.sim.obs).all())
    # joints stay inside the URDF limits, quaternion-free fixed bases untouched
    q = a.sim.obs[:, 0:6]
    assert float(q.abs().max()) < 2 * np.pi


@pytest.mark.parametrize('cfg,B,steps', [
    (os.path.join(ROOT, 'examples', 'r2d2_maze', 'r2d2_maze.yaml'), 4096, 30),              # BASELINE config 2 at its size
    (os.path.join(ROOT, 'examples', 'from_the_readme', 'from_the_readme.yaml'), 1024, 30),   # config 5 (contacts by step ~15)
    (os.path.join(ROOT, 'examples', 'drone_pilot', 'drone_pilot.yaml'), 16384, 20),          # config 4, one GPU's shard
])
def test_full_size_configs_are_deterministic_and_finite(cfg, B, steps):
    # size-independent properties at the BASELINE sizes: bitwise repeatability and finite state / outputs
    a = run(cfg, B, steps)
    b = run(cfg, B, steps)
    assert torch.equal(a.sim.state, b.sim.state) and torch.equal(a.sim.obs, b.sim.obs)
    assert bool(torch.isfinite(a.sim.state[:, :B]).all()) and bool(torch.isfinite(a.sim.obs).all())


def test_shards_equal_whole_batch():
    whole = run(DRONE, 512, 15).sim.get_state()
    parts = [run(DRONE, 256, 15, base=b, total=512).sim.get_state() for b in (0, 256)]
    assert np.array_equal(np.concatenate(parts, axis=0), whole)


def test_a_shard_in_another_workspace_mode_agrees_with_the_whole_batch():
    """dg_world_create picks the workspace mode from the scene AND the batch (from_the_readme: one env per wavefront up to four
    wavefronts per CU, four envs per wavefront above), so a shard can run another solver path -- another summation order --
    than the whole batch it is cut from: bit-for-bit replay holds within one mode (test_shards_equal_whole_batch; pin it with
    DG_MAX_LANES), ACROSS modes the shard agrees to rounding.  25 steps of the settling scene: 2e-3 on the observations."""
    import diy_gym_amd.examples  # noqa: F401
    from diy_gym_amd import DIYGym
    from test_parity_gpu import action_bounds
    readme = os.path.join(ROOT, 'examples', 'from_the_readme', 'from_the_readme.yaml')
    whole = DIYGym(readme, num_envs=2048, device='cuda:0', seed=3)
    shard = DIYGym(readme, num_envs=512, device='cuda:0', seed=3, env_index_base=1024)
    assert whole.sim.lanes != shard.sim.lanes, (whole.sim.lanes, shard.sim.lanes)
    lo, hi = action_bounds(whole); gen = torch.Generator().manual_seed(9)
    for _ in range(25):
        act = lo + (hi - lo) * torch.rand((2048, lo.numel()), generator=gen)
        whole.sim.step(whole._all_slots, act.to('cuda:0')); shard.sim.step(shard._all_slots, act[1024:1536].to('cuda:0').contiguous())
    assert bool(torch.isfinite(whole.sim.obs).all())
    assert float((whole.sim.obs[1024:1536] - shard.sim.obs).abs().max()) < 2e-3


def test_ragged_batch_sizes():
    # batch sizes that do not fill the last wavefront, and a single env
    for B in (1, 63, 65, 130):
        env = run(UR, B, 3)
        assert env.sim.obs.shape == (B, 27) and bool(torch.isfinite(env.sim.obs).all())
        ref = run(UR, 130, 3).sim.obs[:B] if B != 130 else env.sim.obs
        # env i of a small batch == env i of a bigger one (same seed, same actions prefix is not guaranteed)
    a, b = run(UR, 63, 3, total=130), run(UR, 130, 3, total=130)
    assert torch.equal(a.sim.obs, b.sim.obs[:63])


def test_dict_action_path_matches_flat_path():
    from diy_gym_amd import DIYGym
    e1 = DIYGym(UR, num_envs=8, device='cuda:0')
    e2 = DIYGym(UR, num_envs=8, device='cuda:0')
    gen = torch.Generator().manual_seed(1)
    for _ in range(5):
        flat = (torch.rand((8, 12), generator=gen) * 0.02 - 0.01).to('cuda:0')
        e1.sim.step(e1._all_slots, flat)
        act = {'ur5_l': {'controller': {'linear': flat[:, 0:3], 'rotation': flat[:, 3:6]}},
               'ur5_r': {'controller': {'linear': flat[:, 6:9], 'rotation': flat[:, 9:12]}}}
        obs, rew, term, _ = e2.step(act)
    assert torch.equal(e1.sim.obs, e2.sim.obs)
    assert obs['ur5_l']['joint_state']['position'].shape == (8, 6) and term.dtype == torch.bool


def test_many_bodies_take_the_lds_row_path(tmp_path):
    """7 marbles = 42 DoF > 32: contact rows fall back from the register-resident dense form to per-body LDS blocks
    (two-sided rows, marble-marble contacts).  Checked against the oracle like everything else."""
    from diy_gym_amd import DIYGym
    from oracle_backend import OracleBackend
    lines = ['plane: {model: grass/plane.urdf}']
    for i in range(7):
        lines.append('m%d:\n  model: sphere2.urdf\n  scale: 0.3\n  xyz: [%g, %g, 0.16]\n  push: {addon: external_force}\n  respawn: {addon: respawn}'
                     % (i, 0.32 * (i % 4), 0.33 * (i // 4)))
    cfg = tmp_path / 'many.yaml'
    cfg.write_text('\n'.join(lines) + '\n')
    gpu = DIYGym(str(cfg), num_envs=11, device='cuda:0')
    cpu = DIYGym(str(cfg), num_envs=11, backend_factory=OracleBackend)
    gen = torch.Generator().manual_seed(4)
    for _ in range(25):
        act = torch.rand((11, 21), generator=gen) * 6 - 3
        gpu.sim.step(gpu._all_slots, act.to('cuda:0')); cpu.sim.step(cpu._all_slots, act)
    a, b = phys_state(gpu), phys_state(cpu)
    assert int(gpu.sim.enable_diagnostics().shape[0]) == 11
    assert np.abs(a - b).max() < 5e-3
    assert cpu.sim.contacts(0) >= 7


def test_many_bodies_with_a_large_contact_budget_run_from_the_global_workspace(tmp_path):
    """10 marbles = 60 DoF (not dense) with a 32-contact budget: the per-env scratch no longer fits LDS even at 16 envs
    per wavefront, so the generic per-body rows live in the [workgroup][slot][lane] device buffer (mode 0)."""
    from diy_gym_amd import DIYGym
    from oracle_backend import OracleBackend
    lines = ['max_contacts: 32', 'plane: {model: grass/plane.urdf}']
    for i in range(10):
        lines.append('m%d:\n  model: sphere2.urdf\n  scale: 0.3\n  xyz: [%g, %g, 0.16]\n  push: {addon: external_force}\n  respawn: {addon: respawn}'
                     % (i, 0.32 * (i % 4), 0.33 * (i // 4)))
    cfg = tmp_path / 'many10.yaml'
    cfg.write_text('\n'.join(lines) + '\n')
    gpu = DIYGym(str(cfg), num_envs=70, device='cuda:0')     # two workgroups, the second one ragged
    cpu = DIYGym(str(cfg), num_envs=70, backend_factory=OracleBackend)
    assert gpu.sim.lanes == 0
    gen = torch.Generator().manual_seed(4)
    for _ in range(20):
        act = torch.rand((70, 30), generator=gen) * 6 - 3
        gpu.sim.step(gpu._all_slots, act.to('cuda:0')); cpu.sim.step(cpu._all_slots, act)
    assert np.abs(phys_state(gpu) - phys_state(cpu)).max() < 5e-3
    assert cpu.sim.contacts(0) >= 10


def test_step_and_masked_reset_are_graph_capturable():
    """No entry point allocates or synchronises: a step + auto-reset pair can be captured once and replayed."""
    import diy_gym_amd.examples  # noqa: F401
    from diy_gym_amd import DIYGym
    envs = [DIYGym(DRONE, num_envs=256, device='cuda:0', seed=1) for _ in range(2)]
    act = torch.rand((256, 4), device='cuda:0')
    for e in envs:  # warm up outside capture (first call sets the motor table)
        e.sim.step(e._all_slots, act); e.sim.reset(e.sim.term_flag)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        with torch.cuda.graph(g, stream=s):
            envs[0].sim.step(envs[0]._all_slots, act)
            envs[0].sim.reset(envs[0].sim.term_flag)
    for _ in range(20):
        g.replay()
        envs[1].sim.step(envs[1]._all_slots, act); envs[1].sim.reset(envs[1].sim.term_flag)
    torch.cuda.synchronize()
    assert torch.equal(envs[0].sim.state, envs[1].sim.state) and torch.equal(envs[0].sim.obs, envs[1].sim.obs)


FLAT_KEYS = 'flatten_actions: yes\nflatten_observations: yes\nsum_rewards: yes\nauto_reset: yes\n'


def _flat_config(tmp_path, name, episode_steps):
    """The reference's YAML plus the trainer-facing keys of diy_gym.py:94-96,114-122 (and the auto_reset extension)."""
    src = {'drone': DRONE, 'ur': os.path.join(ROOT, 'examples', 'ur_high_5', 'ur_high_5_joint.yaml')}[name]
    text = open(src).read().replace('max_episode_steps: 10000', '')
    cfg = tmp_path / (name + '_flat.yaml')
    cfg.write_text(FLAT_KEYS + 'max_episode_steps: %d\n' % episode_steps + text)
    return str(cfg)


@pytest.mark.parametrize('name,tol', [('drone', 2e-3), ('ur', 5e-4)])
def test_flat_env_step_with_timer_auto_reset_matches_oracle_over_three_episodes(tmp_path, name, tol):
    """N3 (reference diy_gym.py:114-122,180-183): env.step() itself -- flat torch tensor in, (obs [B, O], reward [B],
    terminal [B], {}) out -- with flatten_actions + flatten_observations + sum_rewards + terminal_if_any +
    max_episode_steps; the episode timer fires every 7 steps and the masked auto-reset restarts those envs (new
    respawn jitter per episode: the drone's target moves).  Same calls on the HIP backend and on the oracle backend.
    Blind to: everything both share (Bullet constants from recollection, no warm starting)."""
    import diy_gym_amd.examples  # noqa: F401
    from diy_gym_amd import DIYGym
    from oracle_backend import OracleBackend
    cfg = _flat_config(tmp_path, name, 7)
    B = 70
    gpu = DIYGym(cfg, num_envs=B, device='cuda:0', seed=21)
    cpu = DIYGym(cfg, num_envs=B, seed=21, backend_factory=OracleBackend)
    assert gpu._flat_fast and gpu.auto_reset
    lo, hi = torch.as_tensor(gpu.action_space.low), torch.as_tensor(gpu.action_space.high)
    A = lo.numel()
    assert gpu.observation_space.shape == (gpu.layout.obs_dim,) and A == gpu.layout.act_dim
    gen = torch.Generator().manual_seed(2)
    fired, first_obs = 0, gpu.observe().clone()
    ptrs = None
    for step in range(1, 24):
        act = lo + (hi - lo) * torch.rand((B, A), generator=gen)
        act_dev = act.to('cuda:0')
        allocs = torch.cuda.memory_stats()['allocation.all.allocated']
        g_obs, g_rew, g_term, g_info = gpu.step(act_dev)
        assert torch.cuda.memory_stats()['allocation.all.allocated'] == allocs   # the step (and its auto-reset) made no device allocation
        c_obs, c_rew, c_term, _ = cpu.step(act)
        # zero-copy: the returned tensors ARE the backend's persistent buffers, and a step allocates nothing
        assert g_obs.data_ptr() == gpu.sim.obs.data_ptr() and g_rew.data_ptr() == gpu.sim.rew_sum.data_ptr()
        assert g_term.data_ptr() == gpu.sim.term_flag.data_ptr() and g_term.dtype == torch.bool and g_info == {}
        assert g_obs.shape == (B, gpu.layout.obs_dim) and g_rew.shape == (B,) and g_term.shape == (B,)
        if step == 1:
            ptrs = (g_obs.data_ptr(), g_rew.data_ptr(), g_term.data_ptr())
        assert (g_obs.data_ptr(), g_rew.data_ptr(), g_term.data_ptr()) == ptrs
        assert torch.equal(g_term.cpu(), c_term), step
        assert float((g_obs.cpu() - c_obs).abs().max()) < tol and float((g_rew.cpu() - c_rew).abs().max()) < tol, step
        if step % 7 == 0:   # the timer ended every env's episode; obs rows are the new episode's first observation
            assert bool(g_term.all())
            fired += 1
            assert torch.equal(gpu.sim.state[0, :B].cpu(), torch.zeros(B))           # step counters restarted
            assert torch.equal(gpu.sim.state[1, :B].cpu(), torch.full((B,), 1.0 + fired))   # episode counters
        else:
            assert not bool(g_term.all())
    assert fired == 3
    if name == 'drone':   # the target respawned somewhere else in every episode (jitter keyed by episode)
        assert float((gpu.observe() - first_obs).abs().max()) > 0.5


def test_flat_env_step_rejects_a_wrong_width():
    from diy_gym_amd import DIYGym
    from diy_gym_amd.config import Configuration
    conf = Configuration.from_file(UR)
    conf.set('flatten_actions', True)
    env = DIYGym(conf, num_envs=8, device='cuda:0')
    with pytest.raises(ValueError):
        env.step(torch.zeros((8, 11), device='cuda:0'))
    with pytest.raises(ValueError):
        env.sim.step(env._all_slots, torch.zeros((8, 12)))   # CPU tensor
    env.step(torch.zeros((8, 12), device='cuda:0'))


def test_unindexed_device_string_is_accepted():
    """DIYGym(..., device='cuda') -- no index: tensors allocated there report 'cuda:0', and device('cuda') != device('cuda:0'),
    so the backend keeps the INDEXED device.  The flat step path, reset(mask) without a copy and render() all take
    caller tensors through the device check."""
    import yaml
    from diy_gym_amd import DIYGym
    from diy_gym_amd.config import Configuration
    tree = yaml.safe_load(open(os.path.join(ROOT, 'tests', 'golden', 'basic_env.yaml')))
    tree.update(flatten_actions=True, flatten_observations=True, sum_rewards=True, terminal_if_any=True)
    env = DIYGym(Configuration.from_dict('basic_env', tree), num_envs=4, device='cuda')
    assert env.device == torch.device('cuda', torch.cuda.current_device()) and env.sim.obs.device == env.device
    act = torch.zeros((4, env.layout.act_dim), device='cuda')
    obs, rew, term, _ = env.step(act)
    assert obs.device == env.device
    mask = torch.tensor([1, 0, 1, 0], dtype=torch.uint8, device='cuda')
    before = mask.data_ptr()
    env.reset(mask)
    assert mask.data_ptr() == before
    cam = [a for r in env.receptors.values() for a in r.addons.values() if hasattr(a, 'camera_index')][0]
    out = cam.observe()
    assert out['depth'].shape[0] == 4 and bool(torch.isfinite(out['depth']).all())


@pytest.mark.parametrize('max_lanes', [None, '32', '8'])
def test_python_hook_addon_equals_the_compiled_propellor(max_lanes):
    """A user addon that acts on the world from Python: the reference's Propellor (examples/drone_pilot/drone_pilot.py:10-40;
    registry diy_gym/addons/addon.py:80-81, hooks :91-186) written as a plain hook addon on ``env.sim.apply_external_*``
    (dg_world_apply_wrench) against the compiled DG_OP_PROPELLOR, same dict actions, 40 steps incl. terminal resets.
    Rotor speeds and the drone's whole state: bit-identical with the one-launch form (the compiled ops and the entry point
    run the same device function with contraction off; the spool-up filter is two separately rounded fp32 operations on
    both sides).  With pybullet's two separate calls per rotor the base torque is summed in another order: 1e-5 relative."""
    import yaml
    import diy_gym_amd.examples  # noqa: F401
    from diy_gym_amd import DIYGym
    from diy_gym_amd.addons.addon import AddonFactory
    from diy_gym_amd.config import Configuration
    from user_addons import PyPropellor, PyPropellorTwoCalls
    AddonFactory.register_addon('py_propellor', PyPropellor)
    AddonFactory.register_addon('py_propellor2', PyPropellorTwoCalls)
    tree = yaml.safe_load(open(DRONE))
    motors = sorted(k for k, v in tree['drone'].items() if isinstance(v, dict) and v.get('addon') == 'propellor')
    assert len(motors) == 4
    B = 257

    def make(addon_name):
        t = yaml.safe_load(open(DRONE))
        for m in motors:
            t['drone'][m]['addon'] = addon_name
        return DIYGym(Configuration.from_dict('drone_pilot', t), num_envs=B, device='cuda:0', seed=4)

    if max_lanes:   # (the wrench entry point in the narrow workspace modes -- drone_pilot takes 32 envs per wavefront at 16 384 envs)
        os.environ['DG_MAX_LANES'] = max_lanes
    try:
        compiled, hooked, hooked2 = make('propellor'), make('py_propellor'), make('py_propellor2')
    finally:
        os.environ.pop('DG_MAX_LANES', None)
    assert max_lanes is None or compiled.sim.lanes == int(max_lanes) == hooked.sim.lanes
    assert not compiled._hook_addons and len(hooked._hook_addons) == 4 and compiled.layout.addon_off == hooked.layout.addon_off
    L = compiled.layout
    so = L.body_state_off[[i for i in range(L.n_bodies) if not L.body_fixed[i]][0]]   # the drone (the target's respawn jitter is keyed by op index)
    gen = torch.Generator().manual_seed(1)
    for step in range(40):
        act = {'drone': {m: torch.rand((B, 1), generator=gen).to('cuda:0') for m in motors}}
        oc, _, tc, _ = compiled.step(act); oh, _, th, _ = hooked.step(act); o2, _, _, _ = hooked2.step(act)
        for m in motors:
            assert torch.equal(oc['drone'][m], oh['drone'][m]), (step, m)   # rotor speeds
        a = compiled.sim.state[so:so + 13, :B]
        assert torch.equal(a, hooked.sim.state[so:so + 13, :B]), step                # one launch per rotor: the same bits
        b = hooked2.sim.state[so:so + 13, :B]                                         # pybullet's two calls: another summation order
        assert float(((a - b).abs() / (1.0 + a.abs())).max()) < 1e-5, step
        done = compiled.sim.term_flag.clone()
        assert torch.equal(done, hooked.sim.term_flag)
        compiled.reset(done); hooked.reset(done); hooked2.reset(done)
    assert float(compiled.sim.state[so + 2, :B].min()) < 0.45   # they flew / fell: the forces did something
