"""GPU parity tests proper: the HIP path (through the C-ABI) against the fp64 CPU
oracle on the same seeded inputs.  fp32 tolerance is stated per test."""
import os

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu

CONFIGS = {
    'marbles': os.path.join(ROOT, 'tests', 'golden', 'basic_env_nocam.yaml'),
    'drone': os.path.join(ROOT, 'examples', 'drone_pilot', 'drone_pilot.yaml'),
    'ur_ik': os.path.join(ROOT, 'examples', 'ur_high_5', 'ur_high_5.yaml'),
    'ur_joint': os.path.join(ROOT, 'examples', 'ur_high_5', 'ur_high_5_joint.yaml'),
    'cart_tree': os.path.join(ROOT, 'tests', 'golden', 'cart_tree.yaml'),
}


def make_pair(name, B, seed=5):
    import diy_gym_amd.examples  # noqa: F401  registers propellor / fell_over
    from diy_gym_amd import DIYGym
    from oracle_backend import OracleBackend
    gpu = DIYGym(CONFIGS[name], num_envs=B, device='cuda:0', seed=seed)
    cpu = DIYGym(CONFIGS[name], num_envs=B, seed=seed, backend_factory=OracleBackend)
    return gpu, cpu


def action_bounds(env):
    from diy_gym_amd.utils import flatten, get_bounds_for_space
    lo = flatten(get_bounds_for_space(env.action_space, True))
    hi = flatten(get_bounds_for_space(env.action_space, False))
    return torch.as_tensor(lo, dtype=torch.float32), torch.as_tensor(hi, dtype=torch.float32)


def rollout(gpu, cpu, steps, scale=1.0, seed=0):
    gen = torch.Generator().manual_seed(seed)
    lo, hi = action_bounds(gpu)
    B = gpu.num_envs
    worst = dict(obs=0.0, rew=0.0, state=0.0, term_mismatch=0)
    for _ in range(steps):
        act = (lo + (hi - lo) * torch.rand((B, lo.numel()), generator=gen)) * scale
        gpu.sim.step(gpu._all_slots, act.to(gpu.device))
        cpu.sim.step(cpu._all_slots, act)
        worst['obs'] = max(worst['obs'], float((gpu.sim.obs.cpu() - cpu.sim.obs).abs().max()))
        worst['rew'] = max(worst['rew'], float((gpu.sim.rew.cpu() - cpu.sim.rew).abs().max()))
        worst['term_mismatch'] += int((gpu.sim.term.cpu() != cpu.sim.term).sum())
    worst['state'] = float(np.abs(gpu.sim.get_state() - cpu.sim.get_state()).max())
    return worst


def test_initial_state_and_reset_match():
    for name in CONFIGS:
        gpu, cpu = make_pair(name, 5)
        # after the constructor's reset (respawn + rest joints + 1 hot-start step); efforts are O(100 N m)
        assert np.allclose(gpu.sim.get_state(), cpu.sim.get_state(), rtol=1e-5, atol=1e-4), name
        assert float((gpu.sim.obs.cpu() - cpu.sim.obs).abs().max()) < 1e-5, name


def test_ur_high_5_joint_variant_100_steps():
    # 12 position motors, no contacts: tolerance 2e-4 rad on joint angles / 2e-4 m on poses after 100 steps
    gpu, cpu = make_pair('ur_joint', 67)
    w = rollout(gpu, cpu, 100)
    assert w['obs'] < 2e-4 and w['state'] < 5e-3, w


def test_ur_high_5_ik_100_steps():
    # the reference's own YAML: batched IK + position motors
    gpu, cpu = make_pair('ur_ik', 67)
    w = rollout(gpu, cpu, 100)
    assert w['obs'] < 5e-4 and w['rew'] < 5e-4, w


def test_drone_pilot_60_steps():
    gpu, cpu = make_pair('drone', 33)
    w = rollout(gpu, cpu, 60)
    assert w['obs'] < 2e-3 and w['term_mismatch'] == 0, w


def test_marbles_contacts_200_steps():
    # resting + rolling contacts with friction; chaotic once marbles collide, so compare a short horizon
    gpu, cpu = make_pair('marbles', 9)
    w = rollout(gpu, cpu, 200, scale=1.0)
    assert w['obs'] < 2e-3, w


def test_cart_tree_every_feature_60_steps():
    # floating articulated base (branching tree, prismatic + revolute, limits, damping), sphere contacts,
    # every sensor flag, electricity cost, time penalty, episode timer, terminal_if_all, respawn jitter
    gpu, cpu = make_pair('cart_tree', 37)
    w = rollout(gpu, cpu, 60)
    assert w['obs'] < 5e-3 and w['rew'] < 5e-3 and w['term_mismatch'] == 0, w
    assert torch.equal(gpu.sim.term_flag.cpu(), cpu.sim.term_flag)
    assert float((gpu.sim.rew_sum.cpu() - cpu.sim.rew_sum).abs().max()) < 5e-3


def test_frame_state_getter_matches_oracle():
    gpu, cpu = make_pair('ur_ik', 5)
    rollout(gpu, cpu, 10)
    for body, frame, com in [(0, 7, False), (1, 7, True), (0, 3, False), (1, -1, True)]:
        a = gpu.sim.frame_state(body, frame, com).cpu().numpy()
        b = cpu.sim.frame_state64(body, frame, com)
        assert np.abs(a[:, :3] - b[:, :3]).max() < 1e-4 and np.abs(a[:, 7:] - b[:, 7:]).max() < 1e-3
        assert np.minimum(np.abs(a[:, 3:7] - b[:, 3:7]).max(1), np.abs(a[:, 3:7] + b[:, 3:7]).max(1)).max() < 1e-4


def test_masked_reset_only_touches_masked_envs():
    gpu, cpu = make_pair('drone', 16)
    rollout(gpu, cpu, 20)
    mask = torch.zeros(16, dtype=torch.uint8)
    mask[[1, 5, 11]] = 1
    before = gpu.sim.get_state()
    gpu.sim.reset(mask.to(gpu.device))
    cpu.sim.reset(mask)
    after = gpu.sim.get_state()
    keep = mask.numpy() == 0
    assert np.array_equal(before[keep], after[keep])
    assert np.abs(after - cpu.sim.get_state()).max() < 2e-3
