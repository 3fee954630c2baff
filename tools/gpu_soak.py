"""Long rollouts with random actions and masked auto-reset; fails on any non-finite state / output."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import torch
import diy_gym_amd.examples  # noqa: F401
from diy_gym_amd import DIYGym
import test_parity_gpu as T
for name, B, steps in (('ur_ik', 16384, 20000), ('ur_joint', 16384, 10000), ('drone', 16384, 5000), ('marbles', 4096, 5000), ('maze', 4096, 1500),
                       ('readme', 1024, 400), ('gripper', 1024, 1000), ('touching', 4096, 2000), ('touching_ik', 16384, 2000)):  # (touching*: the hull narrow phase, polytope search included)
    env = DIYGym(T.CONFIGS[name], num_envs=B, device='cuda:0', seed=7)
    lo, hi = T.action_bounds(env)
    gen = torch.Generator().manual_seed(5)
    ring = [(lo + (hi - lo) * torch.rand((B, lo.numel()), generator=gen)).to('cuda:0') for _ in range(16)]
    t0 = time.time(); resets = 0
    for i in range(steps):
        env.sim.step(env._all_slots, ring[i % 16])
        env.sim.reset(env.sim.term_flag)
        if i % 500 == 499:
            resets += int(env.sim.term_flag.sum())
            assert bool(torch.isfinite(env.sim.state[:, :B]).all()) and bool(torch.isfinite(env.sim.obs).all()) and bool(torch.isfinite(env.sim.rew).all()), (name, i)
    torch.cuda.synchronize()
    ok = bool(torch.isfinite(env.sim.state[:, :B]).all())
    print('%-9s %6d envs x %6d steps: finite=%s  %.1f s  (terminals seen at checkpoints: %d)' % (name, B, steps, ok, time.time() - t0, resets), flush=True)
    assert ok
