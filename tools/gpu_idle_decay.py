"""Quantities that decay geometrically towards the denormal range in an idle rollout (rotor speeds of drone_pilot under zero
commands: x 0.9 per step) must not poison anything on their way through 1e-38 .. 1e-45."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import torch
import diy_gym_amd.examples  # noqa: F401
from diy_gym_amd import DIYGym
import test_parity_gpu as T
for name in ('drone', 'marbles', 'cart_tree', 'admittance'):
    B = 256
    env = DIYGym(T.CONFIGS[name], num_envs=B, device='cuda:0', seed=3)
    lo, hi = T.action_bounds(env)
    gen = torch.Generator().manual_seed(1)
    for i in range(30):
        env.sim.step(env._all_slots, (lo + (hi - lo) * torch.rand((B, lo.numel()), generator=gen)).to('cuda:0'))
    zero = torch.zeros((B, lo.numel()), device='cuda:0')
    first = None
    for i in range(1300):
        env.sim.step(env._all_slots, zero)
        if i % 50 == 49 and first is None and not bool(torch.isfinite(env.sim.state[:, :B]).all() and torch.isfinite(env.sim.obs).all()):
            first = i
    print('%-10s idle for 1300 steps: %s' % (name, 'finite' if first is None else 'NON-FINITE by step %d' % first), flush=True)
