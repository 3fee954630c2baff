import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import torch
import diy_gym_amd.examples  # noqa: F401
from diy_gym_amd import DIYGym
import test_parity_gpu as T
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
env = DIYGym(T.CONFIGS['maze'], num_envs=B, device='cuda:0', seed=11)
lo, hi = T.action_bounds(env)
gen = torch.Generator().manual_seed(5)
ring = [((lo + (hi - lo) * torch.rand((B, lo.numel()), generator=gen)) * 10.0).to('cuda:0') for _ in range(16)]
g2 = torch.Generator(device='cuda:0').manual_seed(9)
epw = env.sim.lanes
for i in range(80):
    env.sim.step(env._all_slots, ring[i % 16])
    if i % 7 == 3:
        mask = (torch.rand(B, device='cuda:0', generator=g2) < 0.03).to(torch.uint8)
        before = env.sim.state[:, :B].clone()
        env.sim.reset(mask)
        after = env.sim.state[:, :B]
        keep = mask == 0
        diff = (before != after) & keep[None, :]
        if bool(diff.any()):
            envs = torch.nonzero(diff.any(0)).flatten().tolist(); rows = torch.nonzero(diff.any(1)).flatten().tolist()
            resets = torch.nonzero(mask).flatten().tolist()
            print('step', i, 'lanes', epw, 'changed untouched envs', envs[:12], 'rows', rows[:20], 'state_dim', env.layout.state_dim)
            print('  same wave as a reset env:', [any(e // epw == r // epw for r in resets) for e in envs[:12]])
            e = envs[0]; r = rows[0]
            print('  env', e, 'row', r, 'before', float(before[r, e]), 'after', float(after[r, e]))
            break
else:
    print('no difference')
