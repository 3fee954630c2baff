"""Step-kernel time of ur_high_5 as a function of the IK iteration count: the slope is the cost of one IK iteration on the
critical path, the intercept everything else."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import torch
from diy_gym_amd import DIYGym
import test_parity_gpu as T
B = 16384
for iters in (80, 40, 20):
    env = DIYGym(T.CONFIGS['ur_ik'], num_envs=B, device='cuda:0', seed=7, engine=dict(ik_iterations=iters))
    lo, hi = T.action_bounds(env)
    gen = torch.Generator().manual_seed(5)
    ring = [(lo + (hi - lo) * torch.rand((B, lo.numel()), generator=gen)).to('cuda:0') for _ in range(8)]
    for i in range(30): env.sim.step(env._all_slots, ring[i % 8])
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(100)]
    for i in range(100):
        ev[i][0].record(); env.sim.step(env._all_slots, ring[i % 8]); ev[i][1].record()
    torch.cuda.synchronize()
    ms = sorted(a.elapsed_time(b) for a, b in ev)
    print('ik_iterations %2d: step kernel median %.1f us' % (iters, ms[50] * 1e3), flush=True)
