"""Settle a scene, then run a fixed number of steps (for rocprofv3 --pmc / --kernel-trace runs):
   python3 tools/gpu_scene_loop.py <config key of tests/test_parity_gpu.CONFIGS> <envs> <settle steps> <steps> [action scale]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import torch
import diy_gym_amd.examples  # noqa: F401
from diy_gym_amd import DIYGym
import test_parity_gpu as T
name, B, settle, steps = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
scale = float(sys.argv[5]) if len(sys.argv) > 5 else 1.0
env = DIYGym(T.CONFIGS[name], num_envs=B, device='cuda:0')
lo, hi = T.action_bounds(env)
act = ((lo + (hi - lo) * torch.rand((B, lo.numel()), generator=torch.Generator().manual_seed(3))) * scale).to('cuda:0')
for _ in range(settle + steps):
    env.sim.step(env._all_slots, act)
torch.cuda.synchronize()
print('done', name, B, settle, steps)
