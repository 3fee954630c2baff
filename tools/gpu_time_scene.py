"""Step time of a scene per engine setting: python tools/gpu_time_scene.py <config name> <envs> [key=value ...]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import torch
import diy_gym_amd.examples
from diy_gym_amd import DIYGym
import test_parity_gpu as T
name, B = sys.argv[1], int(sys.argv[2])
variants = [dict()] + [dict([kv.split('=')]) for kv in sys.argv[3:]]
for eng in variants:
    eng = {k: float(v) for k, v in eng.items()}
    env = DIYGym(T.CONFIGS[name], num_envs=B, device='cuda:0', engine=eng)
    lo, hi = T.action_bounds(env); gen = torch.Generator().manual_seed(1)
    ring = [(lo + (hi - lo) * torch.rand((B, lo.numel()), generator=gen)).to('cuda:0') for _ in range(8)]
    for i in range(30): env.sim.step(env._all_slots, ring[i % 8])
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for i in range(64): env.sim.step(env._all_slots, ring[i % 8])
    b.record(); torch.cuda.synchronize()
    d = env.sim.enable_diagnostics()
    for i in range(8): env.sim.step(env._all_slots, ring[i % 8])
    torch.cuda.synchronize()
    it = d[:, 1].float()
    print(name, B, eng, 'lanes', env.sim.lanes, '%.4f ms/step' % (a.elapsed_time(b) / 64), 'iters mean %.1f wave-max mean %.1f' % (it.mean(), it.reshape(-1, env.sim.envs_per_wave).max(1).values.mean() if B % env.sim.envs_per_wave == 0 else -1))
