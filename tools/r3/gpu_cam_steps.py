"""Diagnostic: foreground fraction of the gripper camera per step, HIP (per build of the render kernel) vs oracle."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import torch, numpy as np
from diy_gym_amd import DIYGym
from oracle_backend import OracleBackend
from test_parity_gpu import CONFIGS
envs = {}
for wpe in ('1', '2', '3'):
    os.environ['DG_RENDER_WPE'] = wpe
    envs[wpe] = DIYGym(CONFIGS['readme'], num_envs=3, device='cuda:0', seed=5)
cpu = DIYGym(CONFIGS['readme'], num_envs=3, seed=5, backend_factory=OracleBackend)
def fg(e):
    e._tick += 1
    d = e.models['r2d2'].addons['arm_camera'].observe()['depth']
    return round(float((d > -99.0).float().mean()), 4)
for step in range(0, 31):
    if step % 3 == 0: print('step', step, 'oracle', fg(cpu), {k: fg(e) for k, e in envs.items()})
    z = torch.zeros((3, cpu.layout.act_dim))
    cpu.sim.step(cpu._all_slots, z)
    for e in envs.values(): e.sim.step(e._all_slots, z.to('cuda:0'))
