"""Diagnostic: gripper-camera depth of from_the_readme (3 envs: one 8-row band per workgroup) against the oracle, per render switch."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import torch, numpy as np
from test_parity_gpu import make_pair, rollout
gpu, cpu = make_pair('readme', 3)
w = rollout(gpu, cpu, 6); print({k: v for k, v in w.items() if k in ('obs', 'rew')})
cpu._tick += 1
c = cpu.models['r2d2'].addons['arm_camera'].observe()
for diag in (0, 128, 1, 129):
    gpu.sim.set_render_diag(diag); gpu._tick += 1
    g = gpu.models['r2d2'].addons['arm_camera'].observe()
    close = (g['depth'].cpu() - c['depth']).abs() < 5e-3
    fg = c['depth'] > -99.0
    print('diag', diag, 'agree %.4f' % float(close.float().mean()), 'foreground agree %.4f' % float(close[fg].float().mean()), 'gpu fg frac %.4f cpu fg frac %.4f' % (float((g['depth'] > -99.0).float().mean()), float(fg.float().mean())))
    rows = (~close).any(dim=2).nonzero()
    print('   rows with differences (env,row):', rows[:12].tolist())
gpu.sim.set_render_diag(0)
for Bn in (3, 64, 300, 1024):
    from diy_gym_amd import DIYGym
    from test_parity_gpu import CONFIGS
    for wpe in ('2', '3'):
        os.environ['DG_RENDER_WPE'] = wpe
        e = DIYGym(CONFIGS['readme'], num_envs=Bn, device='cuda:0', seed=5)
        for _ in range(6): e.sim.step(e._all_slots, torch.zeros((Bn, e.layout.act_dim), device='cuda:0'))
        e._tick += 1
        d = e.models['r2d2'].addons['arm_camera'].observe()['depth']
        print('B', Bn, 'wpe', wpe, 'lanes', e.sim.lanes, 'foreground fraction %.4f' % float((d > -99.0).float().mean()), 'per env', [(round(float((d[k] > -99.0).float().mean()), 4)) for k in range(min(Bn, 3))])
