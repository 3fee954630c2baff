"""from_the_readme: gripper-camera depth against the oracle's after 6 steps; which body the oracle sees where the GPU sees nothing."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np, torch, yaml
import test_parity_gpu as T
from diy_gym_amd import DIYGym
from diy_gym_amd.config import Configuration
from oracle_backend import OracleBackend
import diy_gym_amd.examples
tree = yaml.safe_load(open(T.CONFIGS['readme']))
tree['r2d2']['arm_camera']['use_segmentation_mask'] = True
for B in (3,):
    gpu = DIYGym(Configuration.from_dict('from_the_readme', tree), num_envs=B, device='cuda:0', seed=5)
    cpu = DIYGym(Configuration.from_dict('from_the_readme', tree), num_envs=B, seed=5, backend_factory=OracleBackend)
    w = T.rollout(gpu, cpu, 6)
    for diag in ('0', '32'):
        os.environ['DG_RENDER_DIAG'] = diag
        gpu._tick += 1; cpu._tick += 1
        g = gpu.models['r2d2'].addons['arm_camera'].observe(); c = cpu.models['r2d2'].addons['arm_camera'].observe()
        d = (g['depth'].cpu() - c['depth']).abs(); fg = c['depth'] > -99.9
        print('B', B, 'lanes', gpu.sim.lanes, 'diag', diag, 'close %.4f' % float((d < 5e-3).float().mean()), 'oracle fg px', int(fg.sum()),
              'oracle seg there', torch.unique(c['segmentation_mask'][fg]).tolist(), 'gpu seg there', torch.unique(g['segmentation_mask'].cpu()[fg]).tolist(),
              'gpu fg px', int((g['depth'].cpu() > -99.9).sum()), 'gpu depth there: min %.5f max %.5f' % (float(g['depth'].cpu()[fg].min()), float(g['depth'].cpu()[fg].max())), 'gpu depth elsewhere fg:', int((g['depth'].cpu()[~fg] > -99.9).sum()))
    del os.environ['DG_RENDER_DIAG']
    print('  bodies:', {i: n for i, n in enumerate(gpu.layout.body_names)} if hasattr(gpu.layout, 'body_names') else gpu.layout.n_bodies)
