import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np, torch, yaml
from diy_gym_amd import DIYGym
from diy_gym_amd.config import Configuration
from oracle_backend import OracleBackend
tree = yaml.safe_load(open(os.path.join(ROOT, 'tests/golden/basic_env.yaml')))
tree['camera']['use_segmentation_mask'] = True
tree['camera']['resolution'] = [64, 64]
tree['green_marble']['eye'] = {'addon': 'camera', 'xyz': [0, -2.0, 0.5], 'rpy': [1.2, 0, 0], 'resolution': [40, 40], 'use_segmentation_mask': True}
B = 5
gpu = DIYGym(Configuration.from_dict('basic_env', tree), num_envs=B, device='cuda:0', seed=2)
cpu = DIYGym(Configuration.from_dict('basic_env', yaml.safe_load(yaml.dump(tree))), num_envs=B, seed=2, backend_factory=OracleBackend)
for rec, name in (('basic_env', 'camera'), ('green_marble', 'eye')):
    g = gpu.receptors[rec].addons[name].observe(); c = cpu.receptors[rec].addons[name].observe()
    torch.cuda.synchronize()
    sg, sc = g['segmentation_mask'].cpu(), c['segmentation_mask']
    print(name, 'seg equal frac', (sg == sc).float().mean().item(), 'gpu uniq', torch.unique(sg).tolist(), 'cpu uniq', torch.unique(sc).tolist())
    print(' depth gpu', g['depth'].min().item(), g['depth'].max().item(), 'cpu', c['depth'].min().item(), c['depth'].max().item())
    print(' idx', gpu.receptors[rec].addons[name].camera_index, cpu.receptors[rec].addons[name].camera_index)
g = gpu.addons['camera'].observe(); c = cpu.addons['camera'].observe()
sg, sc = g['segmentation_mask'].cpu(), c['segmentation_mask']
print(sg[0, ::8, ::8]); print(sc[0, ::8, ::8])
print(sg.dtype, sc.dtype, sg.shape, sc.shape, (sg[0] == sc[0]).sum().item())
print('depth eq', (g['depth'].cpu() - c['depth']).abs().max().item())
