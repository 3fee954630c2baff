import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import torch, yaml
from diy_gym_amd import DIYGym
from diy_gym_amd.config import Configuration
from diy_gym_amd.scene import K
B = 1024
def t(fn, n=10):
    fn(); torch.cuda.synchronize(); t0 = time.time()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.time() - t0) / n * 1e3
tree = yaml.safe_load(open(os.path.join(ROOT, 'tests/golden/basic_env.yaml')))
tree['camera']['resolution'] = [200, 200]
env = DIYGym(Configuration.from_dict('basic_env', tree), num_envs=B, device='cuda:0')
cam = env.addons['camera']; cam.observe(); rgb, depth, seg = cam._buffers
ms = t(lambda: env.sim.render(cam.camera_index, rgb, depth, None))
print('marbles top-down 200x200 x %d envs: %.3f ms -> %.0f GB/s image writes' % (B, ms, B * 200 * 200 * 16 / ms / 1e6))
import diy_gym_amd.examples
env2 = DIYGym(os.path.join(ROOT, 'examples/from_the_readme/from_the_readme.yaml'), num_envs=B, device='cuda:0')
I = env2.layout.I
SI = I[I[K.H_OFF_SHAPE_I]:I[K.H_OFF_SHAPE_I] + I[K.H_N_SHAPES] * K.SI_STRIDE].reshape(-1, K.SI_STRIDE)
print('readme shapes', len(SI), 'types', [int((SI[:, 0] == k).sum()) for k in range(4)], 'hull planes total', I[K.H_N_PLANES], 'max per hull', SI[:, K.SI_N_PLANES].max())
