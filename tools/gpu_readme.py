import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import torch
import diy_gym_amd.examples
from diy_gym_amd import DIYGym
import test_parity_gpu as T
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
env = DIYGym(T.CONFIGS['readme'], num_envs=B, device='cuda:0')
print('lanes', env.sim.lanes, 'lds', env.sim.lds_bytes, 'state', env.sim.state_dim, 'contacts', env.layout.max_contacts)
lo, hi = T.action_bounds(env)
act = (lo + (hi - lo) * torch.rand((B, lo.numel()))).to('cuda:0')
cam = env.models['r2d2'].addons['arm_camera']
def t(fn, n=5):
    fn(); torch.cuda.synchronize(); t0 = time.time()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.time() - t0) / n * 1e3
print('step   %.2f ms' % t(lambda: env.sim.step(env._all_slots, act)))
settle = int(sys.argv[2]) if len(sys.argv) > 2 else 0
for _ in range(settle): env.sim.step(env._all_slots, act * 0.2)
if settle: print('step after %d settling steps  %.2f ms' % (settle, t(lambda: env.sim.step(env._all_slots, act * 0.2))))
def render():
    env._tick += 1; cam.observe()
ms = t(render, 10)
px = B * 200 * 200
print('render %.3f ms  -> %.1f GB/s of image writes (16 B/pixel), %.2f Gpixel/s' % (ms, px * 16 / ms / 1e6, px / ms / 1e6))
d = env.sim.enable_diagnostics(); env.sim.step(env._all_slots, act); torch.cuda.synchronize()
print('contacts mean %.1f max %d, iterations mean %.1f max %d' % (d[:, 0].float().mean(), d[:, 0].max(), d[:, 1].float().mean(), d[:, 1].max()))
rgb, depth, seg = cam._buffers
for name, args in (('depth only', (None, depth, None)), ('rgb only', (rgb, None, None)), ('none (no writes)', (None, None, None)), ('both', (rgb, depth, None))):
    print('%-18s %.3f ms' % (name, t(lambda: env.sim.render(cam.camera_index, *args), 10)))
