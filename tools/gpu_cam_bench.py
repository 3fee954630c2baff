"""Render-kernel timing at BASELINE config 5's size: 1 024 envs x 200 x 200, rgb + depth (16 B per pixel)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import torch, yaml
import diy_gym_amd.examples
from diy_gym_amd import DIYGym
from diy_gym_amd.config import Configuration
B = 1024
def t(fn, n=20):
    fn(); torch.cuda.synchronize()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(n)]
    for a, b in ev:
        a.record(); fn(); b.record()
    torch.cuda.synchronize()
    return sorted(a.elapsed_time(b) for a, b in ev)[n // 2]
tree = yaml.safe_load(open(os.path.join(ROOT, 'examples/from_the_readme/from_the_readme.yaml')))
tree['overview'] = {'addon': 'camera', 'xyz': [1.2, -0.9, 1.4], 'rpy': [0.9, 0.0, 0.9], 'resolution': [200, 200]}
env = DIYGym(Configuration.from_dict('from_the_readme', tree), num_envs=B, device='cuda:0')
for _ in range(30): env.sim.step(env._all_slots, torch.zeros((B, env.layout.act_dim), device='cuda:0'))
for rec, name in (('r2d2', 'arm_camera'), ('from_the_readme', 'overview')):
    cam = env.receptors[rec].addons[name]; cam.observe(); rgb, depth, seg = cam._buffers
    for label, args in (('rgb+depth', (rgb, depth, None)), ('depth only', (None, depth, None))):
        for diag in os.environ.get('DIAGS', '0').split(','):
            env.sim.set_render_diag(int(diag))
            ms = t(lambda: env.sim.render(cam.camera_index, *args))
            nbytes = B * 200 * 200 * (16 if args[0] is not None else 4)
            print('%-10s %-10s diag %s: pose+render %.3f ms -> %.0f GB/s of image writes' % (name, label, diag, ms, nbytes / ms / 1e6), flush=True)

# candidate counters per stage (diagnostic)
import ctypes
lib = env.sim.lib
lib.dg_debug_render_counters.argtypes = [ctypes.POINTER(ctypes.c_uint64), ctypes.c_int32]
buf = (ctypes.c_uint64 * 16)()
env.sim.set_render_diag(16)
for rec, name in (('r2d2', 'arm_camera'), ('from_the_readme', 'overview')):
    cam = env.receptors[rec].addons[name]; rgb, depth, seg = cam._buffers
    lib.dg_debug_render_counters(buf, 1)
    env.sim.render(cam.camera_index, rgb, depth, None); torch.cuda.synchronize()
    lib.dg_debug_render_counters(buf, 1)
    c = list(buf); tiles = 13 * 25 * B
    print(name, 'per tile: after sphere-cone [sphere box capsule hull] %s | band-0 list length %.1f | after separating face [box hull] %s | after frustum [box hull] %s | intersected [sphere box capsule hull] %s' % (
        [round(x / tiles, 2) for x in c[0:4]], c[4] / B, [round(c[5] / tiles, 2), round(c[7] / tiles, 2)], [round(c[9] / tiles, 2), round(c[11] / tiles, 2)], [round(x / tiles, 2) for x in c[12:16]]))
    print('   cycles per workgroup (100 MHz s_memtime ticks): phase A %.0f | to the end of B1 %.0f | B2 makespan %.0f, mean over wavefronts %.0f | queued tiles %.1f' % (c[4] / B, c[6] / B, c[5] / B, c[10] / B / 4, c[8] / B))
    print('   depth background fraction %.4f' % float((depth <= -99.9).float().mean()))
