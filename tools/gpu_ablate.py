"""Cost attribution by ablation: time the step kernel while engine parameters switch parts of the work off."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np, torch, yaml
import diy_gym_amd.examples
from diy_gym_amd import DIYGym
from diy_gym_amd.config import Configuration
import test_parity_gpu as T

def timeit(cfg, B=16384, n=40, engine=None, tweak=None):
    tree = yaml.safe_load(open(cfg))
    if tweak: tree.update(tweak)
    conf = Configuration.from_dict(os.path.splitext(os.path.basename(cfg))[0], tree)
    env = DIYGym(conf, num_envs=B, device='cuda:0', engine=engine or {})
    lo, hi = T.action_bounds(env)
    gen = torch.Generator().manual_seed(1)
    ring = [(lo + (hi - lo) * torch.rand((B, lo.numel()), generator=gen)).to('cuda:0') for _ in range(8)]
    for i in range(10): env.sim.step(env._all_slots, ring[i % 8])
    torch.cuda.synchronize(); t0 = time.time()
    for i in range(n): env.sim.step(env._all_slots, ring[i % 8])
    torch.cuda.synchronize()
    return (time.time() - t0) / n * 1e3, env.sim.lanes, env.sim.lds_bytes

UR, URJ = T.CONFIGS['ur_ik'], T.CONFIGS['ur_joint']
print('ur_ik baseline          %.3f ms (lanes %d lds %d)' % timeit(UR))
print('ur_ik ik_iterations=1   %.3f ms' % timeit(UR, engine=dict(ik_iterations=1))[0])
print('ur_ik ik_iterations=5   %.3f ms' % timeit(UR, engine=dict(ik_iterations=5))[0])
print('ur_ik solver_iters=1    %.3f ms' % timeit(UR, tweak=dict(solver_iterations=1))[0])
print('ur_ik solver_iters=1, ik=1  %.3f ms' % timeit(UR, tweak=dict(solver_iterations=1), engine=dict(ik_iterations=1))[0])
print('ur_ik thr=1e-3 (fewer PGS its) %.3f ms' % timeit(UR, engine=dict(residual_threshold=1e-3))[0])
print('ur_ik update_freq=240 (1 substep) %.3f ms' % timeit(UR, tweak=dict(update_freq=240))[0])
print('ur_joint baseline       %.3f ms' % timeit(URJ)[0])
print('ur_joint solver_iters=1 %.3f ms' % timeit(URJ, tweak=dict(solver_iterations=1))[0])
for B in (4096, 8192, 16384, 32768, 65536):
    print('ur_ik B=%d  %.3f ms' % (B, timeit(UR, B=B)[0]))
