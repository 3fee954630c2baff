"""Time of a full masked reset (every env) per workspace mode: from_the_readme, 1 024 envs."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import torch
import diy_gym_amd.examples  # noqa: F401
from diy_gym_amd import DIYGym
import test_parity_gpu as T
for name, B, env_vars in (('readme', 1024, {}), ('readme', 1024, {'DG_NO_WAVE_ENV': '1'}), ('maze', 4096, {}), ('ur_ik', 16384, {})):
    os.environ.update(env_vars)
    env = DIYGym(T.CONFIGS[name], num_envs=B, device='cuda:0', seed=7)
    for k in env_vars: del os.environ[k]
    for _ in range(20): env.sim.step(env._all_slots, torch.zeros((B, env.layout.act_dim), device='cuda:0'))
    torch.cuda.synchronize()
    ts = []
    for frac in (1.0, 0.01):
        mask = (torch.rand(B, device='cuda:0') < frac).to(torch.uint8)
        torch.cuda.synchronize(); t0 = time.time()
        for _ in range(5): env.sim.reset(mask)
        torch.cuda.synchronize(); ts.append((time.time() - t0) / 5 * 1e3)
    print('%-8s %6d envs lanes %3d hot_start %d: reset of every env %.3f ms, of 1 %% of the envs %.3f ms' % (name, B, env.sim.lanes, env.layout.hot_start if hasattr(env.layout, 'hot_start') else -1, ts[0], ts[1]), flush=True)
