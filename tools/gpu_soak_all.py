"""Every parity config under random actions (several scales) and random masked resets: finite state / outputs."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import torch
import diy_gym_amd.examples  # noqa: F401
from diy_gym_amd import DIYGym
import test_parity_gpu as T
seed = int(sys.argv[1]) if len(sys.argv) > 1 else 1
for name in T.CONFIGS:
    B = 1024 if name in ('readme', 'gripper') else 4096
    env = DIYGym(T.CONFIGS[name], num_envs=B, device='cuda:0', seed=seed)
    lo, hi = T.action_bounds(env)
    gen = torch.Generator().manual_seed(seed)
    g2 = torch.Generator(device='cuda:0').manual_seed(seed + 100)
    t0 = time.time(); steps = 300 if B == 1024 else 800
    for i in range(steps):
        scale = (1.0, 0.1, 3.0, 10.0)[(i // 50) % 4]
        act = ((lo + (hi - lo) * torch.rand((B, lo.numel()), generator=gen)) * scale).to('cuda:0')
        env.sim.step(env._all_slots, act)
        if i % 5 == 2:
            env.sim.reset((torch.rand(B, device='cuda:0', generator=g2) < 0.05).to(torch.uint8))
        else:
            env.sim.reset(env.sim.term_flag)
        if i % 50 == 49:
            ok = bool(torch.isfinite(env.sim.state[:, :B]).all()) and bool(torch.isfinite(env.sim.obs).all()) and bool(torch.isfinite(env.sim.rew).all())
            if not ok:
                bad = torch.nonzero(~torch.isfinite(env.sim.state[:, :B]).all(0)).flatten().tolist()
                print('%-12s NON-FINITE at step %d, envs %s' % (name, i, bad[:8]), flush=True); break
    else:
        torch.cuda.synchronize()
        print('%-12s %5d envs lanes %3d x %4d steps: finite  (%.1f s)' % (name, B, env.sim.lanes, steps, time.time() - t0), flush=True)
