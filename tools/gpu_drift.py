"""Step time of ur_high_5 as a random-action rollout ages (no episode limit in the reference YAML)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import torch
from diy_gym_amd import DIYGym
import test_parity_gpu as T
B = 16384
env = DIYGym(T.CONFIGS['ur_ik'], num_envs=B, device='cuda:0', seed=7)
d = env.sim.enable_diagnostics()
lo, hi = T.action_bounds(env)
gen = torch.Generator().manual_seed(5)
ring = [(lo + (hi - lo) * torch.rand((B, lo.numel()), generator=gen)).to('cuda:0') for _ in range(16)]
done = 0
for upto in (300, 1000, 2000, 4000, 8000, 16000):
    torch.cuda.synchronize(); t0 = time.time()
    for i in range(done, upto):
        env.sim.step(env._all_slots, ring[i % 16]); env.sim.reset(env.sim.term_flag)
    torch.cuda.synchronize(); dt = (time.time() - t0) / (upto - done); done = upto
    q = env.sim.obs[:, 0:6]
    print('steps %5d: %.3f ms/step | contacts mean %.3f max %d | iterations mean %.1f max %d | max |q| %.2f' % (
        upto, dt * 1e3, d[:, 0].float().mean(), d[:, 0].max(), d[:, 1].float().mean(), d[:, 1].max(), float(q.abs().max())), flush=True)
