"""First-contact GPU probe: builds nothing, runs each scene a few steps on cuda:0 next to the oracle and prints errors + timing."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np, torch
import test_parity_gpu as T

for name in ['marbles', 'drone', 'ur_joint', 'ur_ik']:
    gpu, cpu = T.make_pair(name, 64)
    print(name, 'state_dim', gpu.sim.state_dim, 'lds', gpu.sim.lds_bytes, 'lanes', gpu.sim.lanes, flush=True)
    print('  init state err', np.abs(gpu.sim.get_state() - cpu.sim.get_state()).max(), 'obs err', float((gpu.sim.obs.cpu() - cpu.sim.obs).abs().max()), flush=True)
    for steps in (1, 10, 50):
        w = T.rollout(gpu, cpu, steps, seed=steps)
        print('  after +%d steps' % steps, w, flush=True)
for name, B in [('ur_ik', 16384), ('ur_joint', 16384), ('drone', 16384), ('marbles', 4096)]:
    import diy_gym_amd.examples
    from diy_gym_amd import DIYGym
    env = DIYGym(T.CONFIGS[name], num_envs=B, device='cuda:0')
    lo, hi = T.action_bounds(env)
    act = (lo + (hi - lo) * torch.rand((B, lo.numel()))).to('cuda:0')
    for _ in range(5): env.sim.step(env._all_slots, act)
    torch.cuda.synchronize(); t0 = time.time()
    n = 50
    for _ in range(n): env.sim.step(env._all_slots, act)
    torch.cuda.synchronize(); dt = (time.time() - t0) / n
    print('%s B=%d: %.3f ms/step, %.3g env-steps/s (lanes %d, lds %d)' % (name, B, dt * 1e3, B / dt, env.sim.lanes, env.sim.lds_bytes), flush=True)
