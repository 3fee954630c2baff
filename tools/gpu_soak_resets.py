"""Contact scenes under random masked resets in every workspace mode they can take: finite state / outputs, and the
envs a reset does not name keep their state bit for bit."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import torch
import diy_gym_amd.examples  # noqa: F401
from diy_gym_amd import DIYGym
import test_parity_gpu as T
for name, B, steps, env_vars in (('readme', 1024, 600, {}), ('readme', 1024, 300, {'DG_NO_WAVE_ENV': '1'}), ('maze', 4096, 1500, {}), ('maze', 1000, 600, {'DG_MAX_LANES': '1'}),
                                 ('marbles', 16384, 3000, {}), ('gripper', 1024, 600, {}), ('child', 4096, 1000, {})):
    os.environ.update(env_vars)
    env = DIYGym(T.CONFIGS[name], num_envs=B, device='cuda:0', seed=11)
    for k in env_vars: del os.environ[k]
    lo, hi = T.action_bounds(env)
    gen = torch.Generator().manual_seed(5)
    scale = 10.0 if name == 'maze' else 1.0
    ring = [((lo + (hi - lo) * torch.rand((B, lo.numel()), generator=gen)) * scale).to('cuda:0') for _ in range(16)]
    g2 = torch.Generator(device='cuda:0').manual_seed(9)
    t0 = time.time(); checked = 0
    for i in range(steps):
        env.sim.step(env._all_slots, ring[i % 16])
        if i % 7 == 3:
            mask = (torch.rand(B, device='cuda:0', generator=g2) < 0.03).to(torch.uint8)
            if i % 70 == 3:
                before = env.sim.state[:, :B].clone()
                env.sim.reset(mask)
                keep = mask == 0
                assert torch.equal(before[:, keep], env.sim.state[:, :B][:, keep]), (name, i)
                checked += 1
            else:
                env.sim.reset(mask)
        if i % 100 == 99:
            assert bool(torch.isfinite(env.sim.state[:, :B]).all()) and bool(torch.isfinite(env.sim.obs).all()), (name, i)
    torch.cuda.synchronize()
    print('%-8s %6d envs lanes %3d x %5d steps with masked resets: finite, %d untouched-env checks passed, %.1f s' % (name, B, env.sim.lanes, steps, checked, time.time() - t0), flush=True)
