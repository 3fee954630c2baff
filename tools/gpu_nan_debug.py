import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import torch
import diy_gym_amd.examples  # noqa: F401
from diy_gym_amd import DIYGym
import test_parity_gpu as T
B = 4096
do_reset = os.environ.get('RESETS', '1') == '1'
import yaml
from diy_gym_amd.config import Configuration
tree = yaml.safe_load(open(T.CONFIGS['maze']))
if os.environ.get('HOT') is not None: tree['hot_start'] = int(os.environ['HOT'])
env = DIYGym(Configuration.from_dict('r2d2_maze', tree), num_envs=B, device='cuda:0', seed=11)
lo, hi = T.action_bounds(env)
gen = torch.Generator().manual_seed(5)
ring = [((lo + (hi - lo) * torch.rand((B, lo.numel()), generator=gen)) * 10.0).to('cuda:0') for _ in range(16)]
g2 = torch.Generator(device='cuda:0').manual_seed(9)
was_reset = torch.zeros(B, dtype=torch.int32, device='cuda:0')
d = env.sim.enable_diagnostics()
for i in range(60):
    env.sim.step(env._all_slots, ring[i % 16])
    bad = ~torch.isfinite(env.sim.state[:, :B]).all(0)
    if bool(bad.any()):
        e = torch.nonzero(bad).flatten().tolist()
        print('lanes', env.sim.lanes, 'after STEP', i, 'non-finite envs', e[:10], 'previously reset at step', was_reset[e[:10]].tolist(), 'contacts', d[e[:10], 0].tolist(), 'iters', d[e[:10], 1].tolist())
        break
    if do_reset and i % 7 == 3:
        mask = (torch.rand(B, device='cuda:0', generator=g2) < 0.03).to(torch.uint8)
        before = env.sim.state[:, :B].clone()
        env.sim.reset(mask)
        was_reset[mask.bool()] = i
        bad = ~torch.isfinite(env.sim.state[:, :B]).all(0)
        if bool(bad.any()):
            e = torch.nonzero(bad).flatten().tolist()
            print('lanes', env.sim.lanes, 'after RESET at step', i, 'non-finite envs', e[:10], 'in mask', mask[e[:10]].tolist())
            torch.set_printoptions(precision=4, linewidth=200)
            ok = [k for k in torch.nonzero(mask).flatten().tolist() if k not in e][0]
            print('  bad env before reset :', before[:, e[0]].cpu())
            print('  good env before reset:', before[:, ok].cpu())
            print('  bad env after        :', env.sim.state[:, e[0]].cpu())
            print('  good env after       :', env.sim.state[:, ok].cpu())
            print('  contacts/iters bad', d[e[0]].tolist(), 'good', d[ok].tolist())
            break
else:
    print('lanes', env.sim.lanes, 'all finite over 60 steps (resets %s)' % do_reset)
