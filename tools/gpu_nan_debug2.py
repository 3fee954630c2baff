"""The maze env that goes non-finite one step after a masked reset: same state and action through the oracle."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np, torch, yaml
import diy_gym_amd.examples  # noqa: F401
from diy_gym_amd import DIYGym
from diy_gym_amd.config import Configuration
from oracle_backend import OracleBackend
import test_parity_gpu as T
B = 4096
tree = yaml.safe_load(open(T.CONFIGS['maze'])); tree['hot_start'] = 0
env = DIYGym(Configuration.from_dict('r2d2_maze', tree), num_envs=B, device='cuda:0', seed=11)
lo, hi = T.action_bounds(env)
gen = torch.Generator().manual_seed(5)
ring = [((lo + (hi - lo) * torch.rand((B, lo.numel()), generator=gen)) * 10.0).to('cuda:0') for _ in range(16)]
g2 = torch.Generator(device='cuda:0').manual_seed(9)
d = env.sim.enable_diagnostics()
for i in range(4):
    env.sim.step(env._all_slots, ring[i % 16])
mask = (torch.rand(B, device='cuda:0', generator=g2) < 0.03).to(torch.uint8)
env.sim.reset(mask)
st = np.array(env.sim.get_state())
e = 122
env.sim.step(env._all_slots, ring[4])
after = np.array(env.sim.get_state())
print('gpu env', e, 'finite after step:', np.isfinite(after[e]).all(), 'contacts', int(d[e, 0]), 'was reset', int(mask[e]))
import copy
cpu = DIYGym(Configuration.from_dict('r2d2_maze', copy.deepcopy(tree)), num_envs=2, seed=11, backend_factory=OracleBackend)
cpu.sim.set_state(st[[e, e + 1]])
act = ring[4][[e, e + 1]].cpu()
cpu.sim.step(cpu._all_slots, act)
cs = np.array(cpu.sim.get_state())
print('oracle finite:', np.isfinite(cs[0]).all(), 'contacts', cpu.sim.contacts(0))
np.set_printoptions(precision=5, linewidth=220, suppress=False)
print('state in :', st[e])
print('oracle out:', cs[0])
print('gpu out   :', after[e])
print('action', act[0].numpy())
# reproduce in a small batch: every env gets the state of the failing env
for Bs, ev in ((8, {}), (8, {'DG_MAX_LANES': '32'}), (8, {'DG_NO_MINV_SLICES': '1'}), (1, {})):
    os.environ.update(ev)
    small = DIYGym(Configuration.from_dict('r2d2_maze', copy.deepcopy(tree)), num_envs=Bs, device='cuda:0', seed=11)
    for k in ev: del os.environ[k]
    ds = small.sim.enable_diagnostics()
    small.sim.set_state(np.repeat(st[[e]], Bs, axis=0))
    small.sim.step(small._all_slots, ring[4][[e] * Bs].contiguous())
    out = np.array(small.sim.get_state())
    print('small batch', Bs, ev, 'lanes', small.sim.lanes, 'finite', np.isfinite(out).all(1).tolist(), 'contacts', ds[:, 0].tolist())
print('--- variants of the failing state (B = 1)')
small = DIYGym(Configuration.from_dict('r2d2_maze', copy.deepcopy(tree)), num_envs=1, device='cuda:0', seed=11)
ds = small.sim.enable_diagnostics()
L = small.layout
def trial(label, s, a):
    small.sim.set_state(s[None, :].copy()); small.sim.step(small._all_slots, a[None, :].contiguous())
    out = np.array(small.sim.get_state())[0]
    print('  %-46s finite %s  diag %s' % (label, bool(np.isfinite(out).all()), ds[0].tolist()))
s0 = st[e].copy(); a0 = ring[4][e].clone()
trial('as is', s0, a0)
s = s0.copy(); s[np.abs(s) < 1e-6] = 0.0; trial('tiny entries zeroed', s, a0)
trial('zero action', s0, a0 * 0)
for k in range(21, len(s0)):
    if s0[k] != 0.0 and abs(s0[k]) < 1e-6:
        s = s0.copy(); s[k] = 0.0; trial('entry %d (%.3e) zeroed' % (k, s0[k]), s, a0)
s = s0.copy(); s[0] = 5.0; trial('step counter 5', s, a0)
s = s0.copy(); s[1] = 1.0; trial('episode 1', s, a0)
print('link_state_off', list(L.link_state_off), 'body_state_off', list(L.body_state_off), 'state_dim', L.state_dim, 'addon_off', L.addon_off)
