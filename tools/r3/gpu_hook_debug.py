"""Diagnostic: where the Python-hook propellor and the compiled op part ways (ext wrench rows right after the update phase)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import torch, yaml, numpy as np
import diy_gym_amd.examples
from diy_gym_amd import DIYGym
from diy_gym_amd.addons.addon import AddonFactory
from diy_gym_amd.config import Configuration
from user_addons import PyPropellor
AddonFactory.register_addon('py_propellor', PyPropellor)
DRONE = os.path.join(ROOT, 'examples', 'drone_pilot', 'drone_pilot.yaml')
tree = yaml.safe_load(open(DRONE))
motors = sorted(k for k, v in tree['drone'].items() if isinstance(v, dict) and v.get('addon') == 'propellor')
B = 65
def make(name):
    t = yaml.safe_load(open(DRONE))
    for m in motors: t['drone'][m]['addon'] = name
    return DIYGym(Configuration.from_dict('drone_pilot', t), num_envs=B, device='cuda:0', seed=4)
c, h = make('propellor'), make('py_propellor')
print('lanes', c.sim.lanes, h.sim.lanes, 'state dims', c.layout.state_dim, h.layout.state_dim, 'addon_off', c.layout.addon_off, h.layout.addon_off)
print('initial equal', torch.equal(c.sim.state[:c.layout.addon_off], h.sim.state[:h.layout.addon_off]))
print('ops order compiled:', [m for m in c.receptors['drone'].addons], 'motors', motors)
gen = torch.Generator().manual_seed(1)
for step in range(3):
    act = {'drone': {m: torch.rand((B, 1), generator=gen).to('cuda:0') for m in motors}}
    for m in motors: h.receptors['drone'].addons[m].update(act['drone'][m])
    so = h.layout.body_state_off[[i for i in range(h.layout.n_bodies) if not h.layout.body_fixed[i]][0]]
    ext_h = h.sim.state[so + 13:so + 19, :B].clone()
    h.sim.step(0); h._tick += 1
    c.step(act)
    a, b = c.sim.state[:c.layout.addon_off, :B], h.sim.state[:h.layout.addon_off, :B]
    d = (a - b).abs()
    rows = torch.nonzero(d.max(1).values > 0).flatten().tolist()
    print('step', step, 'rows that differ', rows, 'max', float(d.max()), 'body state off', so, 'ext_h env0', ext_h[:, 0].tolist())
    print('   rotor speeds compiled', c.sim.state[c.layout.addon_off:c.layout.addon_off + 4, 0].tolist(), 'hooked', [float(h.receptors['drone'].addons[m].rotor_speed[0]) for m in motors])

# ---- what the compiled ops applied, made visible: DG_DEBUG_KEEP_EXT leaves the external wrench in the state after a step
os.environ['DG_DEBUG_KEEP_EXT'] = '1'
c2, h2 = make('propellor'), make('py_propellor')
del os.environ['DG_DEBUG_KEEP_EXT']
st = np.array(c2.sim.get_state()); so = c2.layout.body_state_off[[i for i in range(c2.layout.n_bodies) if not c2.layout.body_fixed[i]][0]]
rng = np.random.default_rng(0); q = rng.normal(size=(B, 4)) * float(os.environ.get('TILT', '0.2')) + np.array([0, 0, 0, 1.0]); q /= np.linalg.norm(q, axis=1, keepdims=True)
st[:, so + 3:so + 7] = q; st[:, so:so + 3] += rng.normal(size=(B, 3)) * 0.3
sh = np.array(h2.sim.get_state()); sh[:, :c2.layout.addon_off] = st[:, :c2.layout.addon_off]
c2.sim.set_state(st); h2.sim.set_state(sh)
act = {'drone': {m: torch.rand((B, 1), generator=gen).to('cuda:0') for m in motors}}
c2.step(act)
for m in motors: h2.receptors['drone'].addons[m].update(act['drone'][m])
ec = c2.sim.state[so + 13:so + 19, :B]; eh = h2.sim.state[so + 13:so + 19, :B]
print('ext wrench rows equal:', [bool(torch.equal(ec[k], eh[k])) for k in range(6)], 'max abs diff', float((ec - eh).abs().max()), 'max', float(ec.abs().max()))
bad = (ec != eh).any(0).nonzero().flatten()[:3].tolist()
for e in bad: print('  env', e, 'compiled', ec[:, e].tolist(), 'hooked', eh[:, e].tolist())

# ---- the real rollout with the external wrench left in the state (it accumulates: same for both)
os.environ['DG_DEBUG_KEEP_EXT'] = '1'
c3, h3 = make('propellor'), make('py_propellor')
del os.environ['DG_DEBUG_KEEP_EXT']
gen = torch.Generator().manual_seed(1)
for step in range(4):
    act = {'drone': {m: torch.rand((B, 1), generator=gen).to('cuda:0') for m in motors}}
    pre_c = c3.sim.state[so:so + 13, :B].clone(); pre_h = h3.sim.state[so:so + 13, :B].clone()
    c3.step(act)
    for m in motors: h3.receptors['drone'].addons[m].update(act['drone'][m])
    eh = h3.sim.state[so + 13:so + 19, :B].clone()
    h3.sim.step(0)
    ec = c3.sim.state[so + 13:so + 19, :B]
    print('rollout step', step, 'pre-state equal', bool(torch.equal(pre_c, pre_h)), 'ext equal', [bool(torch.equal(ec[k], eh[k])) for k in range(6)], 'post-state equal', bool(torch.equal(c3.sim.state[so:so + 13, :B], h3.sim.state[so:so + 13, :B])),
          'warm caches equal', bool(torch.equal(c3.sim.state[c3.layout.warm_off:, :B], h3.sim.state[h3.layout.warm_off:, :B])))
    if not torch.equal(ec, eh):
        e = int((ec != eh).any(0).nonzero()[0]); print('   env', e, 'compiled ext', ec[:, e].tolist(), 'hooked', eh[:, e].tolist(), 'quat', pre_c[3:7, e].tolist())

# ---- replicate rollout step 1 in fresh worlds and drop motors one at a time
os.environ['DG_DEBUG_KEEP_EXT'] = '1'
c4, h4 = make('propellor'), make('py_propellor')
del os.environ['DG_DEBUG_KEEP_EXT']
gen = torch.Generator().manual_seed(1)
acts = [{'drone': {m: torch.rand((B, 1), generator=gen).to('cuda:0') for m in motors}} for _ in range(2)]
c4.step(acts[0]); h4.step(acts[0])
base_c = np.array(c4.sim.get_state()); base_h = np.array(h4.sim.get_state()); w_h = [h4.receptors['drone'].addons[m].rotor_speed.clone() for m in motors]
print('after step 0: drone rows equal', np.array_equal(base_c[:, so:so + 13], base_h[:, so:so + 13]))
for keep in ([0, 1, 2, 3], [0], [1], [2], [3], [0, 1], [2, 3]):
    sc_ = base_c.copy(); sh_ = base_h.copy(); sc_[:, so + 13:so + 19] = 0; sh_[:, so + 13:so + 19] = 0
    c4.sim.set_state(sc_); h4.sim.set_state(sh_)
    for i, m in enumerate(motors): h4.receptors['drone'].addons[m].rotor_speed = w_h[i].clone()
    a = {'drone': {m: acts[1]['drone'][m] for i, m in enumerate(motors) if i in keep}}
    c4.step(a)
    for i, m in enumerate(motors):
        if i in keep: h4.receptors['drone'].addons[m].update(a['drone'][m])
    ec = c4.sim.state[so + 13:so + 19, :B]; eh = h4.sim.state[so + 13:so + 19, :B]
    nbad = int((ec != eh).any(0).sum())
    print('motors', keep, 'ext rows equal', [bool(torch.equal(ec[k], eh[k])) for k in range(6)], 'envs that differ', nbad)
    if nbad:
        e = int((ec != eh).any(0).nonzero()[0]); print('    env', e, 'compiled', ec[:, e].tolist(), 'hooked', eh[:, e].tolist())
