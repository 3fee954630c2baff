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
