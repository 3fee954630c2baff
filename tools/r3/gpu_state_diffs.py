"""Diagnostic: HIP vs oracle state differences split into kinematic columns (poses, twists, q, qd) and effort columns."""
import copy, os, sys
import numpy as np, torch, yaml
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
from diy_gym_amd import DIYGym
from diy_gym_amd.config import Configuration
from oracle_backend import OracleBackend
from test_parity_gpu import action_bounds, CONFIGS
import diy_gym_amd.examples

def cols(L):
    kin, eff, names = [], [], {}
    for b in range(L.n_bodies):
        so = L.body_state_off[b]
        if so < 0: continue
        n = 7 if L.body_fixed[b] else 13
        for k in range(n): kin.append(so + k); names[so + k] = 'body%d.%s' % (b, ['px','py','pz','qx','qy','qz','qw','vx','vy','vz','wx','wy','wz'][k])
    for i, lo in enumerate(L.link_state_off):
        kin += [lo, lo + 1]; eff.append(lo + 5); names[lo] = 'link%d.q' % i; names[lo + 1] = 'link%d.qd' % i; names[lo + 5] = 'link%d.applied' % i
    return kin, eff, names

def pair(tree, B, name='s', **eng):
    gpu = DIYGym(Configuration.from_dict(name, copy.deepcopy(tree)), num_envs=B, device='cuda:0', seed=5, engine=eng)
    cpu = DIYGym(Configuration.from_dict(name, copy.deepcopy(tree)), num_envs=B, seed=5, backend_factory=OracleBackend, engine=eng)
    return gpu, cpu

def report(tag, gpu, cpu, steps, scale=1.0, every=4, actfix=None, seed=0):
    kin, eff, names = cols(gpu.layout)
    d = gpu.sim.enable_diagnostics(); B = gpu.num_envs
    lo, hi = action_bounds(gpu); gen = torch.Generator().manual_seed(seed)
    print('==', tag, 'lanes', gpu.sim.lanes)
    for i in range(steps):
        act = (lo + (hi - lo) * torch.rand((B, lo.numel()), generator=gen)) * scale
        if actfix: actfix(act)
        gpu.sim.step(gpu._all_slots, act.to(gpu.device)); cpu.sim.step(cpu._all_slots, act)
        if i % every == every - 1 or i == steps - 1:
            a, b = gpu.sim.get_state(), cpu.sim.get_state(); df = np.abs(a - b)
            e, k = np.unravel_index(df[:, kin].argmax(), df[:, kin].shape)
            ee = df[:, eff].max() if eff else 0.0
            print(i, 'kin %.3e (%s env %d: gpu %.5f cpu %.5f)  effort %.3e  obs %.3e' % (df[:, kin].max(), names[kin[k]], e, a[e, kin[k]], b[e, kin[k]], ee, float((gpu.sim.obs.cpu() - cpu.sim.obs).abs().max())),
                  'iters', d[:3, 1].tolist(), [cpu.sim.iterations(q) for q in range(3)], 'cont', d[:3, 0].tolist())

maze = yaml.safe_load(open(CONFIGS['maze']))
for cap, thr in ((150, 1e-7), (4000, 1e-13)):
    t = copy.deepcopy(maze); t['solver_iterations'] = cap
    g, c = pair(t, 19, residual_threshold=thr); report('maze cap %d thr %g' % (cap, thr), g, c, 40, scale=10.0)
mw = yaml.safe_load(open(CONFIGS['marbles'])); mw['solver_iterations'] = 2000
mw['r2d2'] = {'model': 'r2d2.urdf', 'xyz': [0.0, 0.0, 0.5]}
mw['red_marble']['xyz'] = [0.28, 0.12, 0.25]; mw['green_marble']['xyz'] = [-0.28, 0.12, 0.25]; mw['blue_marble']['xyz'] = [0.27, -0.12, 1.2]
def push(act): act[:, 0] = -abs(act[:, 0]) * 20.0
g, c = pair(mw, 7, residual_threshold=1e-13); report('marbles+wheels converged', g, c, 60, every=6, actfix=push, seed=2)
crowd = {'plane': {'model': 'grass/plane.urdf'},
         'crowd': {'addon': 'spawn_multiple', 'num_models': 3,
                   'ball': {'model': 'sphere2.urdf', 'scale': 0.2, 'xyz': [0, 0, 0.4], 'mass': 0.5,
                            'jitter': {'addon': 'respawn', 'position_range': [1.5, 1.5, 0.2]},
                            'pose': {'addon': 'object_state_sensor'}, 'push': {'addon': 'external_force'}}}}
g, c = pair(crowd, 33); report('spawn_multiple', g, c, 100, every=10)
g, c = pair(crowd, 33, residual_threshold=1e-13); report('spawn_multiple converged', g, c, 100, every=10)
