"""Diagnostic: r2d2_maze with the sweeps run to convergence, HIP vs oracle, per step and per state column."""
import copy, os, sys
import numpy as np, torch, yaml
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
from diy_gym_amd import DIYGym
from diy_gym_amd.config import Configuration
from oracle_backend import OracleBackend
from test_parity_gpu import action_bounds

def run(cap, thr, B=19, steps=40):
    tree = yaml.safe_load(open(os.path.join(ROOT, 'examples/r2d2_maze/r2d2_maze.yaml')))
    tree['solver_iterations'] = cap
    eng = dict(residual_threshold=thr)
    gpu = DIYGym(Configuration.from_dict('m', copy.deepcopy(tree)), num_envs=B, device='cuda:0', seed=5, engine=eng)
    cpu = DIYGym(Configuration.from_dict('m', copy.deepcopy(tree)), num_envs=B, seed=5, backend_factory=OracleBackend, engine=eng)
    d = gpu.sim.enable_diagnostics()
    lo, hi = action_bounds(gpu); gen = torch.Generator().manual_seed(0)
    print('cap', cap, 'thr', thr, 'lanes', gpu.sim.lanes, 'state_dim', gpu.layout.state_dim)
    for i in range(steps):
        act = (lo + (hi - lo) * torch.rand((B, lo.numel()), generator=gen)) * 10.0
        gpu.sim.step(gpu._all_slots, act.to(gpu.device)); cpu.sim.step(cpu._all_slots, act)
        a, b = gpu.sim.get_state(), cpu.sim.get_state()
        df = np.abs(a - b)
        e, k = np.unravel_index(df.argmax(), df.shape)
        print(i, 'max diff %.3e at env %d col %d (gpu %.5f cpu %.5f)' % (df.max(), e, k, a[e, k], b[e, k]),
              'iters gpu', d[:4, 1].tolist(), 'cpu', [cpu.sim.iterations(q) for q in range(4)], 'contacts', d[:4, 0].tolist())
    return gpu, cpu

for cap, thr in ((150, 1e-7), (150, 1e-13), (1000, 1e-13), (4000, 1e-13), (4000, 1e-10)):
    run(cap, thr)
