"""How far do the GPU and the oracle stay together on from_the_readme once things touch?"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import torch
import test_parity_gpu as T
gpu, cpu = T.make_pair('readme', 3)
d = gpu.sim.enable_diagnostics()
lo, hi = T.action_bounds(gpu)
gen = torch.Generator().manual_seed(0)
for i in range(int(sys.argv[1]) if len(sys.argv) > 1 else 120):
    act = (lo + (hi - lo) * torch.rand((3, lo.numel()), generator=gen)) * 0.2
    gpu.sim.step(gpu._all_slots, act.to(gpu.device)); cpu.sim.step(cpu._all_slots, act)
    if i % 10 == 9:
        a, b = gpu.sim.get_state(), cpu.sim.get_state()
        print(i, 'obs', float((gpu.sim.obs.cpu() - cpu.sim.obs).abs().max()), 'state', float(abs(a - b).max()), 'contacts gpu', d[:, 0].tolist(), 'cpu', [cpu.sim.contacts(e) for e in range(3)], 'iters', d[:, 1].tolist(), flush=True)
