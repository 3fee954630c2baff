import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np, torch
import test_parity_gpu as T
name = sys.argv[1] if len(sys.argv) > 1 else 'cart_tree'
B = int(sys.argv[2]) if len(sys.argv) > 2 else 37
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 60
gpu, cpu = T.make_pair(name, B)
gpu.sim.enable_diagnostics()
gen = torch.Generator().manual_seed(0)
lo, hi = T.action_bounds(gpu)
for s in range(steps):
    act = (lo + (hi - lo) * torch.rand((B, lo.numel()), generator=gen))
    gpu.sim.step(gpu._all_slots, act.to(gpu.device)); cpu.sim.step(cpu._all_slots, act)
    d = (gpu.sim.obs.cpu() - cpu.sim.obs).abs()
    e, c = divmod(int(d.argmax()), d.shape[1])
    ds = np.abs(gpu.sim.get_state() - cpu.sim.get_state())
    es, cs = divmod(int(ds.argmax()), ds.shape[1])
    diag = gpu.sim.diag.cpu().numpy()
    print(s, 'obs err %.3g env %d col %d (gpu %.5f cpu %.5f) | state err %.3g env %d idx %d | contacts gpu %d cpu %d iters gpu %d cpu %d' % (
        float(d.max()), e, c, float(gpu.sim.obs[e, c]), float(cpu.sim.obs[e, c]), ds.max(), es, cs, diag[es, 0], cpu.sim.contacts(es), diag[es, 1], cpu.sim.iterations(es)), flush=True)
