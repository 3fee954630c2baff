#!/usr/bin/env python3
"""GPU diagnostic for bench.py's legs: per-replay wall time of a leg's graph right after its world is built (clock ramp?
state evolution?), next to the kernel-only graph.  python tools/gpu_leg_debug.py reference_solver_settings drone_pilot"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import bench  # noqa: E402

dev = torch.device('cuda', 0)
torch.cuda.set_device(0)
for name in sys.argv[1:]:
    leg = bench.make_leg(name, dev)
    leg.eager(48); torch.cuda.synchronize()
    leg.capture(16)
    rows = []
    for k in range(40):
        torch.cuda.synchronize(); t0 = time.perf_counter(); leg.graph.replay(); torch.cuda.synchronize(); a = (time.perf_counter() - t0) / leg.R * 1e3
        torch.cuda.synchronize(); t0 = time.perf_counter(); leg.graph_kernel.replay(); torch.cuda.synchronize(); b = (time.perf_counter() - t0) / leg.R * 1e3
        rows.append((a, b))
    print(name, 'ms per step by replay (loop graph / kernel-only graph):', ' '.join('%.3f/%.3f' % r for r in rows[:6]), '...', ' '.join('%.3f/%.3f' % r for r in rows[-4:]))
    t = leg.timed(304)
    print(name, 'timed 304 steps: %.4f ms per step; episodes finished %d' % (t / 304 * 1e3, int(leg.sim.state[1, :leg.B].sum().item()) - leg.B), flush=True)
    leg.close()
