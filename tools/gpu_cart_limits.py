#!/usr/bin/env python3
"""GPU diagnostic: cart_tree driven into its joint limits (top of the action range), kernels against the oracle step by step:
which observation column / state entry differs, from which step on, with and without the starting guesses."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import test_parity_gpu as T  # noqa: E402

for engine in (dict(residual_threshold=1e-13), dict(residual_threshold=1e-13, limit_guess=0.0), dict(residual_threshold=1e-13, motor_guess=0.0), dict()):
    gpu, cpu = T.make_pair('cart_tree', 9, **engine)
    lo, hi = T.action_bounds(gpu)
    act = hi[None].repeat(9, 1)
    print('engine', engine, 'lanes', gpu.sim.lanes, 'kernel', gpu.sim.kernel_name)
    d = gpu.sim.enable_diagnostics()
    for i in range(60):
        gpu.sim.step(gpu._all_slots, act.to(gpu.device)); cpu.sim.step(cpu._all_slots, act)
        do = (gpu.sim.obs.cpu() - cpu.sim.obs).abs()
        ds = np.abs(T.phys_state(gpu) - T.phys_state(cpu))
        if i % 6 == 5 or (i > 8 and i < 20):
            e, c = np.unravel_index(int(do.argmax()), do.shape); es, cs = np.unravel_index(int(ds.argmax()), ds.shape)
            print('  step %2d obs diff %.3e at env %d col %d (gpu %.5f cpu %.5f) | state diff %.3e at env %d entry %d (gpu %.5f cpu %.5f) | iters gpu %s cpu %s' % (
                i, float(do.max()), e, c, float(gpu.sim.obs[e, c]), float(cpu.sim.obs[e, c]), float(ds.max()), es, cs, T.phys_state(gpu)[es, cs], T.phys_state(cpu)[es, cs],
                d[:3, 1].tolist(), [cpu.sim.iterations(k) for k in range(3)]))
