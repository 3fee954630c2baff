"""Prints HIP-vs-oracle differences per state block for the loosely asserted scenes, to set test tolerances from data."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np, torch
import test_parity_gpu as T
for name, B, steps, scale, engine in [('maze', 19, 40, 10.0, {}), ('cart_tree', 37, 30, 1.0, {}), ('cart_tree', 37, 30, 1.0, dict(residual_threshold=1e-13)),
                                      ('cart_tree', 37, 12, 1.0, dict(residual_threshold=1e-13)), ('gripper', 5, 40, 0.5, {}), ('touching', 37, 30, 0.3, {})]:
    gpu, cpu = T.make_pair(name, B, **engine)
    w = T.rollout(gpu, cpu, steps, scale=scale)
    a, b = gpu.sim.get_state(), cpu.sim.get_state()
    L = gpu.layout
    print(name, engine, 'steps', steps, w)
    for body in range(L.n_bodies):
        so = L.body_state_off[body]
        if so < 0: continue
        n = 7 if L.body_fixed[body] else 13
        print('  body %d base pose %.2e' % (body, np.abs(a[:, so:so + 7] - b[:, so:so + 7]).max()), 'vel %.2e' % (np.abs(a[:, so + 7:so + 13] - b[:, so + 7:so + 13]).max() if n == 13 else 0))
    qs = [o for o in L.link_state_off]
    if qs:
        dq = np.abs(a[:, qs] - b[:, qs]); dqd = np.abs(a[:, [o + 1 for o in qs]] - b[:, [o + 1 for o in qs]])
        print('  joints q max %.2e per joint %s' % (dq.max(), np.array2string(dq.max(0), precision=1)))
        print('  joints qd max %.2e per joint %s | |qd| max %.1f' % (dqd.max(), np.array2string(dqd.max(0), precision=1), np.abs(b[:, [o + 1 for o in qs]]).max()))
