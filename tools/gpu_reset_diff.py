"""Prints the GPU-vs-oracle state difference after the constructor's reset for each parity config."""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'tests'))
import numpy as np
import test_parity_gpu as T

for name in T.CONFIGS:
    gpu, cpu = T.make_pair(name, 5)
    a, b = np.asarray(gpu.sim.get_state()), np.asarray(cpu.sim.get_state())
    d = np.abs(a - b)
    k = np.unravel_index(np.argmax(d), d.shape)
    v = d - (5e-4 + 1e-4 * np.abs(b))
    for kk in zip(*np.nonzero(v > 0)):
        print('   violates', kk, a[kk], b[kk])
    print('%-14s max|d| %.3e at %s gpu %.6f cpu %.6f  rel-ok %s' % (name, d.max(), k, a[k], b[k], np.allclose(a, b, rtol=1e-4, atol=5e-4)))
