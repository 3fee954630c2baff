"""Committed golden vectors (tests/golden/vectors.npz, made by tests/golden/make_vectors.py from the oracle):
the oracle must keep reproducing them (CPU) and the HIP path must match them (GPU) -- independent of the oracle
binary built on the box."""
import os
import sys

import numpy as np
import pytest
import torch  # noqa: F401

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, 'golden'))
import make_vectors  # noqa: E402

V = np.load(os.path.join(HERE, 'golden', 'vectors.npz'))


@pytest.mark.parametrize('name', sorted(make_vectors.SCENES))
def test_oracle_reproduces_golden(name):
    from oracle_backend import OracleBackend
    r = make_vectors.run(name, backend_factory=OracleBackend)
    assert np.array_equal(r['actions'], V[name + '/actions'])
    assert np.allclose(r['obs'], V[name + '/obs'], rtol=0, atol=1e-6)
    assert np.allclose(r['state'], V[name + '/state'], rtol=1e-9, atol=1e-9)
    assert np.array_equal(r['term'], V[name + '/term'])


@pytest.mark.parametrize('name', make_vectors.REF_SCENES)
def test_oracle_reproduces_golden_at_reference_solver_settings(name):
    """The second set: motor rows started from zero, contact rows warm started with 0.85 (make_vectors.REFERENCE_SETTINGS)."""
    from oracle_backend import OracleBackend
    r = make_vectors.run(name, backend_factory=OracleBackend, engine=make_vectors.REFERENCE_SETTINGS)
    assert np.allclose(r['obs'], V['ref/' + name + '/obs'], rtol=0, atol=1e-6)
    assert np.allclose(r['state'], V['ref/' + name + '/state'], rtol=1e-9, atol=1e-9)
    assert np.array_equal(r['term'], V['ref/' + name + '/term'])


@pytest.mark.gpu
@pytest.mark.parametrize('name,tol', [('marbles', 2e-3), ('ur_ik', 5e-4), ('ur_joint', 5e-4), ('cart_tree', 5e-3), ('maze', 5e-3), ('readme', 2e-3),
                                      ('touching', 3e-3), ('pressing', 2e-3), ('gripper', 3e-3)])
def test_hip_matches_golden_at_reference_solver_settings(name, tol):
    r = make_vectors.run(name, device='cuda:0', engine=make_vectors.REFERENCE_SETTINGS)
    obs, ref = r['obs'], V['ref/' + name + '/obs']
    if name == 'cart_tree':  # efforts (columns 6..8) are only determined to the solver's residual threshold
        keep = np.ones(obs.shape[-1], dtype=bool); keep[6:9] = False
        obs, ref = obs[..., keep], ref[..., keep]
    if obs.size:
        assert np.abs(obs - ref).max() < tol
    a, b = r['state'][:, :V['ref/' + name + '/state'].shape[1]], V['ref/' + name + '/state']
    if name == 'maze':   # (no sensors in the scene: the state is what there is to compare; velocities at the iteration cap are loose)
        assert np.abs(a[:, :9] - b[:, :9]).max() < 5e-3
    assert np.array_equal(r['term'], V['ref/' + name + '/term'])


@pytest.mark.gpu
@pytest.mark.parametrize('name,tol', [('marbles', 2e-3), ('drone', 2e-3), ('ur_ik', 5e-4), ('ur_joint', 5e-4), ('cart_tree', 5e-3), ('maze', 5e-3),
                                      ('admittance', 3e-3), ('readme', 2e-3), ('touching', 3e-3), ('pressing', 2e-3), ('gripper', 3e-3)])
def test_hip_matches_golden(name, tol):
    r = make_vectors.run(name, device='cuda:0')
    obs, ref = r['obs'], V[name + '/obs']
    # gripper: the observation includes motor efforts of O(5 N m) (relative difference 2e-4); positions agree to 5e-5
    if name == 'cart_tree':  # efforts (columns 6..8) are only determined to the solver's residual threshold
        keep = np.ones(obs.shape[-1], dtype=bool); keep[6:9] = False
        obs, ref = obs[..., keep], ref[..., keep]
    assert np.abs(obs - ref).max() < tol
    assert np.array_equal(r['term'], V[name + '/term'])
