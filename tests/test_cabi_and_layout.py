"""The C-ABI library loads and exports every symbol include/diygym_hip.h declares
(no compute without a GPU), and the product never touches the oracle."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(ROOT, 'diy_gym_amd', 'csrc', 'libdiygym_hip.so')


def declared_symbols():
    text = open(os.path.join(ROOT, 'include', 'diygym_hip.h')).read()
    text = re.sub(r'/\*.*?\*/', '', text, flags=re.S)
    return sorted(set(re.findall(r'\b(dg_[a-z_0-9]+)\s*\(', text)))


def test_header_and_binding_agree():
    from diy_gym_amd import backend
    assert declared_symbols() == sorted(backend.SYMBOLS)


@pytest.mark.skipif(not os.path.isfile(LIB), reason='run __graft_entry__.build() first')
def test_library_exports_every_declared_symbol():
    lib = ctypes.CDLL(LIB)
    for name in declared_symbols():
        assert hasattr(lib, name), name
    lib.dg_version.restype = ctypes.c_int32
    assert lib.dg_version() >= 3
    from diy_gym_amd import backend
    backend.load_library()


def test_backend_fails_loudly_without_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip('GPU present')
    from diy_gym_amd import DIYGym
    with pytest.raises(RuntimeError, match='no CPU fallback'):
        DIYGym(os.path.join(ROOT, 'tests', 'golden', 'basic_env_nocam.yaml'), num_envs=2)


def test_product_never_references_the_oracle():
    pkg = os.path.join(ROOT, 'diy_gym_amd')
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith(('.py', '.h', '.hip', '.cpp')):
                text = open(os.path.join(dirpath, f), errors='ignore').read()
                assert 'dgsim_oracle' not in text and 'oracle_backend' not in text and 'dgo_' not in text, os.path.join(dirpath, f)


def test_layout_required_by_the_contract():
    for p in ['bench.py', '__graft_entry__.py', 'DESIGN.md', 'INTEGRATION.md', 'include/diygym_hip.h', 'oracle/dgsim_oracle.c',
              'tests/golden', 'profiles', 'diy-gym_amd']:
        assert os.path.exists(os.path.join(ROOT, p)), p
