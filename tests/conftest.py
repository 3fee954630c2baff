import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')


@pytest.fixture(scope='session', autouse=True)
def _built_oracle():
    """The oracle is test infrastructure: build it on demand (gcc, < 5 s)."""
    so = os.path.join(ROOT, 'oracle', 'libdgsim_oracle.so')
    src = os.path.join(ROOT, 'oracle', 'dgsim_oracle.c')
    if not os.path.isfile(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(['make', '-C', os.path.join(ROOT, 'oracle')], stdout=subprocess.DEVNULL)
    return so
