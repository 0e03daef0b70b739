import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, 'tests')):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')


@pytest.fixture(scope='session')
def oracle_abi():
    """The C oracle (same C ABI, host pointers) wrapped for torch-CPU tensors."""
    from abi_call import Abi
    from oracle import cbind
    return Abi(cbind.load(), 'cpu')


@pytest.fixture(scope='session')
def hip_abi():
    """libacgan_hip.so on cuda:0; fails loudly (no fallback) when the library or the GPU is missing."""
    import torch
    from abi_call import Abi
    from action_conditioned_gans_amd import _lib
    assert torch.cuda.is_available(), 'gpu-marked test needs a GPU'
    return Abi(_lib.get(), 'cuda:0')
