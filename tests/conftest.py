import os
import sys

import pytest

try:
    # PyTorch-ROCm bundles its own copy of the HIP runtime.  Tests that use torch tensors next to libfpx (the sharded
    # path) need ONE runtime in the process: whichever library loads first provides it, and torch only finds the GPU
    # when it is its own.  Importing torch before libfpx (as bench.py does) guarantees that for any test selection.
    import torch  # noqa: F401
except Exception:      # torch is optional for everything but the sharded tests
    pass

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def orc():
    from oracle import oracle
    oracle.build()
    oracle.lib()
    return oracle
