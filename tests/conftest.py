import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def ciao():
    """The product package, loaded under its importable alias."""
    import ciao_loader
    return ciao_loader.load()


@pytest.fixture(scope="session")
def ctx(ciao):
    """One library context on cuda:0 for the whole GPU session (fails loudly without the HIP extension / a GPU)."""
    import torch
    assert torch.cuda.is_available(), "GPU tests need a GPU"
    from ciaoalgorithms_jl_amd.device import Context
    c = Context(0)
    yield c
    c.synchronize()
    c.close()
