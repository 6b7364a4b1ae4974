import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    import pyoracle
    pyoracle.build()
    return pyoracle


@pytest.fixture(scope="session")
def caller():
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    import lofreq_amd as la
    c = la.SnvCaller(0)
    yield c
    c.close()
