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


def pytest_terminal_summary(terminalreporter):
    """largest p-value deviation observed per regime (tests/util.py::assert_pvalue_close)"""
    try:
        import util
    except ImportError:
        return
    rows = [(k, v) for k, v in util.PV_ERR_MAX.items() if v[1]]
    if rows:
        terminalreporter.write_line("p-value parity, max |dlog p| observed (tolerance 1e-10 up to |log p| = 600; beyond, the noise "
                                    "bound of the reference's own arithmetic, %g * ulp(|log p|) * N):" % util.PV_NOISE_A)
        for k, v in rows:
            terminalreporter.write_line("    %-16s %.3g over %d records%s" % (
                k, v[0], v[1], (", at most %.2f of the bound" % v[2]) if len(v) > 2 else ""))
