import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    """The CPU oracle (oracle/oracle.c through oracle/oracle.py): the checker, never the thing under test."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import oracle as O

    O.build()
    return O


@pytest.fixture(scope="session")
def dmx():
    import dmx_compressor_amd as d

    return d


@pytest.fixture(scope="session")
def cuda(dmx):
    import torch

    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    if not os.path.exists(dmx.LIB_PATH):
        pytest.fail(f"{dmx.LIB_PATH} missing on a GPU box: run __graft_entry__.build() — no fallback exists")
    return torch.device("cuda:0")
