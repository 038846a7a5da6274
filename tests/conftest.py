import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run by the driver with -m gpu)")


@pytest.fixture(scope="session")
def co():
    """C++ CPU oracle (test infrastructure; built on demand with g++)."""
    from oracle import coracle
    coracle.build()
    coracle.lib()
    return coracle


@pytest.fixture(scope="session")
def golden():
    def load(name):
        return np.load(os.path.join(GOLDEN, name + ".npz"))
    return load


@pytest.fixture(scope="session")
def gpu_ctx():
    """libpcdhip.so context on cuda:0.  No fallback: a missing library or GPU is an error, not a skip."""
    # torch first, as in bench.py: its wheel carries its own HIP runtime, and whichever runtime is loaded first owns the
    # device for the process (the device-resident exchange test hands torch tensors to the library)
    import torch
    torch.zeros(1, device="cuda:0")
    from pcd_amd import capi
    ctx = capi.Context(0)
    yield ctx
    ctx.close()
