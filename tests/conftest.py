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


AT_SIZE = os.path.join(GOLDEN, "at_size.npz")


class Expect:
    """Expected values of the at-size tests (BASELINE configs at their stated sizes).

    tests/golden/at_size.npz holds what the CPU oracle computes for each case -- affine MSM results, proof bytes, digests of long
    vectors -- written in the build container by tests/golden/gen_at_size.py from the same seeded case builders the tests use
    (each test module lists them in AT_SIZE).  `expect(key, fn)` returns the stored value; with PCD_RECOMPUTE=1, or for a key the
    file does not hold, it runs `fn` (the oracle) and, when both exist, requires them to be equal -- so the fixture can be
    re-derived on any box, and a stale one cannot pass.  Values: one ndarray or a tuple of ndarrays."""

    def __init__(self, path=AT_SIZE):
        self.path = path
        self.recompute = os.environ.get("PCD_RECOMPUTE") == "1"
        self._npz = np.load(path) if os.path.exists(path) else None
        self.served, self.computed = [], []

    def stored(self, key):
        z = self._npz
        if z is None:
            return None
        if key in z.files:
            return z[key]
        if key + "/len" in z.files:
            return tuple(z[f"{key}/{i}"] for i in range(int(z[key + "/len"])))
        return None

    def __call__(self, key, fn):
        have = self.stored(key)
        if have is not None and not self.recompute:
            self.served.append(key)
            return have
        val = fn()
        self.computed.append(key)
        if have is not None:
            a, b = (have, val) if isinstance(have, tuple) else ((have,), (val,))
            assert len(a) == len(b) and all(np.array_equal(x, np.asarray(y)) for x, y in zip(a, b)), f"tests/golden/at_size.npz[{key}] differs from the oracle"
        return val

    @staticmethod
    def digest(a):
        """sha256 of an array's bytes as uint8[32] (long vectors -- a witness map's h is 42 MB -- are stored and compared as digests)"""
        import hashlib
        return np.frombuffer(hashlib.sha256(np.ascontiguousarray(a).tobytes()).digest(), dtype=np.uint8).copy()


@pytest.fixture(scope="session")
def expect():
    return Expect()
