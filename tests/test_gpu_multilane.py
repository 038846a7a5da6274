"""GPU: the multi-lane group operations of the bucket reduction (round 6) -- EC4 (four lanes per addition / doubling, prime-field groups) and
EC2 (two lanes, and two HALVES of L lanes for the lane-split Fq2 / Fq3 groups, inlined 298-bit and mailbox 753-bit forms) -- against the
plain one-item operations, under random active-item masks and the branches a bucket reduction rarely meets: equal operands (doubling inside
an addition), opposite operands (cancellation), identities on either side (tests/gpucheck/multilane_check.hip).  The plain forms are pinned
to the oracle by tests/test_gpu_msm.py and tests/test_hostcheck.py."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GC = os.path.join(ROOT, "tests", "gpucheck")
NAMES = ["add", "dbl", "dbl(add(dbl))", "add(P,P)", "add(P,-P)", "identities"]
WHICH = {0: "G1-298 EC4", 1: "G1-298 EC2", 2: "G1-753 mailbox EC4", 3: "G1-753 mailbox EC2", 4: "Fq2-298 halves", 5: "Fq3-298 halves",
         6: "Fq2-753 mailbox halves", 7: "Fq3-753 mailbox halves"}


def _lib():
    so = os.path.join(GC, "libgpucheck_ml.so")
    if not os.path.exists(so):   # (built by __graft_entry__.build(); only a missing library is built here: see tests/test_gpu_mailbox.py)
        subprocess.check_call(["make", "-C", GC, "libgpucheck_ml.so"], stdout=subprocess.DEVNULL)
    lib = C.CDLL(so)
    lib.gc_multilane_check.argtypes = [C.c_int, C.c_int, C.c_uint32, C.c_void_p]
    return lib


@pytest.mark.gpu
@pytest.mark.parametrize("which", sorted(WHICH))
def test_multilane_ops_equal_plain_under_masks(which):
    lib = _lib()
    for seed in (2026, 31337, 5):
        out = np.zeros(8, dtype=np.uint32)
        rc = lib.gc_multilane_check(which, 32, seed, out.ctypes.data_as(C.c_void_p))
        assert rc == 0, WHICH[which]
        assert out[7] > 0, "the check kernel did not run"
        assert not out[:6].any(), (WHICH[which], {n: int(v) for n, v in zip(NAMES, out) if v})
