"""GPU: the LDS-mailbox variants of the lane-split 753-bit extension fields (what the accumulate / tail / single-item kernels of the
MNT4-753 and MNT6-753 G2 MSMs compute in) against the plain lane-split variants, under random active-item masks and divergent
add / double / cancel / infinity branches (tests/gpucheck/mailbox_check.hip), plus the round-3 reproducer of the dropped copy
(one item, both operands finite, result as stored).  The plain form itself is pinned to the oracle by tests/test_gpu_msm.py."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GC = os.path.join(ROOT, "tests", "gpucheck")
NAMES = ["mul", "sqr", "madd", "add", "dbl+add", "add(P,P)", "add(P,-P)", "infinity"]


def _lib():
    so = os.path.join(GC, "libgpucheck.so")
    # built by __graft_entry__.build() where the sources' timestamps are right; on the GPU box the snapshot's timestamps are not, so `make`
    # would recompile for three minutes: only a missing library is built here
    if not os.path.exists(so):
        subprocess.check_call(["make", "-C", GC, "libgpucheck.so"], stdout=subprocess.DEVNULL)
    lib = C.CDLL(so)
    lib.gc_mailbox_check.argtypes = [C.c_int, C.c_int, C.c_int, C.c_uint32, C.c_void_p]
    lib.gc_mailbox_merge.argtypes = [C.c_int, C.c_uint32]
    return lib


def test_product_flags_disable_copyprop():
    """the product (and this harness) must be built without MachineCopyPropagation: ROCm 7.2 drops a live copy with it"""
    for mk in (os.path.join(ROOT, "pcd_amd", "csrc", "Makefile"), os.path.join(GC, "Makefile")):
        text = open(mk).read()
        assert "-mllvm -disable-copyprop" in text, mk


@pytest.mark.gpu
@pytest.mark.parametrize("which,lanes", [(3, 63), (3, 3), (3, 30), (3, 9), (2, 64), (2, 2), (2, 22)])
def test_mailbox_equals_plain_under_masks(which, lanes):
    lib = _lib()
    for seed in (12345, 777, 20261003):
        out = np.zeros(8, dtype=np.uint32)
        rc = lib.gc_mailbox_check(which, 24, lanes, seed, out.ctypes.data_as(C.c_void_p))
        assert rc == 0
        assert not out.any(), {n: int(v) for n, v in zip(NAMES, out) if v}


@pytest.mark.gpu
@pytest.mark.parametrize("which", [3, 2])
def test_single_item_addition_as_stored(which):
    lib = _lib()
    for seed in (4242, 1, 99991, 31337):
        assert lib.gc_mailbox_merge(which, seed) == 0
