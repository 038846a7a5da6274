#!/usr/bin/env python3
"""Writes tests/golden/at_size.npz: what the CPU oracle computes for every at-size test case of the GPU suite (BASELINE configs[1]-[4]
at their stated sizes, the 2^20 MSMs, the 2^22 merge proof, the witness-like 2^20 proof).

Run in the BUILD container (no GPU needed; 8 cores: about an hour, most of it the 2^22 MNT4-753 proof):

    python tests/golden/gen_at_size.py            # computes the keys the file does not hold yet
    python tests/golden/gen_at_size.py --force    # everything again
    python tests/golden/gen_at_size.py KEY ...    # these keys only

The cases are NOT restated here: each GPU test module lists its case builders in AT_SIZE (key -> builder(co) -> (seeded inputs, thunk)),
and the tests obtain the thunk's value through conftest.Expect -- from this file, or (PCD_RECOMPUTE=1) by running the thunk on the spot,
in which case the stored value must agree.  The fixture is DATA derived from the oracle (affine points, proof bytes, sha256 digests of
long vectors); the oracle itself is pinned at small sizes by tests/golden/*.npz and the KAT bridge (tests/test_kat_bridge.py), and
tests/test_oracle_golden.py::test_at_size_fixture_spot_check re-derives the cheap entries in the CPU suite."""
import importlib
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
MODULES = ("test_gpu_at_size", "test_gpu_config4", "test_gpu_witness_like")
OUT = os.path.join(ROOT, "tests", "golden", "at_size.npz")


def cases():
    out = {}
    for m in MODULES:
        mod = importlib.import_module(m)
        for k, b in mod.AT_SIZE.items():
            assert k not in out, k
            out[k] = b
    return out


def main():
    from oracle import coracle as co
    co.build(); co.lib()
    args = [a for a in sys.argv[1:] if not a.startswith("-")]
    force = "--force" in sys.argv
    have = dict(np.load(OUT)) if os.path.exists(OUT) and not force else {}
    all_cases = cases()
    for key, build in all_cases.items():
        if args and key not in args:
            continue
        if not args and (key in have or key + "/len" in have):
            continue
        t0 = time.time()
        _, want = build(co)
        t1 = time.time()
        val = want()
        for k in [k for k in have if k == key or k.startswith(key + "/")]:
            del have[k]
        if isinstance(val, tuple):
            have[key + "/len"] = np.array(len(val))
            for i, a in enumerate(val):
                have[f"{key}/{i}"] = np.asarray(a)
        else:
            have[key] = np.asarray(val)
        np.savez_compressed(OUT, **have)   # (after every case: an interrupted run keeps what it has)
        print(f"{key}: inputs {t1 - t0:.1f} s, oracle {time.time() - t1:.1f} s, threads {os.cpu_count()}", flush=True)
    stale = [k for k in have if k.split("/")[0] not in all_cases]
    assert not stale, f"keys without a case builder: {stale}"


if __name__ == "__main__":
    main()
