#!/usr/bin/env python3
"""Compare what real arkworks computed (rust/tests/kat_outputs.txt, written by `cargo test --test kat`) with the golden vectors
this repository's oracle and HIP path are pinned to (tests/golden/*.npz).  Agreement on every line turns DESIGN.md's "parity
unpinned" into "pinned against upstream".  See tools/kat_export.py for the three-step recipe."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    path = os.path.join(ROOT, "rust", "tests", "kat_outputs.txt")
    if not os.path.exists(path):
        raise SystemExit(f"{path} is missing: run `cargo test --release --test kat` in rust/ first (needs a Rust toolchain)")
    cache, bad, n = {}, [], 0
    for ln in open(path):
        name, dtype, shape, data = ln.split()
        f, key = name.split(".", 1)
        if f not in cache:
            cache[f] = np.load(os.path.join(ROOT, "tests", "golden", f + ".npz"))
        want = cache[f][key]
        got = np.array([int(v, 16) for v in data.split(",")], dtype=np.dtype(dtype)).reshape(want.shape)
        n += 1
        if not np.array_equal(got, want):
            bad.append(name)
    print(f"{n - len(bad)} of {n} arrays computed by arkworks equal the golden vectors")
    if bad:
        print("MISMATCH:", ", ".join(bad))
        return 1
    return 0


if __name__ == "__main__":
    sys.exit(main())
