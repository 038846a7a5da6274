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
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    import kat_extra
    extra = None
    cache, bad, n, advisory, seen = {}, [], 0, [], set()
    for ln in open(path):
        name, dtype, shape, data = ln.split()
        f, key = name.split(".", 1)
        if f.startswith("x_"):   # the round-4 extension: expected values recomputed from seeds by the oracle (tools/kat_extra.py)
            if extra is None:
                extra = kat_extra.expected()
            want = extra.get(name)
            if want is None:
                bad.append(name + " (no expected value)")
                continue
        else:
            if f not in cache:
                cache[f] = np.load(os.path.join(ROOT, "tests", "golden", f + ".npz"))
            want = cache[f][key]
        seen.add(name)
        got = np.array([int(v, 16) for v in data.split(",")], dtype=np.dtype(dtype)).reshape(want.shape)
        n += 1
        if not np.array_equal(got, np.asarray(want).astype(got.dtype)):
            why = next((w for k, w in kat_extra.ADVISORY.items() if name.endswith(k)), None)
            (advisory if why else bad).append(name if not why else f"{name}: differs -- {why}")
    if extra is not None:
        missing = sorted(k for k in extra if k not in seen)
        if missing:
            bad.append("not written by kat.rs: " + ", ".join(missing[:6]))
    # ... and every OUTPUT array of the golden files (tools/kat_export.golden_outputs: what is not handed over as an input, plus the key's
    # queries, which are inputs of the proof KAT and outputs of the generate_parameters KAT)
    import kat_export
    for name in kat_export.golden_outputs():
        if name not in seen:
            bad.append(name + " (a golden output kat.rs did not write)")
    for a in advisory:
        print("NOTE:", a)
    print(f"{n - len(bad) - len(advisory)} of {n} arrays computed by arkworks equal the golden vectors / the oracle")
    if bad:
        print("MISMATCH:", ", ".join(bad))
        return 1
    return 0


if __name__ == "__main__":
    sys.exit(main())
