#!/usr/bin/env python3
"""Known-answer-test bridge to the real arkworks (for whoever has `cargo`; there is none in the build container).

    python tools/kat_export.py          # tests/golden/*.npz  ->  rust/tests/kat_inputs.txt   (inputs only, as text)
    (cd rust && cargo test --release --test kat -- --nocapture)   # arkworks computes: rust/tests/kat_outputs.txt
    python tools/check_kat.py           # compares kat_outputs.txt with the expected values in tests/golden/*.npz

Text format, one array per line:  <file>.<key> <dtype> <dim0>x<dim1>.. <hex,hex,...>   (row-major, u64 / u32 / u8 as hex)."""
import os
import re

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden")
# what the Rust side needs as INPUT (expected outputs stay in the npz files and are compared by check_kat.py)
INPUT_KEYS = {
    "fields": lambda k: re.fullmatch(r"f\d_[ab]", k) is not None,
    "msm": lambda k: re.fullmatch(r"c\d_g\d_(bases|inf|scalars)", k) is not None,
    "fft": lambda k: k.endswith("_in"),
    "pairing": lambda k: k.endswith(("_p", "_q")),
    # (the key's queries are INPUTS of the proof KAT and, since round 4, also OUTPUTS of the generate_parameters KAT: kat.rs writes them back)
    "groth16": lambda k: not k.endswith(("_h", "_proof", "_proof_inf")),
    "wire": lambda k: k.endswith(("_xy", "_inf")),
}


def is_setup_output(k):
    """groth16.npz keys that are inputs of the proof KAT AND outputs of the generate_parameters KAT"""
    stem = k.split("_", 1)[1]
    return stem.endswith(("_query", "_query_inf")) or stem in ("alpha_g1", "beta_g1", "delta_g1", "beta_g2", "gamma_g2", "delta_g2",
                                                                "gamma_abc_g1", "gamma_abc_g1_inf")


def golden_outputs():
    """names `<file>.<key>` of every array in tests/golden/*.npz that arkworks must reproduce (rust/tests/kat.rs writes one line each)"""
    out = []
    for f, is_input in INPUT_KEYS.items():
        g = np.load(os.path.join(GOLDEN, f + ".npz"))
        out += [f"{f}.{k}" for k in g.files if not is_input(k) or (f == "groth16" and is_setup_output(k))]
    return out


def line(name, a):
    a = np.ascontiguousarray(a)
    shape = "x".join(str(d) for d in a.shape) or "1"
    return f"{name} {a.dtype} {shape} " + ",".join(format(int(v), "x") for v in a.reshape(-1))


def main():
    out = []
    for f, want in INPUT_KEYS.items():
        g = np.load(os.path.join(GOLDEN, f + ".npz"))
        for k in g.files:
            if want(k):
                out.append(line(f"{f}.{k}", g[k]))
    # what the product only recalls of upstream (mixed-radix domains, the domain choice past the 2-adicity, FixedBaseMSM, constants):
    # inputs from seeds, expected values recomputed by tools/check_kat.py (tools/kat_extra.py)
    import sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    import kat_extra
    for name, arr in kat_extra.inputs():
        out.append(line(name, arr))
    path = os.path.join(ROOT, "rust", "tests", "kat_inputs.txt")
    with open(path, "w") as fh:
        fh.write("\n".join(out) + "\n")
    print(f"wrote {path}: {len(out)} arrays, {os.path.getsize(path) // 1024} KiB")


if __name__ == "__main__":
    main()
