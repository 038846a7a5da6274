"""CPU: the known-answer bridge to real arkworks (tools/kat_export.py -> rust/tests/kat.rs -> tools/check_kat.py) cannot run here (no Rust
toolchain), but its Python half can be exercised end to end: export the inputs, play the Rust side's part with the oracle's own values
(every line kat.rs is written to emit: the same names, dtypes and shapes), and let check_kat.py compare -- it must accept that file and
reject a corrupted one.  This pins the file format and the name set both sides agree on."""
import os
import re
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))


def _line(name, a):
    a = np.ascontiguousarray(a)
    shape = "x".join(str(d) for d in a.shape) or "1"
    return f"{name} {a.dtype} {shape} " + ",".join(format(int(v), "x") for v in a.reshape(-1))


def test_kat_round_trip_with_the_oracle_in_place_of_arkworks(co):
    import kat_extra
    out_path = os.path.join(ROOT, "rust", "tests", "kat_outputs.txt")
    saved = open(out_path).read() if os.path.exists(out_path) else None
    try:
        subprocess.check_call([sys.executable, os.path.join(ROOT, "tools", "kat_export.py")], stdout=subprocess.DEVNULL)
        names_in = {ln.split(" ", 1)[0] for ln in open(os.path.join(ROOT, "rust", "tests", "kat_inputs.txt"))}
        src = open(os.path.join(ROOT, "rust", "tests", "kat.rs")).read()
        # every input kat.rs indexes by a literal name exists in the export (format!-built names are checked through the x_ prefixes below)
        for lit in re.findall(r'a\[&?"([a-z_0-9.]+)"\]', src):
            assert lit in names_in, lit
        exp = kat_extra.expected()
        import kat_export
        golden = {}
        for name in kat_export.golden_outputs():   # EVERY output array of the golden files, as arkworks would write it (names file.key)
            f, k = name.split(".", 1)
            golden[name] = np.load(os.path.join(ROOT, "tests", "golden", f + ".npz"))[k]
        glines = [_line(k, v) for k, v in golden.items()]
        lines = [_line(k, v) for k, v in exp.items()] + glines
        open(out_path, "w").write("\n".join(lines) + "\n")
        ok = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "check_kat.py")], capture_output=True, text=True)
        assert ok.returncode == 0, ok.stdout + ok.stderr
        # corrupt one limb of h and one setup query: both must be reported
        bad = dict(exp)
        h = bad["x_wm.h"].copy(); h[123, 0] ^= 1; bad["x_wm.h"] = h
        lines = [_line(k, v) for k, v in bad.items()]
        q = golden["groth16.c0_l_query"].copy(); q[0, 0] ^= 1
        lines += [_line(k, (q if k == "groth16.c0_l_query" else v)) for k, v in golden.items()]
        open(out_path, "w").write("\n".join(lines) + "\n")
        res = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "check_kat.py")], capture_output=True, text=True)
        assert res.returncode == 1 and "x_wm.h" in res.stdout and "groth16.c0_l_query" in res.stdout, res.stdout
        # a golden output that arkworks did not write is a failure too (round 5: no golden array without a KAT line)
        open(out_path, "w").write("\n".join([_line(k, v) for k, v in exp.items()] + [ln for ln in glines if not ln.startswith("fields.f2_mul ")]) + "\n")
        res = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "check_kat.py")], capture_output=True, text=True)
        assert res.returncode == 1 and "fields.f2_mul" in res.stdout, res.stdout
        # a generator that differs is advisory (the setup entry point takes generators as arguments), not a failure
        adv = dict(exp)
        gq = adv["x_consts.c2_g2_generator"].copy(); gq[0] ^= 1; adv["x_consts.c2_g2_generator"] = gq
        open(out_path, "w").write("\n".join([_line(k, v) for k, v in adv.items()] + glines) + "\n")
        res = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "check_kat.py")], capture_output=True, text=True)
        assert res.returncode == 0 and "NOTE: x_consts.c2_g2_generator" in res.stdout, res.stdout
    finally:
        if saved is None:
            if os.path.exists(out_path):
                os.remove(out_path)
        else:
            open(out_path, "w").write(saved)


def test_no_golden_array_without_a_kat_line():
    """the manifest of rust/tests/kat.rs (`EMITS`, which the Rust test asserts against what it writes) names a line family
    for EVERY output array of tests/golden/*.npz and for everything tools/kat_extra.expected() checks -- a golden vector that arkworks is
    never asked to reproduce fails here"""
    import kat_export
    import kat_extra
    src = open(os.path.join(ROOT, "rust", "tests", "kat.rs")).read()
    body = src[src.index("const EMITS"):]
    emits = set(re.findall(r'"([a-z_0-9.{}]+)"', body[:body.index("];")]))
    fam = lambda k: k.split(".", 1)[0] + "." + re.sub(r"\d+", "{}", k.split(".", 1)[1])   # (digits after the file prefix)
    want = {fam(k) for k in kat_export.golden_outputs()} | {fam(k) for k in kat_extra.expected()}
    assert want <= emits, sorted(want - emits)
    assert emits <= want, sorted(emits - want)   # (and nothing is claimed that nobody checks)
    # the functions that write them are called from kat()
    for fn in ("fields::<", "msm_inf::<", "setup_inf::<", "setup::<", "wire::<", "groth16::<"):
        assert fn in src[src.index("fn kat()"):], fn
