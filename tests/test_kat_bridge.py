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
        lines = [_line(k, v) for k, v in exp.items()]
        # a few of the golden-file outputs as arkworks would write them (names file.key)
        g = np.load(os.path.join(ROOT, "tests", "golden", "groth16.npz"))
        for k in ("c0_a_query", "c0_a_query_inf", "c0_h_query", "c1_b_g2_query", "c0_proof", "c1_gamma_abc_g1"):
            lines.append(_line("groth16." + k, g[k]))
        open(out_path, "w").write("\n".join(lines) + "\n")
        ok = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "check_kat.py")], capture_output=True, text=True)
        assert ok.returncode == 0, ok.stdout + ok.stderr
        # corrupt one limb of h and one setup query: both must be reported
        bad = dict(exp)
        h = bad["x_wm.h"].copy(); h[123, 0] ^= 1; bad["x_wm.h"] = h
        lines = [_line(k, v) for k, v in bad.items()]
        q = g["c0_l_query"].copy(); q[0, 0] ^= 1
        lines.append(_line("groth16.c0_l_query", q))
        open(out_path, "w").write("\n".join(lines) + "\n")
        res = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "check_kat.py")], capture_output=True, text=True)
        assert res.returncode == 1 and "x_wm.h" in res.stdout and "groth16.c0_l_query" in res.stdout, res.stdout
        # a generator that differs is advisory (the setup entry point takes generators as arguments), not a failure
        adv = dict(exp)
        gq = adv["x_consts.c2_g2_generator"].copy(); gq[0] ^= 1; adv["x_consts.c2_g2_generator"] = gq
        open(out_path, "w").write("\n".join(_line(k, v) for k, v in adv.items()) + "\n")
        res = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "check_kat.py")], capture_output=True, text=True)
        assert res.returncode == 0 and "NOTE: x_consts.c2_g2_generator" in res.stdout, res.stdout
    finally:
        if saved is None:
            if os.path.exists(out_path):
                os.remove(out_path)
        else:
            open(out_path, "w").write(saved)


def test_kat_rs_emits_every_expected_name():
    """the x_* names kat.rs builds with format! cover exactly what tools/kat_extra.expected() knows how to check"""
    import kat_extra
    src = open(os.path.join(ROOT, "rust", "tests", "kat.rs")).read()
    emitted = set(re.findall(r'emit\(out, &format!\("(x_[^"]+)"', src)) | set(re.findall(r'emit\(out, "(x_[^"]+)"', src))
    # "{}_i{}c{}" with stem = the input name minus "_in" covers the x_mixed family
    fam = lambda k: re.sub(r"\d+", "{}", k)
    want = {fam(k) for k in kat_extra.expected()}
    got = {fam(e) for e in emitted} | {"x_mixed.f{}_n{}_i{}c{}"}
    assert want <= got, sorted(want - got)
