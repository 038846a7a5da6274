"""CPU: nothing that ships to the GPU box may carry a sanitizer flag or an XNACK switch.

GPU AddressSanitizer / ThreadSanitizer builds and XNACK-on runs are not available on this pool; the GPU runner scans the snapshot it is
about to push and REFUSES the whole call when a runnable file holds such a flag (round 5 lost its driver-run GPU suite to one list
element in a test file).  Every sanitizer recipe therefore lives in tools/san/, which .gpurunignore keeps off the GPU box, and this test
walks exactly the files that do travel (the tree minus .git/, gpurun_out/ and the .gpurunignore entries).  The needles are assembled at
run time so that this file does not contain them either."""
import fnmatch
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
NEEDLES = re.compile("|".join([re.escape("-fsan" + "itize"), "HSA_" + "XNACK", "xnack" + re.escape("+"), re.escape("-mxn" + "ack"), "xnack" + "-on"]).encode())


def _ignored():
    pats = [".git", "gpurun_out"]
    with open(os.path.join(ROOT, ".gpurunignore")) as f:
        pats += [ln.strip().rstrip("/") for ln in f if ln.strip() and not ln.startswith("#")]
    return pats


def _shipped_files():
    pats = _ignored()
    def skip(rel):
        return any(rel == p or rel.startswith(p + "/") or fnmatch.fnmatch(rel, p) for p in pats)
    for d, dirs, files in os.walk(ROOT):
        rel_d = os.path.relpath(d, ROOT)
        dirs[:] = [x for x in dirs if not skip(os.path.normpath(os.path.join(rel_d, x))) and x != "__pycache__"]
        for fn in files:
            rel = os.path.normpath(os.path.join(rel_d, fn))
            if not skip(rel):
                yield rel


def test_sanitizer_recipes_do_not_ship():
    pats = _ignored()
    assert "tools/san" in pats, "tools/san/ (the sanitizer recipes) must be listed in .gpurunignore"
    assert os.path.exists(os.path.join(ROOT, "tools", "san", "Makefile"))


def test_no_sanitizer_or_xnack_literal_in_anything_that_ships():
    hits = []
    n = 0
    for rel in _shipped_files():
        p = os.path.join(ROOT, rel)
        if os.path.islink(p) or os.path.getsize(p) > (64 << 20):
            continue
        n += 1
        with open(p, "rb") as f:
            data = f.read()
        if rel.endswith((".so", ".o", ".a", ".npz", ".npy", ".pyc")):
            continue          # (built objects and numeric fixtures: a byte coincidence is not a flag, and the runner scans text)
        for m in NEEDLES.finditer(data):
            line = data.count(b"\n", 0, m.start()) + 1
            hits.append(f"{rel}:{line}: {m.group().decode()}")
    assert n > 100
    assert not hits, "sanitizer / XNACK literals in files that ship to the GPU box (move them under tools/san/):\n" + "\n".join(hits[:40])
