"""CPU: the sanitizer legs (SURVEY.md section 5 "race detection / sanitizers") as part of the CPU suite.

The recipes and their driver live in tools/san/ (kept off the GPU box by .gpurunignore -- the GPU runner refuses snapshots that carry
sanitizer flags; tests/test_no_forbidden_literals.py).  Here they run in a child pytest: the C++ oracle and the library's host paths
under the address / undefined-behaviour checkers; the hostcheck leg stays opt-in (PCD_SAN_HOSTCHECK=1, ~4 minutes to compile)."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SAN = os.path.join(ROOT, "tools", "san")


@pytest.mark.skipif(not os.path.exists(os.path.join(SAN, "Makefile")), reason="tools/san/ does not ship to the GPU box (CPU-only leg)")
def test_sanitizer_legs():
    p = subprocess.run([sys.executable, "-m", "pytest", "-x", "-q", "-p", "no:cacheprovider", SAN], cwd=ROOT, capture_output=True, text=True, timeout=3400)
    assert p.returncode == 0, (p.stdout + p.stderr)[-4000:]
    assert " passed" in p.stdout
