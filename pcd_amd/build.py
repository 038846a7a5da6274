"""Build driver for libpcdhip.so (hipcc, gfx950 only; cross-compiles without a GPU)."""
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
_ROOT = os.path.dirname(_HERE)


def library_path():
    # PCDHIP_LIB: developer knob for A/B runs of experimental builds (tools/); the default is the in-tree library
    return os.environ.get("PCDHIP_LIB") or os.path.join(_HERE, "libpcdhip.so")


def build_library(jobs=None, verbose=False):
    jobs = jobs or min(8, os.cpu_count() or 1)
    cmd = ["make", "-C", os.path.join(_HERE, "csrc"), f"-j{jobs}", "../libpcdhip.so"]
    env = dict(os.environ)
    env.setdefault("HIPCC", "hipcc")
    out = None if verbose else subprocess.DEVNULL
    subprocess.check_call(cmd, env=env, stdout=out)
    if not os.path.exists(library_path()):
        raise RuntimeError("libpcdhip.so was not produced")
    return library_path()
