"""Sanitizers on the CPU side (SURVEY.md section 5 "race detection / sanitizers").  GPU AddressSanitizer is not available
on this pool, so what can be instrumented is what runs on the host:

  * the C++ oracle (`make -C tools/san oracle`: AddressSanitizer + UndefinedBehaviorSanitizer, -fno-sanitize-recover) -- the golden-vector and
    property tests of the oracle run again in a child python with the sanitizer runtimes preloaded and ORACLE_LIB pointing at that build;
    its MSM and FFT are multi-threaded, so this also walks the thread joins;
  * the library's HOST paths that need no GPU -- the wire format (pcd_amd/csrc/wire.hip: parsing of untrusted bytes, Tonelli-Shanks, subgroup
    checks), argument checking, the no-device error paths -- from `make -C tools/san lib` (host-only compile of the same sources under
    AddressSanitizer), again in a child python.

A report from either runtime fails the test (non-zero exit of the child or a sanitizer banner in its output)."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
BANNERS = ("ERROR: AddressSanitizer", "runtime error:", "ERROR: LeakSanitizer", "SUMMARY: UndefinedBehaviorSanitizer")


def _run(env_extra, args, timeout):
    env = dict(os.environ, **env_extra)
    env["ASAN_OPTIONS"] = "detect_leaks=0:abort_on_error=0:exitcode=97:verify_asan_link_order=0"   # (python itself leaks by design at exit)
    env["UBSAN_OPTIONS"] = "print_stacktrace=1:halt_on_error=1:exitcode=98"
    p = subprocess.run([sys.executable, "-m", "pytest", "-x", "-q", "-p", "no:cacheprovider"] + args, cwd=ROOT, env=env, capture_output=True, text=True,
                       timeout=timeout)
    out = p.stdout + p.stderr
    assert not any(b in out for b in BANNERS), out[-3000:]
    assert p.returncode == 0, out[-3000:]
    return out


def test_oracle_under_asan_ubsan():
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "tools", "san"), "oracle"], stdout=subprocess.DEVNULL)
    so = os.path.join(ROOT, "oracle", "_san", "liboracle_san.so")
    pre = [subprocess.check_output(["gcc", f"-print-file-name={n}"], text=True).strip() for n in ("libasan.so", "libubsan.so")]
    assert all(os.path.isabs(x) and os.path.exists(x) for x in pre), pre
    out = _run({"ORACLE_LIB": so, "LD_PRELOAD": ":".join(pre)},
               ["tests/test_oracle_golden.py", "tests/test_oracle_properties.py", "tests/test_synthetic_keys.py", "-m", "not gpu"], timeout=1500)
    assert " passed" in out


def test_library_host_paths_under_asan():
    san = os.path.join(ROOT, "build", "san", "libpcdhip_san.so")
    r = subprocess.run(["make", "-j4", "-C", os.path.join(ROOT, "tools", "san"), "lib"], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    rt = subprocess.check_output(["hipcc", "-print-file-name=libclang_rt.asan-x86_64.so"], text=True).strip()
    if not (os.path.isabs(rt) and os.path.exists(rt)):
        pytest.skip("no shared AddressSanitizer runtime next to hipcc")
    out = _run({"PCDHIP_LIB": san, "LD_PRELOAD": rt},
               ["tests/test_wire_format.py", "tests/test_capi_and_dist.py", "-m", "not gpu", "-k", "not two_ranks and not failure_still"], timeout=1500)
    assert " passed" in out


@pytest.mark.skipif(os.environ.get("PCD_SAN_HOSTCHECK") != "1", reason="opt-in (PCD_SAN_HOSTCHECK=1): compiling the harness under AddressSanitizer takes "
                    "~4 minutes, more than the CPU suite may spend; last run: profiles/r05_sanitizers.txt")
def test_hostcheck_under_asan():
    """the kernels' own __host__ __device__ arithmetic (28-bit-limb fields, lazily reduced additions, group law, pairing VM programs) on the
    host under AddressSanitizer: tests/test_hostcheck.py in a child python with the harness built into build/san"""
    rt = subprocess.check_output(["hipcc", "-print-file-name=libclang_rt.asan-x86_64.so"], text=True).strip()
    out = _run({"HOSTCHECK_SAN": "1", "LD_PRELOAD": rt}, ["tests/test_hostcheck.py"], timeout=3000)
    assert " passed" in out
