"""The C-ABI from plain C (tests/c_driver/driver.c): the header compiles as C and links against libpcdhip.so here (CPU); on the GPU the
program proves from a blob -- key, matrices, assignment, r, s repacked exactly as rust/src/prover.rs does -- and must reproduce the
oracle's proof bytes."""
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "build", "c_driver")


def _build():
    os.makedirs(os.path.dirname(EXE), exist_ok=True)
    lib_dir = os.path.join(ROOT, "pcd_amd")
    subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Werror", "-O1", "-I", os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "tests", "c_driver", "driver.c"), "-L", lib_dir, "-lpcdhip",
                           f"-Wl,-rpath,{lib_dir}", "-o", EXE])


def test_header_is_c_and_links():
    from pcd_amd import capi
    capi.lib()          # the library exists (built by __graft_entry__.build())
    _build()
    assert os.path.exists(EXE)


@pytest.mark.gpu
def test_c_driver_proves(co, tmp_path):
    _build()
    curve = 1
    fr = co.CURVE_FR[curve]
    r = co.synthetic_r1cs(fr, 700, 2, seed=1201)
    keys = co.groth16_setup(curve, r, co.gen_field(fr, 5, seed=1202), nthreads=8)
    rs = co.gen_field(fr, 2, seed=1203)
    want, winf = co.groth16_prove(keys, r, rs[0], rs[1], nthreads=8)
    m = r.num_vars
    sc = co.gen_scalars(fr, m, seed=1204, dist=1)
    msm_want = co.to_affine(curve, 1, co.msm(curve, 1, keys.a_query, sc, inf=keys.a_inf, nthreads=8))[0]
    header = np.array([curve, m, r.num_inputs, keys.domain_size, keys.h_query.shape[0], keys.l_query.shape[0], r.num_constraints,
                       len(r.col_a), len(r.col_b), len(r.col_c), 0, 0], dtype=np.uint64)
    parts = [header, keys.alpha_g1, keys.beta_g1, keys.delta_g1, keys.beta_g2, keys.delta_g2, keys.a_query, keys.a_inf, keys.b_g1_query,
             keys.b_g1_inf, keys.b_g2_query, keys.b_g2_inf, keys.h_query, keys.h_inf, keys.l_query, keys.l_inf,
             r.rp_a, r.col_a, r.coeff_a, r.rp_b, r.col_b, r.coeff_b, r.rp_c, r.col_c, r.coeff_c, r.z, rs[0], rs[1], want, winf, sc, msm_want]
    blob = tmp_path / "prove.blob"
    with open(blob, "wb") as f:
        for a in parts:
            b = np.ascontiguousarray(a).tobytes()
            f.write(b + b"\0" * (-len(b) % 8))
    out = subprocess.run([EXE, str(blob)], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "c driver ok" in out.stdout


@pytest.mark.gpu
def test_c_driver_s2_sequence(co, tmp_path):
    """seam S2 from plain C: the calls rust/src/s2.rs makes for a Marlin prover (tests/mnt4_marlin.rs:72-75) -- one upload of the
    committer key, MSMs over prefixes of it with host scalars, in-place transforms of host vectors -- against the oracle"""
    _build()
    curve, n = 0, 1 << 13
    fr = co.CURVE_FR[curve]
    pts = co.gen_points(curve, 1, n, seed=1301)
    inf = np.zeros(n, dtype=np.uint8)
    lens = [n, n // 2 + 17, 4097, n - 1]
    parts = [np.array([curve, n, len(lens), 14, fr, 0, 0, 0], dtype=np.uint64), pts, inf]
    for i, m in enumerate(lens):
        sc = co.gen_scalars(fr, m, seed=1310 + i, dist=i % 2)
        want = co.to_affine(curve, 1, co.msm(curve, 1, pts[:m], sc, nthreads=8))[0]
        parts += [np.array([m], dtype=np.uint64), sc, want]
    v = co.gen_field(fr, 1 << 14, seed=1320)
    parts += [v, co.fft(fr, v, nthreads=8), co.fft(fr, v, inverse=True, nthreads=8)]
    blob = tmp_path / "s2.blob"
    with open(blob, "wb") as f:
        for a in parts:
            b = np.ascontiguousarray(a).tobytes()
            f.write(b + b"\0" * (-len(b) % 8))
    out = subprocess.run([EXE, str(blob), "s2"], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "S2 sequence" in out.stdout
