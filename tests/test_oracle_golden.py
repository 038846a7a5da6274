"""CPU: the C++ oracle (oracle/*.hpp, restatement of the upstream ark-ec / ark-poly / ark-groth16
algorithms) against the committed golden vectors produced by the independent pure-Python big-integer
oracle (tests/golden/gen_golden.py).  The reference itself holds no vectors for this path."""
import os

import numpy as np
import pytest


@pytest.mark.parametrize("fid", [0, 1, 2, 3])
def test_field_ops(co, golden, fid):
    g = golden("fields")
    a, b = g[f"f{fid}_a"], g[f"f{fid}_b"]
    assert np.array_equal(co.fp_op(fid, "add", a, b), g[f"f{fid}_add"])
    assert np.array_equal(co.fp_op(fid, "sub", a, b), g[f"f{fid}_sub"])
    assert np.array_equal(co.fp_op(fid, "mul", a, b), g[f"f{fid}_mul"])
    assert np.array_equal(co.fp_op(fid, "inv", b), g[f"f{fid}_inv_b"])
    assert np.array_equal(co.fp_op(fid, "to_canonical", a), g[f"f{fid}_a_canonical"])
    assert np.array_equal(co.fp_op(fid, "from_canonical", g[f"f{fid}_a_canonical"]), a)


@pytest.mark.parametrize("cid", [0, 1, 2, 3])
@pytest.mark.parametrize("grp", [1, 2])
def test_msm_pippenger(co, golden, cid, grp):
    g = golden("msm")
    pre = f"c{cid}_g{grp}_"
    bases, inf, sc = g[pre + "bases"], g[pre + "inf"], g[pre + "scalars"]
    for nthreads, c_override in ((1, 0), (4, 0), (2, 5), (1, 13)):
        out = co.msm(cid, grp, bases, sc, inf=inf, nthreads=nthreads, c_override=c_override)
        xy, oinf = co.to_affine(cid, grp, out)
        assert np.array_equal(xy[0], g[pre + "result_xy"]) and oinf[0] == g[pre + "result_inf"][0]
    ones = np.zeros_like(sc)
    ones[:, 0] = 1
    xy, oinf = co.to_affine(cid, grp, co.msm(cid, grp, bases, ones, inf=inf))
    assert np.array_equal(xy[0], g[pre + "ones_xy"]) and oinf[0] == g[pre + "ones_inf"][0]
    xy, oinf = co.to_affine(cid, grp, co.msm(cid, grp, bases, np.zeros_like(sc), inf=inf))
    assert oinf[0] == 1


@pytest.mark.parametrize("fid", [0, 1, 2, 3])
def test_fft(co, golden, fid):
    g = golden("fft")
    for log_n in (0, 1, 4, 8, 11):
        x = g[f"f{fid}_n{log_n}_in"]
        for inv in (0, 1):
            for coset in (0, 1):
                for nthreads in (1, 3):
                    got = co.fft(fid, x, inverse=bool(inv), coset=bool(coset), nthreads=nthreads)
                    assert np.array_equal(got, g[f"f{fid}_n{log_n}_i{inv}c{coset}"]), (fid, log_n, inv, coset)


def _r1cs(co, g, cid):
    pre = f"c{cid}_"
    return co.R1CS(co.CURVE_FR[cid], int(g[pre + "num_inputs"][0]), g[pre + "rp_a"], g[pre + "col_a"], g[pre + "coeff_a"],
                   g[pre + "rp_b"], g[pre + "col_b"], g[pre + "coeff_b"], g[pre + "rp_c"], g[pre + "col_c"],
                   g[pre + "coeff_c"], np.ascontiguousarray(g[pre + "z"]))


@pytest.mark.parametrize("cid", [0, 1, 2, 3])
def test_witness_map(co, golden, cid):
    g = golden("groth16")
    r = _r1cs(co, g, cid)
    assert np.array_equal(co.witness_map(r, nthreads=2), g[f"c{cid}_h"])


@pytest.mark.parametrize("cid", [0, 1])
def test_groth16_setup_prove_verify(co, golden, cid):
    """Counterpart of tests/mnt4_groth16.rs:84-87,119 at the SNARK level: setup -> prove -> verify accepts,
    a wrong public input rejects; plus byte parity of keys and proof with the big-int oracle."""
    g = golden("groth16")
    pre = f"c{cid}_"
    r = _r1cs(co, g, cid)
    keys = co.groth16_setup(cid, r, g[pre + "toxic"], nthreads=4)
    for nm in ("alpha_g1", "beta_g1", "delta_g1", "beta_g2", "delta_g2", "gamma_g2", "a_query", "b_g1_query",
               "b_g2_query", "h_query", "l_query", "gamma_abc_g1"):
        assert np.array_equal(getattr(keys, nm), g[pre + nm]), nm
    proof, inf = co.groth16_prove(keys, r, g[pre + "r"], g[pre + "s"], nthreads=4)
    assert np.array_equal(proof, g[pre + "proof"]) and not inf.any()
    pub = np.ascontiguousarray(r.z[1:r.num_inputs])
    assert co.groth16_verify(keys, pub, proof)
    bad = pub.copy()
    bad[0] = co.fp_op(co.CURVE_FR[cid], "add", bad[:1], r.z[:1])[0]  # x + 1
    assert not co.groth16_verify(keys, bad, proof)


@pytest.mark.parametrize("cid", [0, 1, 2, 3])
def test_pairing(co, golden, cid):
    g = golden("pairing")
    assert np.array_equal(co.pairing(cid, g[f"c{cid}_p"], g[f"c{cid}_q"]), g[f"c{cid}_gt"])


# ---- tests/golden/at_size.npz: the oracle's results for the at-size GPU tests (tests/golden/gen_at_size.py; conftest.Expect) ----
def _at_size_cases():
    import importlib
    import sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    out = {}
    for m in ("test_gpu_at_size", "test_gpu_config4", "test_gpu_witness_like"):
        out.update(importlib.import_module(m).AT_SIZE)
    return out


def test_at_size_fixture_is_complete():
    """every at-size case of the GPU suite has its expectation in the committed file, and the file holds nothing else: the GPU run then spends no
    oracle time on them (PCD_RECOMPUTE=1 brings it back) and cannot silently fall back to recomputing a missing one"""
    from conftest import Expect
    e = Expect()
    cases = _at_size_cases()
    missing = [k for k in cases if e.stored(k) is None]
    assert not missing, f"run tests/golden/gen_at_size.py: no expectation for {missing}"
    stale = {k.split("/")[0] for k in e._npz.files} - set(cases)
    assert not stale, f"at_size.npz holds keys without a case builder: {stale}"


@pytest.mark.parametrize("key", ["prove_c1_2p16", "wm_skewed_2p20_sha256"])
def test_at_size_fixture_spot_check(co, key):
    """the cheap entries re-derived here, in the CPU suite: same case builder, the oracle on the spot, equal to the committed value"""
    from conftest import Expect
    e = Expect()
    have = e.stored(key)
    assert have is not None
    _, want_fn = _at_size_cases()[key](co)
    got = want_fn()
    a, b = (have, got) if isinstance(have, tuple) else ((have,), (got,))
    assert len(a) == len(b) and all(np.array_equal(x, np.asarray(y)) for x, y in zip(a, b))
