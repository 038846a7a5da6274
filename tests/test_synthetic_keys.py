"""CPU: the synthetic Groth16 keys the at-size tests and the bench prove with carry exactly the points at infinity a setup over the same
R1CS would leave in them (oracle/coracle.py synthetic_keys(consistent=True)): a_query[i] is the identity when no row of A mentions
variable i (public inputs excepted: the input-consistency rows of the QAP), b_g1 / b_g2 likewise for B -- checked against a REAL setup
(coracle.groth16_setup) on a small R1CS of the same shape, where the flags come out of the arithmetic."""
import numpy as np


def test_consistent_flags_match_a_real_setup(co):
    cid = 0
    fr = co.CURVE_FR[cid]
    r = co.skewed_r1cs(fr, 900, 3, seed=4711)
    real = co.groth16_setup(cid, r, co.gen_field(fr, 5, seed=4712), nthreads=8)
    syn = co.synthetic_keys(cid, r, seed=4713)
    dense = co.synthetic_keys(cid, r, seed=4713, consistent=False)
    assert 0.05 < real.a_inf.mean() < 0.7 and 0.1 < real.b_g2_inf.mean() < 0.8     # (the shape leaves a good share of the variables out)
    for name in ("a_inf", "b_g1_inf", "b_g2_inf"):
        assert np.array_equal(getattr(syn, name), getattr(real, name)), name
        assert not getattr(dense, name).any()
    assert np.array_equal(real.b_g1_inf, real.b_g2_inf)
    assert not syn.l_inf.any() and not syn.h_inf.any()


def test_consistent_flags_match_a_real_setup_witness_shape(co):
    """the same for the witness-like generator (coracle.witness_r1cs: booleanity rows put every bit into A AND B, a packed word only
    into C): flags from the matrices == flags out of a real setup's arithmetic"""
    cid = 1
    fr = co.CURVE_FR[cid]
    r = co.witness_r1cs(fr, 700, 2, seed=4811)
    real = co.groth16_setup(cid, r, co.gen_field(fr, 5, seed=4812), nthreads=8)
    syn = co.synthetic_keys(cid, r, seed=4813)
    assert 0.02 < real.a_inf.mean() < 0.5
    for name in ("a_inf", "b_g1_inf", "b_g2_inf"):
        assert np.array_equal(getattr(syn, name), getattr(real, name)), name
