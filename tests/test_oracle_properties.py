"""CPU: size-independent properties of the C++ oracle at sizes the Python oracle cannot reach."""
import numpy as np
import pytest


@pytest.mark.parametrize("fid,log_n", [(0, 12), (1, 14), (2, 10), (3, 12)])
def test_fft_roundtrip_and_linearity(co, fid, log_n):
    n = 1 << log_n
    x, y = co.gen_field(fid, n, 1), co.gen_field(fid, n, 2)
    for coset in (False, True):
        fx = co.fft(fid, x, coset=coset, nthreads=4)
        assert np.array_equal(co.fft(fid, fx, inverse=True, coset=coset, nthreads=4), x)
        fy = co.fft(fid, y, coset=coset, nthreads=4)
        fxy = co.fft(fid, co.fp_op(fid, "add", x, y), coset=coset, nthreads=4)
        assert np.array_equal(fxy, co.fp_op(fid, "add", fx, fy))


def test_fft_above_two_adicity_is_refused(co):
    # F298A has 2-adicity 17: a radix-2 domain of 2^18 does not exist (upstream switches to mixed radix)
    import ctypes as C
    buf = np.zeros((1, 5), dtype=np.uint64)
    assert co.lib().orc_fft(0, buf.ctypes.data_as(C.c_void_p), 18, 0, 0, 1) == -3


@pytest.mark.parametrize("cid,grp,n", [(0, 1, 3000), (1, 1, 3000), (0, 2, 700), (1, 2, 500), (2, 1, 400), (3, 2, 60)])
def test_msm_window_independence_and_split(co, cid, grp, n):
    """The MSM value does not depend on the window size, the thread count, or on splitting the range
    (the multi-GPU shard + combine path)."""
    fr = co.CURVE_FR[cid]
    pts = co.gen_points(cid, grp, n, seed=3)
    assert all(co.on_curve(cid, grp, p) for p in pts[:: max(1, n // 7)])
    for dist in (0, 1):
        sc = co.gen_scalars(fr, n, seed=4, dist=dist)
        ref, _ = co.to_affine(cid, grp, co.msm(cid, grp, pts, sc, nthreads=4))
        alt, _ = co.to_affine(cid, grp, co.msm(cid, grp, pts, sc, nthreads=1, c_override=7))
        assert np.array_equal(ref, alt)
        h = n // 3
        parts = co.jac_add(cid, grp, co.msm(cid, grp, pts[:h], sc[:h]), co.msm(cid, grp, pts[h:], sc[h:], nthreads=2))
        sp, _ = co.to_affine(cid, grp, parts)
        assert np.array_equal(ref, sp)


@pytest.mark.parametrize("cid", [0, 1, 2, 3])
def test_group_order_and_pairing_bilinearity(co, cid):
    fr = co.CURVE_FR[cid]
    k = co.gen_scalars(fr, 2, seed=9)
    # r * G = O in both groups: scalar r-1 then add G
    g1, g2 = co.generator(cid, 1), co.generator(cid, 2)
    a_g1, _ = co.to_affine(cid, 1, co.scalar_mul(cid, 1, g1, k[0]))
    b_g2, _ = co.to_affine(cid, 2, co.scalar_mul(cid, 2, g2, k[1]))
    # e(aP, bQ) == e(abP, Q)
    ab = co.fp_op(fr, "to_canonical", co.fp_op(fr, "mul", co.fp_op(fr, "from_canonical", k[:1]),
                                               co.fp_op(fr, "from_canonical", k[1:2])))[0]
    ab_g1, _ = co.to_affine(cid, 1, co.scalar_mul(cid, 1, g1, ab))
    assert np.array_equal(co.pairing(cid, a_g1[0], b_g2[0]), co.pairing(cid, ab_g1[0], g2))


@pytest.mark.parametrize("fid,q", [(0, 7), (2, 5)])
def test_mixed_radix_domain_vs_bigint(co, fid, q):
    """ark-poly MixedRadixEvaluationDomain (sizes 2^a q^b): C++ restatement == naive big-integer DFT."""
    import random
    from oracle import pyoracle as O
    f = O.FIELDS[fid]
    rnd = random.Random(fid)
    for m, a in ((q, 3), (q * q, 2), (q, 0)):
        xs = [rnd.randrange(f.p) for _ in range(m << a)]
        X = O.pack_fp(f, xs)
        for inv in (False, True):
            for coset in (False, True):
                want = O.dft_general(f, xs, inverse=inv, coset=coset)
                assert O.unpack_fp(f, co.fft_general(fid, X, m, inverse=inv, coset=coset, nthreads=2)) == want
    assert co.domain_size(fid, (1 << f.two_adicity) + 1) == O.best_mixed_domain_size(f, (1 << f.two_adicity) + 1, q)
    assert co.domain_size(fid, 1000) == 1024


@pytest.mark.parametrize("cid,group", [(0, 1), (0, 2), (1, 1), (1, 2), (2, 1), (3, 2)])
def test_fixed_base_msm_matches_double_and_add(co, cid, group):
    """The oracle's restatement of ark-ec FixedBaseMSM (window table + mixed additions + batch normalisation)
    against plain double-and-add, including 0, 1, r - 1 and both window-size regimes (n < 32 and n >= 32)."""
    fr = co.CURVE_FR[cid]
    base = co.generator(cid, group)
    for n in (5, 40):
        sc = co.gen_scalars(fr, n, seed=90 + n)
        sc[0] = 0
        sc[1] = 0; sc[1, 0] = 1
        one = np.zeros_like(sc[:1]); one[0, 0] = 1
        sc[2] = co.fp_op(fr, "to_canonical", co.fp_op(fr, "neg", co.fp_op(fr, "from_canonical", one)))[0]
        got, inf = co.fixed_base_mul(cid, group, base, sc, nthreads=4)
        for i in range(n if n < 32 else 6):
            want, winf = co.to_affine(cid, group, co.scalar_mul(cid, group, base, sc[i])[None])
            assert inf[i] == winf[0] and np.array_equal(got[i], want[0])
        assert inf[0] == 1 and np.array_equal(got[1], base)


def test_skewed_r1cs_generator(co):
    """coracle.skewed_r1cs (what the mat-vec kernels are measured on): satisfied, power-law row lengths with the forced long rows,
    >= 75 % unit coefficients, every column index in range"""
    from oracle import pyoracle as po
    r = co.skewed_r1cs(1, 9000, 2, seed=9)
    f = po.FIELDS[1]
    z = [f.from_mont(f.unlimbs(x)) for x in r.z]

    def row(rp, col, cf, j):
        return sum(f.from_mont(f.unlimbs(cf[k])) * z[col[k]] for k in range(int(rp[j]), int(rp[j + 1]))) % f.p
    for j in list(range(40)) + [3000, 6000, 4500, 8999]:
        assert row(r.rp_a, r.col_a, r.coeff_a, j) * row(r.rp_b, r.col_b, r.coeff_b, j) % f.p == row(r.rp_c, r.col_c, r.coeff_c, j)
    la, lb = np.diff(r.rp_a.astype(np.int64)), np.diff(r.rp_b.astype(np.int64))
    assert la.max() > 4096 and lb.max() == 300 and la.min() >= 1 and np.median(la) <= 2
    one, mone = f.to_mont(1), f.to_mont(f.p - 1)
    vals = [f.unlimbs(x) for x in r.coeff_a[:20000]]
    assert sum(v in (one, mone) for v in vals) / len(vals) > 0.75
    assert max(r.col_a.max(), r.col_b.max(), r.col_c.max()) < r.num_vars and r.num_vars == r.z.shape[0]


def test_witness_r1cs_generator(co):
    """coracle.witness_r1cs (round 5: the assignment a verifier circuit produces -- runs of bits with booleanity rows, packing rows, a few
    products; /root/reference src/ec_cycle_pcd/data_structures.rs:269-304): every row satisfied on two fields, >= 70 % of z is 0 or 1 with
    both values frequent, the rest spread over the field, every column in range, one variable per row"""
    from oracle import pyoracle as po
    for field, nc in ((1, 6000), (3, 700)):
        r = co.witness_r1cs(field, nc, 2, seed=11 + field)
        f = po.FIELDS[field]
        z = [f.from_mont(f.unlimbs(x)) for x in r.z]

        def row(rp, col, cf, j):
            return sum(f.from_mont(f.unlimbs(cf[k])) * z[col[k]] for k in range(int(rp[j]), int(rp[j + 1]))) % f.p
        for j in range(nc):
            assert row(r.rp_a, r.col_a, r.coeff_a, j) * row(r.rp_b, r.col_b, r.coeff_b, j) % f.p == row(r.rp_c, r.col_c, r.coeff_c, j), j
        zeros, ones = sum(v == 0 for v in z), sum(v == 1 for v in z)
        assert (zeros + ones) / len(z) >= 0.70 and zeros / len(z) >= 0.35 and ones / len(z) >= 0.25
        big = [v for v in z if v.bit_length() > f.p.bit_length() - 8]
        assert len(big) >= len(z) // 20
        assert max(r.col_a.max(), r.col_b.max(), r.col_c.max()) < r.num_vars and r.num_vars == r.z.shape[0] == nc + 2 + 4
        assert r.rp_c[nc] < r.rp_a[nc]   # booleanity rows have an empty C row
