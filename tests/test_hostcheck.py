"""CPU: the __host__ __device__ field / curve / pairing templates of pcd_amd/csrc (the code the HIP kernels are
made of) compiled for the HOST by tests/hostcheck/hostcheck.hip and checked against the oracles.  This is a
test harness only -- the product has no CPU path -- but it lets the 28-bit-limb arithmetic, the C-ABI <-> device
image conversions, the group law and the pairing formulas be verified in a GPU-less container."""
import ctypes as C
import os
import random
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HC = os.path.join(ROOT, "tests", "hostcheck")


class _Libs:
    """symbols of the two harness libraries (compiled side by side) under one object"""

    def __init__(self, libs):
        self._libs = libs

    def __getattr__(self, name):
        for lib in self._libs:
            try:
                return getattr(lib, name)
            except AttributeError:
                pass
        raise AttributeError(name)


@pytest.fixture(scope="module")
def hc():
    csrc = os.path.join(ROOT, "pcd_amd", "csrc")
    hdrs = [os.path.join(csrc, f) for f in ("fp.hip.h", "ec.hip.h", "pairing.hip.h", "pairing_vm.hip.h", "pairing_vm_gen.h", "vm_tables.h",
                                            "params_gen.h", "params28_gen.h")]
    # HOSTCHECK_SAN=1 (tools/san/test_sanitizers.py, opt-in leg): load the harness the sanitizer recipes of tools/san built into build/san
    # (this file only picks the directory; every compiler flag of that build lives in tools/san/Makefile, which does not ship to the GPU box)
    if os.environ.get("HOSTCHECK_SAN") == "1":
        out_dir = os.path.join(ROOT, "build", "san")
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "tools", "san"), "hostcheck"], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    else:
        out_dir = HC
        jobs = []
        for stem in ("hostcheck", "hostcheck_pairing"):
            so, src = os.path.join(out_dir, f"lib{stem}.so"), os.path.join(HC, f"{stem}.hip")
            if not os.path.exists(so) or any(os.path.getmtime(s) > os.path.getmtime(so) for s in [src] + hdrs):
                jobs.append(subprocess.Popen(["hipcc", "--offload-arch=gfx950", "-O1", "-std=c++17", "-fPIC", "-shared", "-DPCD_LZ_CHECK", src, "-o", so],
                                             stderr=subprocess.DEVNULL))
        for j in jobs:
            assert j.wait() == 0
    return _Libs([C.CDLL(os.path.join(out_dir, "libhostcheck.so")), C.CDLL(os.path.join(out_dir, "libhostcheck_pairing.so"))])


def P(a):
    return a.ctypes.data_as(C.c_void_p)


@pytest.mark.parametrize("fid", [0, 1, 2, 3])
def test_field_ops_and_abi_conversion(hc, fid):
    from oracle import pyoracle as O
    f = O.FIELDS[fid]
    rnd = random.Random(5 + fid)
    cases = [(0, 0), (f.p - 1, f.p - 1), (1, f.p - 1), (f.p - 1, 0), (2, 1), ((f.p - 1) // 2, 3), ((f.p + 1) // 2, 4), (1 << (f.p.bit_length() - 1), 5)]
    cases += [(1 << k, 7) for k in range(0, f.p.bit_length() - 1, 37)] + [(rnd.randrange(f.p), rnd.randrange(f.p)) for _ in range(60)]
    for a, b in cases:
        A, B = O.pack_fp(f, [a])[0], O.pack_fp(f, [b])[0]
        out = np.zeros(9 * f.n64, dtype=np.uint64)
        assert hc.hc_field_ops(fid, P(A), P(B), P(out)) == 0
        got = out.reshape(9, f.n64)
        exp = [a * b % f.p, (a + b) % f.p, (a - b) % f.p, pow(a, -1, f.p) if a else 0, a * 17 % f.p]
        assert O.unpack_fp(f, got[:5]) == exp, (a, b)
        assert O.unpack_fp(f, got[5:6], mont=False)[0] == a
        assert O.unpack_fp(f, got[6:7])[0] == ((-(2 * ((a + b + b - a - a) % f.p))) * 121 * a) % f.p
        assert O.unpack_fp(f, got[7:9]) == [exp[3], exp[3]], a   # Fp::inv_gcd (divsteps) == a^(p-2)


@pytest.mark.parametrize("fid", [0, 1, 2, 3])
def test_add_sub_on_every_representative(hc, fid):
    """Fp::operator+ / operator- decide from the two top limbs whether 2p comes off / goes on and fall back to an exact carry chain
    when those cannot tell: operands are raw device images (any representative in [0, 2p)), aimed at the undecidable band."""
    from oracle import pyoracle as O
    f = O.FIELDS[fid]
    p, N, B = f.p, (11 if fid < 2 else 27), 1 << 28
    rnd = random.Random(77 + fid)
    lo = B ** (N - 2)                      # weight of the second limb from the top
    pairs = []
    def both(a, b):
        if 0 <= a < 2 * p and 0 <= b < 2 * p:
            pairs.append((a, b))
    for _ in range(2000):
        both(rnd.randrange(2 * p), rnd.randrange(2 * p))
    for a in [0, 1, p - 1, p, p + 1, 2 * p - 1] + [rnd.randrange(2 * p) for _ in range(40)]:
        for d in (-2 * lo, -lo - 1, -lo, -lo + 1, -3, -2, -1, 0, 1, 2, 3, lo - 1, lo, lo + 1, 2 * lo):
            both(a, 2 * p - a + d)         # sums around 2p: the band the estimate leaves open
            both(a, a + d)                 # differences around 0
            both(a + d, a)
        both(a, a); both(a, 0); both(0, a); both(a, 2 * p - 1); both(2 * p - 1, a)
    for _ in range(300):                   # differences whose two top limbs cancel
        a = rnd.randrange(2 * p)
        both(a, a - (a % lo) + rnd.randrange(lo))
        both(a, (2 * p - a) - ((2 * p - a) % lo) + rnd.randrange(lo))
    def limbs(x):
        return [(x >> (28 * i)) & (B - 1) for i in range(N - 1)] + [x >> (28 * (N - 1))]
    A = np.array([limbs(a) for a, _ in pairs], dtype=np.uint32)
    Bv = np.array([limbs(b) for _, b in pairs], dtype=np.uint32)
    out = np.zeros((len(pairs), 2, N), dtype=np.uint32)
    assert hc.hc_addsub_raw(fid, P(A), P(Bv), len(pairs), P(out)) == 0
    for k, (a, b) in enumerate(pairs):
        for j, want in enumerate((a + b - 2 * p if a + b >= 2 * p else a + b, a - b + 2 * p if a < b else a - b)):
            got = [int(w) for w in out[k, j]]
            assert all(w < B for w in got[:-1]), (a, b, j)
            assert sum(w << (28 * i) for i, w in enumerate(got)) == want, (a, b, j)


@pytest.mark.parametrize("cid", [0, 1, 2, 3])
@pytest.mark.parametrize("grp", [1, 2])
def test_group_law(hc, co, cid, grp):
    n = 6
    pts = co.gen_points(cid, grp, n, seed=3)
    pts[4] = pts[3]
    sc = co.gen_scalars(co.CURVE_FR[cid], n, seed=4)
    sc[1] = 0
    sc[1, 0] = 1
    sc[3] = sc[4]
    out = np.zeros(3 * co.point_words(cid, grp) // 2, dtype=np.uint64)
    assert hc.hc_msm_naive(cid * 2 + grp - 1, P(pts), P(sc), n, P(out)) == 0
    want, _ = co.to_affine(cid, grp, co.msm(cid, grp, pts, sc))
    got, _ = co.to_affine(cid, grp, out)
    assert np.array_equal(want, got)


@pytest.mark.parametrize("cid", [0, 1, 2, 3])
def test_pairing(hc, golden, cid):
    g = golden("pairing")
    p, q = np.ascontiguousarray(g[f"c{cid}_p"]), np.ascontiguousarray(g[f"c{cid}_q"])
    out = np.zeros_like(g[f"c{cid}_gt"])
    assert hc.hc_pairing(cid, P(p), P(q), P(out)) == 0
    assert np.array_equal(out, g[f"c{cid}_gt"])


@pytest.mark.parametrize("cid", [0, 1, 2, 3])
def test_vm_pairing(hc, golden, co, cid):
    """the wave-per-pairing VM (pairing_vm.hip.h): generated programs + the device's MUL / LIN arithmetic, interpreted on the host,
    against the golden pairing, and a product of three pairings (one of them with a negated point) against the oracle"""
    g = golden("pairing")
    p, q = np.ascontiguousarray(g[f"c{cid}_p"]), np.ascontiguousarray(g[f"c{cid}_q"])
    out = np.zeros_like(g[f"c{cid}_gt"])
    assert hc.hc_vm_pairing(cid, P(p), None, P(q), 1, P(out)) == 0
    assert np.array_equal(out, g[f"c{cid}_gt"])
    fr = co.CURVE_FR[cid]
    k = co.gen_scalars(fr, 3, seed=91 + cid)
    g1, g2 = co.generator(cid, 1), co.generator(cid, 2)
    ps = np.stack([co.to_affine(cid, 1, co.scalar_mul(cid, 1, g1, k[i]))[0][0] for i in range(3)])
    qs = np.stack([co.to_affine(cid, 2, co.scalar_mul(cid, 2, g2, k[(i + 1) % 3]))[0][0] for i in range(3)])
    assert hc.hc_vm_pairing(cid, P(ps), None, P(qs), 3, P(out)) == 0
    want = np.zeros_like(out)
    assert hc.hc_pairing(cid, P(ps[0]), P(qs[0]), P(want)) == 0          # (the lane-per-pairing templates: golden-checked above)
    assert np.array_equal(co.pairing(cid, ps[0], qs[0]), want)
    # e(P0,Q0) e(P1,Q1) e(P2,Q2) through the oracle's Fq^k product is not exposed: check bilinearity instead -- e(aG, bH)e(bG, cH)e(cG, aH)
    # is symmetric under swapping the roles of the scalars
    ps2 = np.stack([ps[1], ps[2], ps[0]]); qs2 = np.stack([qs[2], qs[0], qs[1]])
    out2 = np.zeros_like(out)
    assert hc.hc_vm_pairing(cid, P(ps2), None, P(qs2), 3, P(out2)) == 0
    assert np.array_equal(out, out2) and out.any()
    assert hc.hc_vm_pairing(cid, P(ps), None, P(qs), 0, P(out2)) == 0     # the empty product: one
    one = out2.reshape(-1, co.FIELD_N64[co.CURVE_FQ[cid]])
    assert one[1:].any() == False and one[0].any()  # noqa: E712
    # Jacobian G1 points (X, Y | Z) = (x z^2, y z^3 | z): the same pairing without any inversion for P
    fq = co.CURVE_FQ[cid]
    L = co.FIELD_N64[fq]
    z = co.gen_field(fq, 3, seed=17 + cid)
    z2 = co.fp_op(fq, "mul", z, z)
    xs = co.fp_op(fq, "mul", np.ascontiguousarray(ps[:, :L]), z2)
    ys = co.fp_op(fq, "mul", np.ascontiguousarray(ps[:, L:]), co.fp_op(fq, "mul", z2, z))
    pj = np.ascontiguousarray(np.concatenate([xs, ys], axis=1))
    assert hc.hc_vm_pairing(cid, P(pj), P(np.ascontiguousarray(z)), P(qs), 3, P(out2)) == 0
    assert np.array_equal(out, out2)


@pytest.mark.parametrize("fid", [0, 1, 2, 3])
def test_signed_sum_reduction(hc, fid):
    """Fp::from_signed_sum (the LIN instruction of the pairing VM, the small-coefficient entries of the mat-vec): sums of up to 16
    terms with coefficient weight up to 4000 (= 2000 on operands below 2p) -- all positive, all negative, alternating, on operands 0, 1, p - 1 and random ones --
    against Python integers"""
    from oracle import pyoracle as O
    f = O.FIELDS[fid]
    rnd = random.Random(40 + fid)
    cases = []
    for T in (1, 2, 8, 16):
        for mode in ("pos", "neg", "alt", "rand"):
            for vals in ("max", "rand", "small"):
                a = [f.p - 1 if vals == "max" else rnd.randrange(f.p) if vals == "rand" else rnd.randrange(3) for _ in range(T)]
                w = 4000 // T   # (canonical operands here: twice the weight stands in for operands up to 2p)
                c = [w if mode == "pos" else -w if mode == "neg" else (w if i % 2 else -w) if mode == "alt" else rnd.randrange(-w, w + 1) for i in range(T)]
                cases.append((a, c))
    for a, c in cases:
        A = O.pack_fp(f, a)
        cc = np.array(c, dtype=np.int32)
        out = np.zeros(f.n64, dtype=np.uint64)
        assert hc.hc_signed_sum(fid, P(A), P(cc), len(a), P(out)) == 0
        assert O.unpack_fp(f, out.reshape(1, -1))[0] == sum(x * y for x, y in zip(a, c)) % f.p, (a, c)


@pytest.mark.parametrize("cid", [0, 1])
def test_lazy_madd_chain(hc, co, cid):
    """EC::madd_lz (the bucket-accumulation step with unreduced X, Y between additions) against the ordinary madd and
    the oracle, over chains long enough for the coordinate bounds to reach their steady state, with repeated points
    (doubling branch), P then -P (cancellation to infinity, then restart) and infinity entries in the stream."""
    w = co.point_words(cid, 1)
    n = 600
    pts = co.gen_points(cid, 1, n, seed=21 + cid)
    pts[7] = pts[6]                      # acc + P where P was just added ... not yet equal to acc
    neg = pts[10].copy()
    one = np.zeros((1, w // 2), dtype=np.uint64); one[0, 0] = 1
    neg[w // 2:] = co.fp_op(co.CURVE_FQ[cid], "neg", pts[10][None, w // 2:])[0]
    seqs = {
        "random": pts,
        "double": np.concatenate([pts[:1], pts[:1], pts[1:50]]),               # acc = P, then + P: doubling branch
        "cancel": np.concatenate([pts[10:11], neg[None], pts[20:60]]),         # P - P = infinity, then continue
        "inf": np.concatenate([pts[:5], np.zeros((2, w), dtype=np.uint64), pts[5:30]]),
        "cancel_mid": np.concatenate([pts[:9], pts[10:11], pts[30:40]]),
    }
    for name, seq in seqs.items():
        seq = np.ascontiguousarray(seq)
        out = np.zeros(2 * 3 * w // 2, dtype=np.uint64)
        assert hc.hc_madd_chain(cid, P(seq), seq.shape[0], P(out)) == 0
        lazy, plain = out[:3 * w // 2], out[3 * w // 2:]
        sc = np.zeros((seq.shape[0], co.FIELD_N64[co.CURVE_FR[cid]]), dtype=np.uint64); sc[:, 0] = 1
        inf = np.array([0 if r.any() else 1 for r in seq], dtype=np.uint8)
        want, winf = co.to_affine(cid, 1, co.msm(cid, 1, seq, sc, inf=inf))
        for got in (lazy, plain):
            g, ginf = co.to_affine(cid, 1, got)
            assert ginf[0] == winf[0] and np.array_equal(g, want), name


@pytest.mark.parametrize("cid,grp", [(0, 1), (0, 2), (1, 2), (2, 1), (3, 2)])
def test_xyzz_madd_chain(hc, co, cid, grp):
    """EC::madd_x (mixed addition in XYZZ coordinates: the bucket-accumulation step of every group without the lazy form) against the
    Jacobian madd and the oracle, with the doubling branch, a cancellation to infinity and infinity entries in the stream."""
    w = co.point_words(cid, grp)
    n = (300 if grp == 2 else 40) if cid < 2 else 14   # (MNT4-298 G2: the lazily reduced form, long enough for its bounds to settle)
    pts = co.gen_points(cid, grp, n, seed=41 + cid)
    neg = pts[3].copy()
    deg = w // 2 // co.FIELD_N64[co.CURVE_FQ[cid]]
    ycoef = pts[3][w // 2:].reshape(deg, -1)
    neg[w // 2:] = co.fp_op(co.CURVE_FQ[cid], "neg", np.ascontiguousarray(ycoef)).reshape(-1)
    seqs = {
        "random": pts,
        "double": np.concatenate([pts[:1], pts[:1], pts[1:8]]),
        "double_late": np.concatenate([pts[:6], pts[6:7], pts[6:7], pts[7:12]]),   # (acc + P)... the same point twice in a row is NOT a doubling of acc; kept as an ordinary case
        "cancel": np.concatenate([pts[3:4], neg[None], pts[4:10]]),
        "inf": np.concatenate([pts[:3], np.zeros((2, w), dtype=np.uint64), pts[3:9]]),
    }
    gi = 2 * cid + (grp - 1)
    for name, seq in seqs.items():
        seq = np.ascontiguousarray(seq)
        out = np.zeros(2 * 3 * w // 2, dtype=np.uint64)
        assert hc.hc_maddx_chain(gi, P(seq), seq.shape[0], P(out)) == 0
        xyzz, plain = out[:3 * w // 2], out[3 * w // 2:]
        sc = np.zeros((seq.shape[0], co.FIELD_N64[co.CURVE_FR[cid]]), dtype=np.uint64); sc[:, 0] = 1
        inf = np.array([0 if r.any() else 1 for r in seq], dtype=np.uint8)
        want, winf = co.to_affine(cid, grp, co.msm(cid, grp, seq, sc, inf=inf))
        for got in (xyzz, plain):
            g, ginf = co.to_affine(cid, grp, got)
            assert ginf[0] == winf[0] and np.array_equal(g, want), name
