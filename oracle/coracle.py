"""TEST INFRASTRUCTURE ONLY -- ctypes/numpy binding of oracle/liboracle.so (the C++ CPU restatement
of the upstream ark-ec / ark-poly / ark-groth16 algorithms; PARITY UNPINNED).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
All arrays are numpy uint64, little-endian limbs, Montgomery form unless stated (encodings of
include/pcdhip.h)."""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

FIELD_N64 = [5, 5, 12, 12]
CURVE_FQ = [0, 1, 2, 3]
CURVE_FR = [1, 0, 3, 2]
CURVE_G2_DEG = [2, 3, 2, 3]
CURVE_NAMES = ["MNT4_298", "MNT6_298", "MNT4_753", "MNT6_753"]


def build(force=False):
    so = os.path.join(_HERE, "liboracle.so")
    srcs = [os.path.join(_HERE, f) for f in os.listdir(_HERE) if f.endswith((".hpp", ".cpp"))]
    if force or not os.path.exists(so) or any(os.path.getmtime(s) > os.path.getmtime(so) for s in srcs):
        subprocess.check_call(["make", "-C", _HERE, "liboracle.so"], stdout=subprocess.DEVNULL)
    return so


def lib():
    global _LIB
    if _LIB is None:
        so = os.environ.get("ORACLE_LIB") or os.path.join(_HERE, "liboracle.so")   # (ORACLE_LIB: the sanitizer build, tools/san/test_sanitizers.py)
        if not os.path.exists(so):
            build()
        _LIB = C.CDLL(so)
        _LIB.orc_synthetic_r1cs_num_vars.restype = C.c_size_t
        _LIB.orc_domain_size.restype = C.c_size_t
    return _LIB


def _p(a):
    if a is None:
        return None
    assert a.flags["C_CONTIGUOUS"]
    return a.ctypes.data_as(C.c_void_p)


def _u64(a):
    return np.ascontiguousarray(a, dtype=np.uint64)


def point_words(curve, group):
    """u64 words per affine point."""
    deg = 1 if group == 1 else CURVE_G2_DEG[curve]
    return 2 * deg * FIELD_N64[CURVE_FQ[curve]]


def fp_op(field, op, a, b=None):
    ops = dict(add=0, sub=1, mul=2, inv=3, from_canonical=4, to_canonical=5, neg=6, sqr=7)
    a = _u64(a)
    out = np.empty_like(a)
    bb = _u64(b) if b is not None else None
    rc = lib().orc_fp_op(field, ops[op], _p(a), _p(bb), _p(out), C.c_size_t(a.shape[0]))
    assert rc == 0
    return out


def msm(curve, group, bases, scalars, inf=None, nthreads=1, c_override=0):
    bases, scalars = _u64(bases), _u64(scalars)
    n = min(bases.shape[0], scalars.shape[0])
    out = np.zeros(3 * point_words(curve, group) // 2, dtype=np.uint64)
    infp = np.ascontiguousarray(inf, dtype=np.uint8) if inf is not None else None
    rc = lib().orc_msm(curve, group, _p(bases), _p(infp), _p(scalars), C.c_size_t(n), nthreads, c_override, _p(out))
    assert rc == 0
    return out


def msm_window(n):
    return lib().orc_msm_window(C.c_size_t(n))


def to_affine(curve, group, xyz):
    xyz = _u64(xyz).reshape(-1, 3 * point_words(curve, group) // 2)
    n = xyz.shape[0]
    xy = np.zeros((n, point_words(curve, group)), dtype=np.uint64)
    inf = np.zeros(n, dtype=np.uint8)
    rc = lib().orc_to_affine(curve, group, _p(xyz), C.c_size_t(n), _p(xy), _p(inf))
    assert rc == 0
    return xy, inf


def jac_add(curve, group, a, b):
    a, b = _u64(a), _u64(b)
    out = np.zeros_like(a)
    assert lib().orc_jac_add(curve, group, _p(a), _p(b), _p(out)) == 0
    return out


def scalar_mul(curve, group, xy, scalar_canonical):
    xy, k = _u64(xy), _u64(scalar_canonical)
    out = np.zeros(3 * point_words(curve, group) // 2, dtype=np.uint64)
    assert lib().orc_scalar_mul(curve, group, _p(xy), _p(k), _p(out)) == 0
    return out


def fixed_base_mul(curve, group, base_xy, scalars_canonical, nthreads=1):
    """FixedBaseMSM::multi_scalar_mul + batch normalisation: (n affine points, flags)."""
    sc = _u64(scalars_canonical).reshape(-1, FIELD_N64[CURVE_FR[curve]])
    n = sc.shape[0]
    out = np.zeros((n, point_words(curve, group)), dtype=np.uint64)
    inf = np.zeros(n, dtype=np.uint8)
    assert lib().orc_fixed_base_mul(curve, group, _p(_u64(base_xy)), _p(sc), C.c_size_t(n), nthreads, _p(out), _p(inf)) == 0
    return out, inf


def on_curve(curve, group, xy):
    return lib().orc_on_curve(curve, group, _p(_u64(xy))) == 1


def generator(curve, group):
    out = np.zeros(point_words(curve, group), dtype=np.uint64)
    assert lib().orc_generator(curve, group, _p(out)) == 0
    return out


def gen_points(curve, group, n, seed):
    out = np.zeros((n, point_words(curve, group)), dtype=np.uint64)
    assert lib().orc_gen_points(curve, group, C.c_size_t(n), C.c_uint64(seed), _p(out)) == 0
    return out


def gen_points_mt(curve, group, n, seed, threads=None):
    """n seeded on-curve points made as independent blocks of 2^16 (block k = gen_points(.., seed + k)) on a thread pool -- ctypes
    releases the GIL, and the single walk of gen_points takes minutes for 2^22 points of a 753-bit group."""
    from concurrent.futures import ThreadPoolExecutor
    blk = 1 << 16
    if n <= blk:
        return gen_points(curve, group, n, seed)
    out = np.zeros((n, point_words(curve, group)), dtype=np.uint64)
    starts = list(range(0, n, blk))

    def work(k):
        lo = starts[k]
        cnt = min(blk, n - lo)
        assert lib().orc_gen_points(curve, group, C.c_size_t(cnt), C.c_uint64(seed + k), _p(out[lo:lo + cnt])) == 0
    with ThreadPoolExecutor(max_workers=threads or min(64, os.cpu_count() or 1)) as ex:
        list(ex.map(work, range(len(starts))))
    return out


def gen_scalars(field, n, seed, dist=0):
    """canonical scalars; dist 0 uniform, 1 witness-like (45% zero / 35% one / 20% uniform)."""
    out = np.zeros((n, FIELD_N64[field]), dtype=np.uint64)
    assert lib().orc_gen_scalars(field, C.c_size_t(n), C.c_uint64(seed), dist, _p(out)) == 0
    return out


def gen_field(field, n, seed):
    out = np.zeros((n, FIELD_N64[field]), dtype=np.uint64)
    assert lib().orc_gen_field(field, C.c_size_t(n), C.c_uint64(seed), _p(out)) == 0
    return out


def fft(field, data, inverse=False, coset=False, nthreads=1):
    data = _u64(data).copy()
    n = data.shape[0]
    log_n = n.bit_length() - 1
    assert 1 << log_n == n
    rc = lib().orc_fft(field, _p(data), log_n, int(inverse), int(coset), nthreads)
    assert rc == 0, rc
    return data


SMALL_SUBGROUP = {0: (7, 2), 2: (5, 2)}  # field id -> (q, adicity) of ark-ff SMALL_SUBGROUP_BASE (help fields only)


def fft_general(field, data, m, inverse=False, coset=False, nthreads=1):
    """transform of size n = m * 2^a (m = 1: radix-2; m = q or q^2: MixedRadixEvaluationDomain)."""
    data = _u64(data).copy()
    n = data.shape[0]
    a = (n // m).bit_length() - 1
    assert m << a == n
    rc = lib().orc_fft_general(field, _p(data), C.c_size_t(m), a, int(inverse), int(coset), nthreads)
    assert rc == 0, rc
    return data


class R1CS:
    """CSR triple + assignment in the C-ABI layout."""

    def __init__(self, field, num_inputs, rp_a, col_a, coeff_a, rp_b, col_b, coeff_b, rp_c, col_c, coeff_c, z):
        self.field, self.num_inputs = field, num_inputs
        self.rp_a, self.col_a, self.coeff_a = rp_a, col_a, coeff_a
        self.rp_b, self.col_b, self.coeff_b = rp_b, col_b, coeff_b
        self.rp_c, self.col_c, self.coeff_c = rp_c, col_c, coeff_c
        self.z = z
        self.num_constraints = len(rp_a) - 1
        self.num_vars = z.shape[0]

    @property
    def domain_log(self):
        need = self.num_constraints + self.num_inputs
        lg = 0
        while (1 << lg) < need:
            lg += 1
        return lg

    def csr_args(self):
        return [_p(self.rp_a), _p(self.col_a), _p(self.coeff_a), _p(self.rp_b), _p(self.col_b), _p(self.coeff_b),
                _p(self.rp_c), _p(self.col_c), _p(self.coeff_c)]


def synthetic_r1cs(field, nc, num_inputs, seed):
    L = FIELD_N64[field]
    m = lib().orc_synthetic_r1cs_num_vars(C.c_size_t(nc), C.c_size_t(num_inputs))
    rp_ab = np.zeros(nc + 1, dtype=np.uint64)
    rp_c = np.zeros(nc + 1, dtype=np.uint64)
    col_a = np.zeros(3 * nc, dtype=np.uint32)
    col_b = np.zeros(3 * nc, dtype=np.uint32)
    col_c = np.zeros(nc, dtype=np.uint32)
    ca = np.zeros((3 * nc, L), dtype=np.uint64)
    cb = np.zeros((3 * nc, L), dtype=np.uint64)
    cc = np.zeros((nc, L), dtype=np.uint64)
    z = np.zeros((m, L), dtype=np.uint64)
    rc = lib().orc_synthetic_r1cs(field, C.c_size_t(nc), C.c_size_t(num_inputs), C.c_uint64(seed), _p(rp_ab), _p(col_a),
                                  _p(ca), _p(col_b), _p(cb), _p(rp_c), _p(col_c), _p(cc), _p(z))
    assert rc == 0
    return R1CS(field, num_inputs, rp_ab, col_a, ca, rp_ab.copy(), col_b, cb, rp_c, col_c, cc, z)


def _generated_r1cs(fn, field, nc, num_inputs, seed):
    L = FIELD_N64[field]
    m = lib().orc_synthetic_r1cs_num_vars(C.c_size_t(nc), C.c_size_t(num_inputs))
    cap = 8 * nc + 16384
    rp = [np.zeros(nc + 1, dtype=np.uint64) for _ in range(3)]
    col = [np.zeros(cap, dtype=np.uint32) for _ in range(3)]
    cf = [np.zeros((cap, L), dtype=np.uint64) for _ in range(3)]
    z = np.zeros((m, L), dtype=np.uint64)
    nnz = np.zeros(3, dtype=np.uint64)
    rc = fn(field, C.c_size_t(nc), C.c_size_t(num_inputs), C.c_uint64(seed), C.c_size_t(cap), _p(rp[0]), _p(col[0]),
            _p(cf[0]), _p(rp[1]), _p(col[1]), _p(cf[1]), _p(rp[2]), _p(col[2]), _p(cf[2]), _p(z), _p(nnz))
    assert rc == 0, rc
    tr = lambda k: (rp[k], np.ascontiguousarray(col[k][:int(nnz[k])]), np.ascontiguousarray(cf[k][:int(nnz[k])]))
    (ra, ca, fa), (rb, cb, fb), (rc_, cc, fc) = tr(0), tr(1), tr(2)
    return R1CS(field, num_inputs, ra, ca, fa, rb, cb, fb, rc_, cc, fc, z)


def skewed_r1cs(field, nc, num_inputs, seed):
    """a satisfied constraint system with the shape of a finalised verifier circuit: power-law row lengths (two rows above 4096
    entries when nc >= 8192), >= 80 % unit coefficients, ~12 % small integers, the rest random (oracle_capi.cpp orc_skewed_r1cs);
    the assignment is uniformly random field elements"""
    return _generated_r1cs(lib().orc_skewed_r1cs, field, nc, num_inputs, seed)


def witness_r1cs(field, nc, num_inputs, seed):
    """a satisfied constraint system whose ASSIGNMENT looks like a verifier circuit's (data_structures.rs:269-304: bit decompositions):
    runs of K bits with their booleanity rows, a packing row, a few product rows -- ~45 % of z is 0, ~35 % is 1, ~20 % neither
    (oracle_capi.cpp orc_witness_r1cs)"""
    return _generated_r1cs(lib().orc_witness_r1cs, field, nc, num_inputs, seed)


def domain_size(field, min_size):
    """size of GeneralEvaluationDomain::new(min_size): radix-2, or mixed radix beyond the field's 2-adicity."""
    return lib().orc_domain_size(field, C.c_size_t(min_size))


def witness_map(r, nthreads=1):
    n = domain_size(r.field, r.num_constraints + r.num_inputs)
    h = np.zeros((n, FIELD_N64[r.field]), dtype=np.uint64)
    rc = lib().orc_witness_map(r.field, C.c_size_t(r.num_constraints), C.c_size_t(r.num_inputs), *r.csr_args(),
                               _p(r.z), nthreads, _p(h))
    assert rc == 0
    return h


class G16PkHost(C.Structure):
    """Mirror of `pcdhip_g16_pk_host` (include/pcdhip.h) == `orc_g16_pk`."""
    _fields_ = [("curve_id", C.c_uint32), ("_pad", C.c_uint32), ("num_vars", C.c_uint64), ("num_inputs", C.c_uint64),
                ("domain_size", C.c_uint64),
                ("alpha_g1", C.c_void_p), ("beta_g1", C.c_void_p), ("delta_g1", C.c_void_p),
                ("beta_g2", C.c_void_p), ("delta_g2", C.c_void_p),
                ("a_query", C.c_void_p), ("a_inf", C.c_void_p),
                ("b_g1_query", C.c_void_p), ("b_g1_inf", C.c_void_p),
                ("b_g2_query", C.c_void_p), ("b_g2_inf", C.c_void_p),
                ("h_query", C.c_void_p), ("h_inf", C.c_void_p), ("h_len", C.c_uint64),
                ("l_query", C.c_void_p), ("l_inf", C.c_void_p), ("l_len", C.c_uint64)]


class Keys:
    """Groth16 proving + verifying key as numpy arrays (kept alive for the ctypes struct)."""

    def __init__(self, curve, r, arrays):
        self.curve = curve
        self.num_vars, self.num_inputs = r.num_vars, r.num_inputs
        self.domain_size = 1 << r.domain_log
        self.__dict__.update(arrays)

    def host_struct(self):
        s = G16PkHost()
        s.curve_id = self.curve
        s.num_vars, s.num_inputs, s.domain_size = self.num_vars, self.num_inputs, self.domain_size
        for name in ("alpha_g1", "beta_g1", "delta_g1", "beta_g2", "delta_g2", "a_query", "a_inf", "b_g1_query",
                     "b_g1_inf", "b_g2_query", "b_g2_inf", "h_query", "h_inf", "l_query", "l_inf"):
            arr = getattr(self, name)
            setattr(s, name, arr.ctypes.data if arr is not None else None)
        s.h_len = self.h_query.shape[0]
        s.l_len = self.l_query.shape[0]
        return s


def synthetic_keys(curve, r, seed, mt=False, consistent=True):
    """Groth16 key made of seeded on-curve points, sized for the domain `GeneralEvaluationDomain::new` picks (radix-2 or
    mixed radix).  A real trusted setup at 2^20 takes minutes on the CPU and the prover arithmetic does not depend on the
    key being consistent; proofs made with such a key are compared bit for bit with this oracle's, not verified.
    mt: the points come from gen_points_mt (other points than with mt=False; for the 2^20+ keys of the 753-bit curves).
    consistent: the queries carry the points at infinity a real setup puts there -- a_query[i] = a_i(tau) G is the identity when no
    row of A mentions variable i (a_i is the zero polynomial), likewise b_g1 / b_g2 for B (ark-groth16 `generate_parameters`; what
    groth16_setup above produces).  False: every entry finite, the densest key there can be."""
    m, ni = r.num_vars, r.num_inputs
    n = domain_size(r.field, r.num_constraints + ni)
    gen = gen_points_mt if mt else gen_points
    g1 = gen(curve, 1, 2 * m + (n - 1) + (m - ni) + 3, seed=seed)
    g2 = gen(curve, 2, m + 2, seed=seed + (1 << 20 if mt else 1))
    z8 = lambda k: np.zeros(k, dtype=np.uint8)
    o = [0]

    def take(k):
        v = np.ascontiguousarray(g1[o[0]:o[0] + k])
        o[0] += k
        return v
    A = dict(a_query=take(m), b_g1_query=take(m), h_query=take(n - 1), l_query=take(m - ni))
    A.update(alpha_g1=take(1)[0], beta_g1=take(1)[0], delta_g1=take(1)[0])
    A.update(b_g2_query=np.ascontiguousarray(g2[:m]), beta_g2=np.ascontiguousarray(g2[m]), delta_g2=np.ascontiguousarray(g2[m + 1]),
             gamma_g2=np.ascontiguousarray(g2[m + 1]), gamma_abc_g1=np.ascontiguousarray(g1[:ni]), gamma_abc_inf=z8(ni),
             a_inf=z8(m), b_g1_inf=z8(m), b_g2_inf=z8(m), h_inf=z8(n - 1), l_inf=z8(m - ni))
    if consistent:
        for (rp, mat, cf), names in (((r.rp_a, r.col_a, r.coeff_a), ("a_inf",)), ((r.rp_b, r.col_b, r.coeff_b), ("b_g1_inf", "b_g2_inf"))):
            col = np.asarray(mat, dtype=np.int64)
            live = np.asarray(cf).any(axis=1)                        # (an entry with coefficient zero mentions nothing)
            # entries of one row on the same column add up: a pair c, -c (the generator makes some) leaves the variable out of the row
            rows = np.repeat(np.arange(len(rp) - 1, dtype=np.int64), np.diff(np.asarray(rp).astype(np.int64)))
            key = rows * m + col
            order = np.argsort(key, kind="stable")
            ks = key[order]
            dup = np.flatnonzero(ks[1:] == ks[:-1])
            if len(dup):
                starts = np.flatnonzero(np.concatenate(([True], ks[1:] != ks[:-1])))
                ends = np.concatenate((starts[1:], [len(ks)]))
                big = np.flatnonzero(ends - starts > 1)
                for s0, e0 in zip(starts[big], ends[big]):
                    if True:
                        idx = order[s0:e0]
                        acc = np.ascontiguousarray(np.asarray(cf)[idx[:1]])
                        for j in idx[1:]:
                            acc = fp_op(r.field, "add", acc, np.ascontiguousarray(np.asarray(cf)[j:j + 1]))
                        live[idx] = bool(acc.any())
            absent = np.ones(m, dtype=np.uint8)
            absent[col[live]] = 0
            if names[0] == "a_inf":
                absent[:ni] = 0   # (the input-consistency rows of the QAP put every public input into A)
            for nm in names:
                A[nm] = absent.copy()
    keys = Keys(curve, r, A)
    keys.domain_size = n
    return keys


def groth16_setup(curve, r, toxic_mont, nthreads=1):
    """toxic_mont: (5, L) Montgomery limbs of (alpha, beta, gamma, delta, tau)."""
    w1, w2 = point_words(curve, 1), point_words(curve, 2)
    m, ni, n = r.num_vars, r.num_inputs, 1 << r.domain_log
    z64 = lambda *s: np.zeros(s, dtype=np.uint64)
    z8 = lambda k: np.zeros(k, dtype=np.uint8)
    A = dict(alpha_g1=z64(w1), beta_g1=z64(w1), delta_g1=z64(w1), beta_g2=z64(w2), delta_g2=z64(w2), gamma_g2=z64(w2),
             a_query=z64(m, w1), a_inf=z8(m), b_g1_query=z64(m, w1), b_g1_inf=z8(m), b_g2_query=z64(m, w2),
             b_g2_inf=z8(m), h_query=z64(n - 1, w1), h_inf=z8(n - 1), l_query=z64(m - ni, w1), l_inf=z8(m - ni),
             gamma_abc_g1=z64(ni, w1), gamma_abc_inf=z8(ni))
    toxic_mont = _u64(toxic_mont)
    rc = lib().orc_groth16_setup(curve, C.c_size_t(r.num_constraints), C.c_size_t(m), C.c_size_t(ni), *r.csr_args(),
                                 _p(toxic_mont), nthreads,
                                 *[_p(A[k]) for k in ("alpha_g1", "beta_g1", "delta_g1", "beta_g2", "delta_g2", "gamma_g2",
                                                      "a_query", "a_inf", "b_g1_query", "b_g1_inf", "b_g2_query", "b_g2_inf",
                                                      "h_query", "h_inf", "l_query", "l_inf", "gamma_abc_g1", "gamma_abc_inf")])
    assert rc == 0
    return Keys(curve, r, A)


def groth16_prove(keys, r, r_mont, s_mont, nthreads=1):
    curve = keys.curve
    w1, w2 = point_words(curve, 1), point_words(curve, 2)
    proof = np.zeros(2 * w1 + w2, dtype=np.uint64)
    inf = np.zeros(3, dtype=np.uint8)
    s = keys.host_struct()
    rc = lib().orc_groth16_prove(C.byref(s), C.c_size_t(r.num_constraints), *r.csr_args(), _p(r.z), _p(_u64(r_mont)),
                                 _p(_u64(s_mont)), nthreads, _p(proof), _p(inf))
    assert rc == 0
    return proof, inf


def groth16_verify(keys, public_inputs_mont, proof, proof_inf=None):
    pi = _u64(public_inputs_mont)
    rc = lib().orc_groth16_verify(keys.curve, _p(keys.alpha_g1), _p(keys.beta_g2), _p(keys.gamma_g2), _p(keys.delta_g2),
                                  _p(keys.gamma_abc_g1), _p(keys.gamma_abc_inf), C.c_size_t(keys.num_inputs), _p(pi),
                                  _p(_u64(proof)), _p(proof_inf))
    assert rc in (0, 1)
    return rc == 1


def pairing(curve, g1_xy, g2_xy):
    k = 2 * CURVE_G2_DEG[curve]
    out = np.zeros(k * FIELD_N64[CURVE_FQ[curve]], dtype=np.uint64)
    assert lib().orc_pairing(curve, _p(_u64(g1_xy)), _p(_u64(g2_xy)), _p(out)) == 0
    return out
