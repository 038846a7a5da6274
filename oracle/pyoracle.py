"""TEST INFRASTRUCTURE ONLY -- pure-Python big-integer oracle for the PCD prover hot path.

PARITY UNPINNED: the reference (/root/reference) contains neither the
arithmetic of this path (it lives in un-vendored, un-pinned arkworks git dependencies,
Cargo.toml:16-42) nor a single golden vector for it (tests/*.rs are prove->verify round
trips, e.g. tests/mnt4_groth16.rs:86-87,119), and no Rust toolchain exists here to run it.
This module therefore pins results *mathematically*: every function below computes the
unique value the upstream function must return (an MSM is a group element, a DFT over a
stated (omega, g) is a vector, a reduced pairing is an element of GT, a Groth16 proof for
given (r, s) is three affine points), with Python `int` arithmetic and textbook affine
formulas -- sharing no code, limb layout or formula with oracle/*.hpp (the C++ restatement
of the upstream *algorithms*) or with the HIP kernels.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this.

Call sites in the reference that reach the functions mirrored here:
  src/ec_cycle_pcd/mod.rs:171,179   MainSNARK::prove / HelpSNARK::prove  -> witness_map, msm
  src/ec_cycle_pcd/mod.rs:239       HelpSNARK::verify                    -> pairing
  src/ec_cycle_pcd/mod.rs:69,78     circuit_specific_setup               -> groth16_setup
"""
import json
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
with open(os.path.join(_HERE, "params.json")) as _fh:
    PARAMS = json.load(_fh)


def _int(v):
    return int(v) if isinstance(v, str) else v


class Field:
    """Prime field descriptor (ids as in tools/gen_params.py)."""

    def __init__(self, d):
        self.name = d["name"]
        self.p = _int(d["p"])
        self.n64 = d["n64"]
        self.bits = d["bits"]
        self.two_adicity = d["two_adicity"]
        self.generator = _int(d["generator"])
        self.root = _int(d["root"])
        self.R = 1 << (64 * self.n64)

    # Montgomery <-> canonical, limb packing (little-endian u64 limbs, as ark-ff BigInteger)
    def to_mont(self, x):
        return x * self.R % self.p

    def from_mont(self, x):
        return x * pow(self.R, -1, self.p) % self.p

    def limbs(self, x):
        return [(x >> (64 * i)) & 0xFFFFFFFFFFFFFFFF for i in range(self.n64)]

    def unlimbs(self, ls):
        return sum(int(v) << (64 * i) for i, v in enumerate(ls))

    def domain_root(self, log_n):
        assert log_n <= self.two_adicity
        return pow(self.root, 1 << (self.two_adicity - log_n), self.p)

    def domain_root_general(self, n):
        """generator^((p-1)/n): the group generator of a radix-2 OR mixed-radix domain of size n (upstream:
        TWO_ADIC_ROOT / LARGE_SUBGROUP_ROOT_OF_UNITY are both powers of GENERATOR)."""
        assert (self.p - 1) % n == 0
        return pow(self.generator, (self.p - 1) // n, self.p)


FIELDS = [Field(d) for d in PARAMS["fields"]]


class Ext:
    """F_p[u]/(u^d - nr); d = 1 is the prime field (elements are 1-tuples)."""

    def __init__(self, p, d, nr):
        self.p, self.d, self.nr = p, d, nr

    def zero(self):
        return (0,) * self.d

    def one(self):
        return (1,) + (0,) * (self.d - 1)

    def lift(self, x):
        return (x % self.p,) + (0,) * (self.d - 1)

    def add(self, a, b):
        return tuple((x + y) % self.p for x, y in zip(a, b))

    def sub(self, a, b):
        return tuple((x - y) % self.p for x, y in zip(a, b))

    def neg(self, a):
        return tuple((-x) % self.p for x in a)

    def mul(self, a, b):
        d = self.d
        if d == 1:
            return (a[0] * b[0] % self.p,)
        r = [0] * (2 * d - 1)
        for i in range(d):
            if a[i]:
                for j in range(d):
                    r[i + j] += a[i] * b[j]
        for k in range(2 * d - 2, d - 1, -1):
            r[k - d] += r[k] * self.nr
        return tuple(x % self.p for x in r[:d])

    def pow(self, a, e):
        r = self.one()
        while e:
            if e & 1:
                r = self.mul(r, a)
            a = self.mul(a, a)
            e >>= 1
        return r

    def inv(self, a):
        p, nr = self.p, self.nr
        if self.d == 1:
            return (pow(a[0], -1, p),)
        if self.d == 2:  # 1/(a0 + a1 u) = (a0 - a1 u)/(a0^2 - nr a1^2)
            ninv = pow((a[0] * a[0] - nr * a[1] * a[1]) % p, -1, p)
            return (a[0] * ninv % p, (-a[1]) * ninv % p)
        if self.d == 3:  # adjugate / norm
            a0, a1, a2 = a
            t0 = (a0 * a0 - nr * a1 * a2) % p
            t1 = (nr * a2 * a2 - a0 * a1) % p
            t2 = (a1 * a1 - a0 * a2) % p
            ninv = pow((a0 * t0 + nr * (a2 * t1 + a1 * t2)) % p, -1, p)
            return (t0 * ninv % p, t1 * ninv % p, t2 * ninv % p)
        return self.pow(a, p ** self.d - 2)


class Curve:
    def __init__(self, d):
        self.name = d["name"]
        self.fq = FIELDS[d["fq"]]
        self.fr = FIELDS[d["fr"]]
        self.k = d["k"]
        self.nr = d["nr"]
        self.a = d["a"]
        self.b = _int(d["b"])
        p = self.fq.p
        self.F1 = Ext(p, 1, 0)
        self.F2 = Ext(p, self.k // 2, self.nr)  # twist field
        self.Fk = Ext(p, self.k, self.nr)       # F_{q^k} = Fq[v]/(v^k - nr), v^2 = u
        self.a1 = (self.a,)
        self.b1 = (self.b,)
        self.a2 = tuple(_int(v) for v in d["a2"])
        self.b2 = tuple(_int(v) for v in d["b2"])
        self.g1 = tuple((_int(v),) for v in d["g1"])
        self.g2 = tuple(tuple(_int(v) for v in comp) for comp in d["g2"])
        self.ate_loop = _int(d["ate_loop"])
        self.ate_neg = d["ate_neg"]

    def group(self, g):
        """(field, a) of G1 (g=1) or G2 (g=2)."""
        return (self.F1, self.a1) if g == 1 else (self.F2, self.a2)


CURVES = [Curve(d) for d in PARAMS["curves"]]
CURVE_BY_NAME = {c.name: c for c in CURVES}


# ----------------------------------------------------------------------------- group law (affine, textbook)
def ec_add(F, a, P, Q):
    if P is None:
        return Q
    if Q is None:
        return P
    x1, y1 = P
    x2, y2 = Q
    if x1 == x2:
        if F.add(y1, y2) == F.zero():
            return None
        xx = F.mul(x1, x1)
        num = F.add(F.add(F.add(xx, xx), xx), a)
        den = F.add(y1, y1)
    else:
        num, den = F.sub(y2, y1), F.sub(x2, x1)
    lam = F.mul(num, F.inv(den))
    x3 = F.sub(F.sub(F.mul(lam, lam), x1), x2)
    return (x3, F.sub(F.mul(lam, F.sub(x1, x3)), y1))


def ec_neg(F, P):
    return None if P is None else (P[0], F.neg(P[1]))


def ec_mul(F, a, k, P):
    R = None
    while k:
        if k & 1:
            R = ec_add(F, a, R, P)
        P = ec_add(F, a, P, P)
        k >>= 1
    return R


def msm_naive(F, a, bases, scalars):
    """sum_i k_i * P_i by double-and-add: the unique value VariableBaseMSM::multi_scalar_mul
    [ark-ec msm/variable_base.rs, reached from mod.rs:171,179] must return."""
    acc = None
    for P, k in zip(bases, scalars):
        if k and P is not None:
            acc = ec_add(F, a, acc, ec_mul(F, a, k, P))
    return acc


# ----------------------------------------------------------------------------- DFT / domains
def dft_naive(fld, xs, inverse=False):
    n = len(xs)
    log_n = n.bit_length() - 1
    assert 1 << log_n == n
    p = fld.p
    w = fld.domain_root(log_n)
    if inverse:
        w = pow(w, -1, p)
    out = []
    for k in range(n):
        wk = pow(w, k, p)
        acc, cur = 0, 1
        for j in range(n):
            acc += xs[j] * cur
            cur = cur * wk % p
        out.append(acc % p)
    if inverse:
        ninv = pow(n, -1, p)
        out = [v * ninv % p for v in out]
    return out


def fft(fld, xs, inverse=False, coset=False):
    """In-order radix-2 DFT with the upstream domain definition (Radix2EvaluationDomain):
    fft: X_k = sum_j x_j w^{jk};  ifft: w^-1 and scale by 1/n;  coset_fft: x_j *= g^j first;
    coset_ifft: ifft then x_j *= g^-j."""
    n = len(xs)
    log_n = n.bit_length() - 1
    assert 1 << log_n == n
    p = fld.p
    g = fld.generator
    xs = list(xs)
    if coset and not inverse:
        cur = 1
        for j in range(n):
            xs[j] = xs[j] * cur % p
            cur = cur * g % p
    w = fld.domain_root(log_n)
    if inverse:
        w = pow(w, -1, p)

    def rec(v, w):
        m = len(v)
        if m == 1:
            return v
        e = rec(v[0::2], w * w % p)
        o = rec(v[1::2], w * w % p)
        out = [0] * m
        cur = 1
        h = m // 2
        for i in range(h):
            t = cur * o[i] % p
            out[i] = (e[i] + t) % p
            out[i + h] = (e[i] - t) % p
            cur = cur * w % p
        return out

    out = rec(xs, w)
    if inverse:
        ninv = pow(n, -1, p)
        out = [v * ninv % p for v in out]
        if coset:
            ginv = pow(g, -1, p)
            cur = 1
            for j in range(n):
                out[j] = out[j] * cur % p
                cur = cur * ginv % p
    return out


def dft_general(fld, xs, inverse=False, coset=False):
    """Naive DFT over the size-n subgroup for ANY n | p-1 (mixed-radix domains n = 2^a q^b of
    ark-poly MixedRadixEvaluationDomain), same conventions as `fft`."""
    n = len(xs)
    p, g = fld.p, fld.generator
    w = fld.domain_root_general(n)
    xs = list(xs)
    if coset and not inverse:
        xs = [x * pow(g, j, p) % p for j, x in enumerate(xs)]
    if inverse:
        w = pow(w, -1, p)
    pw = [pow(w, i, p) for i in range(n)]
    out = [sum(xs[j] * pw[(j * k) % n] for j in range(n)) % p for k in range(n)]
    if inverse:
        ninv = pow(n, -1, p)
        out = [v * ninv % p for v in out]
        if coset:
            ginv = pow(g, -1, p)
            out = [v * pow(ginv, j, p) % p for j, v in enumerate(out)]
    return out


def best_mixed_domain_size(fld, min_size, q, q_adicity=2):
    """ark-poly `best_mixed_domain_size`: smallest 2^a q^b >= min_size with b <= q_adicity, a <= two-adicity."""
    best = None
    for b in range(q_adicity + 1):
        r, a = q ** b, 0
        while r < min_size:
            r *= 2
            a += 1
        if a <= fld.two_adicity and (best is None or r < best):
            best = r
    return best


# ----------------------------------------------------------------------------- R1CS / QAP / Groth16
class R1CS:
    """A, B, C as lists of rows; a row is a list of (coeff, column).  z = [1, inputs..., witness...]."""

    def __init__(self, fld, num_inputs, A, B, C, z):
        self.fld, self.num_inputs, self.A, self.B, self.C, self.z = fld, num_inputs, A, B, C, z
        self.num_constraints = len(A)
        self.num_vars = len(z)

    def domain_log(self):
        need = self.num_constraints + self.num_inputs
        log_n = 0
        while (1 << log_n) < need:
            log_n += 1
        return log_n

    def is_satisfied(self):
        p = self.fld.p
        dot = lambda row: sum(c * self.z[j] for c, j in row) % p
        return all(dot(a) * dot(b) % p == dot(c) for a, b, c in zip(self.A, self.B, self.C))


def synthetic_r1cs(fld, num_constraints, num_inputs, seed):
    """Banded synthetic R1CS: constraint j multiplies two random 3-term combinations of earlier
    variables and defines a new witness variable as the product (satisfying by construction)."""
    import random
    rnd = random.Random(seed)
    p = fld.p
    z = [1] + [rnd.randrange(p) for _ in range(num_inputs - 1)]
    while len(z) < max(num_inputs, 4):
        z.append(rnd.randrange(p))  # a few free witness variables so every row has 3 sources
    A, B, C = [], [], []
    for _ in range(num_constraints):
        m = len(z)
        lo = max(0, m - 8)
        ra = [(rnd.randrange(1, p), rnd.randrange(lo, m)) for _ in range(3)]
        rb = [(rnd.randrange(1, p), rnd.randrange(lo, m)) for _ in range(3)]
        va = sum(c * z[j] for c, j in ra) % p
        vb = sum(c * z[j] for c, j in rb) % p
        z.append(va * vb % p)
        A.append(ra)
        B.append(rb)
        C.append([(1, m)])
    return R1CS(fld, num_inputs, A, B, C, z)


def _poly_mul(p, a, b):
    out = [0] * (len(a) + len(b) - 1)
    for i, x in enumerate(a):
        if x:
            for j, y in enumerate(b):
                out[i + j] = (out[i + j] + x * y) % p
    return out


def witness_map_naive(r):
    """h(X) = (A(X)B(X) - C(X)) / Z(X) by interpolation + exact polynomial division.  This is the
    unique polynomial R1CSToQAP::witness_map (libsnark reduction) must return (n coefficients,
    top one zero) -- independent of the coset generator used by the FFT pipeline."""
    fld = r.fld
    p = fld.p
    log_n = r.domain_log()
    n = 1 << log_n
    dot = lambda row: sum(c * r.z[j] for c, j in row) % p
    a = [dot(row) for row in r.A] + [r.z[j] for j in range(r.num_inputs)]
    b = [dot(row) for row in r.B] + [0] * r.num_inputs
    c = [dot(row) for row in r.C] + [0] * r.num_inputs
    pad = lambda v: v + [0] * (n - len(v))
    ac = fft(fld, pad(a), inverse=True)
    bc = fft(fld, pad(b), inverse=True)
    cc = fft(fld, pad(c), inverse=True)
    prod = _poly_mul(p, ac, bc)
    for i, v in enumerate(cc):
        prod[i] = (prod[i] - v) % p
    # divide by X^n - 1 : prod = h * (X^n - 1)  =>  prod[i] = h[i-n] - h[i]
    deg = len(prod)
    h = [0] * n
    rem = list(prod) + [0] * max(0, 2 * n - deg)
    for i in range(2 * n - 1, n - 1, -1):
        if i - n < n:
            h[i - n] = rem[i] % p
            rem[i - n] = (rem[i - n] + rem[i]) % p
            rem[i] = 0
    assert all(v % p == 0 for v in rem[:n]), "R1CS not satisfied: division by Z is not exact"
    return h


def lagrange_at(fld, log_n, tau):
    """[L_0(tau), ..., L_{n-1}(tau)] over the radix-2 domain."""
    p = fld.p
    n = 1 << log_n
    w = fld.domain_root(log_n)
    zt = (pow(tau, n, p) - 1) % p
    assert zt != 0
    out = []
    wi = 1
    ninv = pow(n, -1, p)
    for _ in range(n):
        # L_i(tau) = Z(tau) * w^i / (n * (tau - w^i))
        out.append(zt * wi % p * ninv % p * pow((tau - wi) % p, -1, p) % p)
        wi = wi * w % p
    return out


def groth16_setup(curve, r, toxic):
    """generate_parameters [ark-groth16 generator.rs, reached from mod.rs:69,78] with fixed toxic
    waste (alpha, beta, gamma, delta, tau) and the fixed group generators."""
    fld = r.fld
    assert fld is curve.fr
    p = fld.p
    alpha, beta, gamma, delta, tau = toxic
    log_n = r.domain_log()
    n = 1 << log_n
    L = lagrange_at(fld, log_n, tau)
    m = r.num_vars
    At, Bt, Ct = [0] * m, [0] * m, [0] * m
    for j, (ra, rb, rc) in enumerate(zip(r.A, r.B, r.C)):
        for c, col in ra:
            At[col] = (At[col] + c * L[j]) % p
        for c, col in rb:
            Bt[col] = (Bt[col] + c * L[j]) % p
        for c, col in rc:
            Ct[col] = (Ct[col] + c * L[j]) % p
    for i in range(r.num_inputs):
        At[i] = (At[i] + L[r.num_constraints + i]) % p
    F1, a1 = curve.group(1)
    F2, a2 = curve.group(2)
    g1m = lambda k: ec_mul(F1, a1, k % p, curve.g1)
    g2m = lambda k: ec_mul(F2, a2, k % p, curve.g2)
    zt = (pow(tau, n, p) - 1) % p
    dinv, ginv = pow(delta, -1, p), pow(gamma, -1, p)
    pk = dict(
        alpha_g1=g1m(alpha), beta_g1=g1m(beta), delta_g1=g1m(delta),
        beta_g2=g2m(beta), delta_g2=g2m(delta), gamma_g2=g2m(gamma),
        a_query=[g1m(v) for v in At], b_g1_query=[g1m(v) for v in Bt], b_g2_query=[g2m(v) for v in Bt],
        h_query=[g1m(pow(tau, i, p) * zt % p * dinv) for i in range(n - 1)],
        l_query=[g1m((beta * At[i] + alpha * Bt[i] + Ct[i]) % p * dinv) for i in range(r.num_inputs, m)],
        gamma_abc_g1=[g1m((beta * At[i] + alpha * Bt[i] + Ct[i]) % p * ginv) for i in range(r.num_inputs)],
    )
    return pk


def groth16_prove(curve, pk, r, rr, ss):
    """create_proof [ark-groth16 prover.rs] -- SURVEY.md Appendix A.1, with (r, s) as inputs."""
    p = r.fld.p
    F1, a1 = curve.group(1)
    F2, a2 = curve.group(2)
    h = witness_map_naive(r)
    add1 = lambda P, Q: ec_add(F1, a1, P, Q)
    add2 = lambda P, Q: ec_add(F2, a2, P, Q)
    h_acc = msm_naive(F1, a1, pk["h_query"], h[:len(pk["h_query"])])
    aux = r.z[r.num_inputs:]
    l_acc = msm_naive(F1, a1, pk["l_query"], aux)
    asg = r.z[1:]
    g_a = add1(add1(add1(ec_mul(F1, a1, rr, pk["delta_g1"]), pk["a_query"][0]),
                    msm_naive(F1, a1, pk["a_query"][1:], asg)), pk["alpha_g1"])
    g1_b = add1(add1(add1(ec_mul(F1, a1, ss, pk["delta_g1"]), pk["b_g1_query"][0]),
                     msm_naive(F1, a1, pk["b_g1_query"][1:], asg)), pk["beta_g1"])
    g2_b = add2(add2(add2(ec_mul(F2, a2, ss, pk["delta_g2"]), pk["b_g2_query"][0]),
                     msm_naive(F2, a2, pk["b_g2_query"][1:], asg)), pk["beta_g2"])
    rs_delta = ec_mul(F1, a1, rr * ss % p, pk["delta_g1"])
    g_c = add1(add1(add1(add1(ec_mul(F1, a1, ss, g_a), ec_mul(F1, a1, rr, g1_b)),
                         ec_neg(F1, rs_delta)), l_acc), h_acc)
    return g_a, g2_b, g_c


# ----------------------------------------------------------------------------- pairing (textbook reduced ate)
def _embed_twist(curve, e):
    """F_{q^(k/2)} (basis u^j) -> F_{q^k} = Fq[v]/(v^k - nr) with u = v^2."""
    out = [0] * curve.k
    for j, c in enumerate(e):
        out[2 * j] = c
    return tuple(out)


def miller_loop(curve, P, Q):
    """f_{|T|, psi(Q)}(P) with T = q - r (trace - 1), affine slopes on the twist, lines evaluated in
    F_{q^k}; vertical lines omitted (they lie in a proper subfield).  Inverted when T < 0."""
    Fk, F2 = curve.Fk, curve.F2
    if P is None or Q is None:
        return Fk.one()
    k = curve.k
    v = tuple(1 if i == 1 else 0 for i in range(k))
    vinv = Fk.inv(v)
    vinv3 = Fk.mul(vinv, Fk.mul(vinv, vinv))
    xP, yP = Fk.lift(P[0][0]), Fk.lift(P[1][0])

    def line(R, lam):
        # untwisted: y_P - lam/v * x_P + (lam*x_R - y_R)/v^3
        lam_k = _embed_twist(curve, lam)
        c = _embed_twist(curve, F2.sub(F2.mul(lam, R[0]), R[1]))
        t = Fk.sub(yP, Fk.mul(Fk.mul(lam_k, vinv), xP))
        return Fk.add(t, Fk.mul(c, vinv3))

    f = Fk.one()
    R = Q
    bits = bin(curve.ate_loop)[3:]
    for b in bits:
        xx = F2.mul(R[0], R[0])
        lam = F2.mul(F2.add(F2.add(F2.add(xx, xx), xx), curve.a2), F2.inv(F2.add(R[1], R[1])))
        f = Fk.mul(Fk.mul(f, f), line(R, lam))
        R = ec_add(F2, curve.a2, R, R)
        if b == "1":
            if R[0] == Q[0]:
                R = ec_add(F2, curve.a2, R, Q)  # vertical or tangent: cannot happen below the order
                continue
            lam = F2.mul(F2.sub(Q[1], R[1]), F2.inv(F2.sub(Q[0], R[0])))
            f = Fk.mul(f, line(R, lam))
            R = ec_add(F2, curve.a2, R, Q)
    if curve.ate_neg:
        f = Fk.inv(f)
    return f


def final_exponentiation(curve, f):
    q, r, k = curve.fq.p, curve.fr.p, curve.k
    assert (q ** k - 1) % r == 0
    return curve.Fk.pow(f, (q ** k - 1) // r)


def pairing(curve, P, Q):
    return final_exponentiation(curve, miller_loop(curve, P, Q))


def fk_to_tower(curve, e):
    """F_{q^k} flat (coefficients of v^i) -> tower order used by the C-ABI:
    (c0, c1) with c_i in F_{q^(k/2)} = (a_i0, a_i1[, a_i2]);  coefficient of v^(2j+i) is a_ij."""
    d = curve.k // 2
    return [[e[2 * j + i] for j in range(d)] for i in range(2)]


def groth16_verify(curve, pk, public_inputs, proof):
    """e(A,B) == e(alpha,beta) * e(sum x_i gamma_abc_i, gamma) * e(C, delta)  (x_0 = 1);
    the check HelpSNARK::verify performs at mod.rs:239."""
    F1, a1 = curve.group(1)
    Fk = curve.Fk
    A, B, C = proof
    acc = pk["gamma_abc_g1"][0]
    for x, g in zip(public_inputs, pk["gamma_abc_g1"][1:]):
        acc = ec_add(F1, a1, acc, ec_mul(F1, a1, x, g))
    lhs = pairing(curve, A, B)
    rhs = Fk.mul(Fk.mul(pairing(curve, pk["alpha_g1"], pk["beta_g2"]), pairing(curve, acc, pk["gamma_g2"])),
                 pairing(curve, C, pk["delta_g2"]))
    return lhs == rhs


# ----------------------------------------------------------------------------- limb packing (numpy)
def pack_fp(fld, vals, mont=True):
    """list of ints -> (n, N64) uint64 array (Montgomery form unless mont=False)."""
    import numpy as np
    out = np.zeros((len(vals), fld.n64), dtype=np.uint64)
    for i, v in enumerate(vals):
        out[i] = fld.limbs(fld.to_mont(v) if mont else v)
    return out


def unpack_fp(fld, arr, mont=True):
    import numpy as np
    arr = np.asarray(arr, dtype=np.uint64).reshape(-1, fld.n64)
    vals = [fld.unlimbs(row) for row in arr]
    return [fld.from_mont(v) for v in vals] if mont else vals


def pack_points(curve, group, pts):
    """affine points (ext-tuples or None) -> ((n, words) uint64 Montgomery, (n,) uint8 infinity flags)."""
    import numpy as np
    fld = curve.fq
    d = 1 if group == 1 else curve.k // 2
    out = np.zeros((len(pts), 2 * d * fld.n64), dtype=np.uint64)
    inf = np.zeros(len(pts), dtype=np.uint8)
    for i, P in enumerate(pts):
        if P is None:
            inf[i] = 1
            continue
        flat = list(P[0]) + list(P[1])
        out[i] = pack_fp(fld, flat).reshape(-1)
    return out, inf


def unpack_points(curve, group, arr, inf=None):
    import numpy as np
    fld = curve.fq
    d = 1 if group == 1 else curve.k // 2
    arr = np.asarray(arr, dtype=np.uint64).reshape(-1, 2 * d * fld.n64)
    out = []
    for i, row in enumerate(arr):
        if inf is not None and inf[i]:
            out.append(None)
            continue
        v = unpack_fp(fld, row.reshape(2 * d, fld.n64))
        out.append((tuple(v[:d]), tuple(v[d:])))
    return out


def unpack_jacobian(curve, group, xyz):
    """X||Y||Z Montgomery limbs -> affine ext-tuples (or None), via big-int division."""
    import numpy as np
    fld = curve.fq
    d = 1 if group == 1 else curve.k // 2
    F, _ = curve.group(group)
    v = unpack_fp(fld, np.asarray(xyz, dtype=np.uint64).reshape(3 * d, fld.n64))
    X, Y, Z = tuple(v[:d]), tuple(v[d:2 * d]), tuple(v[2 * d:])
    if Z == F.zero():
        return None
    zi = F.inv(Z)
    zi2 = F.mul(zi, zi)
    return (F.mul(X, zi2), F.mul(Y, F.mul(zi2, zi)))


# ----------------------------------------------------------------------------- wire format (ark-serialize CanonicalSerialize, restated)
def _ser_elem(curve, e, flags=0):
    nb = (curve.fq.bits + 7) // 8
    out = bytearray()
    for c in e:
        out += int(c).to_bytes(nb, "little")
    out[-1] |= flags
    return bytes(out)


def _y_is_larger(F, y):
    ny = F.neg(y)
    return tuple(reversed(y)) > tuple(reversed(ny))   # most significant coefficient first, integers per coefficient


def serialize_point(curve, group, P, compressed=True):
    """affine point (ext-tuples) or None -> bytes: flags 0x80 = y is the larger root, 0x40 = infinity, in the top bits of the last byte"""
    F, _ = curve.group(group)
    if P is None:
        return _ser_elem(curve, F.zero(), 0x40) if compressed else _ser_elem(curve, F.zero()) + _ser_elem(curve, F.one(), 0x40)
    x, y = P
    if compressed:
        return _ser_elem(curve, x, 0x80 if _y_is_larger(F, y) else 0)
    return _ser_elem(curve, x) + _ser_elem(curve, y)


def serialize_proof(curve, proof, compressed=True):
    A, B, Cc = proof
    return serialize_point(curve, 1, A, compressed) + serialize_point(curve, 2, B, compressed) + serialize_point(curve, 1, Cc, compressed)


def serialize_vk(curve, alpha_g1, beta_g2, gamma_g2, delta_g2, gamma_abc_g1, compressed=True):
    out = serialize_point(curve, 1, alpha_g1, compressed)
    for Q in (beta_g2, gamma_g2, delta_g2):
        out += serialize_point(curve, 2, Q, compressed)
    out += len(gamma_abc_g1).to_bytes(8, "little")
    for P in gamma_abc_g1:
        out += serialize_point(curve, 1, P, compressed)
    return out
