#!/usr/bin/env python3
"""Derive every field / curve constant of the MNT4/MNT6 298- and 753-bit cycles from
first principles (the four primes, the curve coefficients, and the generators) and emit

  * pcd_amd/csrc/params_gen.h   -- 32-bit-limb tables for the HIP kernels
  * oracle/params_gen.hpp       -- 64-bit-limb tables for the CPU oracle
  * oracle/params.json          -- decimal values for the Python big-int oracle

Nothing here is copied from the reference (which contains no constants at all: the
arithmetic lives in un-vendored upstream crates, Cargo.toml:16-42).  Every value is
checked mathematically below (primality is assumed from SURVEY.md Appendix B; on-curve
and group-order checks are run here), so a wrong recalled constant fails this script.

Field ids   0: F298A = MNT4-298 Fq = MNT6-298 Fr     (2-adicity 17)
            1: F298B = MNT4-298 Fr = MNT6-298 Fq     (2-adicity 34)
            2: F753A = MNT4-753 Fq = MNT6-753 Fr     (2-adicity 15)
            3: F753B = MNT4-753 Fr = MNT6-753 Fq     (2-adicity 30)
Curve ids   0: MNT4_298 (Fq=F298A, Fr=F298B)   1: MNT6_298 (Fq=F298B, Fr=F298A)
            2: MNT4_753 (Fq=F753A, Fr=F753B)   3: MNT6_753 (Fq=F753B, Fr=F753A)
"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

Q298 = 475922286169261325753349249653048451545124879242694725395555128576210262817955800483758081
R298 = 475922286169261325753349249653048451545124878552823515553267735739164647307408490559963137
Q753 = 41898490967918953402344214791240637128170709919953949071783502921025352812571106773058893763790338921418070971888253786114353726529584385201591605722013126468931404347949840543007986327743462853720628051692141265303114721689601
R753 = 41898490967918953402344214791240637128170709919953949071783502921025352812571106773058893763790338921418070971888458477323173057491593855069696241854796396165721416325350064441470418137846398469611935719059908164220784476160001

# name, modulus, multiplicative generator (upstream `GENERATOR`; primitive-root test below)
FIELDS = [
    ("F298A", Q298, 17),
    ("F298B", R298, 10),
    ("F753A", Q753, 17),
    ("F753B", R753, 17),
]

# curve: name, base field idx, scalar field idx, a, b, embedding degree k, twist non-residue
CURVES = [
    dict(name="MNT4_298", fq=0, fr=1, a=2, k=4, nr=17,
         b=423894536526684178289416011533888240029318103673896002803341544124054745019340795360841685,
         g1=(60760244141852568949126569781626075788424196370144486719385562369396875346601926534016838,
             363732850702582978263902770815145784459747722357071843971107674179038674942891694705904306),
         g2=((438374926219350099854919100077809681842783509163790991847867546339851681564223481322252708,
              37620953615500480110935514360923278605464476459712393277679280819942849043649216370485641),
             (37437409008528968268352521034936931842973546441370663118543015118291998305624025037512482,
              424621479598893882672393190337420680597584695892317197646113820787463109735345923009077489))),
    dict(name="MNT6_298", fq=1, fr=0, a=11, k=6, nr=5,
         b=106700080510851735677967319632585352256454251201367587890185989362936000262606668469523074,
         g1=(336685752883082228109289846353937104185698209371404178342968838739115829740084426881123453,
             402596290139780989709332707716568920777622032073762749862342374583908837063963736098549800),
         g2=((421456435772811846256826561593908322288509115489119907560382401870203318738334702321297427,
              103072927438548502463527009961344915021167584706439945404959058962657261178393635706405114,
              143029172143731852627002926324735183809768363301149009204849580478324784395590388826052558),
             (464673596668689463130099227575639512541218133445388869383893594087634649237515554342751377,
              100642907501977375184575075967118071807821117960152743335603284583254620685343989304941678,
              123019855502969896026940545715841181300275180157288044663051565390506010149881373807142903))),
    dict(name="MNT4_753", fq=2, fr=3, a=2, k=4, nr=13,
         b=28798803903456388891410036793299405764940372360099938340752576406393880372126970068421383312482853541572780087363938442377933706865252053507077543420534380486492786626556269083255657125025963825610840222568694137138741554679540,
         g1=None, g2=None),
    dict(name="MNT6_753", fq=3, fr=2, a=11, k=6, nr=11,
         b=0x7DA285E70863C79D56446237CE2E1468D14AE9BB64B2BB01B10E60A5D5DFE0A25714B7985993F62F03B22A9A3C737A1A1E0FCF2C43D7BF847957C34CCA1E3585F9A80A95F401867C4E80F4747FDE5ABA7505BA6FCF2485540B13DFC8468A,
         g1=None, g2=None),
]


# ----------------------------------------------------------------------------- big-int helpers
def sqrt_mod(n, p):
    n %= p
    if n == 0:
        return 0
    if pow(n, (p - 1) // 2, p) != 1:
        return None
    s, t = 0, p - 1
    while t % 2 == 0:
        s += 1
        t //= 2
    z = 2
    while pow(z, (p - 1) // 2, p) != p - 1:
        z += 1
    c, x, b, m = pow(z, t, p), pow(n, (t + 1) // 2, p), pow(n, t, p), s
    while b != 1:
        i, bb = 0, b
        while bb != 1:
            bb = bb * bb % p
            i += 1
        g = pow(c, 1 << (m - i - 1), p)
        x, c, m = x * g % p, g * g % p, i
        b = b * c % p
    return x


class Ext:
    """Fp[u]/(u^d - nr), elements are d-tuples (c0, c1, ...)."""

    def __init__(self, p, d, nr):
        self.p, self.d, self.nr = p, d, nr

    def zero(self):
        return (0,) * self.d

    def one(self):
        return (1,) + (0,) * (self.d - 1)

    def add(self, a, b):
        return tuple((x + y) % self.p for x, y in zip(a, b))

    def sub(self, a, b):
        return tuple((x - y) % self.p for x, y in zip(a, b))

    def neg(self, a):
        return tuple((-x) % self.p for x in a)

    def mul(self, a, b):
        d = self.d
        r = [0] * (2 * d - 1)
        for i in range(d):
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
        return self.pow(a, self.p ** self.d - 2)

    def sqrt(self, a):
        """Square root in F_{p^d} for odd d via exponent trick when p^d = 3 mod 4 is not
        available: use Tonelli-Shanks over the extension."""
        if a == self.zero():
            return a
        qd = self.p ** self.d
        if self.pow(a, (qd - 1) // 2) != self.one():
            return None
        s, t = 0, qd - 1
        while t % 2 == 0:
            s += 1
            t //= 2
        # find a non-residue deterministically
        z = None
        c = 2
        while z is None:
            cand = (c, 1) + (0,) * (self.d - 2)
            if self.pow(cand, (qd - 1) // 2) != self.one():
                z = cand
            c += 1
        cc = self.pow(z, t)
        x = self.pow(a, (t + 1) // 2)
        b = self.pow(a, t)
        m = s
        while b != self.one():
            i, bb = 0, b
            while bb != self.one():
                bb = self.mul(bb, bb)
                i += 1
            g = cc
            for _ in range(m - i - 1):
                g = self.mul(g, g)
            x = self.mul(x, g)
            cc = self.mul(g, g)
            b = self.mul(b, cc)
            m = i
        return x


class Prime(Ext):
    def __init__(self, p):
        super().__init__(p, 1, 0)


def ec_add(F, P, Q, a):
    if P is None:
        return Q
    if Q is None:
        return P
    x1, y1 = P
    x2, y2 = Q
    if x1 == x2:
        if F.add(y1, y2) == F.zero():
            return None
        x1s = F.mul(x1, x1)
        num = F.add(F.add(F.add(x1s, x1s), x1s), a)
        den = F.add(y1, y1)
    else:
        num, den = F.sub(y2, y1), F.sub(x2, x1)
    lam = F.mul(num, F.inv(den))
    x3 = F.sub(F.sub(F.mul(lam, lam), x1), x2)
    return (x3, F.sub(F.mul(lam, F.sub(x1, x3)), y1))


def ec_mul(F, k, P, a):
    R = None
    while k:
        if k & 1:
            R = ec_add(F, R, P, a)
        P = ec_add(F, P, P, a)
        k >>= 1
    return R


def on_curve(F, P, a, b):
    x, y = P
    return F.mul(y, y) == F.add(F.add(F.mul(F.mul(x, x), x), F.mul(a, x)), b)


def twist_coeffs(c, p):
    """G2 twist: E'/F_{q^(k/2)}: y^2 = x^3 + a'x + b' with twist element u.
    MNT4 (Fq2, u^2 = nr):  a' = a*u^2 = (a*nr, 0)        b' = b*u^3 = (0, b*nr)
    MNT6 (Fq3, u^3 = nr):  a' = a*u^2 = (0, 0, a)        b' = b*u^3 = (b*nr, 0, 0)"""
    a, b, nr = c["a"], c["b"], c["nr"]
    if c["k"] == 4:
        return (a * nr % p, 0), (0, b * nr % p)
    return (0, 0, a), (b * nr % p, 0, 0)


def derive_g1(c, p, order):
    F = Prime(p)
    x = 1
    while True:
        y = sqrt_mod(x ** 3 + c["a"] * x + c["b"], p)
        if y is not None and y != 0:
            y = min(y, p - y)
            P = ((x,), (y,))
            assert ec_mul(F, order, P, (c["a"],)) is None
            return (x, y)
        x += 1


def derive_g2(c, p, order):
    """Hash-by-increment onto the twist, then clear the cofactor."""
    d = c["k"] // 2
    F = Ext(p, d, c["nr"])
    a2, b2 = twist_coeffs(c, p)
    t = p + 1 - order  # trace of E over Fq (#E(Fq) = order, cofactor 1)
    if d == 2:
        td = t * t - 2 * p
    else:
        td = t ** 3 - 3 * p * t
    n_plus, n_minus = p ** d + 1 + td, p ** d + 1 - td
    cands = [n for n in (n_plus, n_minus) if n % order == 0]
    x0 = 1
    while True:
        x = (x0, 1) + (0,) * (d - 2)
        rhs = F.add(F.add(F.mul(F.mul(x, x), x), F.mul(a2, x)), b2)
        y = F.sqrt(rhs)
        if y is not None:
            assert F.mul(y, y) == rhs
            P = (x, y)
            for n in cands:
                Q = ec_mul(F, n // order, P, a2)
                if Q is not None and ec_mul(F, order, Q, a2) is None and on_curve(F, Q, a2, b2):
                    return Q
        x0 += 1


# ----------------------------------------------------------------------------- emitters
def limbs(x, n, bits):
    mask = (1 << bits) - 1
    return [(x >> (bits * i)) & mask for i in range(n)]


def c_arr(vals, bits):
    suf = "u" if bits == 32 else "ull"
    w = 8 if bits == 32 else 16
    return "{" + ", ".join(f"0x{v:0{w}x}{suf}" for v in vals) + "}"


def small_prime_factors(n, bound=10000):
    fs = []
    d = 2
    while d < bound:
        if n % d == 0:
            fs.append(d)
            while n % d == 0:
                n //= d
        d += 1
    return fs, n


def main():
    out_json = {"fields": [], "curves": []}
    fields = []
    for name, p, gen in FIELDS:
        bits = p.bit_length()
        n64 = (bits + 63) // 64
        n32 = 2 * n64
        R = 1 << (64 * n64)
        s, t = 0, p - 1
        while t % 2 == 0:
            s += 1
            t //= 2
        # primitive-root test on the small prime factors of p-1
        # (p-1 has a large cofactor we cannot factor here, so this is the small-factor part of the
        # primitive-root test; what the FFT needs -- root has exact order 2^s and gen^(2^s) != 1, so
        # the coset g*<w> is disjoint from <w> -- is asserted exactly.)
        fs, _rest = small_prime_factors(p - 1)
        assert pow(gen, 1 << s, p) != 1
        for f in fs:
            assert pow(gen, (p - 1) // f, p) != 1, (name, "generator not primitive", f)
        root = pow(gen, t, p)  # 2^s-th primitive root of unity = GENERATOR^T
        assert pow(root, 1 << s, p) == 1 and pow(root, 1 << (s - 1), p) == p - 1
        f = dict(name=name, p=p, bits=bits, n64=n64, n32=n32, R=R % p, R2=R * R % p, R3=R * R * R % p,
                 inv32=(-pow(p, -1, 1 << 32)) % (1 << 32), inv64=(-pow(p, -1, 1 << 64)) % (1 << 64),
                 two_adicity=s, generator=gen, root=root, t=t)
        fields.append(f)
        out_json["fields"].append({k: (str(v) if isinstance(v, int) and v > 2 ** 53 else v) for k, v in f.items()})

    curves = []
    for c in CURVES:
        p = fields[c["fq"]]["p"]
        order = fields[c["fr"]]["p"]
        F1 = Prime(p)
        if c["g1"] is None:
            c["g1"] = derive_g1(c, p, order)
        g1 = ((c["g1"][0],), (c["g1"][1],))
        assert on_curve(F1, g1, (c["a"],), (c["b"],)), c["name"]
        assert ec_mul(F1, order, g1, (c["a"],)) is None, c["name"]
        d = c["k"] // 2
        F2 = Ext(p, d, c["nr"])
        # nr must be a non-d-th-residue
        assert pow(c["nr"], (p - 1) // d, p) != 1
        a2, b2 = twist_coeffs(c, p)
        if c["g2"] is None:
            c["g2"] = derive_g2(c, p, order)
        assert on_curve(F2, c["g2"], a2, b2), c["name"]
        assert ec_mul(F2, order, c["g2"], a2) is None, c["name"]
        # embedding degree
        assert pow(p, c["k"], order) == 1 and all(pow(p, j, order) != 1 for j in range(1, c["k"]))
        # ate loop count |t-1| = |q - r| and final-exponent split
        loop = p - order
        c["ate_neg"] = loop < 0
        c["ate_loop"] = abs(loop)
        k = c["k"]
        if k == 4:
            hard = (p * p + 1) // order
            assert (p * p + 1) % order == 0
            w0 = hard - p  # hard = q*1 + w0
        else:
            hard = (p * p - p + 1) // order
            assert (p * p - p + 1) % order == 0
            w0 = hard - p
        c["w0_neg"] = w0 < 0
        c["w0"] = abs(w0)
        c["a2"], c["b2"] = a2, b2
        curves.append(c)
        out_json["curves"].append(dict(
            name=c["name"], fq=c["fq"], fr=c["fr"], a=c["a"], b=str(c["b"]), k=k, nr=c["nr"],
            g1=[str(v) for v in c["g1"]], g2=[[str(v) for v in comp] for comp in c["g2"]],
            a2=[str(v) for v in a2], b2=[str(v) for v in b2],
            ate_loop=str(c["ate_loop"]), ate_neg=c["ate_neg"], w0=str(c["w0"]), w0_neg=c["w0_neg"]))

    hdr = ["// GENERATED by tools/gen_params.py -- do not edit.", "#pragma once", "#include <stdint.h>", ""]
    dev = list(hdr)
    ora = list(hdr)
    for i, f in enumerate(fields):
        p = f["p"]
        for (lines, bits, n) in ((dev, 32, f["n32"]), (ora, 64, f["n64"])):
            ty = "uint32_t" if bits == 32 else "uint64_t"
            pre = f"PCD_{f['name']}"
            lines.append(f"// field {i}: {f['name']}  ({f['bits']} bits, 2-adicity {f['two_adicity']}, generator {f['generator']})")
            lines.append(f"#define {pre}_ID {i}")
            lines.append(f"#define {pre}_BITS {f['bits']}")
            lines.append(f"#define {pre}_N{bits} {n}")
            lines.append(f"#define {pre}_TWO_ADICITY {f['two_adicity']}")
            lines.append(f"#define {pre}_INV{bits} 0x{(f['inv32'] if bits == 32 else f['inv64']):x}{'u' if bits == 32 else 'ull'}")
            mont = lambda x: x * (1 << (64 * f["n64"])) % p
            for nm, val in (("MOD", p), ("R", f["R"]), ("R2", f["R2"]), ("R3", f["R3"]),
                            ("GEN_MONT", mont(f["generator"])), ("ROOT_MONT", mont(f["root"])),
                            ("MOD_MINUS_2", p - 2)):
                lines.append(f"#define {pre}_{nm} {c_arr(limbs(val, n, bits), bits)}")
            lines.append("")
    for i, c in enumerate(curves):
        f = fields[c["fq"]]
        p = f["p"]
        mont = lambda x: x * (1 << (64 * f["n64"])) % p
        for (lines, bits, n) in ((dev, 32, f["n32"]), (ora, 64, f["n64"])):
            pre = f"PCD_{c['name']}"
            lines.append(f"// curve {i}: {c['name']}: y^2 = x^3 + {c['a']}x + b over field {c['fq']}, order = field {c['fr']} modulus, k = {c['k']}")
            lines.append(f"#define {pre}_ID {i}")
            lines.append(f"#define {pre}_A_MONT {c_arr(limbs(mont(c['a']), n, bits), bits)}")
            lines.append(f"#define {pre}_B_MONT {c_arr(limbs(mont(c['b']), n, bits), bits)}")
            lines.append(f"#define {pre}_NR_MONT {c_arr(limbs(mont(c['nr']), n, bits), bits)}")
            lines.append(f"#define {pre}_NR_SMALL {c['nr']}")
            lines.append(f"#define {pre}_A_SMALL {c['a']}")
            flat = lambda tup: sum((limbs(mont(v), n, bits) for v in tup), [])
            lines.append(f"#define {pre}_TWIST_A_MONT {c_arr(flat(c['a2']), bits)}")
            lines.append(f"#define {pre}_TWIST_B_MONT {c_arr(flat(c['b2']), bits)}")
            lines.append(f"#define {pre}_G1_MONT {c_arr(flat(c['g1']), bits)}")
            lines.append(f"#define {pre}_G2_MONT {c_arr(flat(c['g2'][0]) + flat(c['g2'][1]), bits)}")
            nl = (c["ate_loop"].bit_length() + bits - 1) // bits
            lines.append(f"#define {pre}_ATE_LOOP_BITS {c['ate_loop'].bit_length()}")
            lines.append(f"#define {pre}_ATE_LOOP_NLIMBS {nl}")
            lines.append(f"#define {pre}_ATE_LOOP {c_arr(limbs(c['ate_loop'], nl, bits), bits)}")
            lines.append(f"#define {pre}_ATE_NEG {int(c['ate_neg'])}")
            nw = (c["w0"].bit_length() + bits - 1) // bits
            lines.append(f"#define {pre}_W0_BITS {c['w0'].bit_length()}")
            lines.append(f"#define {pre}_W0_NLIMBS {nw}")
            lines.append(f"#define {pre}_W0 {c_arr(limbs(c['w0'], nw, bits), bits)}")
            lines.append(f"#define {pre}_W0_NEG {int(c['w0_neg'])}")
            lines.append("")

    os.makedirs(os.path.join(ROOT, "pcd_amd", "csrc"), exist_ok=True)
    os.makedirs(os.path.join(ROOT, "oracle"), exist_ok=True)
    with open(os.path.join(ROOT, "pcd_amd", "csrc", "params_gen.h"), "w") as fh:
        fh.write("\n".join(dev) + "\n")
    with open(os.path.join(ROOT, "oracle", "params_gen.hpp"), "w") as fh:
        fh.write("\n".join(ora) + "\n")
    with open(os.path.join(ROOT, "oracle", "params.json"), "w") as fh:
        json.dump(out_json, fh, indent=1)
    print("ok: wrote params_gen.h, params_gen.hpp, params.json")
    for c in curves:
        print(c["name"], "ate_loop bits", c["ate_loop"].bit_length(), "neg", c["ate_neg"],
              "w0 bits", c["w0"].bit_length(), "neg", c["w0_neg"])


if __name__ == "__main__":
    sys.exit(main())
