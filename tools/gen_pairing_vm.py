#!/usr/bin/env python3
"""Compile the MNT4 / MNT6 ate pairing into straight-line programs for the wave-wide field VM (pcd_amd/csrc/pairing_vm.hip.h).

K6 of SURVEY.md section 8: `PairingEngine::{miller_loop, final_exponentiation}` behind `Groth16::verify`
(/root/reference src/ec_cycle_pcd/mod.rs:239) and `process_vk` (mod.rs:71).  One pairing in one lane is a chain of ~13 000 dependent
field products; here ONE WAVE runs one pairing: every value lives in an LDS register, a program is a list of steps, and in a step
up to 64 lanes execute one instruction each --

    MUL   dst = (sum_{t < T} A_t * B_t) / R'  mod p        (T <= TMAX products into one column accumulator, one Montgomery reduction)
    LIN   dst = sum_{t < 8} c_t * A_t        mod p        (small signed integer coefficients)

both closed on [0, 2p).  Fq4 / Fq6 are the binomial extensions Fq[v]/(v^k - nr) (u = v^2 spans the twist field Fq2 / Fq3), so every
tower operation is a sparse polynomial product whose coefficient sums of equal weight become MUL instructions on sibling lanes and
whose weights (1, 2, nr, 2 nr) are applied by the LIN that follows.  This script traces the formulas (the ones of pairing.hip.h, i.e.
ark-ec's flipped Miller loop with extended Jacobian coordinates, written once over that polynomial ring), levels the dependency
graph (LIN on even slots, MUL on odd ones), allocates registers (state registers are double-banked per slot so that a program may
overwrite its inputs), evaluates the programs with Python integers -- tests/test_pairing_vm.py compares that evaluation with the
textbook pairing of the test oracle -- and writes pcd_amd/csrc/pairing_vm_gen.h.

    python tools/gen_pairing_vm.py            # regenerate the header
"""
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
B28 = 28
TMAX = 1          # products per MUL instruction (1: every product on a lane of its own; a step then costs 2 N^2 multiply-adds, not 3 N^2)
LIN_TERMS = 16    # terms per LIN instruction (more than 8: the instruction takes a second, continuation slot)
LIN_WEIGHT = 2000  # sum of |coefficients| a LIN may carry (inputs below 2p: Fp::from_signed_sum reduces values below 2^12 p)
LANES = 64

K_LIN, K_MUL, K_SQR = 0, 1, 2   # (K_SQR: a MUL step whose instructions are all squares: N (N + 1) / 2 + N^2 multiply-adds)
SP_REG, SP_IN, SP_OUT, SP_TAB = 0, 1, 2, 3   # operand spaces: plain register, state slot (current / other bank), entry `sel` of a table
SET_SEL = 0xF0   # script entries >= SET_SEL are not programs: they set the table selector to (entry - SET_SEL)
SCRIPT_INV = 0xEF   # script entry: register ft1_0 <- 1 / register nrm0, by lane 0 (Fp::inv), not a program


def params():
    return json.load(open(os.path.join(ROOT, "oracle", "params.json")))


# ------------------------------------------------------------------------------------------------ program builder
class Prog:
    """One straight-line program.  Values are node ids (None = zero).  Leaves: state slots and plain registers (constants, inputs)."""

    def __init__(self, env, name):
        self.env, self.name = env, name
        self.nodes = []       # (kind, payload): ('leaf', (space, idx)) | ('mul', [(a, b)]) | ('lin', [(c, a)])
        self.leaf_cache = {}
        self.outputs = []     # (state slot, node)
        self.reg_outputs = [] # (plain register index, node): values handed to later programs of the same kernel without banking

    def leaf(self, space, idx):
        key = (space, idx)
        if key not in self.leaf_cache:
            self.nodes.append(("leaf", key))
            self.leaf_cache[key] = len(self.nodes) - 1
        return self.leaf_cache[key]

    def state(self, name):
        return self.leaf(SP_IN, self.env.slot(name))

    def reg(self, name):
        return self.leaf(SP_REG, self.env.reg(name))

    def const(self, value):
        return self.leaf(SP_REG, self.env.const(value))

    def tab(self, table, off):
        """coefficient `off` of the entry the script selected in `table` (0: powers of nrm, 1: powers of the w0 base)"""
        return self.leaf(SP_TAB, (table, off))

    def mul(self, terms):
        terms = [(a, b) for a, b in terms if a is not None and b is not None]
        if not terms:
            return None
        assert len(terms) <= TMAX
        self.nodes.append(("mul", terms))
        return len(self.nodes) - 1

    def lin(self, terms):
        """sum c * a with integer c; LIN operands that are themselves LIN nodes are expanded (one LIN level between products)"""
        p = self.env.p
        flat = {}

        def put(c, a):
            if a is None or c % p == 0:
                return
            flat[a] = flat.get(a, 0) + c
        for c, a in terms:
            if a is None:
                continue
            kind, pay = self.nodes[a]
            if kind == "lin":
                for c2, a2 in pay:
                    put(c * c2, a2)
            else:
                put(c, a)
        out = [(c, a) for a, c in flat.items() if c != 0]
        if not out:
            return None
        if len(out) == 1 and out[0][0] == 1:
            return out[0][1]
        if len(out) > LIN_TERMS or sum(abs(c) for c, _ in out) > LIN_WEIGHT:
            # too wide to expand: keep the LIN operands as values of their own
            out2 = {}
            for c, a in terms:
                if a is not None and c != 0:
                    out2[a] = out2.get(a, 0) + c
            out = [(c, a) for a, c in out2.items() if c != 0]
            assert len(out) <= LIN_TERMS and sum(abs(c) for c, _ in out) <= LIN_WEIGHT, (self.name, out)
        self.nodes.append(("lin", out))
        return len(self.nodes) - 1

    def out(self, name, node):
        self.outputs.append((self.env.slot(name), node))

    def out_reg(self, name, node):
        self.reg_outputs.append((self.env.reg(name), node))


class X:
    """element of Fq[v]/(v^k - nr) with sparse coefficients (node ids of a Prog, None = 0)"""

    def __init__(self, P, c):
        self.P, self.c = P, list(c)

    @property
    def k(self):
        return len(self.c)

    def _lin2(self, o, ca, cb):
        return X(self.P, [self.P.lin([(ca, a), (cb, b)]) for a, b in zip(self.c, o.c)])

    def __add__(self, o):
        return self._lin2(o, 1, 1)

    def __sub__(self, o):
        return self._lin2(o, 1, -1)

    def neg(self):
        return X(self.P, [self.P.lin([(-1, a)]) for a in self.c])

    def times(self, m):
        return X(self.P, [self.P.lin([(m, a)]) for a in self.c])

    def dbl(self):
        return self.times(2)

    def shift(self, power, coeff=1):
        """self * coeff * v^power"""
        nr, k = self.P.env.nr, self.k
        out = [None] * k
        for j, a in enumerate(self.c):
            m = (j + power) % k
            wraps = (j + power) // k
            out[m] = self.P.lin([(coeff * nr ** wraps, a)])
        return X(self.P, out)

    def __mul__(self, o):
        P, k, nr = self.P, self.k, self.P.env.nr
        same = self.c == o.c
        groups = [dict() for _ in range(k)]   # output coefficient -> weight -> list of products
        for i, a in enumerate(self.c):
            for j, b in enumerate(o.c):
                if a is None or b is None:
                    continue
                w = nr if i + j >= k else 1
                if same:
                    if j < i:
                        continue
                    if j > i:
                        w *= 2
                groups[(i + j) % k].setdefault(w, []).append((a, b))
        out = []
        for m in range(k):
            terms = []
            for w, prods in sorted(groups[m].items()):
                for s in range(0, len(prods), TMAX):
                    terms.append((w, P.mul(prods[s:s + TMAX])))
            out.append(P.lin(terms))
        return X(P, out)

    def sqr(self):
        return self * self

    def frob(self, i):
        """x^(q^i): coefficient j times w^(i j), w = nr^((q - 1) / k)"""
        env = self.P.env
        out = []
        for j, a in enumerate(self.c):
            e = (i * j) % self.k
            out.append(a if (a is None or e == 0) else self.P.mul([(a, self.P.const(env.frob_w[e]))]))
        return X(self.P, out)

    def even(self):
        return X(self.P, [a if j % 2 == 0 else None for j, a in enumerate(self.c)])

    def odd_over_v(self):
        """A1 with self = A0(v^2) + v A1(v^2), as an even polynomial"""
        out = [None] * self.k
        for j, a in enumerate(self.c):
            if j % 2 == 1:
                out[j - 1] = a
        return X(self.P, out)

    def conj(self):
        return X(self.P, [a if j % 2 == 0 else self.P.lin([(-1, a)]) for j, a in enumerate(self.c)])


# ------------------------------------------------------------------------------------------------ one curve
class Env:
    def __init__(self, cid):
        PR = params()
        c = PR["curves"][cid]
        f = PR["fields"][int(c["fq"])]
        self.cid, self.name = cid, c["name"]
        self.p = int(f["p"])
        self.field_name = f["name"]
        self.N = 11 if f["bits"] < 320 else 27
        self.k = int(c["k"])
        self.nr = int(c["nr"])
        self.a = int(c["a"])
        self.ate_loop, self.ate_neg = int(c["ate_loop"]), c["ate_neg"] in (True, "True")
        self.w0, self.w0_neg = int(c["w0"]), c["w0_neg"] in (True, "True")
        assert (self.p - 1) % self.k == 0
        w = pow(self.nr, (self.p - 1) // self.k, self.p)
        self.frob_w = [pow(w, i, self.p) for i in range(self.k)]
        self.slots, self.regs, self.consts = {}, {}, {}
        self.nslots = 0
        self.scripts = {}
        self.const_values = []   # plain registers 0 .. : constants first, then named registers
        self.progs = {}

    def slot(self, name):
        # (the final exponentiation's slots reuse the numbers of the Miller loop's: two kernels, two register files -- and the bank word
        #  holds 31 slots)
        if name not in self.slots:
            m = re.fullmatch(r"(acc|pw)(\d+)", name)
            if m:
                self.slots[name] = self.slot(("f" if m.group(1) == "acc" else "ln") + m.group(2))
            else:
                self.slots[name] = self.nslots
                self.nslots += 1
        return self.slots[name]

    def const(self, value):
        value %= self.p
        if value not in self.consts:
            self.consts[value] = len(self.const_values)
            self.const_values.append(value)
        return ("c", self.consts[value])

    def reg(self, name):
        if name not in self.regs:
            self.regs[name] = len(self.regs)
        return ("r", self.regs[name])

    def prog(self, name):
        P = Prog(self, name)
        self.progs[name] = P
        return P

    # ---- named groups of registers / slots
    def st(self, P, name, idxs):
        return X(P, [P.state(f"{name}{j}") if j in idxs else None for j in range(self.k)])

    def rg(self, P, name, idxs):
        return X(P, [P.reg(f"{name}{j}") if j in idxs else None for j in range(self.k)])

    def put(self, P, name, x, idxs):
        for j in idxs:
            node = x.c[j]
            if node is None:
                node = P.lin([(0, None)]) or P.const(0)
            P.out(f"{name}{j}", node)

    def put_reg(self, P, name, x, idxs):
        for j in idxs:
            node = x.c[j] if x.c[j] is not None else P.const(0)
            P.out_reg(f"{name}{j}", node)


def build(cid):
    env = Env(cid)
    k, nr = env.k, env.nr
    EV = list(range(0, k, 2))    # coefficient positions of a twist-field element (even powers of v)
    ALL = list(range(k))
    one = lambda P: X(P, [P.const(1)] + [None] * (k - 1))

    def mul_by_a(x):             # times the twist's a' = a u^2 = a v^4
        return x.shift(4, env.a)

    # ---------------- Miller loop (one wave per pair) ----------------
    # plain registers: px0, py0 (G1 point), qx*, qy* (G2 point, even positions), derived once by `setup`:
    #   qyo* = qy / twist, l1c* = px - qx / twist      (twist = u = v^2;  1 / u = u^(d-1) / nr)
    # The G1 point arrives in JACOBIAN form (px, py, pz): x = px / pz^2, y = py / pz^3 (pz = 1 for an affine point).  Every line is
    # evaluated times pz^3 -- an element of Fq*, which the final exponentiation kills -- so no inversion is ever needed for P:
    #   doubling line  (c_l - 4c) pz^3 - c_j (px pz) u,   c_h py u          addition line  oz py u,  -(qyo pz^3 oz + (px pz - qxo pz^3) l1)
    P = env.prog("setup")
    px, py, pz = P.reg("px0"), P.reg("py0"), P.reg("pz0")
    qx, qy = env.rg(P, "qx", EV), env.rg(P, "qy", EV)
    inv_nr = pow(nr, -1, env.p)
    pz2 = P.mul([(pz, pz)])
    pz3 = P.mul([(pz2, pz)])
    pxz = P.mul([(px, pz)])
    # x / u = x * v^(k-2) / nr
    def over_twist(x):
        sh = x.shift(k - 2)
        return X(P, [None if a is None else P.mul([(a, P.const(inv_nr))]) for a in sh.c])
    qxo, qyo = over_twist(qx), over_twist(qy)
    scale = lambda x, f: X(P, [None if a is None else P.mul([(a, f)]) for a in x.c])
    l1c = X(P, [pxz] + [None] * (k - 1)) - scale(qxo, pz3)
    env.put_reg(P, "qyo", scale(qyo, pz3), EV)
    env.put_reg(P, "l1c", l1c, EV)
    P.out_reg("pxz0", pxz)
    P.out_reg("pz30", pz3)
    # state: r = (x, y, z, t = z^2, c = z^3) <- (qx, qy, 1, 1, 1), f <- 1
    env.put(P, "rx", qx, EV); env.put(P, "ry", qy, EV)
    env.put(P, "rz", one(P), EV); env.put(P, "rt", one(P), EV); env.put(P, "rc", one(P), EV)
    # f travels as a PAIR: the state slots f (call it s) and ln (the line of the step before, not multiplied in yet); the Miller value is
    # s * ln.  A doubling then is  s <- (s ln)^2, ln <- its own line: the product s ln and the squaring sit on the first two product levels,
    # beside the point arithmetic whose line is only complete after the third -- the program is three product levels deep where
    # f <- f^2 * line behind the line made it four (7 steps instead of 9, on two thirds of the loop).  `flush` multiplies the last line in.
    env.put(P, "f", one(P), ALL)
    env.put(P, "ln", one(P), ALL)

    def line_regs(P):
        pxt = X(P, [None, None, P.reg("pxz0")] + [None] * (k - 3))   # px pz * twist
        pyt = X(P, [None, None, P.reg("py0")] + [None] * (k - 3))    # py * twist
        return pxt, pyt

    P = env.prog("dbl")
    x, y, z, t = (env.st(P, n, EV) for n in ("rx", "ry", "rz", "rt"))
    f, ln = env.st(P, "f", ALL), env.st(P, "ln", ALL)
    pxt, pyt = line_regs(P)
    # (upstream's doubling -- ark-ec mnt4 / mnt6 `doubling_step_for_flipped_miller_loop` -- takes its mixed terms from squares, e.g.
    #  2 y z = (y + z)^2 - y^2 - z^2: one product saved in a lane.  Here products on sibling lanes are free and every term of a LIN
    #  costs time, so the mixed terms are products: the values are the same, the sum in front of the first level disappears (6 steps
    #  instead of 7) and the widest LIN has 5 terms instead of 8.)
    a, b, c = t.sqr(), x.sqr(), y.sqr()
    d = c.sqr()
    e = (x * c).dbl()
    fq = b.times(3) + mul_by_a(a)
    g = fq.sqr()
    ox = g - e.times(4)
    oy = fq * (e.dbl() - ox) - d.times(8)
    oz = (y * z).dbl()
    ot = oz.sqr()
    c_h = (oz * t).dbl()
    c_j = (fq * t).dbl()
    c_l = (fq * x).dbl()
    pz3 = P.reg("pz30")
    c_l4 = c_l - c.times(4)
    g_rr = (X(P, [None if a_ is None else P.mul([(a_, pz3)]) for a_ in c_l4.c]) - c_j * pxt) + (c_h * pyt).shift(1)
    for n, v in (("rx", ox), ("ry", oy), ("rz", oz), ("rt", ot), ("rc", oz * ot)):
        env.put(P, n, v, EV)
    env.put(P, "f", (f * ln).sqr(), ALL)
    env.put(P, "ln", g_rr, ALL)

    # mixed addition (madd-2007-bl on the extended coordinates), THREE product levels deep like the doubling.  Upstream's order
    # (ark-ec mnt4 / mnt6 `mixed_addition_step_for_flipped_miller_loop`) is four: d = ((z + qy)^2 - qy^2 - t) t and
    # y3 = l1 (v - x3) - 2 y j both wait for a product of the level before.  Here d = 2 qy z^3 comes straight from the cached cube
    # (state rc; (z + qy)^2 - qy^2 - z^2 = 2 z qy), so l1 = d - 2y is known after the first level, and y3 is expanded in products of
    # second-level values:  y3 = l1 (3v + j - l1^2) - 2 y j = (12 l1 x + 4 l1 h - 8 y h) i - l1^2 l1   (i = h^2, v = 4 x i, j = 4 h i).
    # The same values as upstream's, coefficient by coefficient (tests/test_pairing_vm.py: the pairing against the textbook one).
    P = env.prog("add")
    x, y, z, t, zc = (env.st(P, n, EV) for n in ("rx", "ry", "rz", "rt", "rc"))
    f, ln = env.st(P, "f", ALL), env.st(P, "ln", ALL)
    pxt, pyt = line_regs(P)
    qx, qy = env.rg(P, "qx", EV), env.rg(P, "qy", EV)
    qyo, l1c = env.rg(P, "qyo", EV), env.rg(P, "l1c", EV)
    h = t * qx - x
    l1 = (zc * qy).dbl() - y.dbl()
    i = h.sqr()
    l1sq = l1.sqr()
    ox = l1sq - (h.times(4) + x.times(8)) * i      # l1^2 - j - 2v
    oy = ((l1 * x).times(12) + (l1 * h).times(4) - (y * h).times(8)) * i - l1sq * l1   # (sums before the product with i: fewer products on the level)
    oz = (z * h).dbl()
    ot = oz.sqr()
    oc = ((zc * h) * i).times(8)                   # oz = 2 z h, so oz^3 = 8 z^3 h^3
    line = (oz * pyt) + (qyo * oz + l1c * l1).neg().shift(1)
    for n, vv in (("rx", ox), ("ry", oy), ("rz", oz), ("rt", ot), ("rc", oc)):
        env.put(P, n, vv, EV)
    env.put(P, "f", f * ln, ALL)
    env.put(P, "ln", line, ALL)

    P = env.prog("flush")       # the Miller value itself: the pending line multiplied in
    env.put(P, "f", env.st(P, "f", ALL) * env.st(P, "ln", ALL), ALL)

    # loop count < 0: upstream multiplies by the line through R and -R and inverts.  That line is (0, 4 l1c r.y) exactly (the mixed
    # addition of R and its own affine negation has h = 0, so o.z = 0 and l1 = -4 r.y: no inversion of r.z is needed to know it),
    # and the inversion of f is deferred: the final exponentiation of an inverse is the conjugate of the final exponentiation.
    P = env.prog("negfix")
    y = env.st(P, "ry", EV)
    f = env.st(P, "f", ALL)
    l1c = env.rg(P, "l1c", EV)
    line = (l1c * y).times(4).shift(1)
    env.put(P, "f", f * line, ALL)

    # ---------------- final exponentiation (one wave per product of Miller values) ----------------
    # state: acc (the product, then powers), plain registers: g (the next factor), v, vi, fst, fsti, nrm (Fq), ninv (Fq)
    for j in range(1, 1 << WIN_FQ, 2):      # the two tables the scripts select from: consecutive registers, entry (j - 1) / 2 = power j
        env.reg(f"ft{j}_0")
    for j in range(1, 1 << WIN_POW, 2):
        for c_ in range(k):
            env.reg(f"pb{j}_{c_}")
    P = env.prog("fe_mul")      # acc *= g
    env.put(P, "acc", env.st(P, "acc", ALL) * env.rg(P, "g", ALL), ALL)

    P = env.prog("fe_norm")     # the norm of acc down to Fq: acc^-1 = conj-like / norm
    v = env.st(P, "acc", ALL)
    A0, A1 = v.even(), v.odd_over_v()
    n_e = A0.sqr() - A1.sqr().shift(2)                 # in the twist field (even positions): A0^2 - u A1^2
    d = k // 2
    ne = [n_e.c[2 * j] for j in range(d)]              # coefficients over u
    if d == 2:
        t0, t1 = ne[0], P.lin([(-1, ne[1])])
        nrm = P.lin([(1, P.mul([(ne[0], ne[0])])), (-nr, P.mul([(ne[1], ne[1])]))])
        cof = [t0, t1]
    else:
        m = lambda a_, b_: P.mul([(a_, b_)])
        t0 = P.lin([(1, m(ne[0], ne[0])), (-nr, m(ne[1], ne[2]))])
        t1 = P.lin([(nr, m(ne[2], ne[2])), (-1, m(ne[0], ne[1]))])
        t2 = P.lin([(1, m(ne[1], ne[1])), (-1, m(ne[0], ne[2]))])
        nrm = P.lin([(1, m(ne[0], t0)), (nr, m(ne[2], t1)), (nr, m(ne[1], t2))])
        cof = [t0, t1, t2]
    P.out_reg("nrm0", nrm)
    for j in range(d):
        P.out_reg(f"cof{2 * j}", cof[j])                # inverse of n_e = cof / nrm
    env.put_reg(P, "v", v, ALL)

    # The one field inversion, of nrm0, is not a program: the script entry SCRIPT_INV has lane 0 run Fp::inv (divsteps, fp.hip.h:
    # ~45 products' worth against the ~940 dependent steps of a sliding window over p - 2) from register nrm0 into register ft1_0,
    # entry 0 of table 0, which "fq_init" then moves into the state slot pw0.
    P = env.prog("fq_init")
    P.out("pw0", P.tab(0, 0))

    P = env.prog("fe_easy")     # vi = conj_v(v) * (cof * ninv); first = v^(q^(k/2)) * vi ...
    v = env.rg(P, "v", ALL)
    ninv = P.state("pw0")
    ne_inv = X(P, [None] * k)
    for j in range(d):
        ne_inv.c[2 * j] = P.mul([(P.reg(f"cof{2 * j}"), ninv)])
    A0, A1 = v.even(), v.odd_over_v()
    vi = (A0 * ne_inv) + (A1 * ne_inv).neg().shift(1)
    if k == 4:
        first, first_inv = v.frob(2) * vi, vi.frob(2) * v
    else:
        aa, ai = v.frob(3) * vi, vi.frob(3) * v
        first, first_inv = aa.frob(1) * aa, ai.frob(1) * ai
    base = first_inv if env.w0_neg else first
    env.put_reg(P, "pb1_", base, ALL)                   # the base of the w0 power
    env.put_reg(P, "ff", first.frob(1), ALL)            # first^q

    P = env.prog("pow_tab")     # odd powers of the base for the sliding window over w0 (table 1)
    b1 = env.rg(P, "pb1_", ALL)
    b2 = b1.sqr()
    cur = b1
    for j in range(3, 1 << WIN_POW, 2):
        cur = cur * b2
        env.put_reg(P, f"pb{j}_", cur, ALL)
    P = env.prog("pow_sqr")
    env.put(P, "acc", env.st(P, "acc", ALL).sqr(), ALL)
    sel = lambda P: X(P, [P.tab(1, j) for j in range(k)])
    P = env.prog("pow_mul")
    env.put(P, "acc", env.st(P, "acc", ALL) * sel(P), ALL)
    P = env.prog("pow_init")
    env.put(P, "acc", sel(P), ALL)
    P = env.prog("fe_last")     # result = first^q * pow;  conjugated when the Miller values were left un-inverted (loop count < 0)
    res = env.rg(P, "ff", ALL) * env.st(P, "acc", ALL)
    if env.ate_neg:
        res = res.conj()
    env.put(P, "acc", res, ALL)

    # ---------------- scripts: the order in which the kernels run the programs (everything data-dependent in a pairing is constant)
    bits = bin(env.ate_loop)[3:]
    mil = ["setup"]
    for bch in bits:
        mil.append("dbl")
        if bch == "1":
            mil.append("add")
    mil.append("flush")
    if env.ate_neg:
        mil.append("negfix")
    env.scripts["miller"] = mil
    env.scripts["final_exp"] = (["fe_norm", ("inv", 0), ("sel", 0), "fq_init", "fe_easy", "pow_tab"]
                                + window_schedule(env.w0, WIN_POW, "pow_init", "pow_sqr", "pow_mul") + ["fe_last"])
    # which kernel needs which programs (each kernel stages only its own in LDS)
    env.standalone = {"final_exp": ["fe_mul"]}     # programs a kernel also runs outside its script
    env.sets = {"miller": ["setup", "dbl", "add", "flush", "negfix"],
                "final_exp": ["fe_mul"] + [n for n in env.progs if n.startswith(("fe_n", "fq_", "fe_e", "pow_", "fe_l"))]}
    # the table registers are consecutive: entry (j - 1) / 2 of table 0 is ft<j>_0, of table 1 is pb<j>_0 .. pb<j>_(k-1)
    return env


WIN_FQ, WIN_POW = 1, 3   # (table 0 is one register now: the inverse that SCRIPT_INV leaves there)


def window_schedule(e, WIN, init, sqr, mul):
    """left-to-right sliding window (odd digits below 2^WIN) for x^e: program names; ("sel", i) selects table entry i = (digit - 1) / 2
    for the `init` / `mul` that follows"""
    bits = bin(e)[2:]
    out, i, first = [], 0, True
    while i < len(bits):
        if bits[i] == "0":
            out.append(sqr)
            i += 1
            continue
        j = min(i + WIN, len(bits))
        while bits[j - 1] == "0":
            j -= 1
        val = int(bits[i:j], 2)
        if first:
            out += [("sel", (val - 1) // 2), init]
            first = False
        else:
            out += [sqr] * (j - i) + [("sel", (val - 1) // 2), mul]
        i = j
    return out


# ------------------------------------------------------------------------------------------------ scheduling / allocation
def compile_prog(env, P, nconst_regs):
    """-> dict(steps=[(kind, [instr])], written=[slots], ntemp) with instr = (dst, [(x, y)]) operands encoded as (space, index)"""
    nodes = P.nodes
    wanted = {}
    for slot, node in P.outputs:
        wanted.setdefault(node, []).append((SP_OUT, slot))
    for reg, node in P.reg_outputs:
        wanted.setdefault(node, []).append((SP_REG, reg))
    # liveness from the outputs
    live = set()
    stack = list(wanted)
    while stack:
        n = stack.pop()
        if n in live:
            continue
        live.add(n)
        kind, pay = nodes[n]
        if kind == "mul":
            for a, b in pay:
                stack += [a, b]
        elif kind == "lin":
            stack += [a for _, a in pay]
    # an output that is a leaf (or shared by two destinations) needs a copy: a LIN with one unit term
    extra = []
    for node, dests in list(wanted.items()):
        kind, _ = nodes[node]
        first = 0 if kind != "leaf" else None
        for di, dest in enumerate(dests):
            if first is not None and di == first:
                continue
            extra.append((dest, node))
        wanted[node] = [dests[first]] if first is not None else []
    order = [n for n in range(len(nodes)) if n in live]
    preds_of = {n: [x for t in nodes[n][1] for x in (t if nodes[n][0] == "mul" else (t[1],))] for n in order if nodes[n][0] != "leaf"}
    par_of = lambda n: 1 if nodes[n][0] == "mul" else 0

    def fit(lo, par):
        return lo if lo % 2 == par else lo + 1
    # as soon as possible (unbounded lanes): the length of the critical path
    asap = {}
    for n in order:
        asap[n] = -1 if nodes[n][0] == "leaf" else fit(max(asap[x] for x in preds_of[n]) + 1, par_of(n))
    depth = max([v for v in asap.values()] + [0])
    # as late as possible within that length: the slack of a node is alap - asap
    alap = {}
    for n in reversed(order):
        if nodes[n][0] == "leaf":
            continue
        hi = alap.get(n, depth)
        if hi % 2 != par_of(n):
            hi -= 1
        alap[n] = hi
        for x in preds_of[n]:
            if nodes[x][0] != "leaf":
                alap[x] = min(alap.get(x, depth), hi - 1)
    # list scheduling, a step = at most LANES instruction slots: when a level holds more (the first level of an MNT6 doubling: 66
    # products), the nodes with slack wait for a later step of their kind instead of the level running as two steps
    slot_of = {n: -1 for n in order if nodes[n][0] == "leaf"}
    todo = [n for n in order if nodes[n][0] != "leaf"]
    sl = 0
    while todo:
        par = sl % 2
        ready = sorted((n for n in todo if par_of(n) == par and all(slot_of.get(x, 1 << 30) < sl for x in preds_of[n])), key=lambda n: (alap[n], n))
        used = 0
        for n in ready:
            need = 2 if (par == 0 and len(nodes[n][1]) > 8) else 1
            if used + need > LANES:
                continue
            used += need
            slot_of[n] = sl
        todo = [n for n in todo if n not in slot_of]
        sl += 1
    copies = []   # (slot time, dest, src node)
    for dest, node in extra:
        lo = slot_of[node] + 1
        if lo % 2 != 0:
            lo += 1
        copies.append((lo, dest, node))
    last = max([slot_of[n] for n in order] + [c[0] for c in copies] + [0])
    # last use of every node (in slot time)
    last_use = {}
    for n in order:
        kind, pay = nodes[n]
        if kind == "leaf":
            continue
        preds = [x for t in pay for x in (t if kind == "mul" else (t[1],))]
        for x in preds:
            last_use[x] = max(last_use.get(x, -1), slot_of[n])
    for lo, dest, node in copies:
        last_use[node] = max(last_use.get(node, -1), lo)
    # registers: outputs go straight to their destination; everything else gets a temporary, freed after its last use
    loc = {}
    free, ntemp = [], 0
    by_slot = {}
    for n in order:
        if nodes[n][0] != "leaf":
            by_slot.setdefault(slot_of[n], []).append(n)
    for lo, dest, node in copies:
        by_slot.setdefault(lo, []).append(("copy", dest, node))
    for n in order:
        if nodes[n][0] == "leaf":
            loc[n] = nodes[n][1]
    expiring = {}
    steps = []
    tbase = None
    for s in range(last + 1):
        items = by_slot.get(s, [])
        # temporaries whose last reader ran in an EARLIER slot are free again (readers of slot s read before anyone writes in s)
        for n in expiring.pop(s - 1, []):
            free.append(loc[n][1][1])
        instrs = []
        for it in items:
            if isinstance(it, tuple):
                _, dest, node = it
                instrs.append((dest, [(1, loc[node])], K_LIN))
                continue
            n = it
            kind, pay = nodes[n]
            if wanted.get(n):
                loc[n] = wanted[n][0]
            else:
                if free:
                    r = free.pop()
                else:
                    r = ntemp
                    ntemp += 1
                loc[n] = (SP_REG, ("t", r))
                expiring.setdefault(last_use.get(n, s), []).append(n)
            if kind == "mul":
                instrs.append((loc[n], [(loc[a], loc[b]) for a, b in pay], K_MUL if any(a != b for a, b in pay) else K_SQR))
            else:
                instrs.append((loc[n], [(c, loc[a]) for c, a in pay], K_LIN))
        if not instrs:
            continue
        kinds = {i[2] for i in instrs}
        kind = K_LIN if kinds == {K_LIN} else K_SQR if kinds == {K_SQR} else K_MUL   # (a squaring among products runs as a product)
        assert K_LIN not in kinds or kinds == {K_LIN}
        instrs = [(d, t, kind) for d, t, _ in instrs]
        chunk, used = [], 0
        for ins in instrs:   # a LIN of more than 8 terms occupies two instruction slots (= lanes)
            need = 2 if (kind == K_LIN and len(ins[1]) > 8) else 1
            if used + need > LANES:
                steps.append((kind, chunk))
                chunk, used = [], 0
            chunk.append(ins)
            used += need
        steps.append((kind, chunk))
    written = sorted({d[1] for d, _, _ in sum((i for _, i in steps), []) if d[0] == SP_OUT})
    return dict(steps=steps, written=written, ntemp=ntemp)


def compile_env(env):
    out = {}
    for name, P in env.progs.items():
        out[name] = compile_prog(env, P, len(env.const_values))
    env.compiled = out
    env.ntemp = max(c["ntemp"] for c in out.values())
    return env


# register file layout: [2 * nslots state registers][constants][named registers][temporaries]
def reg_index(env, op):
    space, idx = op
    if space in (SP_IN, SP_OUT):
        return space, idx
    if space == SP_TAB:
        return space, (idx[0] << 8) | idx[1]
    tag, i = idx
    base = 2 * env.nslots
    if tag == "c":
        return SP_REG, base + i
    if tag == "r":
        return SP_REG, base + len(env.const_values) + i
    return SP_REG, base + len(env.const_values) + len(env.regs) + i


# ------------------------------------------------------------------------------------------------ evaluation with Python integers
class Machine:
    """what the device interpreter does, on integers mod p (no Montgomery form, no limbs): program logic, schedule, banking"""

    def __init__(self, env):
        self.env = env
        self.nregs = 2 * env.nslots + len(env.const_values) + len(env.regs) + env.ntemp
        self.r = [0] * self.nregs
        self.bank = 0
        base = 2 * env.nslots
        for i, v in enumerate(env.const_values):
            self.r[base + i] = v
        self.steps_run = 0
        self.sel = 0

    def tab_base(self, table):
        name, stride = (("ft1_0", 1), ("pb1_0", self.env.k))[table]
        return self.addr((SP_REG, ("r", self.env.regs[name]))), stride

    def addr(self, op, writing=False):
        space, idx = reg_index(self.env, op)
        if space == SP_REG:
            return idx
        if space == SP_TAB:
            base, stride = self.tab_base(idx >> 8)
            return base + self.sel * stride + (idx & 0xFF)
        cur = (self.bank >> idx) & 1
        return 2 * idx + (cur ^ (1 if space == SP_OUT else 0))

    def set_reg(self, name, value):
        self.r[self.addr((SP_REG, ("r", self.env.regs[name])))] = value % self.env.p

    def get_reg(self, name):
        return self.r[self.addr((SP_REG, ("r", self.env.regs[name])))]

    def set_state(self, name, value):
        self.r[self.addr((SP_IN, self.env.slots[name]))] = value % self.env.p

    def get_state(self, name):
        return self.r[self.addr((SP_IN, self.env.slots[name]))]

    def run_script(self, name):
        for e in self.env.scripts[name]:
            if isinstance(e, tuple) and e[0] == "inv":
                self.set_reg("ft1_0", pow(self.get_reg("nrm0"), self.env.p - 2, self.env.p))
            elif isinstance(e, tuple):
                self.sel = e[1]
            else:
                self.run(e)

    def run(self, name):
        p = self.env.p
        c = self.env.compiled[name]
        for kind, instrs in c["steps"]:
            vals = []
            for dst, terms, _ in instrs:       # every lane reads ...
                if kind != K_LIN:
                    assert kind == K_MUL or all(a == b for a, b in terms)
                    vals.append(sum(self.r[self.addr(a)] * self.r[self.addr(b)] for a, b in terms) % p)
                else:
                    vals.append(sum(cf * self.r[self.addr(a)] for cf, a in terms) % p)
            for (dst, _, _), v in zip(instrs, vals):   # ... before any lane writes
                self.r[self.addr(dst)] = v
            self.steps_run += 1
        for s in c["written"]:
            self.bank ^= 1 << s


def eval_miller(env, P1, Q2):
    """P1 = (x, y) in Fq, Q2 = (x coefficients, y coefficients) over u -> the k flat coefficients of the (un-inverted) Miller value"""
    M = Machine(env)
    M.set_reg("px0", P1[0]); M.set_reg("py0", P1[1]); M.set_reg("pz0", P1[2] if len(P1) > 2 else 1)
    for j in range(env.k // 2):
        M.set_reg(f"qx{2 * j}", Q2[0][j]); M.set_reg(f"qy{2 * j}", Q2[1][j])
    M.run_script("miller")
    return [M.get_state(f"f{j}") for j in range(env.k)], M.steps_run


def eval_final_exp(env, fs):
    """fs: list of flat Miller values -> flat GT element"""
    M = Machine(env)
    for j in range(env.k):
        M.set_state(f"acc{j}", fs[0][j])
    for g in fs[1:]:
        for j in range(env.k):
            M.set_reg(f"g{j}", g[j])
        M.run("fe_mul")
    M.run_script("final_exp")
    return [M.get_state(f"acc{j}") for j in range(env.k)], M.steps_run


# ------------------------------------------------------------------------------------------------ header
def limbs28(x, n):
    return [(x >> (B28 * i)) & ((1 << B28) - 1) for i in range(n)]


ZERO_BIT = 31   # a bit of the bank word that is never set (fewer than 32 state slots): what plain registers "select" with


def enc_operand(env, op):
    """16 bits: A (8, base register) | B (6, a bit of {bank, ~bank}: added to A) | f (2: add sel * stride of table 0 / table 1).
    register = A + bit B + sel * stride -- four instructions on the device where a decode by operand space was twenty"""
    space, idx = reg_index(env, op)
    named = 2 * env.nslots + len(env.const_values)
    if space == SP_REG:
        A, B, f = idx, ZERO_BIT, 0
    elif space == SP_IN:
        A, B, f = 2 * idx, idx, 0
    elif space == SP_OUT:
        A, B, f = 2 * idx, 32 + idx, 0
    else:
        table, off = idx >> 8, idx & 0xFF
        A, B, f = named + env.regs[("ft1_0", "pb1_0")[table]] + off, ZERO_BIT, 1 << table
    assert A < 256 and env.nslots <= ZERO_BIT
    return A | (B << 8) | (f << 14)


def emit(envs):
    L = ["// GENERATED by tools/gen_pairing_vm.py -- do not edit.  Programs of the wave-wide field VM (pairing_vm.hip.h).",
         "#pragma once", "#include <stdint.h>", "", "namespace pcd { namespace vmgen {", "",
         f"constexpr int VM_TMAX = {TMAX}, VM_LIN_TERMS = {LIN_TERMS}, VM_LIN_WEIGHT = {LIN_WEIGHT};",
         "// instruction = 12 words: w0 = kind | terms << 8 | dst << 16;  MUL / SQR: w[1 + t] = a_t | b_t << 16;  LIN: w[1 + t / 2] holds operand t in its",
         "// low / high half, w[5 + t / 2] the signed 16-bit coefficient; a LIN of more than 8 terms continues in the next slot (w0 = 0xFF, whose lane",
         "// idles).  Operand (16 bits) = A | B << 8 | f << 14: register A + bit B of {bank, ~bank} + sel * (f & 1 ? TAB0_STRIDE : f & 2 ? TAB1_STRIDE : 0) --",
         "// plain register r: A = r, B = 31 (never set); state slot j: A = 2 j, B = j (current bank) or 32 + j (the other bank); table t entry: A = TABt_BASE + offset.",
         "// kind: 0 LIN, 1 MUL, 2 SQR (a step of squarings only).  Scripts: the program ids a kernel runs, in order.", ""]
    for env in envs:
        N = env.N
        Rp = 1 << (B28 * N)
        nm = env.name
        assert env.nslots <= 32
        nregs = 2 * env.nslots + len(env.const_values) + len(env.regs) + env.ntemp
        named = 2 * env.nslots + len(env.const_values)
        L.append(f"// ---- {nm}: {env.nslots} state slots, {len(env.const_values)} constants, {len(env.regs)} named registers, {env.ntemp} temporaries")
        L.append(f"struct {nm} {{")
        L.append(f"  static constexpr int NSLOTS = {env.nslots}, NCONST = {len(env.const_values)}, NNAMED = {len(env.regs)}, NTEMP = {env.ntemp}, NREGS = {nregs};")
        L.append(f"  static constexpr int CONST_BASE = {2 * env.nslots}, NAMED_BASE = {named};")
        L.append(f"  // operand space 3: register = TAB_BASE[table] + sel * TAB_STRIDE[table] + offset (the script sets sel)")
        L.append(f"  static constexpr int TAB0_BASE = {named + env.regs['ft1_0']}, TAB0_STRIDE = 1, TAB1_BASE = {named + env.regs['pb1_0']}, TAB1_STRIDE = {env.k};")
        for sname, si in env.slots.items():
            L.append(f"  static constexpr int S_{sname.upper()} = {si};")
        for rname, ri in env.regs.items():
            L.append(f"  static constexpr int R_{rname.upper()} = {named + ri};")
        L.append("};")
        for setname, names in env.sets.items():
            words, steps, progs = [], [], []
            for pname in names:
                c = env.compiled[pname]
                first = len(steps)
                for kind, instrs in c["steps"]:
                    first_word = len(words)
                    for dst, terms, _ in instrs:
                        w = [0] * 12
                        w[0] = kind | (len(terms) << 8) | (enc_operand(env, dst) << 16)
                        if kind != K_LIN:
                            for t, (a, b) in enumerate(terms):
                                w[1 + t] = enc_operand(env, a) | (enc_operand(env, b) << 16)
                            words += w
                            continue
                        w2 = [0] * 12
                        w2[0] = 0xFF   # continuation slot: its lane idles, the lane before it reads terms 8 .. 15 here
                        for t, (cf, a) in enumerate(terms):
                            assert -32768 <= cf < 32768
                            ww, tt = (w, t) if t < 8 else (w2, t - 8)
                            ww[1 + tt // 2] |= enc_operand(env, a) << (16 * (tt % 2))
                            ww[5 + tt // 2] |= (cf & 0xFFFF) << (16 * (tt % 2))
                        words += w
                        if len(terms) > 8:
                            words += w2
                    nslots = (len(words) - first_word) // 12
                    assert nslots <= LANES
                    # (LIN steps carry their largest term count in bits 8 up: the kernel runs the terms wave-uniformly)
                    tmax = max(len(terms) for _, terms, _ in instrs) if kind == K_LIN else 0
                    steps.append((kind | (tmax << 8), first_word // 12, nslots))
                mask = 0
                for sl in c["written"]:
                    mask |= 1 << sl
                progs.append((pname, first, len(steps) - first, mask))
            # the kernel's whole run as ONE list of step records {kind | tmax << 8 | flags << 16, first slot, slots, bank flip}: the
            # script with its programs expanded.  flags: 1 = last step of a program (flip the banks of the slots it wrote, word 3),
            # 2 = select table entry (flags >> 4) before the step, 4 = the field inversion (SCRIPT_INV) before the step.  The device
            # interpreter walks this list with scalar loads, prefetching across program boundaries; programs a kernel runs on their
            # own (fe_mul) follow the script's records.
            by_name = {pname: (first, cnt, mask) for pname, first, cnt, mask in progs}

            def records(pname, pending=0):
                first, cnt, mask = by_name[pname]
                out = []
                for q in range(cnt):
                    k_, o, n_ = steps[first + q]
                    fl = (pending if q == 0 else 0) | (1 if q == cnt - 1 else 0)
                    out.append((k_ | (fl << 16), o, n_, mask if q == cnt - 1 else 0))
                return out
            flat, pending = [], 0
            for e in env.scripts[setname]:
                if isinstance(e, tuple) and e[0] == "inv":
                    pending |= 4
                elif isinstance(e, tuple):
                    assert e[1] < 16
                    pending = (pending & ~0xF2) | 2 | (e[1] << 4)
                else:
                    flat += records(e, pending)
                    pending = 0
            assert pending == 0
            nflat_script = len(flat)
            alone = {}
            for pname in env.standalone.get(setname, []):
                alone[pname] = (len(flat), by_name[pname][1])
                flat += records(pname)
            pid = {pname: i for i, (pname, _, _, _) in enumerate(progs)}
            assert len(progs) < SCRIPT_INV
            ids = [SCRIPT_INV if isinstance(e, tuple) and e[0] == "inv" else (SET_SEL + e[1]) if isinstance(e, tuple) else pid[e]
                   for e in env.scripts[setname] if isinstance(e, tuple) or e in pid]
            assert len(ids) == len(env.scripts[setname])
            packed = [sum(ids[q + j] << (8 * j) for j in range(4) if q + j < len(ids)) for q in range(0, len(ids), 4)]
            T = f"{nm}_{setname}"
            L.append(f"// {T}: " + ", ".join(f"{i} {pname}" for i, (pname, _, _, _) in enumerate(progs)))
            for i, (pname, _, _, _) in enumerate(progs):
                L.append(f"static const int {T}_P_{pname.upper()} = {i};")
            L.append(f"static const uint32_t {T}_script_len = {len(ids)};")
            L.append(f"static const uint32_t {T}_script[{len(packed)}] = {{  // program ids in running order, four per word; 0xF0 + i: select table entry i")
            for q in range(0, len(packed), 12):
                L.append("  " + ", ".join(f"0x{w:08x}u" for w in packed[q:q + 12]) + ",")
            L.append("};")
            L.append(f"static const uint32_t {T}_progs[{len(progs)}][3] = {{  // first step, steps, mask of the state slots written")
            L.append("  " + ", ".join(f"{{{first}, {cnt}, 0x{mask:08x}u}}" for _, first, cnt, mask in progs) + "};")
            L.append(f"static const uint32_t {T}_steps[{len(steps)}][3] = {{  // kind | largest LIN term count << 8, first instruction slot, slots")
            for q in range(0, len(steps), 8):
                L.append("  " + ", ".join(f"{{0x{k_:x}, {o}, {n_}}}" for k_, o, n_ in steps[q:q + 8]) + ",")
            L.append("};")
            L.append(f"static const uint32_t {T}_flat_script_len = {nflat_script};")
            for pname, (o, c_) in alone.items():
                L.append(f"static const uint32_t {T}_flat_{pname.upper()}_off = {o}, {T}_flat_{pname.upper()}_len = {c_};")
            L.append(f"static const uint32_t {T}_flat[{len(flat)}][4] = {{  // the run as step records: kind | tmax << 8 | flags << 16, first slot, slots, bank flip")
            for q in range(0, len(flat), 6):
                L.append("  " + ", ".join(f"{{0x{a:x}, {b}, {c_}, 0x{d:x}}}" for a, b, c_, d in flat[q:q + 6]) + ",")
            L.append("};")
            L.append(f"static const uint32_t {T}_code[{len(words)}] = {{")
            for q in range(0, len(words), 12):
                L.append("  " + ", ".join(f"0x{w:08x}u" for w in words[q:q + 12]) + ",")
            L.append("};")
        L.append(f"static const uint32_t {nm}_consts[{len(env.const_values)}][{N}] = {{  // Montgomery form (R' = 2^{B28 * N}), 28-bit limbs")
        for v in env.const_values:
            L.append("  {" + ", ".join(f"0x{x:07x}u" for x in limbs28(v * Rp % env.p, N)) + "},")
        L.append("};")
        L.append("")
    L.append("} }  // namespace pcd::vmgen")
    return "\n".join(L) + "\n"


def main():
    envs = [compile_env(build(c)) for c in range(4)]
    for env in envs:
        # the kernels address these families as base + j
        for j in range(3, 1 << WIN_FQ, 2):
            assert env.regs[f"ft{j}_0"] == env.regs["ft1_0"] + (j - 1) // 2
        for j in range(1, 1 << WIN_POW, 2):
            assert all(env.regs[f"pb{j}_{c}"] == env.regs["pb1_0"] + (j - 1) // 2 * env.k + c for c in range(env.k))
        for fam, step in (("qx", 2), ("qy", 2), ("g", 1), ("v", 1)):
            assert all(env.regs[f"{fam}{step * j}"] == env.regs[f"{fam}0"] + j for j in range(env.k // step)), fam
        for fam in ("f", "acc"):
            assert all(env.slots[f"{fam}{j}"] == env.slots[f"{fam}0"] + j for j in range(env.k)), fam
        tot = {n: (len(c["steps"]), sum(len(i) for _, i in c["steps"])) for n, c in env.compiled.items()}
        print(env.name, "slots", env.nslots, "consts", len(env.const_values), "named", len(env.regs), "temps", env.ntemp, tot)
    path = os.path.join(ROOT, "pcd_amd", "csrc", "pairing_vm_gen.h")
    with open(path, "w") as fh:
        fh.write(emit(envs))
    print("wrote", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
