"""Developer probe: check the rocgdb trace of tools/debug/k2_trace.gdb numerically (Python integers).
Every product / square of the failing Fq3-753 mailbox addition is recomputed from the operands found in LDS at entry and compared with
what sits in slot 0 at return; then the operands of every call are compared with what add-2007-bl says they should be."""
import re, sys
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__)))))
from oracle import pyoracle as O

p = O.FIELDS[3].p          # F753B = MNT6-753 Fq
NR = 11
N = 27
Rp = 1 << (28 * N)
Rinv = pow(Rp, -1, p)

def parse(path):
    ev = []
    cur = None
    for line in open(path):
        m = re.match(r"@@ (ENTER|LEAVE) (mul|sqr) ret=(0x[0-9a-f]+)", line)
        if m:
            cur = {"kind": m.group(1), "op": m.group(2), "ret": int(m.group(3), 16), "mem": {}}
            ev.append(cur)
            continue
        m = re.match(r"local#(0x[0-9a-f]+):\s+(.*)", line)
        if m and cur is not None:
            a = int(m.group(1), 16)
            for j, w in enumerate(m.group(2).split()):
                cur["mem"][a + 4 * j] = int(w, 16)
    return ev

def elem(mem, slot, lane):
    v = 0
    limbs = []
    for k in range(7):
        base = ((slot * 7 + k) * 64 + lane) * 16
        for c in range(4):
            i = 4 * k + c
            if i < N:
                limbs.append(mem[base + 4 * c])
    val = sum(l << (28 * i) for i, l in enumerate(limbs))
    return val, limbs

def f3(mem, slot):
    return [elem(mem, slot, l)[0] for l in range(3)]

def mont_mul3(a, b):  # Montgomery product in Fq3 = Fq[u]/(u^3 - NR), residues mod p
    c0 = a[0] * b[0] + NR * (a[1] * b[2] + a[2] * b[1])
    c1 = a[0] * b[1] + a[1] * b[0] + NR * a[2] * b[2]
    c2 = a[0] * b[2] + a[1] * b[1] + a[2] * b[0]
    return [c * Rinv % p for c in (c0, c1, c2)]

def eq(a, b):
    return all((x - y) % p == 0 for x, y in zip(a, b))
def sub(a, b): return [(x - y) % p for x, y in zip(a, b)]
def add(a, b): return [(x + y) % p for x, y in zip(a, b)]
def dbl(a): return add(a, a)

ev = parse(sys.argv[1])
calls = []
for i in range(0, len(ev), 2):
    e, l = ev[i], ev[i + 1]
    assert e["kind"] == "ENTER" and l["kind"] == "LEAVE" and e["ret"] == l["ret"]
    a = f3(e["mem"], 0)
    b = f3(e["mem"], 1) if e["op"] == "mul" else a
    r = f3(l["mem"], 0)
    ok = eq(r, mont_mul3(a, b))
    bounds = all(x < 2 * p for x in a + b + r)
    calls.append((e["op"], a, b, r))
    print(f"call {i // 2:2d} {e['op']} ret={e['ret'] & 0xfffff:#x} product correct: {ok}  operands/result < 2p: {bounds}")

# add-2007-bl: which call is which
# 0: Z1Z1 = Z1^2   1: Z2Z2 = Z2^2   2,3: U1 = X1 Z2Z2, U2 = X2 Z1Z1   4..7: S1 = Y1 Z2 Z2Z2, S2 = Y2 Z1 Z1Z1
names = {}
Z1 = calls[0][1]; Z2 = calls[1][1]; Z1Z1 = calls[0][3]; Z2Z2 = calls[1][3]
def which(x, cands):
    for n, c in cands.items():
        if eq(x, c): return n
    return "?"
known = {"Z1": Z1, "Z2": Z2, "Z1Z1": Z1Z1, "Z2Z2": Z2Z2}
for i, (op, a, b, r) in enumerate(calls):
    na, nb = which(a, known), which(b, known)
    print(f"call {i:2d} {op}: a = {na}, b = {nb}")
    known[f"r{i}"] = r
    # derived candidates that the formula uses as operands
    if i == 7:
        # identify U1, U2, S1, S2 by operand names
        pass
# explicit reconstruction
U = {}
for i in (2, 3):
    op, a, b, r = calls[i]
    names_i = (which(a, {"Z1Z1": Z1Z1, "Z2Z2": Z2Z2}), which(b, {"Z1Z1": Z1Z1, "Z2Z2": Z2Z2}))
    if "Z2Z2" in names_i: U["U1"] = r; X1 = b if names_i[0] == "Z2Z2" else a
    else: U["U2"] = r; X2 = b if names_i[0] == "Z1Z1" else a
print("found", list(U))
# S1 = (Y1 * Z2) * Z2Z2 or Y1 * (Z2 * Z2Z2) ... find the two results that were multiplied last with Z2Z2 / Z1Z1 chains: take calls 4..7
S = {}
for i in range(4, 8):
    op, a, b, r = calls[i]
    print("  call", i, "operands:", which(a, known), which(b, known))
H_candidates = {}
if "U1" in U and "U2" in U:
    H = sub(U["U2"], U["U1"])
    known["H"] = H; known["2H"] = dbl(H); known["U1"] = U["U1"]; known["U2"] = U["U2"]
    op, a, b, r = calls[8]
    print("call 8 operand is 2H:", eq(a, dbl(H)))
    I = r
    known["I"] = I
    for i in range(9, 16):
        op, a, b, r = calls[i]
        print(f"  call {i} {op}: a = {which(a, known)}, b = {which(b, known)}")
        known[f"r{i}"] = r

# ---- the rest of the formula
S1, S2 = calls[5][3], calls[7][3]
r_ = dbl(sub(S2, S1))
print("call 11 operand is r = 2 (S2 - S1):", eq(calls[11][1], r_))
J, V = calls[9][3], calls[10][3]
X3 = sub(sub(calls[11][3], J), dbl(V))
a12, b12 = calls[12][1], calls[12][2]
print("call 12 operands are r and V - X3:", (eq(a12, r_) and eq(b12, sub(V, X3))) or (eq(b12, r_) and eq(a12, sub(V, X3))),
      "| a is r:", eq(a12, r_), "b is r:", eq(b12, r_), "a is V-X3:", eq(a12, sub(V, X3)), "b is V-X3:", eq(b12, sub(V, X3)))
print("call 14 operand is Z1 + Z2:", eq(calls[14][1], add(Z1, Z2)))
t = sub(sub(calls[14][3], Z1Z1), Z2Z2)
print("call 15 operands are (..) and H:", eq(calls[15][1], t), eq(calls[15][2], H))
Y3 = sub(calls[12][3], dbl(calls[13][3]))
Y3_true = sub(mont_mul3(r_, sub(V, X3)), dbl(mont_mul3(S1, J)))
print("Y3 from the traced products equals the formula's value:", eq(Y3, Y3_true))
# what the kernel stored (k2 prints Y.c0 of both forms)
for line in open(sys.argv[1]):
    m = re.match(r"\s*Y\.c0 (plain|mailbox)\s*:\s*(.*)", line)
    if m:
        limbs = [int(w, 16) for w in m.group(2).split()]
        if len(limbs) == N:
            val = sum(l << (28 * i) for i, l in enumerate(limbs))
            print(f"stored Y.c0 ({m.group(1)}): equals Y3.c0 mod p: {(val - Y3[0]) % p == 0}   limbs < 2^28: {all(l < (1 << 28) for l in limbs[:-1])}  value < 2p: {val < 2 * p}")
            if (val - Y3[0]) % p:
                # which single limb would have to change?
                for i in range(N):
                    d = (Y3[0] - (val - (limbs[i] << (28 * i)))) % p
                    for kmul in range(3):
                        cand = d + 0  # want limb value x with (rest + x 2^(28 i)) = Y3 + k p
                        pass
        break_after = False
