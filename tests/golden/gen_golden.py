#!/usr/bin/env python3
"""Generate the committed golden vectors under tests/golden/ with the pure-Python big-integer oracle
(oracle/pyoracle.py).  The reference holds no vectors for this path (its tests are
prove->verify round trips) and cannot be run here (Rust, no toolchain), so these vectors pin the
*mathematical* values: textbook affine group law, naive DFT, polynomial division, textbook ate pairing.

    python tests/golden/gen_golden.py          # rewrites tests/golden/*.npz  (a few minutes)
"""
import os
import random
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import pyoracle as O  # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))
SEED = 0x5043443031  # "PCD01"


def gen_fields():
    rnd = random.Random(SEED)
    out = {}
    for fid, f in enumerate(O.FIELDS):
        a = [0, 1, f.p - 1, 2, f.p - 2] + [rnd.randrange(f.p) for _ in range(11)]
        b = [f.p - 1, f.p - 1, f.p - 1, 0, 1] + [rnd.randrange(f.p) for _ in range(11)]
        out[f"f{fid}_a"] = O.pack_fp(f, a)
        out[f"f{fid}_b"] = O.pack_fp(f, b)
        out[f"f{fid}_add"] = O.pack_fp(f, [(x + y) % f.p for x, y in zip(a, b)])
        out[f"f{fid}_sub"] = O.pack_fp(f, [(x - y) % f.p for x, y in zip(a, b)])
        out[f"f{fid}_mul"] = O.pack_fp(f, [x * y % f.p for x, y in zip(a, b)])
        out[f"f{fid}_inv_b"] = O.pack_fp(f, [pow(y, -1, f.p) if y else 0 for y in b])
        out[f"f{fid}_a_canonical"] = O.pack_fp(f, a, mont=False)
    np.savez_compressed(os.path.join(OUT, "fields.npz"), **out)


def gen_msm():
    rnd = random.Random(SEED + 1)
    out = {}
    for cid, c in enumerate(O.CURVES):
        r = c.fr.p
        for g in (1, 2):
            F, a = c.group(g)
            gen = c.g1 if g == 1 else c.g2
            big = cid >= 2
            n = (9 if g == 2 else 17) if big else 33
            pts = [O.ec_mul(F, a, rnd.randrange(1, 1 << 40), gen) for _ in range(n)]
            pts[3] = pts[2]            # duplicate base (forces the doubling branch of the mixed add)
            pts[5] = None              # point at infinity (flag)
            edge = [0, 1, r - 1, (1 << 13) - 1, 1 << 13, (1 << 15) + 1, (1 << 16) - 1, 1 << 17, 2]
            sc = (edge + [rnd.randrange(r) for _ in range(n)])[:n]
            sc[2], sc[3] = 7, 7        # same bucket, same point: acc == base when the second one arrives
            want = O.msm_naive(F, a, pts, sc)
            xy, inf = O.pack_points(c, g, pts)
            wxy, winf = O.pack_points(c, g, [want])
            out[f"c{cid}_g{g}_bases"] = xy
            out[f"c{cid}_g{g}_inf"] = inf
            out[f"c{cid}_g{g}_scalars"] = O.pack_fp(c.fr, sc, mont=False)
            out[f"c{cid}_g{g}_result_xy"] = wxy[0]
            out[f"c{cid}_g{g}_result_inf"] = winf
            # all-zero and all-one scalar vectors
            ones = O.msm_naive(F, a, pts, [1] * n)
            oxy, oinf = O.pack_points(c, g, [ones])
            out[f"c{cid}_g{g}_ones_xy"] = oxy[0]
            out[f"c{cid}_g{g}_ones_inf"] = oinf
            print("msm", c.name, g, flush=True)
    np.savez_compressed(os.path.join(OUT, "msm.npz"), **out)


def gen_fft():
    rnd = random.Random(SEED + 2)
    out = {}
    for fid, f in enumerate(O.FIELDS):
        xs16 = [rnd.randrange(f.p) for _ in range(16)]
        assert O.fft(f, xs16) == O.dft_naive(f, xs16)
        assert O.fft(f, xs16, inverse=True) == O.dft_naive(f, xs16, inverse=True)
        for log_n in (0, 1, 4, 8, 11):
            xs = [rnd.randrange(f.p) for _ in range(1 << log_n)]
            out[f"f{fid}_n{log_n}_in"] = O.pack_fp(f, xs)
            for inv in (0, 1):
                for coset in (0, 1):
                    out[f"f{fid}_n{log_n}_i{inv}c{coset}"] = O.pack_fp(f, O.fft(f, xs, inverse=bool(inv), coset=bool(coset)))
        print("fft", f.name, flush=True)
    np.savez_compressed(os.path.join(OUT, "fft.npz"), **out)


def csr_arrays(fld, rows):
    rp = np.zeros(len(rows) + 1, dtype=np.uint64)
    cols, coeffs = [], []
    for j, row in enumerate(rows):
        for cf, col in row:
            cols.append(col)
            coeffs.append(cf)
        rp[j + 1] = len(cols)
    return rp, np.array(cols, dtype=np.uint32), O.pack_fp(fld, coeffs)


def gen_groth16():
    rnd = random.Random(SEED + 3)
    out = {}
    for cid in (0, 1, 2, 3):
        c = O.CURVES[cid]
        fr = c.fr
        nc = 13 if cid < 2 else 5
        r = O.synthetic_r1cs(fr, nc, 2, seed=SEED + cid)
        assert r.is_satisfied()
        h = O.witness_map_naive(r)
        pre = f"c{cid}_"
        for nm, rows in (("a", r.A), ("b", r.B), ("c", r.C)):
            rp, col, coeff = csr_arrays(fr, rows)
            out[pre + f"rp_{nm}"], out[pre + f"col_{nm}"], out[pre + f"coeff_{nm}"] = rp, col, coeff
        out[pre + "z"] = O.pack_fp(fr, r.z)
        out[pre + "num_inputs"] = np.array([r.num_inputs], dtype=np.uint64)
        out[pre + "h"] = O.pack_fp(fr, h)
        if cid >= 2:
            continue  # 753-bit: witness map only (pure-Python setup over Fq3 G2 is too slow to regenerate)
        tox = [rnd.randrange(1, fr.p) for _ in range(5)]
        pk = O.groth16_setup(c, r, tox)
        rr, ss = rnd.randrange(fr.p), rnd.randrange(fr.p)
        proof = O.groth16_prove(c, pk, r, rr, ss)
        assert O.groth16_verify(c, pk, r.z[1:r.num_inputs], proof)
        assert not O.groth16_verify(c, pk, [(r.z[1] + 1) % fr.p], proof)
        out[pre + "toxic"] = O.pack_fp(fr, tox)
        out[pre + "r"] = O.pack_fp(fr, [rr])[0]
        out[pre + "s"] = O.pack_fp(fr, [ss])[0]
        for nm, g in (("alpha_g1", 1), ("beta_g1", 1), ("delta_g1", 1), ("beta_g2", 2), ("delta_g2", 2), ("gamma_g2", 2)):
            out[pre + nm] = O.pack_points(c, g, [pk[nm]])[0][0]
        for nm, g in (("a_query", 1), ("b_g1_query", 1), ("b_g2_query", 2), ("h_query", 1), ("l_query", 1), ("gamma_abc_g1", 1)):
            xy, inf = O.pack_points(c, g, pk[nm])
            out[pre + nm], out[pre + nm + "_inf"] = xy, inf
        pa, _ = O.pack_points(c, 1, [proof[0]])
        pb, _ = O.pack_points(c, 2, [proof[1]])
        pc, _ = O.pack_points(c, 1, [proof[2]])
        out[pre + "proof"] = np.concatenate([pa[0], pb[0], pc[0]])
        print("groth16", c.name, flush=True)
    np.savez_compressed(os.path.join(OUT, "groth16.npz"), **out)


def gen_pairing():
    rnd = random.Random(SEED + 4)
    out = {}
    for cid, c in enumerate(O.CURVES):
        F1, a1 = c.group(1)
        F2, a2 = c.group(2)
        P = O.ec_mul(F1, a1, rnd.randrange(1, c.fr.p), c.g1)
        Q = O.ec_mul(F2, a2, rnd.randrange(1, c.fr.p), c.g2)
        e = O.pairing(c, P, Q)
        assert e != c.Fk.one() and c.Fk.pow(e, c.fr.p) == c.Fk.one()
        tower = O.fk_to_tower(c, e)
        out[f"c{cid}_p"] = O.pack_points(c, 1, [P])[0][0]
        out[f"c{cid}_q"] = O.pack_points(c, 2, [Q])[0][0]
        out[f"c{cid}_gt"] = O.pack_fp(c.fq, tower[0] + tower[1]).reshape(-1)
        print("pairing", c.name, flush=True)
    np.savez_compressed(os.path.join(OUT, "pairing.npz"), **out)


def gen_wire():
    """CanonicalSerialize images (oracle/pyoracle.py serialize_*): per curve and group six points -- both signs of y, the point at
    infinity -- compressed and uncompressed, plus one proof-shaped triple and one verifying-key-shaped tuple."""
    rnd = random.Random(SEED + 5)
    out = {}
    for cid, c in enumerate(O.CURVES):
        pts = {}
        for g in (1, 2):
            F, a = c.group(g)
            gen = c.g1 if g == 1 else c.g2
            ps = [O.ec_mul(F, a, rnd.randrange(1, 1 << 64), gen) for _ in range(3)]
            ps = [ps[0], O.ec_neg(F, ps[0]), None, ps[1], ps[2], O.ec_neg(F, ps[2])]
            pts[g] = ps
            xy, inf = O.pack_points(c, g, ps)
            out[f"c{cid}_g{g}_xy"] = xy
            out[f"c{cid}_g{g}_inf"] = inf
            for comp in (0, 1):
                out[f"c{cid}_g{g}_ser{comp}"] = np.frombuffer(b"".join(O.serialize_point(c, g, P, bool(comp)) for P in ps), dtype=np.uint8)
        proof = (pts[1][0], pts[2][3], pts[1][4])
        vk = (pts[1][3], pts[2][0], pts[2][1], pts[2][4], [pts[1][0], pts[1][1], pts[1][5]])
        for comp in (0, 1):
            out[f"c{cid}_proof_ser{comp}"] = np.frombuffer(O.serialize_proof(c, proof, bool(comp)), dtype=np.uint8)
            out[f"c{cid}_vk_ser{comp}"] = np.frombuffer(O.serialize_vk(c, *vk, compressed=bool(comp)), dtype=np.uint8)
    np.savez_compressed(os.path.join(OUT, "wire.npz"), **out)


if __name__ == "__main__":
    which = sys.argv[1:] or ["fields", "msm", "fft", "groth16", "pairing", "wire"]
    for w in which:
        globals()["gen_" + w]()
    print("done")
