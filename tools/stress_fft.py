"""Randomised differential soak of the transforms against the CPU oracle (developer tool, not a test): random field, random radix-2 size
(1 .. 2^16, or up to the field's 2-adicity when that is smaller) or mixed-radix size 2^a q^b (q = 7 on the MNT4-298 base-side field,
5 on the MNT4-753 one), forward / inverse, plain / coset, plus the round trip; and random witness maps (banded and skewed matrices, sizes
that push the help fields onto mixed-radix domains).   python tools/stress_fft.py [seconds = 200] [seed = 1]"""
import os, sys, time, random
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from oracle import coracle as co
from pcd_amd import capi

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 200.0
rnd = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
ctx = capi.Context(0)
ADICITY = {0: 17, 1: 34, 2: 15, 3: 30}   # two-adicity of the four fields (F298A, F298B, F753A, F753B)
MIXED = {0: 7, 2: 5}
t0 = time.time()
n_fft = n_mixed = n_wm = 0
while time.time() - t0 < budget:
    kind = rnd.random()
    fid = rnd.randrange(4)
    inv, coset = rnd.random() < 0.5, rnd.random() < 0.5
    if kind < 0.55:
        logn = rnd.randrange(1, min(ADICITY[fid], 16 if fid < 2 else 13) + 1)
        x = co.gen_field(fid, 1 << logn, seed=rnd.randrange(1 << 30))
        want = co.fft(fid, x, inverse=inv, coset=coset, nthreads=8)
        got = ctx.fft(fid, x, inverse=inv, coset=coset)
        back = ctx.fft(fid, got, inverse=not inv, coset=coset)
        if not (np.array_equal(got, want) and np.array_equal(back, x)):
            print("FFT MISMATCH", dict(fid=fid, logn=logn, inv=inv, coset=coset), flush=True); sys.exit(1)
        n_fft += 1
    elif kind < 0.8:
        fid = rnd.choice((0, 2))
        q = MIXED[fid]
        m = q if rnd.random() < 0.6 else q * q
        a = rnd.randrange(1, min(ADICITY[fid], 12 if fid == 0 else 9) + 1)
        n = m << a
        x = co.gen_field(fid, n, seed=rnd.randrange(1 << 30))
        want = co.fft_general(fid, x, m, inverse=inv, coset=coset, nthreads=8)
        got = ctx.fft_general(fid, x, inverse=inv, coset=coset)
        if not np.array_equal(got, want):
            print("MIXED-RADIX MISMATCH", dict(fid=fid, m=m, a=a, inv=inv, coset=coset), flush=True); sys.exit(1)
        n_mixed += 1
    else:
        cid = rnd.randrange(4)
        fr = co.CURVE_FR[cid]
        nc = rnd.randrange(10, 3000 if cid < 2 else 600)
        if rnd.random() < 0.15:   # beyond the help field's 2-adicity: the mixed-radix domain
            nc = {0: rnd.randrange(10, 3000), 1: (1 << 17) + rnd.randrange(1, 2000), 2: rnd.randrange(10, 600), 3: (1 << 15) + rnd.randrange(1, 500)}[cid]
        u = rnd.random()
        make = co.synthetic_r1cs if nc < 50 or u < 0.35 else co.skewed_r1cs if u < 0.65 else co.witness_r1cs
        r = make(fr, nc, rnd.randrange(2, 5), seed=rnd.randrange(1 << 30))
        want = co.witness_map(r, nthreads=16)
        got = ctx.witness_map(fr, r)
        if not (got.shape == want.shape and np.array_equal(got, want)):
            print("WITNESS MAP MISMATCH", dict(cid=cid, nc=nc, make=make.__name__), flush=True); sys.exit(1)
        n_wm += 1
print(f"stress ok: {n_fft} radix-2 transforms (with round trips), {n_mixed} mixed-radix transforms, {n_wm} witness maps in {time.time() - t0:.0f} s", flush=True)
