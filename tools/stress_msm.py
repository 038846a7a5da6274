"""Randomised differential soak of the MSM paths against the CPU oracle (developer tool, not a test): random group, size, scalar
distribution, precompute mode, sort strategy, accumulation form (running sums / pair tree with random chunk and batch floor), with
duplicate points, P next to -P, flagged infinities and runs of equal scalars thrown in.  Stops at the first mismatch and prints the
case.   python tools/stress_msm.py [seconds = 300] [seed = 1]"""
import os, sys, time, random
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from oracle import coracle as co
from pcd_amd import capi

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 300.0
rnd = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
ctx = capi.Context(0)
t0 = time.time()
cases = 0
per_group = {}
while time.time() - t0 < budget:
    cid, grp = rnd.randrange(4), rnd.choice((1, 1, 2))
    big = cid >= 2
    nmax = (1500 if grp == 1 else 300) if big else (6000 if grp == 1 else 1500)
    n = rnd.choice((1, 2, 3, 31, 32, 33, 63, 64, 65)) if rnd.random() < 0.15 else rnd.randrange(1, nmax)
    fr = co.CURVE_FR[cid]
    pts = co.gen_points(cid, grp, n, seed=rnd.randrange(1 << 30))
    w = pts.shape[1] // 2
    inf = np.zeros(n, dtype=np.uint8)
    sc = co.gen_scalars(fr, n, seed=rnd.randrange(1 << 30), dist=rnd.randrange(2))
    if n > 8:
        for _ in range(rnd.randrange(4)):                       # copies of a point, sometimes with the same scalar
            i, j = rnd.randrange(n), rnd.randrange(n)
            pts[j] = pts[i]
            if rnd.random() < 0.5: sc[j] = sc[i]
        for _ in range(rnd.randrange(3)):                       # P and -P with the same scalar: cancellation inside a bucket
            i, j = rnd.randrange(n), rnd.randrange(n)
            if i != j:
                pts[j] = pts[i]
                L = 5 if cid < 2 else 12                      # u64 limbs of a base-field element: y is `deg` of them
                pts[j, w:] = co.fp_op(co.CURVE_FQ[cid], "neg", np.ascontiguousarray(pts[i, w:].reshape(-1, L))).reshape(-1)
                sc[j] = sc[i]
        for _ in range(rnd.randrange(3)):
            inf[rnd.randrange(n)] = 1
        if rnd.random() < 0.3:                                  # a run of equal scalars: one long bucket run
            a = rnd.randrange(n); b = min(n, a + rnd.randrange(1, max(2, n // 2)))
            sc[a:b] = sc[a]
        if rnd.random() < 0.2:
            k = rnd.randrange(n); sc[k] = 0; sc[k, 0] = 1
    pre = rnd.choice((-1, 0, 2, 3))
    sort = rnd.randrange(3)
    acc = rnd.choice(((0, 0, 0), (1, 0, 0), (2, rnd.randrange(2, 300), rnd.randrange(0, 20)), (2, 0, 0)))
    want = co.to_affine(cid, grp, co.msm(cid, grp, pts, sc, inf=inf, nthreads=8))
    ctx.set_precompute(pre); ctx.msm_set_sort(sort); ctx.msm_set_accumulate(*acc)
    b = ctx.bases_upload(cid, grp, pts, inf)
    got = co.to_affine(cid, grp, ctx.msm(b, sc))
    b.free()
    if not (np.array_equal(want[0], got[0]) and np.array_equal(want[1], got[1])):
        print("MISMATCH", dict(cid=cid, grp=grp, n=n, pre=pre, sort=sort, acc=acc, case=cases), flush=True)
        sys.exit(1)
    cases += 1
    per_group[(cid, grp)] = per_group.get((cid, grp), 0) + 1
print(f"stress ok: {cases} random MSM cases in {time.time() - t0:.0f} s, per (curve, group): {sorted(per_group.items())}", flush=True)
