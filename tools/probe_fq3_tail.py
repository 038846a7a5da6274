"""Developer probe: tiny MNT6-753 G2 MSMs against the oracle, to localise a failing reduction kernel (not a test)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from oracle import coracle as co
from pcd_amd import capi
ctx = capi.Context(0)
cid, grp = 3, 2
fr = co.CURVE_FR[cid]
pts = co.gen_points(cid, grp, 64, seed=5)
def run(tag, n, sc, c=0):
    ctx.msm_config(c, 0)
    ctx.set_precompute(-1)
    b = ctx.bases_upload(cid, grp, pts[:n])
    sb = ctx.buf_upload(fr, sc[:n])
    got = co.to_affine(cid, grp, ctx.msm(b, sb))[0]
    want = co.to_affine(cid, grp, co.msm(cid, grp, pts[:n], sc[:n], nthreads=4))[0]
    print(tag, "n", n, "c", c, "ok", bool(np.array_equal(got, want)), flush=True)
    b.free(); sb.free()
small = np.zeros((64, co.gen_scalars(fr, 1, seed=1).shape[1]), dtype=np.uint64)
for k in (1, 2, 3, 5):
    s = small.copy(); s[:, 0] = k
    run(f"scalar={k}", 1, s); run(f"scalar={k}", 2, s)
s = small.copy(); s[:, 0] = np.arange(1, 65)
run("scalars 1..n", 8, s); run("scalars 1..n", 64, s)
rnd = co.gen_scalars(fr, 64, seed=9)
for n in (1, 2, 8, 64):
    run("random", n, rnd)
for c in (6, 8, 10, 12):
    run("random", 64, rnd, c)
pts = co.gen_points(cid, grp, 4096, seed=6)
ones = np.zeros((4096, small.shape[1]), dtype=np.uint64); ones[:, 0] = 1
for n in (3, 64, 65, 200, 1000, 4096):
    run("all ones", n, ones)
mix = co.gen_scalars(fr, 4096, seed=11); mix[::2] = ones[::2]
for n in (64, 200, 1000, 4096):
    run("half ones", n, mix)
same = co.gen_scalars(fr, 4096, seed=12); same[:] = same[0]
for n in (64, 200, 1000, 4096):
    run("one scalar repeated (big buckets)", n, same)
d1 = co.gen_scalars(fr, 4096, seed=2, dist=1)
for n in (90, 1000, 4096):
    run("witness-like", n, d1)
print("--- merge probes (c = 8)")
for other in (2, 3, 257, 258):
    s = small.copy(); s = np.zeros((4096, small.shape[1]), dtype=np.uint64); s[:, 0] = 1; s[1::2, 0] = other
    for n in (2, 8, 64):
        run(f"ones and {other}", n, s, 8)
