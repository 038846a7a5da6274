"""Developer sweep of MSM window bits / chunk size on the headline workload (not a test)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from oracle import coracle as co
from pcd_amd import capi
ctx = capi.Context(0)
ctx.msm_profile(True)
cid, grp = int(sys.argv[1]) if len(sys.argv) > 1 else 0, int(sys.argv[2]) if len(sys.argv) > 2 else 1
logn = int(sys.argv[3]) if len(sys.argv) > 3 else 20
n = 1 << logn
fr = co.CURVE_FR[cid]
pts = co.gen_points(cid, grp, n, seed=1)
sc = co.gen_scalars(fr, n, seed=2, dist=0)
sb = ctx.buf_upload(fr, sc)
want = co.to_affine(cid, grp, co.msm(cid, grp, pts, sc, nthreads=32))[0]
cs = [int(x) for x in os.environ.get('SWEEP_C', '19,20,21').split(',')]
chunks = [int(x) for x in os.environ.get('SWEEP_CHUNK', '32,40,48,64').split(',')]
for c in cs:
    for chunk in chunks:
        ctx.msm_config(c, chunk)
        ctx.set_precompute(-1)
        b = ctx.bases_upload(cid, grp, pts)
        got = ctx.msm(b, sb)
        ok = np.array_equal(co.to_affine(cid, grp, got)[0], want)
        best = None
        for _ in range(7):
            ctx.msm(b, sb)
            tm = ctx.msm_last_timings()
            if best is None or tm["total"] < best["total"]:
                best = tm
        print(f"c={c} chunk={chunk} ok={ok} " + " ".join(f"{k}={v:.3f}" for k, v in best.items()), flush=True)
        b.free()
