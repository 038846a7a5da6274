"""Developer A/B of the two accumulation forms of the 753-bit G1 MSM (running sums against the pair tree of affine additions,
pcdhip_msm_set_accumulate), not a test: stage times of both on the same resident inputs, results compared with each other and
(at the smaller size) with the oracle.  Usage: python tools/tree_bench.py [logn ...]"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from oracle import coracle as co
from pcd_amd import capi

logs = [int(a) for a in sys.argv[1:]] or [18, 20]
ctx = capi.Context(0)
ctx.msm_profile(True)
for cid in (2,):
    fr = co.CURVE_FR[cid]
    for logn in logs:
        n = 1 << logn
        pts = co.gen_points_mt(cid, 1, n, seed=1) if hasattr(co, "gen_points_mt") else co.gen_points(cid, 1, n, seed=1)
        b = ctx.bases_upload(cid, 1, pts)
        for dist in (0, 1):
            sc = co.gen_scalars(fr, n, seed=2, dist=dist)
            sb = ctx.buf_upload(fr, sc)
            ref = None
            variants = [(1, 0, 0), (2, 0, 0), (2, 0, 24), (2, 288, 0)]
            for mode, chunk, mp in variants:
                ctx.msm_set_accumulate(mode, chunk, mp)
                got = co.to_affine(cid, 1, ctx.msm(b, sb))
                if ref is None:
                    ref = got
                    if logn <= 18:
                        want = co.to_affine(cid, 1, co.msm(cid, 1, pts, sc, nthreads=64))
                        assert np.array_equal(want[0], got[0]), "running sums differ from the oracle"
                same = np.array_equal(ref[0], got[0]) and np.array_equal(ref[1], got[1])
                best = None
                for _ in range(3):
                    ctx.msm(b, sb); tm = ctx.msm_last_timings()
                    if best is None or tm["total"] < best["total"]: best = tm
                print(f"curve={cid} G1 n=2^{logn} dist={dist} mode={mode} chunk={chunk} min_pairs={mp} same={same}: " +
                      " ".join(f"{k}={v:.2f}" for k, v in best.items()), flush=True)
            sb.free()
        b.free()
ctx.msm_set_accumulate(0)
