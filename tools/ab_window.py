"""Developer A/B: headline MSM (MNT4-298 G1, 2^20, resident) at forced window bits, one at a time and four in flight, same box."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from oracle import coracle as co
from pcd_amd import capi
cid, grp, logn = (int(x) for x in (sys.argv[1:4] if len(sys.argv) > 3 else (0, 1, 20)))
cs = [int(x) for x in os.environ.get("AB_C", "0,19,20").split(",")]
ctx = capi.Context(0)
n = 1 << logn
fr = co.CURVE_FR[cid]
pts = co.gen_points(cid, grp, n, seed=1)
sb = ctx.buf_upload(fr, co.gen_scalars(fr, n, seed=2))
for rep in range(2):
    for c in cs:
        ctx.msm_config(c, 0)
        b = ctx.bases_upload(cid, grp, pts)
        for _ in range(8): ctx.msm(b, sb)
        K = 30
        t = time.perf_counter()
        for _ in range(K): ctx.msm(b, sb)
        one = (time.perf_counter() - t) / K * 1e3
        for tk in [ctx.msm_submit(b, sb) for _ in range(4)]: ctx.msm_collect(tk)
        t = time.perf_counter(); pend = []
        for _ in range(K):
            pend.append(ctx.msm_submit(b, sb))
            if len(pend) >= 4: ctx.msm_collect(pend.pop(0))
        while pend: ctx.msm_collect(pend.pop(0))
        four = (time.perf_counter() - t) / K * 1e3
        print(f"curve={cid} G{grp} 2^{logn} c={c or 'auto'} plan={ctx.bases_info(b)}: one at a time {one:.3f} ms ({n / one / 1e3:.1f} M/s), four in flight {four:.3f} ms ({n / four / 1e3:.1f} M/s)", flush=True)
        b.free()
