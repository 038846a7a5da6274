"""Developer measurement: MSM stage times per group, checked against the CPU oracle.  Not a test.

    python tools/perf_msm.py 1:2:17 3:2:16 0:2:20        # curve:group:log_n ...
    PERF_NOCHECK=1 python tools/perf_msm.py 0:1:20       # skip the oracle leg
"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from oracle import coracle as co
from pcd_amd import capi

ctx = capi.Context(0)
ctx.msm_profile(True)
nocheck = os.environ.get("PERF_NOCHECK") == "1"
dist = int(os.environ.get("PERF_DIST", "0"))
for spec in sys.argv[1:]:
    cid, grp, logn = (int(x) for x in spec.split(":"))
    n = 1 << logn
    fr = co.CURVE_FR[cid]
    pts = co.gen_points(cid, grp, n, seed=1)
    sc = co.gen_scalars(fr, n, seed=2, dist=dist)
    sb = ctx.buf_upload(fr, sc)
    t = time.time(); b = ctx.bases_upload(cid, grp, pts); tu = time.time() - t
    got = ctx.msm(b, sb)
    ok = "unchecked"
    if not nocheck:
        want = co.msm(cid, grp, pts, sc, nthreads=min(os.cpu_count() or 1, 64))
        ok = bool(np.array_equal(co.to_affine(cid, grp, got)[0], co.to_affine(cid, grp, want)[0]))
    runs = []
    for _ in range(5):
        ctx.msm(b, sb)
        runs.append(ctx.msm_last_timings())
    med = {k: float(np.median([r[k] for r in runs])) for k in runs[0]}
    c, W, copies = ctx.bases_info(b)
    print(f"msm curve={cid} G{grp} n=2^{logn} ok={ok} c={c} W={W} copies={copies} upload+precompute={tu:.2f}s  " +
          " ".join(f"{k}={v:.3f}" for k, v in med.items()), flush=True)
    b.free(); sb.free()
