"""Stand-alone device time of the MSM shapes of one main Groth16 proof (MNT4-298, n = 2^20): what the concurrent
schedule of pcdhip_groth16_prove should be compared with."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from oracle import coracle as co
from pcd_amd import capi

ctx = capi.Context(0)
ctx.msm_profile(True)
curve = int(sys.argv[1]) if len(sys.argv) > 1 else 0
n = 1 << (int(sys.argv[2]) if len(sys.argv) > 2 else 20)
fr = co.CURVE_FR[curve]
out = {}
for group in (1, 2):
    pts = co.gen_points(curve, group, n, seed=5)
    bases = ctx.bases_upload(curve, group, pts)
    for dist in (0, 1):
        sc = co.gen_scalars(fr, n, seed=6, dist=dist)
        sbuf = ctx.buf_upload(fr, sc)
        for _ in range(3):
            ctx.msm(bases, sbuf)
        tms = []
        for _ in range(5):
            ctx.msm(bases, sbuf)
            tms.append(ctx.msm_last_timings())
        best = min(tms, key=lambda t: t["total"])
        out[f"g{group}_dist{dist}"] = {k: round(float(v), 3) for k, v in best.items()}
        print(f"g{group}_dist{dist}", json.dumps(out[f"g{group}_dist{dist}"]), flush=True)
        sbuf.free()
    bases.free()
