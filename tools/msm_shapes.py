"""Stand-alone device time of the MSM shapes of one main Groth16 proof (MNT4-298, n = 2^20): what the concurrent
schedule of pcdhip_groth16_prove should be compared with."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from oracle import coracle as co
from pcd_amd import capi

ctx = capi.Context(0)
ctx.msm_profile(True)
n = 1 << 20
out = {}
for group in (1, 2):
    pts = co.gen_points(0, group, n, seed=5)
    bases = ctx.bases_upload(0, group, pts)
    for dist in (0, 1):
        sc = co.gen_scalars(1, n, seed=6, dist=dist)
        sbuf = ctx.buf_upload(1, sc)
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
