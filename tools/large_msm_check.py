"""Large-size sanity run (not a test): one G1 MSM at n = 2^22 on MNT4-298 and at 2^21 on MNT4-753 (the per-GPU shard sizes
of BASELINE configs[4]: 2^22 pairs sharded over 8 GPUs leave 2^19 per GPU; these are 8x / 4x that), result compared with
the CPU oracle, stage timings printed."""
import sys, os, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from oracle import coracle as co
from pcd_amd import capi
ctx = capi.Context(0)
ctx.msm_profile(True)
for cid, grp, logn in ((0, 1, 22), (2, 1, 21)):
    n = 1 << logn
    fr = co.CURVE_FR[cid]
    t = time.time(); pts = co.gen_points(cid, grp, n, seed=1); tg = time.time() - t
    sc = co.gen_scalars(fr, n, seed=2)
    sb = ctx.buf_upload(fr, sc)
    t = time.time(); b = ctx.bases_upload(cid, grp, pts); tu = time.time() - t
    got = ctx.msm(b, sb)
    t = time.time(); want = co.msm(cid, grp, pts, sc, nthreads=64); tc = time.time() - t
    ok = bool(np.array_equal(co.to_affine(cid, grp, got)[0], co.to_affine(cid, grp, want)[0]))
    ctx.msm(b, sb)
    tm = {k: round(float(v), 2) for k, v in ctx.msm_last_timings().items()}
    print(json.dumps({"curve": cid, "group": grp, "log_n": logn, "ok_vs_oracle": ok, "gpu_ms": tm, "Mscalar_mul_per_s": round(n / tm["total"] / 1e3, 1),
                      "cpu_port_s": round(tc, 2), "upload_precompute_s": round(tu, 2), "gen_s": round(tg, 1)}), flush=True)
    b.free(); sb.free()
