"""Developer measurement of the 753-bit workloads (BASELINE configs[2] shapes), not a test."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from oracle import coracle as co
from pcd_amd import capi
ctx = capi.Context(0)
ctx.msm_profile(True)
for cid, grp, logn in ((2, 1, 18), (2, 1, 20), (2, 2, 18), (3, 2, 16)):
    n = 1 << logn
    fr = co.CURVE_FR[cid]
    t = time.time(); pts = co.gen_points(cid, grp, n, seed=1); tg = time.time() - t
    sc = co.gen_scalars(fr, n, seed=2, dist=0)
    sb = ctx.buf_upload(fr, sc)
    t = time.time(); b = ctx.bases_upload(cid, grp, pts); tu = time.time() - t
    got = ctx.msm(b, sb)
    t = time.time(); want = co.msm(cid, grp, pts, sc, nthreads=64); tc = time.time() - t
    ok = np.array_equal(co.to_affine(cid, grp, got)[0], co.to_affine(cid, grp, want)[0])
    best = None
    for _ in range(3):
        ctx.msm(b, sb); tm = ctx.msm_last_timings()
        if best is None or tm["total"] < best["total"]: best = tm
    print(f"msm curve={cid} G{grp} n=2^{logn} ok={ok} gen={tg:.1f}s upload+precompute={tu:.2f}s cpu(51 thr)={tc:.2f}s gpu: " + " ".join(f"{k}={v:.2f}" for k, v in best.items()), flush=True)
    b.free(); sb.free()
for fid, logn in ((3, 20), (1, 20)):
    x = co.gen_field(fid, 1 << logn, seed=1)
    xb = ctx.buf_upload(fid, x)
    ctx.fft(fid, xb)
    ctx.timer_start(); ctx.fft(fid, xb); ms = ctx.timer_stop()
    print(f"fft field={fid} n=2^{logn}: {ms:.3f} ms (incl. ABI conversions) passes={ctx.fft_last_timings()}", flush=True)
