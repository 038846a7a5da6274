"""Developer smoke script (not a test): GPU vs CPU-oracle on the first kernels, with timings."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from oracle import coracle as Co
from pcd_amd import capi

ctx = capi.Context(0)
ok = True
def check(name, cond):
    global ok
    print(("PASS " if cond else "FAIL ") + name, flush=True)
    ok &= bool(cond)

which = sys.argv[1] if len(sys.argv) > 1 else "all"
if which in ("all", "msm"):
    for curve, group, sizes in ((0, 1, (1, 33, 1000, 1 << 14)), (1, 1, (1000,)), (0, 2, (300,)), (1, 2, (300,)), (2, 1, (300,)), (3, 1, (300,)), (2, 2, (100,)), (3, 2, (100,))):
        for n in sizes:
            for dist in (0, 1):
                pts = Co.gen_points(curve, group, n, seed=11)
                sc = Co.gen_scalars(Co.CURVE_FR[curve], n, seed=5 + n, dist=dist)
                t = time.time(); want = Co.msm(curve, group, pts, sc, nthreads=8); tc = time.time() - t
                wa, wi = Co.to_affine(curve, group, want)
                for mode in (-1, 0, 3):
                    ctx.set_precompute(mode)
                    t = time.time(); b = ctx.bases_upload(curve, group, pts); tu = time.time() - t
                    t = time.time(); got = ctx.msm(b, sc); tg = time.time() - t
                    ga, gi = Co.to_affine(curve, group, got)
                    check(f"msm curve={curve} G{group} n={n} dist={dist} precomp={mode} cpu={tc:.3f}s upload={tu:.3f}s gpu={tg:.3f}s", np.array_equal(wa, ga) and np.array_equal(wi, gi))
                    b.free()
                ctx.set_precompute(-1)
if which in ("all", "fft"):
    for field in (0, 1, 2, 3):
        for log_n in (1, 4, 10, 11, 13, 15):
            x = Co.gen_field(field, 1 << log_n, seed=log_n)
            for inv in (0, 1):
                for coset in (0, 1):
                    want = Co.fft(field, x, inverse=inv, coset=coset, nthreads=8)
                    got = ctx.fft(field, x, inverse=inv, coset=coset)
                    check(f"fft field={field} log_n={log_n} inv={inv} coset={coset}", np.array_equal(want, got))
if which in ("all", "g16"):
    for curve in (0, 1):
        fr = Co.CURVE_FR[curve]
        r = Co.synthetic_r1cs(fr, 500, 3, seed=2)
        h = ctx.witness_map(fr, r)
        check(f"witness_map curve={curve}", np.array_equal(h, Co.witness_map(r, nthreads=8)))
        tox = Co.gen_field(fr, 5, seed=77)
        keys = Co.groth16_setup(curve, r, tox, nthreads=8)
        rs = Co.gen_field(fr, 2, seed=78)
        want, winf = Co.groth16_prove(keys, r, rs[0], rs[1], nthreads=8)
        pk = ctx.g16_pk_upload(keys.host_struct(), curve)
        t = time.time(); got, ginf = ctx.groth16_prove(pk, r, rs[0], rs[1]); tg = time.time() - t
        check(f"groth16_prove curve={curve} gpu={tg:.3f}s {ctx.groth16_last_timings()}", np.array_equal(want, got))
        check(f"groth16 verify curve={curve}", Co.groth16_verify(keys, r.z[1:r.num_inputs], got))
if which in ("all", "perf"):
    ctx.msm_profile(True)
    for curve, group, logn in ((0, 1, 16), (0, 1, 20), (0, 2, 18), (2, 1, 16), (2, 1, 18)):
        n = 1 << logn
        pts = Co.gen_points(curve, group, n, seed=1)
        for mode in (-1, 0):
            ctx.set_precompute(mode)
            t = time.time(); b = ctx.bases_upload(curve, group, pts); tu = time.time() - t
            for dist in (0, 1):
                sc = Co.gen_scalars(Co.CURVE_FR[curve], n, seed=3, dist=dist)
                sb = ctx.buf_upload(Co.CURVE_FR[curve], sc)
                ctx.msm(b, sb)
                t = time.time(); got = ctx.msm(b, sb); tg = time.time() - t
                tm = {k: round(v, 3) for k, v in ctx.msm_last_timings().items()}
                print(f"perf msm curve={curve} G{group} n=2^{logn} dist={dist} precomp={mode} upload={tu:.2f}s: {tg*1e3:.2f} ms wall  {tm}", flush=True)
                sb.free()
            b.free()
        ctx.set_precompute(-1)
    for field, logn in ((1, 16), (1, 20), (3, 20)):
        x = Co.gen_field(field, 1 << logn, seed=1)
        xb = ctx.buf_upload(field, x)
        ctx.fft(field, xb)
        ctx.timer_start(); ctx.fft(field, xb); ms = ctx.timer_stop()
        print(f"perf fft field={field} n=2^{logn}: {ms:.3f} ms passes={ctx.fft_last_timings()}", flush=True)
print("ALL OK" if ok else "SOME FAILED")
sys.exit(0 if ok else 1)
