"""Developer probe: the accumulate kernel of ONE MSM at a time (pcdhip_msm_profile: HIP events around the launch) over the vectors a proof
multiplies -- the bench's standalone MSM (uniform scalars, every base finite) at the lone MSM's window and at the key's, then the a / b_g1 /
l / h queries of the bench's Groth16 key with the proof's assignment -- with the list length and the entries per lane the device chose.
What a proof's accumulate lane can reach is the SUM of these, not five times the headline kernel.

    python tools/acc_probe.py [curve [log rows]]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
torch.zeros(1, device="cuda:0")
from oracle import coracle as co
from pcd_amd import capi

curve = int(sys.argv[1]) if len(sys.argv) > 1 else 0
logn = int(sys.argv[2]) if len(sys.argv) > 2 else 20
fr = co.CURVE_FR[curve]
ctx = capi.Context(0)


def probe(tag, bases, sbuf, n=None, reps=6):
    for _ in range(3):
        ctx.msm(bases, sbuf, n=n)
    ctx.msm_profile(True)
    acc, tot = [], []
    for _ in range(reps):
        ctx.msm(bases, sbuf, n=n)
        t = ctx.msm_last_timings()
        acc.append(t["accumulate"]); tot.append(t["total"])
    M, chunk = ctx.msm_last_plan()
    ctx.msm_profile(False)
    c, W, copies = ctx.bases_info(bases)
    print(f"{tag:34s} c={c} W={W} entries={M / 1e6:6.2f} M chunk={chunk:2d}  accumulate {np.median(acc):.3f} ms (min {min(acc):.3f})  "
          f"-> {M / np.median(acc) / 1e6:.2f} G entries/s   whole MSM {np.median(tot):.3f} ms", flush=True)


n = 1 << logn
pts = co.gen_points(curve, 1, n, seed=1)
sc = co.gen_scalars(fr, n, seed=2, dist=0)
sbuf = ctx.buf_upload(fr, sc)
c0 = 0
for c in (0, -1):
    if c:
        ctx.msm_config(c0 - 1, 0)
    b = ctx.bases_upload(curve, 1, pts)
    c0 = ctx.bases_info(b)[0]
    probe(f"standalone G1 2^{logn} uniform", b, sbuf)
    if not c:
        b.free()
ctx.msm_config(0, 0)
sw = co.gen_scalars(fr, n, seed=3, dist=1)
swb = ctx.buf_upload(fr, sw)
probe(f"standalone G1 2^{logn} witness-like", b, swb)
b.free()

r = (co.witness_r1cs if os.environ.get("AB_WITNESS") else co.skewed_r1cs)(fr, n - 8, 2, seed=77)
keys = co.synthetic_keys(curve, r, seed=78, mt=True)
z = co.fp_op(fr, "to_canonical", np.ascontiguousarray(r.z))
zb = ctx.buf_upload(fr, z)
m = r.num_vars
win = None
for name, q, inf in (("a_query", keys.a_query, keys.a_inf), ("b_g1_query", keys.b_g1_query, keys.b_g1_inf), ("l_query", keys.l_query, None),
                     ("h_query", keys.h_query, None)):
    cnt = min(q.shape[0], m)
    if win is None:   # the window a key's queries get (one bit under the lone MSM's choice at this size)
        b0 = ctx.bases_upload(curve, 1, q[:cnt], inf=None if inf is None else inf[:cnt])
        win = ctx.bases_info(b0)[0] - (1 if cnt >= (1 << 18) else 0)
        b0.free()
        ctx.msm_config(win, 0)
    b = ctx.bases_upload(curve, 1, q[:cnt], inf=None if inf is None else inf[:cnt])
    frac = 0.0 if inf is None else float(np.mean(inf[:cnt] != 0))
    probe(f"{name} ({100 * frac:.0f} % at infinity) x z", b, zb, n=cnt)
    b.free()
