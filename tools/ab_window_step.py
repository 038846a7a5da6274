"""Developer A/B: one main proof (the bench's key shape, 2^20) with the keys' MSM window fixed at upload time -- is the proof, which is
bound by the SUM of its kernels' work (tails and fix-ups included), better off with a smaller window than the lone MSM is?

    [AB_NC=<constraints>] python tools/ab_window_step.py <curve> <window bits, 0 = the library's choice> ..."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
torch.zeros(1, device="cuda:0")
from oracle import coracle as co
from pcd_amd import capi
curve = int(sys.argv[1])
ctx = capi.Context(0)
fr = co.CURVE_FR[curve]
nc = int(os.environ.get('AB_NC', (1 << 20) - 8))
r = co.skewed_r1cs(fr, nc, 2, seed=77)
keys = co.synthetic_keys(curve, r, seed=78, mt=True)
rs = co.gen_field(fr, 2, seed=79)
r.z = capi.pinned_like(r.z)
first = None
for c in [int(x) for x in sys.argv[2:]]:
    ctx.msm_config(c, 0)
    pk = ctx.g16_pk_upload(keys.host_struct(), curve)
    ctx.g16_pk_set_r1cs(pk, r)
    for _ in range(3):
        proof, _ = ctx.groth16_prove(pk, r, rs[0], rs[1], resident_r1cs=True)
    first = proof if first is None else first
    assert np.array_equal(proof, first)
    w = []
    for _ in range(7):
        t0 = time.perf_counter(); ctx.groth16_prove(pk, r, rs[0], rs[1], resident_r1cs=True); w.append((time.perf_counter() - t0) * 1e3)
    print(f"curve {curve} window {c}: prove wall ms median {np.median(w):.2f} min {min(w):.2f}", flush=True)
    pk.free()
