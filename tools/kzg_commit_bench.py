"""Measurement of BASELINE configs[3]'s shape (SURVEY.md 8d "C4"): the KZG multi-MSM commit pattern of a Marlin
prover over MNT4-298 G1 (K7 of section 8a) -- one resident `powers_of_g` vector of 6n points and one `powers_of_gamma_g`
vector of n points, n = 2^20; a batch of 11 commitments over prefixes of the powers: 7 of length n (w, z_a, z_b, mask,
t, g_1, h_1-like), 2 of length 6n (g_2, h_2-like), 2 of length n (opening witnesses), each with a hiding MSM of
length n over powers_of_gamma_g; plus the FFTs at n and 4n such a round performs.  The shape is an approximation of
Marlin's commit pattern (exact counts come from the Rust host).  One 2^20 commitment is checked against the CPU oracle
and the oracle's time for it, scaled by pair count, is the CPU figure printed next to the GPU time."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from oracle import coracle as co
from pcd_amd import capi

ctx = capi.Context(0)
curve, fr = 0, co.CURVE_FR[0]
log_n = int(sys.argv[1]) if len(sys.argv) > 1 else 20
n = 1 << log_n
t = time.time()
powers = co.gen_points(curve, 1, 6 * n, seed=41)         # stands in for [tau^i] g (any points do for timing / parity)
gamma = co.gen_points(curve, 1, n, seed=42)
polys = co.gen_scalars(fr, 6 * n, seed=43)               # coefficient vectors are read as prefixes of this one
blind = co.gen_scalars(fr, n, seed=44)
gen_s = time.time() - t
t = time.time()
P = ctx.bases_upload(curve, 1, powers)
G = ctx.bases_upload(curve, 1, gamma)
up_s = time.time() - t
S = ctx.buf_upload(fr, polys)
B = ctx.buf_upload(fr, blind)
lengths = [n] * 7 + [6 * n] * 2 + [n] * 2

def batch():
    outs = []
    for L in lengths:
        c = ctx.msm(P, S, offset=0, n=L)                 # commitment to a degree-(L-1) polynomial
        h = ctx.msm(G, B, offset=0, n=n)                 # hiding term
        outs.append(ctx.points_sum(curve, 1, np.stack([c, h])))
    return outs

def batch_pipelined():
    """the same 22 MSMs with up to four in flight (pcdhip_msm_submit / collect): the bucket reduction of one overlaps the next"""
    jobs = []
    for L in lengths:
        jobs.append((P, S, L)); jobs.append((G, B, n))
    res, pending = [], []
    for (bs, sc_, L) in jobs:
        pending.append(ctx.msm_submit(bs, sc_, offset=0, n=L))
        if len(pending) == 4:
            res.append(ctx.msm_collect(pending.pop(0)))
    while pending:
        res.append(ctx.msm_collect(pending.pop(0)))
    return [ctx.points_sum(curve, 1, np.stack([res[2 * i], res[2 * i + 1]])) for i in range(len(lengths))]

batch()
t = time.perf_counter(); outs = batch(); gpu_ms = (time.perf_counter() - t) * 1e3
batch_pipelined()
t = time.perf_counter(); outs_p = batch_pipelined(); gpu_ms_pipe = (time.perf_counter() - t) * 1e3
# (Jacobian representatives differ from run to run -- the order of additions inside a bucket follows the atomics of the sort -- so compare affine)
assert all(np.array_equal(co.to_affine(curve, 1, a)[0], co.to_affine(curve, 1, b)[0]) for a, b in zip(outs, outs_p))
threads = min(os.cpu_count() or 1, 20)
t = time.perf_counter(); want = co.msm(curve, 1, powers[:n], polys[:n], nthreads=threads); cpu_one = time.perf_counter() - t
wanth = co.msm(curve, 1, gamma, blind, nthreads=threads)
ok = np.array_equal(co.to_affine(curve, 1, outs[0])[0], co.to_affine(curve, 1, co.jac_add(curve, 1, want, wanth))[0])
pairs = sum(lengths) + n * len(lengths)
# the FFTs of such a round: domain_h-sized and 4x (product domains), resident data
ffts = {}
for ln in (log_n, log_n + 2):
    x = ctx.buf_upload(fr, co.gen_field(fr, 1 << ln, seed=45))
    ctx.fft(fr, x)
    ctx.sync()
    ctx.timer_start(); ctx.fft(fr, x); ffts[f"fft_2^{ln}_ms"] = round(ctx.timer_stop(), 3)
    x.free()
out = {"workload": f"MNT4-298 G1, n=2^{log_n}: 11 KZG commitments (7+2 of n, 2 of 6n) each with an n-point hiding MSM",
       "ok_vs_oracle_first_commitment": bool(ok), "pairs": pairs, "gpu_batch_ms": round(gpu_ms, 2), "gpu_batch_ms_pipelined": round(gpu_ms_pipe, 2),
       "gpu_Mpairs_per_s": round(pairs / gpu_ms / 1e3, 1),
       "cpu_port_s_scaled": round(cpu_one * pairs / n, 1), "cpu_threads": threads,
       "speedup_vs_cpu_port": round(cpu_one * pairs / n / (gpu_ms / 1e3), 1),
       "key_upload_precompute_s": round(up_s, 1), "input_gen_s": round(gen_s, 1), **ffts}
print(json.dumps(out))
