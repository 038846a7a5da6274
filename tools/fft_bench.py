"""Developer measurement / profiling target: resident radix-2 transforms (pcdhip_fft_dev) over the 298- and 753-bit scalar
fields at n = 2^20 (and 2^22 for the 298-bit field): per-pass device time, achieved HBM GB/s per pass (2 x n x element bytes
per pass: one read + one write of the vector, device image 44 / 108 B per element).  Run under rocprofv3 for profiles/."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from oracle import coracle as co
from pcd_amd import capi

ctx = capi.Context(0)
reps = int(os.environ.get("FFT_REPS", "5"))
cases = [tuple(int(v) for v in c.split(":")) for c in os.environ["FFT_CASES"].split(",")] if os.environ.get("FFT_CASES") else [(1, 20), (3, 20), (1, 22)]
for fid, logn in cases:
    n = 1 << logn
    x = ctx.buf_upload(fid, co.gen_field(fid, n, seed=1))
    ctx.fft(fid, x)
    ctx.sync()
    tot = []
    for _ in range(reps):
        ctx.timer_start(); ctx.fft(fid, x, coset=True); tot.append(ctx.timer_stop())
    passes = ctx.fft_last_timings()
    eb = 44 if fid < 2 else 108
    per = [2 * n * eb / (p * 1e-3) / 1e12 for p in passes]
    print(f"fft field={fid} n=2^{logn}: whole call (incl. ABI<->device-image conversions) {np.median(tot):.3f} ms; passes ms={[round(p, 4) for p in passes]} "
          f"TB/s per pass={[round(v, 2) for v in per]}", flush=True)
    x.free()
