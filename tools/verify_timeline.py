"""Developer tool: the kernel timeline of ONE prepared Groth16 verification (MNT4-298), from a rocprofv3 --kernel-trace run of this
script: which kernels run, how long each takes and the gaps between them (host round trips, launches).

    rocprofv3 --kernel-trace --output-format csv -d /tmp/vt -- python3 tools/verify_timeline.py run
    python3 tools/verify_timeline.py report /tmp/vt"""
import glob, os, sys, time, csv
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def run():
    import numpy as np
    import torch
    torch.zeros(1, device="cuda:0")
    from oracle import coracle as co
    from pcd_amd import capi
    cid = int(os.environ.get("VT_CURVE", "0"))
    ctx = capi.Context(0)
    fr = co.CURVE_FR[cid]
    r = co.synthetic_r1cs(fr, 60, 3, seed=50 + cid)
    keys = co.groth16_setup(cid, r, co.gen_field(fr, 5, seed=51), nthreads=16)
    pk = ctx.g16_pk_upload(keys.host_struct(), cid)
    proof = ctx.groth16_prove(pk, r, *co.gen_field(fr, 2, seed=60))[0]
    pub = co.fp_op(fr, "to_canonical", np.ascontiguousarray(r.z[1:r.num_inputs]))
    pvk = ctx.process_vk(cid, keys.alpha_g1, keys.beta_g2, keys.gamma_g2, keys.delta_g2, keys.gamma_abc_g1)
    P, U = proof[None], pub[None]
    for _ in range(5):
        assert ctx.groth16_verify_prepared(pvk, U, P).all()
    torch.cuda.synchronize()
    time.sleep(0.05)     # a visible gap in the trace in front of the measured call
    t0 = time.perf_counter(); ctx.groth16_verify_prepared(pvk, U, P); dt = (time.perf_counter() - t0) * 1e3
    print(f"VT host wall time of the last verification: {dt:.3f} ms")


def report(d):
    f = glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True)[0]
    rows = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in csv.DictReader(open(f))))
    # the last burst: kernels after the last gap of more than 20 ms
    cut = 0
    for i in range(1, len(rows)):
        if rows[i][0] - rows[i - 1][1] > 20_000_000:
            cut = i
    last = rows[cut:]
    t0 = last[0][0]
    for a, b, name in last:
        print(f"{(a - t0) / 1e3:9.1f} us  +{(b - a) / 1e3:8.1f} us  {name[:110]}")
    print(f"first kernel start -> last kernel end: {(last[-1][1] - t0) / 1e3:.1f} us; kernel time {sum(b - a for a, b, _ in last) / 1e3:.1f} us")


if __name__ == "__main__":
    run() if sys.argv[1] == "run" else report(sys.argv[2])
