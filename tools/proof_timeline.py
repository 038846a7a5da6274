"""Developer tool: the kernel timeline of ONE main Groth16 proof (MNT4-298, 2^20, the bench's key shape) from a rocprofv3 kernel trace:
per 0.5 ms slice, which kernels were running (by family) -- where the device idles or runs latency-bound work alone.

    rocprofv3 --kernel-trace --output-format csv -d /tmp/pt -- python3 tools/proof_timeline.py run
    [PT_SLICE_US=200] python3 tools/proof_timeline.py report /tmp/pt
(PT_CURVE / PT_NC choose the proof; under the tracer every launch costs the host ~50 us, so small proofs come out launch-bound and stretched)"""
import glob, os, sys, time, csv, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def run():
    import numpy as np
    import torch
    torch.zeros(1, device="cuda:0")
    from oracle import coracle as co
    from pcd_amd import capi
    curve = int(os.environ.get("PT_CURVE", "0"))
    ctx = capi.Context(0)
    fr = co.CURVE_FR[curve]
    r = (co.witness_r1cs if os.environ.get("PT_WITNESS") else co.skewed_r1cs)(fr, int(os.environ.get("PT_NC", (1 << 20) - 8)), 2, seed=77)
    keys = co.synthetic_keys(curve, r, seed=78, mt=True)
    rs = co.gen_field(fr, 2, seed=79)
    pk = ctx.g16_pk_upload(keys.host_struct(), curve)
    ctx.g16_pk_set_r1cs(pk, r)
    r.z = capi.pinned_like(r.z)
    if os.environ.get("PT_SCHED"):   # `schedule` or `schedule:reserved CUs` (pcdhip_groth16_set_schedule, pcdhip_set_lane_reserve)
        sched, _, res = os.environ["PT_SCHED"].partition(":")
        ctx.groth16_set_schedule(int(sched))
        if res:
            ctx.set_lane_reserve(int(res))
    for _ in range(4):
        ctx.groth16_prove(pk, r, rs[0], rs[1], resident_r1cs=True)
    torch.cuda.synchronize()
    time.sleep(0.1)
    t0 = time.perf_counter(); ctx.groth16_prove(pk, r, rs[0], rs[1], resident_r1cs=True); dt = (time.perf_counter() - t0) * 1e3
    print(f"PT wall {dt:.2f} ms; device {ctx.groth16_last_timings()}")


def family(name):
    for key, fam in (("msm_accumulate", "ACC"), ("msm_tail", "tail"), ("msm_fixup", "fix"), ("msm_big", "fix"), ("msm_coarse", "sort"), ("msm_bin", "sort"),
                     ("msm_digits", "sort"), ("fft_", "fft"), ("spmv", "spmv"), ("msm_horner", "tail"), ("msm_merge", "fix")):
        if key in name:
            return fam
    return "other"


def report(d):
    f = glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True)[0]
    rows = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in csv.DictReader(open(f))))
    cut = 0
    for i in range(1, len(rows)):
        if rows[i][0] - rows[i - 1][1] > 50_000_000:
            cut = i
    last = rows[cut:]
    t0, t1 = last[0][0], max(b for _, b, _ in last)
    print(f"{len(last)} kernels over {(t1 - t0) / 1e6:.2f} ms")
    # every accumulate kernel and every kernel of 0.3 ms and more: start, duration, grid
    rows_full = {(int(r["Start_Timestamp"]), r["Kernel_Name"]): r for r in csv.DictReader(open(f))}
    for a, b, n in last:
        if "msm_accumulate" in n or b - a >= int(os.environ.get("PT_MIN_US", "300")) * 1000:
            r = rows_full[(a, n)]
            grid = r.get("Grid_Size_X") or r.get("Grid_Size") or "?"
            wg = r.get("Workgroup_Size_X") or r.get("Workgroup_Size") or "?"
            short = n.split("(")[0].replace("void pcd::", "").replace("pcd::", "")[:70]
            print(f"  {(a - t0) / 1e6:7.3f} ms  +{(b - a) / 1e6:6.3f} ms  q{r.get('Queue_Id', '?')} s{r.get('Stream_Id', '?')}  grid {grid}/{wg}  {short}")
    step = int(os.environ.get("PT_SLICE_US", "500")) * 1000
    for s in range(t0, t1, step):
        busy = collections.Counter()
        for a, b, n in last:
            ov = min(b, s + step) - max(a, s)
            if ov > 0:
                busy[family(n) + ("2" if "G2Cfg" in n and "ACC" == family(n) else "")] += ov / step
        print(f"{(s - t0) / 1e6:6.1f} ms  " + "  ".join(f"{k}:{v:.1f}" for k, v in sorted(busy.items())))


if __name__ == "__main__":
    run() if sys.argv[1] == "run" else report(sys.argv[2])
