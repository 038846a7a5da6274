#!/usr/bin/env python3
"""bench.py -- headline measurement of the PCD prover hot path on MI355X.

Metric (BASELINE.json): MSM Mscalar-mul/s on MNT4-298 G1 at n = 2^20 (proving key resident), plus the PCD-step
prover-arithmetic time, reported in the same JSON line:
  pcd_step      main Groth16 proof over MNT4-298 (domain 2^20) + help proof over MNT6-298 (2^16)
  pcd_step_753  BASELINE configs[2] / north_star target: main proof over MNT4-753 (domain 2^20) + help proof over MNT6-753
                (mixed-radix domain 5 * 2^14), with the roofline of its dominant kernel (G1-753 bucket accumulation)
Every prove is timed as the median of 5 and its proof bytes are compared with the CPU oracle before a number is printed.

One "step" = one variable-base MSM with bases AND scalars already resident in HBM (pcdhip_msm_dev); the Jacobian result
returns to the host.  With N GPUs (one process per GPU, RCCL) the pairs are sharded by point range and the only exchange is an
all-gather of one Jacobian point per rank + a local EC-add kernel:
  default   weak scaling: 2^20 pairs PER GPU (`scaling: weak`; at EVERY world size four independent steps are in flight at a time --
            pcdhip_msm_submit / collect on one GPU, pcdhip_msm_submit_partial + the RCCL exchange of earlier steps on N -- and the
            same steps one at a time are reported next to it); the line also carries `strong` -- the same exchange with a
            fixed TOTAL of 2^20 and of 2^22 pairs split over the N ranks, for MNT4-298 G1 and (BASELINE configs[4]) MNT4-753 G1
  also in the line (N = 1): `fft` (per-pass HBM GB/s and multiply-add fraction of the radix-2 passes at 2^20, both scalar fields), `pairing`
            (Groth16 verification latency, single and batch of 8, beside the CPU oracle on 1 and 8 cores), the witness map alone, and
            the G2 / whole-step multiply-add fractions of the 753-bit step
  --strong  the fixed-total run (2^--log-n pairs, default 2^20) is the headline value (`scaling: strong`)

    python bench.py --gpus 1 --steps 20 --warmup 3
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W [--strong]
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

CURVE, GROUP, LOG_N = 0, 1, 20        # MNT4-298 G1, n = 2^20
SEED = 0x5043443031                   # "PCD01"
HBM_PEAK_GBS = 8000.0                 # MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured copy)
MAD_PEAK = 3.42e13                    # v_mad_u64_u32 lane-ops/s measured on MI355X (profiles/r01_k0_int_rates.txt)
# SURVEY.md 8d contract work per pair at n = 2^20 (upstream window rule c = 15: W = 20 / 51 windows x 11 modmul, CIOS
# modmul = 2 L^2 + L 32-bit mads with L = 10 / 24)
CONTRACT = {0: (220, 210, 120), 2: (561, 1176, 288)}   # curve -> (modmul per pair, mads per modmul, bytes per pair)


def madd_mads(curve):
    """32-bit multiply-adds one mixed addition of the G1 accumulate kernel EXECUTES (28-bit limbs: product 2 N^2, square
    N (N + 1) / 2 + N^2, fused two-term product 3 N^2; N = 11 / 27):
      298-bit  lazily reduced XYZZ madd (ec.hip.h madd_lz, madd-2008-s): 2 squares + 6 products + 1 fused two-term product = 2 189
      753-bit  madd-2007-bl: 4 squares + 7 products = 14 634"""
    n = 11 if curve < 2 else 27
    mul, sqr, dot2 = 2 * n * n, n * (n + 1) // 2 + n * n, 3 * n * n
    return 2 * sqr + 6 * mul + dot2 if curve < 2 else 4 * sqr + 7 * mul


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--dist", type=int, default=0, help="scalar distribution: 0 uniform (headline), 1 witness-like")
    ap.add_argument("--strong", action="store_true", help="headline = fixed TOTAL of 2^log-n pairs split over the ranks")
    ap.add_argument("--log-n", type=int, default=LOG_N, help="with --strong: log2 of the total pair count (20 or 22)")
    ap.add_argument("--no-step", action="store_true", help="skip the PCD-step sections (Groth16 main + help, 298- and 753-bit)")
    ap.add_argument("--no-753", action="store_true", help="skip the 753-bit PCD step (about two minutes of input generation and CPU checking)")
    ap.add_argument("--no-cpu", action="store_true", help="skip the CPU baseline leg")
    ap.add_argument("--no-strong", action="store_true", help="skip the fixed-total (strong scaling) section")
    ap.add_argument("--no-pipeline", action="store_true", help="headline = one MSM at a time (no second MSM in flight)")
    ap.add_argument("--no-merge", action="store_true", help="N > 1: skip the PCD-step section (merge-node proof over all devices + DAG branches)")
    ap.add_argument("--merge-log-n", type=int, default=20, help="N > 1: log2 of the merge node's domain (20; 22 = BASELINE configs[4], ~10 min of host work)")
    args = ap.parse_args()

    # stdout carries exactly ONE line, the JSON result: everything else that libraries print there while we run (RCCL's
    # version banner at communicator creation, for one) is sent to stderr by pointing fd 1 at fd 2 until the result is ready
    sys.stdout.flush()
    stdout_fd = os.dup(1)
    os.dup2(2, 1)

    import torch
    import torch.distributed as dist
    from pcd_amd import capi

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus > 1 and world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} needs WORLD_SIZE == {args.gpus} (launch with torch.distributed.run)")
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    use_dist = world > 1 or os.environ.get("PCD_BENCH_FORCE_DIST") == "1"   # the latter: 1-rank dry run of the RCCL path
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        dist.init_process_group("nccl", device_id=device)
    # a HOST-side barrier for the section in which rank 0 alone drives every device (an NCCL barrier would park a spinning kernel on the
    # waiting ranks' GPUs, which that section is busy measuring)
    host_group = dist.new_group(backend="gloo") if use_dist and world > 1 else None

    # ---- synthetic inputs (oracle helpers are test infrastructure: used here only to MAKE inputs and, below,
    # ---- as the CPU baseline / checker -- never inside the timed GPU region)
    from oracle import coracle as co
    n = 1 << LOG_N
    fr = co.CURVE_FR[CURVE]
    pts = co.gen_points(CURVE, GROUP, n, seed=SEED + rank)
    sc = co.gen_scalars(fr, n, seed=SEED + 1000 + rank, dist=args.dist)

    ctx = capi.Context(local_rank)

    def barrier():
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()
        ctx.sync()

    exchange = None
    if use_dist:
        from pcd_amd.dist import DeviceExchange
        exchange = DeviceExchange(ctx, CURVE, GROUP, device)   # partial -> RCCL all-gather -> EC sum, all on the device

    def timed_msm(bases, sbuf, steps, warmup, depth=1):
        """(wall seconds of `steps` MSMs, max over ranks; last result).  Profiling events are OFF inside the timed region.
        depth > 1 (single GPU): `depth` independent MSMs in flight through pcdhip_msm_submit / collect -- every result still returns
        to the host inside the timed region; the bucket reduction of one step overlaps the accumulation of the next."""
        ctx.msm_profile(False)
        step = (lambda: exchange.msm(bases, sbuf)) if use_dist else (lambda: ctx.msm(bases, sbuf))
        submit = (lambda: exchange.submit(bases, sbuf)) if use_dist else (lambda: ctx.msm_submit(bases, sbuf))
        collect = exchange.collect if use_dist else ctx.msm_collect
        res = None
        for _ in range(warmup):
            res = step()
        if depth > 1:   # the side streams' workspaces are allocated on first use: outside the timed region
            for t in [submit() for _ in range(depth)]:
                res = collect(t)
        barrier()
        t0 = time.perf_counter()
        if depth > 1:
            pending = []
            for _ in range(steps):
                pending.append(submit())
                if len(pending) >= depth:
                    res = collect(pending.pop(0))
            while pending:
                res = collect(pending.pop(0))
        else:
            for _ in range(steps):
                res = step()
        barrier()
        elapsed = time.perf_counter() - t0
        if use_dist:
            t = torch.tensor([elapsed], device=device, dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            elapsed = float(t.item())
        return elapsed, res

    def stage_times(bases, sbuf, reps=5):
        """per-stage device time (HIP events on the stream the kernels run on), mean over `reps` MSMs -- outside the timed region"""
        ctx.msm_profile(True)
        acc = []
        for _ in range(reps):
            ctx.msm(bases, sbuf)
            acc.append(ctx.msm_last_timings())
        ctx.msm_profile(False)
        return {k: float(np.mean([a[k] for a in acc])) for k in acc[0]}

    strong_cache = {}

    def strong_run(log_total, steps, warmup, curve=CURVE):
        """fixed TOTAL of 2^log_total pairs, rank r holding the point range [r n / N, (r + 1) n / N) of the key; one MSM at a time
        (the latency of ONE large MSM is what splitting it over the GPUs is for).  curve 2 = MNT4-753 G1: BASELINE configs[4]."""
        nonlocal exchange
        per = (1 << log_total) // world
        cfr = co.CURVE_FR[curve]
        if curve == CURVE and per <= n:
            p, s = pts[:per], sc[:per]
        else:
            big = strong_cache.get(curve)
            if big is None or big[0].shape[0] < per:   # (the 2^20 run takes a prefix of the 2^22 run's inputs)
                big = (co.gen_points_mt(curve, GROUP, per, seed=SEED + 77 + 1000 * curve + rank),
                       co.gen_scalars(cfr, per, seed=SEED + 1077 + 1000 * curve + rank, dist=args.dist))
                strong_cache[curve] = big
            p, s = big[0][:per], big[1][:per]
        b = ctx.bases_upload(curve, GROUP, p)
        sb = ctx.buf_upload(cfr, s)
        saved = exchange
        if use_dist and curve != CURVE:
            from pcd_amd.dist import DeviceExchange
            exchange = DeviceExchange(ctx, curve, GROUP, device)
        el, _ = timed_msm(b, sb, steps, warmup)
        exchange = saved
        plan = ctx.bases_info(b)
        b.free(); sb.free()
        return {"curve": "MNT4-298" if curve == 0 else "MNT4-753", "total_pairs": 1 << log_total, "pairs_per_gpu": per,
                "ms_per_step": round(el / steps * 1e3, 4), "value": round((per * world) * steps / el / 1e6, 3), "unit": "Mscalar-mul/s",
                "window_bits": plan[0], "windows": plan[1]}

    headline_strong = args.strong
    if headline_strong:
        per = (1 << args.log_n) // world
        if per > n:
            pts = co.gen_points(CURVE, GROUP, per, seed=SEED + rank)
            sc = co.gen_scalars(fr, per, seed=SEED + 1000 + rank, dist=args.dist)
        pts, sc, n_local = pts[:per], sc[:per], per
    else:
        n_local = n
    t0 = time.time()
    bases = ctx.bases_upload(CURVE, GROUP, pts)       # includes the one-time window-shifted precomputation
    upload_s = time.time() - t0
    sbuf = ctx.buf_upload(fr, sc)
    depth = 1 if args.no_pipeline else int(os.environ.get("PCD_BENCH_DEPTH", "4"))   # the same at every world size
    for _ in range(10):   # let the clocks settle before the first timed region (the W warm-up steps of each region follow)
        ctx.msm(bases, sbuf)
    elapsed_sync, res = timed_msm(bases, sbuf, args.steps, args.warmup)                 # one MSM at a time: the latency view
    elapsed, res = timed_msm(bases, sbuf, args.steps, args.warmup, depth) if depth > 1 else (elapsed_sync, res)
    stages = stage_times(bases, sbuf)
    c_bits, W, copies = ctx.bases_info(bases)

    # ---- correctness of what was timed (outside the timed region): rank-local partial vs the CPU oracle
    cpu = None
    if rank == 0 and not args.no_cpu:
        Wup = (298 + 14) // 15
        threads = max(1, min(os.cpu_count() or 1, Wup))   # upstream parallelises over windows only
        t0 = time.perf_counter()
        want = co.msm(CURVE, GROUP, pts, sc, nthreads=threads)
        cpu_s = time.perf_counter() - t0
        got = res if world == 1 else ctx.msm(bases, sbuf)
        if not np.array_equal(co.to_affine(CURVE, GROUP, got)[0], co.to_affine(CURVE, GROUP, want)[0]):
            raise SystemExit("GPU MSM result differs from the CPU oracle: refusing to report a number")
        cpu = {"value": round(n_local / cpu_s / 1e6, 4), "unit": "Mscalar-mul/s", "cores": threads, "kind": "port",
               "sample": f"one full MNT4-298 G1 MSM, n={n_local}, same inputs, C++ restatement of ark-ec Pippenger "
                         f"(threads over windows, c=15), {cpu_s:.2f} s; host has {os.cpu_count()} cores"}

    strong = None
    if not headline_strong and not args.no_strong:
        k = max(5, args.steps // 2)
        strong = {f"2^{lt}": strong_run(lt, k, 2) for lt in (20, 22)}
        if not args.no_753:   # BASELINE configs[4]: the merge node's G1 MSM, MNT4-753, fixed totals of 2^22 and 2^20 pairs over the ranks
            for lt in (22, 20):
                strong[f"753_2^{lt}"] = strong_run(lt, 5, 2, curve=2)
            strong_cache.clear()

    # ---- BASELINE's first metric at N > 1: the PCD step with the merge node's proof sharded over ALL devices, and N DAG branches one per
    # ---- device.  Rank 0 drives every device through ONE multi-device context (what a Rust host does: it has no process group); the other
    # ---- ranks free their device memory and wait on the host.  PCD_BENCH_DEVICES="0,0" exercises the same code on one GPU.
    multi_info = None
    dev_env = os.environ.get("PCD_BENCH_DEVICES")
    if (world > 1 or dev_env) and not args.no_merge and not args.no_753:
        bases.free(); sbuf.free()
        bases = sbuf = None
        ctx.sync()
        torch.cuda.empty_cache()
        if host_group is not None:
            dist.barrier(group=host_group)        # every rank has released its device
        if rank == 0:
            devices = [int(x) for x in dev_env.split(",")] if dev_env else list(range(world))
            multi_info = multi_device_step(co, devices, args.merge_log_n)
        if host_group is not None:
            dist.barrier(group=host_group)

    # ---- PCD step (prover arithmetic of main + help Groth16 proofs), N = 1 only
    step_info = step_753 = fft_info = pairing_info = None
    if rank == 0 and world == 1 and not args.no_step:
        if bases is not None:
            bases.free(); sbuf.free()
        fft_info = fft_section(ctx, co)
        pairing_info = pairing_section(ctx, co, (0,) if args.no_753 else (0, 2))
        step_info = pcd_step(ctx, co, (("main_mnt4_298", 0, (1 << 20) - 8), ("help_mnt6_298", 1, (1 << 16) - 8)), 32)
        if not args.no_753:
            step_753 = pcd_step(ctx, co, (("main_mnt4_753", 2, (1 << 20) - 8), ("help_mnt6_753", 3, (1 << 15) + 20000)), 64, roofline_curve=2)

    if rank == 0:
        ms_per_step = elapsed / args.steps * 1e3
        value = world * n_local * args.steps / elapsed / 1e6
        acc = stages["accumulate"]
        mm, mpm, bpp = CONTRACT[CURVE]
        ach_gbs = n_local * bpp / (acc * 1e-3) / 1e9
        traffic, traffic_note = traffic_from_profile()
        executed = n_local * W * madd_mads(CURVE)
        contract = n_local * mm * mpm
        out = {
            "metric": "msm_mscalar_mul_per_s", "value": round(value, 3), "unit": "Mscalar-mul/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms_per_step, 4), "higher_is_better": True,
            "scaling": "strong" if headline_strong else "weak", "vs_baseline": None,
            "dtype": "u32 (11 x 28-bit unsaturated Montgomery limbs, v_mad_u64_u32 with 64-bit column accumulators)",
            "data": "synthetic",
            "config": {"workload": (f"MNT4-298 G1 variable-base MSM, n={n_local} pairs per GPU ({'fixed total 2^%d split over the ranks' % args.log_n if headline_strong else '2^20 per GPU'}), "
                                    f"proving-key bases and scalars resident in HBM, scalar distribution {'uniform' if args.dist == 0 else 'witness-like'}"),
                       "curve": "MNT4-298", "group": "G1", "log_n": args.log_n if headline_strong else LOG_N, "sharding": f"point-range x{world}",
                       "precompute": f"{copies} window-shifted copies of the bases (one per scalar window; one-time, at key upload)",
                       "window_bits": c_bits, "windows": W, "upload_precompute_s": round(upload_s, 3)},
            "roofline": {"bound": "hbm", "achieved": round(ach_gbs, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(ach_gbs / HBM_PEAK_GBS, 5), "traffic": traffic,
                         "traffic_source": traffic_note, "kernel": "msm_accumulate_kernel", "kernel_ms": round(acc, 4),
                         "note": "algorithmic bytes = n x (40 B scalar + 80 B affine base); this kernel is integer-VALU-bound, "
                                 "not HBM-bound: see roofline_int"},
            "roofline_int": {"bound": "valu_int32_mad", "achieved": round(executed / (acc * 1e-3) / 1e12, 3),
                             "peak": round(MAD_PEAK / 1e12, 2), "unit": "T mad/s", "frac": round(executed / (acc * 1e-3) / MAD_PEAK, 4),
                             "executed_mads_per_pair": W * madd_mads(CURVE),
                             "note": f"EXECUTED multiply-adds of the plan that ran: n x W={W} mixed additions (signed digits, c={c_bits}) x "
                                     f"{madd_mads(CURVE)} mads per lazily reduced madd; peak = measured v_mad_u64_u32 issue rate "
                                     "(profiles/r01_k0_int_rates.txt).  This is the hardware fraction.",
                             "upstream_work_rate": {"value": round(contract / (acc * 1e-3) / MAD_PEAK, 4),
                                                    "note": "SURVEY.md 8d contract work (n x 220 modmul x 210 mads: upstream's c=15 / W=20, CIOS) per kernel "
                                                            "second over the same peak -- a speed in units of the upstream algorithm's work, "
                                                            "not a utilisation (signed digits and wider windows do less work per pair)"}},
            "pipelining": {"msms_in_flight": depth,
                           "note": "K independent MSMs, `msms_in_flight` submitted at a time at every world size (pcdhip_msm_submit / collect; with "
                                   "N > 1 pcdhip_msm_submit_partial and the RCCL exchange of earlier steps); every result is read back inside the "
                                   "timed region.  `one_at_a_time` is the same K steps with no overlap: the latency of one MSM -- the figure to "
                                   "set beside cpu_baseline (one MSM at a time as well) and `strong`",
                           "one_at_a_time": {"ms_per_step": round(elapsed_sync / args.steps * 1e3, 4),
                                             "value": round(world * n_local * args.steps / elapsed_sync / 1e6, 3)}},
            "msm_stage_ms": {k: round(float(v), 4) for k, v in stages.items()},
            "whole_step_upstream_work_rate": round(contract / (ms_per_step * 1e-3) / MAD_PEAK, 4),   # VERDICT r01's "whole step" figure: contract mads over the WHOLE step's time
            "whole_step_int_frac": round(executed / (ms_per_step * 1e-3) / MAD_PEAK, 4),   # per GPU: executed mads of the accumulate stage over the WHOLE step's time
            "cpu_baseline": cpu,
        }
        if strong:
            out["strong"] = strong
        if fft_info:
            out["fft"] = fft_info
        if pairing_info:
            out["pairing"] = pairing_info
        if step_info:
            out["pcd_step"] = step_info
        if step_753:
            out["pcd_step_753"] = step_753
        if multi_info:
            out["pcd_step_multi_device"] = multi_info
        sys.stdout.flush()
        os.dup2(stdout_fd, 1)
        print(json.dumps(out), flush=True)
        os.dup2(2, 1)
    if use_dist:
        dist.destroy_process_group()


def source_sha16():
    """identity of the kernel sources the accumulate kernel is made of (the PMC traffic figure is only valid for them)"""
    import hashlib
    h = hashlib.sha256()
    for f in ("msm.hip.h", "ec.hip.h", "fp.hip.h"):
        h.update(open(os.path.join(ROOT, "pcd_amd", "csrc", f), "rb").read())
    return h.hexdigest()[:16]


def traffic_from_profile():
    """HBM bytes per launch of the dominant kernel from the committed PMC passes (rocprofv3 cannot run inside this process).  The
    file records the hash of the kernel sources it was measured on: a figure taken on other sources is NOT reported."""
    tpath = os.path.join(ROOT, "profiles", "traffic_msm_accumulate.json")
    if not os.path.exists(tpath):
        return None, "profiles/traffic_msm_accumulate.json is missing"
    t = json.load(open(tpath))
    if t.get("source_sha16") != source_sha16():
        return None, (f"profiles/traffic_msm_accumulate.json was measured on kernel sources {t.get('source_sha16')}, this build is "
                      f"{source_sha16()}: stale, not reported (re-run tools/profile.sh + tools/traffic_json.py)")
    return t.get("hbm_bytes_per_launch"), "profiles/traffic_msm_accumulate.json (PMC FETCH_SIZE x2 + WRITE_SIZE, separate passes; same kernel sources)"


def fft_section(ctx, co, log_n=20):
    """Radix-2 transform passes at n = 2^20 over the two main scalar fields, resident vector, per pass: device ms (HIP events around
    each pass), achieved HBM GB/s = 2 n s / t (one read and one write of the vector; s = 44 / 108 B device image) and the fraction
    of the multiply-add peak its field products amount to (per element: d / 2 butterfly products of a radix-2^d pass + the
    inter-pass twiddle + the coset factor in the first pass; 2 N^2 mads each, N = 11 / 27)."""
    out = {"n": 1 << log_n, "transform": "coset_fft (three passes: 7 + 7 + 6 layers)", "hbm_peak_GBs": HBM_PEAK_GBS}
    n = 1 << log_n
    for fid, name, eb, N in ((1, "F298B (MNT4-298 Fr)", 44, 11), (3, "F753B (MNT4-753 Fr)", 108, 27)):
        x = ctx.buf_upload(fid, co.gen_field(fid, n, seed=SEED + 40 + fid))
        ctx.fft(fid, x)
        runs = []
        for _ in range(5):
            ctx.fft(fid, x, coset=True)
            runs.append(ctx.fft_last_timings())
        x.free()
        passes = [float(np.median([r[i] for r in runs])) for i in range(len(runs[0]))]
        layers = [log_n // len(passes) + (1 if i < log_n % len(passes) else 0) for i in range(len(passes))]
        prods = [d / 2 + 1 + (1 if i == 0 else 0) for i, d in enumerate(layers)]
        out[name] = {"pass_ms": [round(p, 4) for p in passes], "transform_ms": round(sum(passes), 4),
                     "pass_GBs": [round(2 * n * eb / (p * 1e-3) / 1e9, 1) for p in passes],
                     "pass_hbm_frac": [round(2 * n * eb / (p * 1e-3) / 1e9 / HBM_PEAK_GBS, 4) for p in passes],
                     "pass_mad_frac": [round(n * k * 2 * N * N / (p * 1e-3) / MAD_PEAK, 3) for p, k in zip(passes, prods)]}
    return out


def pairing_section(ctx, co, curves):
    """K6: Groth16 verification (reference call site src/ec_cycle_pcd/mod.rs:239) through process_vk + the prepared verification:
    latency of ONE verification and of a batch of 8 (the prior messages of an arity-8 merge node), wall clock around the C-ABI
    call (host buffers in, answers out), median of 5; beside it the CPU oracle's verify on one core and 8 proofs on 8 cores."""
    from concurrent.futures import ThreadPoolExecutor
    out = {"unit": "ms", "timing": "median of 5 calls, wall clock around the C-ABI call"}
    for cid in curves:
        fr = co.CURVE_FR[cid]
        r = co.synthetic_r1cs(fr, 60, 3, seed=SEED + 50 + cid)
        keys = co.groth16_setup(cid, r, co.gen_field(fr, 5, seed=SEED + 51), nthreads=16)
        pk = ctx.g16_pk_upload(keys.host_struct(), cid)
        proofs = []
        for i in range(8):
            rs = co.gen_field(fr, 2, seed=SEED + 60 + i)
            proofs.append(ctx.groth16_prove(pk, r, rs[0], rs[1])[0])
        pk.free()
        pub_m = np.ascontiguousarray(r.z[1:r.num_inputs])
        pub = co.fp_op(fr, "to_canonical", pub_m)
        pubs, proofs = np.stack([pub] * 8), np.stack(proofs)
        t0 = time.perf_counter()
        pvk = ctx.process_vk(cid, keys.alpha_g1, keys.beta_g2, keys.gamma_g2, keys.delta_g2, keys.gamma_abc_g1)
        pvk_ms = (time.perf_counter() - t0) * 1e3
        rho = np.random.default_rng(5).integers(1, 1 << 62, size=(8, 2), dtype=np.uint64)

        def med(fn):
            fn()
            ts = []
            for _ in range(5):
                t0 = time.perf_counter()
                ok = fn()
                ts.append((time.perf_counter() - t0) * 1e3)
            if not np.all(ok):
                raise SystemExit("GPU Groth16 verification rejected a valid proof")
            return float(np.median(ts))
        one = med(lambda: ctx.groth16_verify_prepared(pvk, pubs[:1], proofs[:1]))
        eight = med(lambda: ctx.groth16_verify_prepared(pvk, pubs, proofs))
        rlc = med(lambda: ctx.groth16_verify_batch_rlc(pvk, pubs, proofs, rho))
        bad = pubs.copy(); bad[3, 0, 0] ^= 1
        if ctx.groth16_verify_prepared(pvk, bad, proofs)[3] or ctx.groth16_verify_batch_rlc(pvk, bad, proofs, rho):
            raise SystemExit("GPU Groth16 verification accepted a wrong public input")
        pvk.free()
        cv = lambda i: co.groth16_verify(keys, pub_m, proofs[i])
        ts = []
        for _ in range(3):
            t0 = time.perf_counter(); ok1 = cv(0); ts.append((time.perf_counter() - t0) * 1e3)
        t0 = time.perf_counter()
        with ThreadPoolExecutor(max_workers=8) as ex:
            ok8 = list(ex.map(cv, range(8)))
        cpu8 = (time.perf_counter() - t0) * 1e3
        if not (ok1 and all(ok8)):
            raise SystemExit("CPU oracle rejected a GPU-made proof")
        out[co.CURVE_NAMES[cid]] = {
            "verify_1_ms": round(one, 3), "verify_batch8_ms": round(eight, 3), "verify_batch8_shared_final_exp_ms": round(rlc, 3),
            "process_vk_ms": round(pvk_ms, 2),
            "cpu_baseline": {"verify_1_ms_1_core": round(float(np.median(ts)), 3), "verify_batch8_ms_8_cores": round(cpu8, 3), "kind": "port",
                             "sample": "the oracle's Groth16 verify (3 Miller loops with e(alpha, beta) recomputed: 4 pairings' worth + 1 final "
                                       "exponentiation) on the same proofs; 8 proofs on 8 threads, one each"},
            "gpu_over_cpu8_batch8": round(cpu8 / eight, 2)}
    return out


def multi_device_step(co, devices, log_n):
    """BASELINE's "PCD-step prover ms ... 1/2/4/8 GPU": one step of an arity-N merge node over MNT4-753 / MNT6-753 (configs[4] shape).
      * the MAIN proof (MNT4-753, domain 2^log_n) through ONE multi-device context over `devices` (pcdhip_init_devices: every query
        sharded by point range, five MSMs per device, the witness map's a / b / c chains on the first three devices, partial sums on
        device 0) -- next to the same proof on devices[0] alone;
      * the HELP proof (MNT6-753, mixed-radix domain 5 * 2^14) on devices[0];
      * len(devices) independent DAG branches (main proof each), one per device: threads x contexts, no exchange.
    The sharded proof must be BIT-IDENTICAL to the single-device one (which tests/ and the N = 1 line pin to the CPU oracle); otherwise
    no number is printed.  Wall clock around the C-ABI call, median of 5."""
    from pcd_amd import capi, dag
    G = len(devices)
    info = {"devices": devices, "unit": "ms", "timing": "median of 5 proves after 3 warm-up proves, wall clock around pcdhip_groth16_prove",
            "check": "sharded proof and every branch's proof == the single-device proof, byte for byte (the single-device path is what tests/ and "
                     "the N = 1 line compare with the CPU oracle; the keys are seeded on-curve points, not a consistent SRS, so there is nothing to verify)"}
    curve, hcurve = 2, 3
    fr, hfr = co.CURVE_FR[curve], co.CURVE_FR[hcurve]
    t0 = time.time()
    r = co.skewed_r1cs(fr, (1 << log_n) - 8, 2, seed=SEED + 200)
    keys = co.synthetic_keys(curve, r, seed=SEED + 201)
    rs = co.gen_field(fr, 2, seed=SEED + 202)
    hr = co.skewed_r1cs(hfr, (1 << 15) + 20000, 2, seed=SEED + 203)
    hkeys = co.synthetic_keys(hcurve, hr, seed=SEED + 204)
    hrs = co.gen_field(hfr, 2, seed=SEED + 205)
    info["input_gen_s"] = round(time.time() - t0, 1)
    r.z = capi.pinned_like(r.z)

    def timed(c, pk, rr, rss, reps=5):
        for _ in range(3):   # (FFT tables, workspaces, and the clocks after half a minute of host-side input generation)
            c.groth16_prove(pk, rr, rss[0], rss[1], resident_r1cs=True)
        walls, proof = [], None
        for _ in range(reps):
            t = time.perf_counter()
            proof, _ = c.groth16_prove(pk, rr, rss[0], rss[1], resident_r1cs=True)
            walls.append((time.perf_counter() - t) * 1e3)
        return float(np.median(walls)), proof, c.groth16_last_timings()

    # one device: the reference proof bytes, the standalone witness map and the help proof
    one = capi.Context(devices[0])
    # at 2^22 the full set of window-shifted copies of one MNT4-753 key is ~195 GB: cap every vector (fewer copies, Horner combine back)
    budget = int(os.environ.get("PCD_BENCH_COPY_BUDGET_GB", "24" if log_n >= 22 else "0")) << 30
    one.set_precompute_budget(budget)
    pk = one.g16_pk_upload(keys.host_struct(), curve)
    one.g16_pk_set_r1cs(pk, r)
    one_ms, proof_one, one_tm = timed(one, pk, r, rs)
    wm = [one.witness_map_resident(pk, r, want_h=False)[1] for _ in range(3)][1:]
    info["witness_map_alone_ms"] = {k: round(float(np.median([w[k] for w in wm])), 3) for k in wm[0]}
    pk.free()
    hpk = one.g16_pk_upload(hkeys.host_struct(), hcurve)
    one.g16_pk_set_r1cs(hpk, hr)
    help_ms, help_proof, _ = timed(one, hpk, hr, hrs)
    hpk.free()
    one.close()

    multi = capi.Context(devices=devices)
    multi.set_precompute_budget(budget)
    t0 = time.time()
    mpk = multi.g16_pk_upload(keys.host_struct(), curve)
    multi.g16_pk_set_r1cs(mpk, r)
    info["sharded_key_upload_s"] = round(time.time() - t0, 2)
    all_ms, proof_all, all_tm = timed(multi, mpk, r, rs)
    mpk.free()
    if not np.array_equal(proof_all, proof_one):
        raise SystemExit("sharded Groth16 proof differs from the single-device proof: refusing to report a number")
    multi.close()
    info["main_mnt4_753"] = {"domain": int(keys.domain_size), "one_device_ms": round(one_ms, 2), "all_devices_ms": round(all_ms, 2),
                             "speedup": round(one_ms / all_ms, 2),
                             "device0_critical_path_ms": {"witness_map_until_h": round(float(all_tm["witness_map"]), 3), "total": round(float(all_tm["total"]), 3)},
                             "one_device_stage_ms": {k: round(float(v), 3) for k, v in one_tm.items()}}
    info["help_mnt6_753"] = {"domain": int(hkeys.domain_size), "one_device_ms": round(help_ms, 2)}
    info["pcd_step_ms"] = round(all_ms + help_ms, 2)
    info["pcd_step_ms_one_device"] = round(one_ms + help_ms, 2)

    # independent DAG branches: one main proof per device, each thread uploads its own key (as a real branch would hold its own step's key)
    def branch(c):
        c.set_precompute_budget(budget)
        bpk = c.g16_pk_upload(keys.host_struct(), curve)
        c.g16_pk_set_r1cs(bpk, r)
        ms, proof, _ = timed(c, bpk, r, rs)
        bpk.free()
        if not np.array_equal(proof, proof_one):
            raise SystemExit("a DAG branch's proof differs from the single-device proof")
        return ms
    t0 = time.perf_counter()
    per = dag.run_branches([branch] * G, devices)
    info["dag_branches"] = {"n": G, "ms_each": [round(v, 2) for v in per], "wall_s_including_key_upload": round(time.perf_counter() - t0, 2),
                            "proofs_per_s": round(G / (max(per) * 1e-3), 3)}
    return info


def median_prove(ctx, pk, r, rs, reps=5):
    """(median wall ms of `reps` proves, proof, device timings of the median-adjacent last run)"""
    walls = []
    proof = None
    for _ in range(reps):
        t0 = time.perf_counter()
        proof, inf = ctx.groth16_prove(pk, r, rs[0], rs[1], resident_r1cs=True)
        walls.append((time.perf_counter() - t0) * 1e3)
    return float(np.median(walls)), proof, ctx.groth16_last_timings(), walls


def pcd_step(ctx, co, proofs, max_threads, roofline_curve=None):
    """Prover arithmetic of one PCD step: main proof + help proof (the help scalar fields have 2-adicity 17 / 15: radix-2 up
    to 2^16 rows at 298 bits, the mixed-radix domain 5 * 2^14 for the 753-bit help circuit); the assignment is uniformly random
    field elements -- the worst case for the MSMs (a real witness is full of 0 / 1 values, which cost nothing / go to the pseudo
    bucket: `--dist 1`); keys and matrices resident, z in page-locked host memory; every timing is the median of 5 proves;
    proof bytes equal to the oracle's or no number is printed."""
    from pcd_amd import capi
    info = {"unit": "ms", "timing": "median of 5 proves per assembly form",
            "what": "witness map + the proof's MSMs (h, l, A, B1 on G1; B on G2) + assembly (s*A, r*B1 chained behind their MSMs or folded "
                    "into two more MSMs, chosen by size), per proof; the MSMs over the assignment overlap the witness map; R1CS synthesis "
                    "(Rust host) excluded.  Keys: seeded points, the a / b queries of variables absent from A / B are the point at infinity "
                    "as a real setup makes them (`query_infinity_frac`; the CPU port times the same key)"}
    total_gpu, total_cpu = 0.0, 0.0
    for name, curve, nc in proofs:
        fr = co.CURVE_FR[curve]
        t0 = time.time()
        # the constraint matrices have the shape `cs.finalize()` leaves of a verifier circuit: power-law row lengths (a few rows above 4096
        # entries), >= 80 % unit coefficients (coracle.skewed_r1cs); the assignment stays uniformly random field elements
        r = co.skewed_r1cs(fr, nc, 2, seed=SEED + curve)
        # the key: seeded points, with the points at infinity a real setup leaves in the a / b queries (variables that no row of A / B
        # mentions: a_i(tau) G = O); PCD_BENCH_DENSE_KEYS=1 makes every entry finite instead (rounds 1-3 measured that: the densest key)
        keys = co.synthetic_keys(curve, r, seed=SEED + 10 + curve, consistent=os.environ.get("PCD_BENCH_DENSE_KEYS") != "1")
        rs = co.gen_field(fr, 2, seed=SEED + 20)
        gen_s = time.time() - t0
        t0 = time.time()
        pk = ctx.g16_pk_upload(keys.host_struct(), curve)
        ctx.g16_pk_set_r1cs(pk, r)                              # matrices are fixed per circuit: resident like the key
        up_s = time.time() - t0
        r.z = capi.pinned_like(r.z)                             # the assignment is handed over in page-locked host memory
        ctx.groth16_prove(pk, r, rs[0], rs[1], resident_r1cs=True)   # warm-up (FFT tables, workspaces)
        wall, proof, tm, walls = median_prove(ctx, pk, r, rs)
        wm = [ctx.witness_map_resident(pk, r, want_h=False)[1] for _ in range(4)][1:]   # the witness map ALONE (nothing else on the device)
        wm = {k: round(float(np.median([w[k] for w in wm])), 3) for k in wm[0]}
        forms = {}
        for mode, label in ((1, "folded"), (2, "chained")):    # the two explicit assembly forms, for the record (the default picks one)
            ctx.groth16_set_assembly(mode)
            ctx.groth16_prove(pk, r, rs[0], rs[1], resident_r1cs=True)
            forms[label] = median_prove(ctx, pk, r, rs)
        ctx.groth16_set_assembly(0)
        threads = min(os.cpu_count() or 1, max_threads)
        t0 = time.perf_counter()
        want, _ = co.groth16_prove(keys, r, rs[0], rs[1], nthreads=threads)
        cpu_ms = (time.perf_counter() - t0) * 1e3
        if not all(np.array_equal(p, want) for p in (proof, forms["folded"][1], forms["chained"][1])):
            raise SystemExit(f"GPU Groth16 proof ({name}) differs from the CPU oracle")
        info[name] = {"gpu_wall_ms": round(wall, 2), "gpu_wall_ms_min_max": [round(min(walls), 2), round(max(walls), 2)],
                      "gpu_wall_ms_folded_assembly": round(forms["folded"][0], 2),
                      "gpu_wall_ms_chained_assembly": round(forms["chained"][0], 2),
                      "gpu_device_ms": {k: round(float(v), 3) for k, v in tm.items()},
                      "witness_map_alone_ms": wm,   # standalone: inside a prove it shares the device with four MSMs (gpu_device_ms.witness_map)
                      "r1cs_entries": [int(len(r.col_a)), int(len(r.col_b)), int(len(r.col_c))],
                      # fraction of the a / b query entries that are the point at infinity (as a setup over this R1CS makes them)
                      "query_infinity_frac": {"a": round(float(np.mean(keys.a_inf)), 4), "b": round(float(np.mean(keys.b_g2_inf)), 4)},
                      "cpu_port_ms": round(cpu_ms, 1), "cpu_threads": threads, "domain": int(keys.domain_size),
                      "key_upload_precompute_s": round(up_s, 2), "input_gen_s": round(gen_s, 2)}
        total_gpu += wall
        total_cpu += cpu_ms
        if roofline_curve is None and curve in (0, 1):
            # the G2 accumulation of the 298-bit step (about 40 % of the main proof's work): a standalone MSM over the key's own b_g2 query,
            # stage events on.  Executed multiply-adds per mixed addition (N = 11): MNT4-298 -- Fq2 over lane pairs, XYZZ with lazily reduced
            # internals (ec.hip.h madd_x_lz2): 56 N^2;  MNT6-298 -- Fq3 over lane triples, XYZZ (8 products of 12 N^2 + 2 squares of 9 N^2): 114 N^2
            n2 = min(1 << 20, int(keys.b_g2_query.shape[0]))
            g2_mads = (56 if curve == 0 else 114) * 11 * 11
            b = ctx.bases_upload(curve, 2, np.ascontiguousarray(keys.b_g2_query[:n2]))
            sb = ctx.buf_upload(fr, co.gen_scalars(fr, n2, seed=SEED + 31))
            ctx.msm_profile(True)
            accs2, tots2 = [], []
            for _ in range(4):
                ctx.msm(b, sb)
                t = ctx.msm_last_timings()
                accs2.append(t["accumulate"]); tots2.append(t["total"])
            ctx.msm_profile(False)
            c2, W2, _ = ctx.bases_info(b)
            b.free(); sb.free()
            acc2 = float(np.median(accs2[1:]))
            info.setdefault("roofline_int_g2", {})[name] = {
                "kernel": "msm_accumulate_kernel (G2 over %s, lane-split)" % ("Fq2, MNT4-298" if curve == 0 else "Fq3, MNT6-298"), "n": n2,
                "kernel_ms": round(acc2, 3), "msm_total_ms": round(float(np.median(tots2[1:])), 3), "bound": "valu_int32_mad",
                "achieved": round(n2 * W2 * g2_mads / (acc2 * 1e-3) / 1e12, 3), "peak": round(MAD_PEAK / 1e12, 2), "unit": "T mad/s",
                "frac": round(n2 * W2 * g2_mads / (acc2 * 1e-3) / MAD_PEAK, 4), "window_bits": c2, "windows": W2, "executed_mads_per_pair": W2 * g2_mads}
        if roofline_curve == curve:
            # the dominant kernel of the step: G1 bucket accumulation of the main proof (four of its five MSMs); measured on one
            # standalone MSM over the key's own h query with the stage events on
            pk.free()
            pk = None
            n = 1 << 20
            hq = np.ascontiguousarray(keys.h_query[:n - 1])
            b = ctx.bases_upload(curve, 1, hq)
            sb = ctx.buf_upload(fr, co.gen_scalars(fr, n - 1, seed=SEED + 30))
            ctx.msm_profile(True)
            accs, tots = [], []
            for _ in range(4):
                ctx.msm(b, sb)
                t = ctx.msm_last_timings()
                accs.append(t["accumulate"]); tots.append(t["total"])
            ctx.msm_profile(False)
            c_bits, W, copies = ctx.bases_info(b)
            b.free(); sb.free()
            acc = float(np.median(accs[1:]))
            executed = (n - 1) * W * madd_mads(curve)
            mm, mpm, bpp = CONTRACT[curve]
            info["roofline_int"] = {"kernel": "msm_accumulate_kernel (G1, MNT4-753)", "kernel_ms": round(acc, 3), "msm_total_ms": round(float(np.median(tots[1:])), 3),
                                    "bound": "valu_int32_mad", "achieved": round(executed / (acc * 1e-3) / 1e12, 3), "peak": round(MAD_PEAK / 1e12, 2),
                                    "unit": "T mad/s", "frac": round(executed / (acc * 1e-3) / MAD_PEAK, 4), "window_bits": c_bits, "windows": W,
                                    "executed_mads_per_pair": W * madd_mads(curve),
                                    "upstream_work_rate": round((n - 1) * mm * mpm / (acc * 1e-3) / MAD_PEAK, 4),
                                    "hbm_algorithmic_GBs": round((n - 1) * bpp / (acc * 1e-3) / 1e9, 2)}
            # the G2 MSM of the same proof (Fq2 twist, points split over lane pairs, XYZZ running sum): 8 products + 2 squares per mixed
            # addition, each ONE fused two-term product of 3 N^2 mads in each of the two lanes -> 60 N^2 executed mads; standalone MSM
            # over the first 2^18 points of the key's own b_g2 query
            n2 = 1 << 18
            b = ctx.bases_upload(curve, 2, np.ascontiguousarray(keys.b_g2_query[:n2]))
            sb = ctx.buf_upload(fr, co.gen_scalars(fr, n2, seed=SEED + 31))
            ctx.msm_profile(True)
            accs2, tots2 = [], []
            for _ in range(4):
                ctx.msm(b, sb)
                t = ctx.msm_last_timings()
                accs2.append(t["accumulate"]); tots2.append(t["total"])
            ctx.msm_profile(False)
            c2, W2, _ = ctx.bases_info(b)
            b.free(); sb.free()
            acc2 = float(np.median(accs2[1:]))
            g2_mads = 60 * 27 * 27
            info["roofline_int_g2"] = {"kernel": "msm_accumulate_kernel (G2 over Fq2, MNT4-753, lane-split)", "n": n2, "kernel_ms": round(acc2, 3),
                                       "msm_total_ms": round(float(np.median(tots2[1:])), 3), "bound": "valu_int32_mad",
                                       "achieved": round(n2 * W2 * g2_mads / (acc2 * 1e-3) / 1e12, 3), "peak": round(MAD_PEAK / 1e12, 2), "unit": "T mad/s",
                                       "frac": round(n2 * W2 * g2_mads / (acc2 * 1e-3) / MAD_PEAK, 4), "window_bits": c2, "windows": W2,
                                       "executed_mads_per_pair": W2 * g2_mads}
            # the whole main proof against the same peak: executed multiply-adds of its five accumulations (h: n - 1 pairs; l', A, B1 on G1 and
            # B on G2: m + 4 pairs each, W windows of the 2^20 plan) + the 7 transforms (3 passes each, ~5 products per element and pass),
            # over the proof's wall time -- fix-up, bucket reduction, sorts and the assembly count as zero work
            # (pairs whose base is the point at infinity are not executed: l' has none, A is in the chained form's shared list whole,
            #  B1 and B leave theirs out)
            m4 = int(keys.a_query.shape[0]) + 4
            fin_b = 1.0 - float(np.mean(keys.b_g2_inf))
            # (the key's queries run one window bit narrower than the standalone plan measured above -- pcdhip_g16_pk_upload -- hence more windows)
            Wk = -(-754 // (c_bits - 1)) if m4 >= (1 << 18) else W
            work = int(((n - 1) + (2 + fin_b) * m4) * Wk * madd_mads(curve) + fin_b * m4 * Wk * g2_mads + 7 * 3 * n * 5 * 2 * 27 * 27)
            info["whole_step_int_frac"] = {"main_mnt4_753": round(work / (info[name]["gpu_wall_ms"] * 1e-3) / MAD_PEAK, 4),
                                           "executed_mads": work, "note": "accumulations + transforms only; everything else counted as zero work"}
        if pk is not None:
            pk.free()
        del keys, r
    info["pcd_step_prover_ms"] = round(total_gpu, 2)
    info["cpu_port_ms"] = round(total_cpu, 1)
    info["speedup_vs_cpu_port"] = round(total_cpu / total_gpu, 2)
    return info


if __name__ == "__main__":
    main()
