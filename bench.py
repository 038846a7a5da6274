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


# executed multiply-adds of one mixed addition of the G2 accumulate kernels (N = 11 / 27 limbs): MNT4-298 Fq2 XYZZ with lazily reduced
# internals 56 N^2; MNT6-298 Fq3 over lane triples 114 N^2; MNT4-753 Fq2 over lane pairs 60 N^2; MNT6-753 Fq3 over lane triples 120 N^2
G2_MADS = {0: 56 * 121, 1: 114 * 121, 2: 60 * 729, 3: 120 * 729}


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
    ap.add_argument("--no-kzg", action="store_true", help="skip the KZG multi-MSM section (BASELINE configs[3])")
    ap.add_argument("--no-variants", action="store_true", help="PCD-step sections: only the uniform assignment under the consistent key (skip the dense key and the witness-like assignment)")
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

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    multi = world > 1 or os.environ.get("PCD_BENCH_FORCE_DIST") == "1"
    if not multi:
        return run(args, stdout_fd)
    # ---- N > 1: first contact with a multi-GPU node must not end without a line.  Every init / barrier is bounded (the
    # ---- process group's timeout, a watchdog thread); whatever fails -- rendezvous, RCCL, a peer copy, a mismatch -- rank 0 runs the N = 1
    # ---- bench in a FRESH child process on its own GPU (never an exec from a process that has touched the GPU), prints that line with the
    # ---- error attached, and exits non-zero; the other ranks just exit non-zero.
    import threading
    limit = float(os.environ.get("PCD_BENCH_DIST_TIMEOUT_S", "1500"))
    done = threading.Event()

    def bail(reason, code=3):
        if done.is_set():
            return
        done.set()
        print(f"[bench rank {rank}] multi-GPU run failed: {reason}", file=sys.stderr, flush=True)
        if rank == 0:
            line = single_gpu_fallback(args, reason, world)
            os.dup2(stdout_fd, 1)
            print(line, flush=True)
        os._exit(code)

    def watchdog():
        if not done.wait(limit + (0 if rank == 0 else 120)):   # (the other ranks leave rank 0 the time to print its line)
            bail(f"no result after {limit:.0f} s (PCD_BENCH_DIST_TIMEOUT_S): a hung rendezvous, collective or peer copy")
    threading.Thread(target=watchdog, daemon=True).start()
    try:
        rc = run(args, stdout_fd)
    except BaseException as e:   # noqa: BLE001 -- SystemExit from a failed check included: the line must still come out
        import traceback
        traceback.print_exc()
        bail(f"{type(e).__name__}: {e}")
        raise
    done.set()
    if rc:
        os._exit(rc)


def single_gpu_fallback(args, reason, world):
    """the N = 1 MSM figures from a fresh child process on this rank's GPU, as one JSON line with the failure attached"""
    import subprocess
    env = {k: v for k, v in os.environ.items()
           if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "LOCAL_WORLD_SIZE", "GROUP_RANK", "ROLE_RANK", "MASTER_ADDR", "MASTER_PORT",
                        "PCD_BENCH_FORCE_DIST", "PCD_BENCH_INJECT_FAILURE", "PCD_BENCH_DEVICES") and not k.startswith("TORCHELASTIC")}
    env.setdefault("HIP_VISIBLE_DEVICES", os.environ.get("LOCAL_RANK", "0"))
    cmd = [sys.executable, os.path.abspath(__file__), "--gpus", "1", "--steps", str(args.steps), "--warmup", str(args.warmup), "--no-step", "--no-strong"]
    out = {"metric": "msm_mscalar_mul_per_s", "value": None, "unit": "Mscalar-mul/s", "n_gpus": 1, "steps": args.steps, "warmup": args.warmup,
           "higher_is_better": True}
    try:
        p = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
        lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
        if p.returncode == 0 and lines:
            out = json.loads(lines[-1])
        else:
            out["fallback_error"] = f"child exited {p.returncode}: {p.stderr[-400:]}"
    except Exception as e:   # noqa: BLE001
        out["fallback_error"] = f"{type(e).__name__}: {e}"
    out["n_gpus_requested"] = world
    out["multi_gpu_error"] = reason
    out["note_fallback"] = "the N > 1 run failed; these are the N = 1 figures of a fresh child process on rank 0's GPU (exit code non-zero)"
    return json.dumps(out)


def preflight(torch, dist, device, world, rank):
    """what a first run on a multi-GPU node should say before anything is timed: which devices can reach which (the library's multi-device
    context copies partial results device to device: hipMemcpyPeerAsync, xGMI where peer access exists) and what a 2-KB RCCL all-gather
    costs (the MSM's only exchange is one Jacobian point per rank: latency, not bandwidth)"""
    info = {}
    ndev = torch.cuda.device_count()
    info["visible_devices"] = ndev
    info["peer_access"] = [[bool(i == j or torch.cuda.can_device_access_peer(i, j)) for j in range(ndev)] for i in range(ndev)]
    send = torch.zeros(256, dtype=torch.int64, device=device)
    recv = torch.zeros(256 * world, dtype=torch.int64, device=device)
    for _ in range(5):
        dist.all_gather_into_tensor(recv, send)
    torch.cuda.synchronize()
    ts = []
    for _ in range(20):
        t0 = time.perf_counter()
        dist.all_gather_into_tensor(recv, send)
        torch.cuda.synchronize()
        ts.append((time.perf_counter() - t0) * 1e6)
    info["allgather_2KB_us"] = {"median": round(float(np.median(ts)), 1), "min": round(float(min(ts)), 1), "max": round(float(max(ts)), 1)}
    if rank == 0:
        print(f"[bench preflight] {json.dumps(info)}", file=sys.stderr, flush=True)
    return info


def run(args, stdout_fd):
    import datetime
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
    inject = os.environ.get("PCD_BENCH_INJECT_FAILURE", "")                 # dry runs of the failure paths: "init", "exchange", "merge"
    pre_info = None
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        if inject == "init":
            raise RuntimeError("injected failure before the process group exists (PCD_BENCH_INJECT_FAILURE=init)")
        # (every collective and barrier below is bounded by this timeout; the watchdog in main() bounds the rest)
        dist.init_process_group("nccl", device_id=device, timeout=datetime.timedelta(seconds=float(os.environ.get("PCD_BENCH_COLLECTIVE_TIMEOUT_S", "300"))))
        pre_info = preflight(torch, dist, device, world, rank)
    # a HOST-side barrier for the section in which rank 0 alone drives every device (an NCCL barrier would park a spinning kernel on the
    # waiting ranks' GPUs, which that section is busy measuring)
    host_group = dist.new_group(backend="gloo", timeout=datetime.timedelta(seconds=3600)) if use_dist and world > 1 else None

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
        if inject == "exchange":
            raise RuntimeError("injected failure of the RCCL exchange (PCD_BENCH_INJECT_FAILURE=exchange)")

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
    # the same MSM with the scalars coming from the HOST on every call (pcdhip_msm: 42 MB over PCIe inside the call) -- what a caller pays whose
    # scalars are not made on the device (`h` is; the assignment z crosses once per proof) -- pinned and pageable memory, median of 7 after a warm-up
    host_scalars = None
    if world == 1 and not headline_strong:
        from pcd_amd import capi as _capi
        host_scalars = {}
        for kind, arr in (("pinned", _capi.pinned_like(np.ascontiguousarray(sc))), ("pageable", np.ascontiguousarray(sc))):
            ctx.msm(bases, arr)
            ts = []
            for _ in range(7):
                t0 = time.perf_counter(); r_host = ctx.msm(bases, arr); ts.append((time.perf_counter() - t0) * 1e3)
            if not np.array_equal(co.to_affine(CURVE, GROUP, r_host)[0], co.to_affine(CURVE, GROUP, res)[0]):   # (same point; the Jacobian representative may differ)
                raise SystemExit("pcdhip_msm (host scalars) and pcdhip_msm_dev (resident scalars) disagree")
            host_scalars[kind] = float(np.median(ts))
    mad_peak_live = ctx.mad_rate()   # the v_mad_u64_u32 issue rate of THIS box, right behind the kernels it prices (boxes differ by ~10 % in sustained clock)
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
    exit_code = 0
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
            try:   # (a failure here -- peer copies between real devices have their first contact in this section -- must not cost the MSM figures)
                if inject == "merge":
                    raise RuntimeError("injected failure of the multi-device section (PCD_BENCH_INJECT_FAILURE=merge)")
                multi_info = multi_device_step(co, devices, args.merge_log_n)
            except BaseException as e:   # noqa: BLE001
                import traceback
                traceback.print_exc()
                multi_info = {"error": f"{type(e).__name__}: {e}", "devices": devices}
                exit_code = 4
        if host_group is not None:
            dist.barrier(group=host_group)

    # ---- PCD step (prover arithmetic of main + help Groth16 proofs), N = 1 only
    step_info = step_753 = fft_info = pairing_info = kzg_info = None
    if rank == 0 and world == 1 and not args.no_step:
        if bases is not None:
            bases.free(); sbuf.free()
        fft_info = fft_section(ctx, co)
        pairing_info = pairing_section(ctx, co, (0,) if args.no_753 else (0, 2))
        kzg_info = None if args.no_kzg else kzg_section(ctx, co)
        step_info = pcd_step(ctx, co, (("main_mnt4_298", 0, (1 << 20) - 8), ("help_mnt6_298", 1, (1 << 16) - 8)), 32, variants=not args.no_variants)
        if not args.no_753:
            step_753 = pcd_step(ctx, co, (("main_mnt4_753", 2, (1 << 20) - 8), ("help_mnt6_753", 3, (1 << 15) + 20000)), 64, roofline_curve=2,
                                variants=not args.no_variants)

    if rank == 0:
        # the headline is SURVEY.md 8d's definition: n / t of ONE MSM at a time (what cpu_baseline times as well); the same K steps with
        # `msms_in_flight` of them submitted at a time are reported under `pipelining`
        ms_per_step = elapsed_sync / args.steps * 1e3
        value = world * n_local * args.steps / elapsed_sync / 1e6
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
            # the BINDING roof of the dominant kernel (integer VALU issue, not HBM and not MFMA); the HBM view is nested under "hbm"
            "roofline": {"bound": "valu_int32_mad", "achieved": round(executed / (acc * 1e-3) / 1e12, 3),
                             "peak": round(MAD_PEAK / 1e12, 2), "unit": "T mad/s", "frac": round(executed / (acc * 1e-3) / MAD_PEAK, 4),
                             "kernel": "msm_accumulate_kernel", "kernel_ms": round(acc, 4),
                             "traffic": traffic, "traffic_source": traffic_note,
                             "traffic_over_algorithmic_bytes": round(traffic / (n_local * bpp), 2) if traffic else None,
                             "hbm": {"bound": "hbm", "achieved": round(ach_gbs, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(ach_gbs / HBM_PEAK_GBS, 5),
                                     "algorithmic_bytes": n_local * bpp,
                                     "note": "secondary, non-binding: algorithmic bytes = n x (40 B scalar + 80 B affine base) over the same kernel time"},
                             "peak_live": round(mad_peak_live / 1e12, 2), "frac_live": round(executed / (acc * 1e-3) / mad_peak_live, 4),
                             "peak_live_note": "pcdhip_mad_rate: the same issue-rate microbenchmark as `peak` (four waves per SIMD, eight chains per lane), run on this "
                                               "box right after the stage timings; `peak` is the round-1 constant every other fraction in this line uses",
                             "executed_mads_per_pair": W * madd_mads(CURVE),
                             "note": f"EXECUTED multiply-adds of the plan that ran: n x W={W} mixed additions (signed digits, c={c_bits}) x "
                                     f"{madd_mads(CURVE)} mads per lazily reduced madd; peak = measured v_mad_u64_u32 issue rate "
                                     "(profiles/r01_k0_int_rates.txt).  This is the hardware fraction.",
                             "upstream_work_rate": {"value": round(contract / (acc * 1e-3) / MAD_PEAK, 4),
                                                    "note": "SURVEY.md 8d contract work (n x 220 modmul x 210 mads: upstream's c=15 / W=20, CIOS) per kernel "
                                                            "second over the same peak -- a speed in units of the upstream algorithm's work, "
                                                            "not a utilisation (signed digits and wider windows do less work per pair)"}},
            "pipelining": {"msms_in_flight": depth,
                           "note": "the same K independent MSMs with `msms_in_flight` submitted at a time at every world size (pcdhip_msm_submit / collect; "
                                   "with N > 1 pcdhip_msm_submit_partial and the RCCL exchange of earlier steps); every result is read back inside the "
                                   "timed region: the throughput of a host that has several commitments to make (a Marlin round: `kzg_multi_msm`).  The "
                                   "top-level `value` is one MSM at a time -- the latency figure SURVEY.md 8d defines, beside cpu_baseline and `strong`",
                           "in_flight": {"ms_per_step": round(elapsed / args.steps * 1e3, 4),
                                         "value": round(world * n_local * args.steps / elapsed / 1e6, 3)}},
            "msm_stage_ms": {k: round(float(v), 4) for k, v in stages.items()},
            "msm_host_scalars_ms": None if host_scalars is None else {
                "pinned": round(host_scalars["pinned"], 4), "pageable": round(host_scalars["pageable"], 4), "resident": round(ms_per_step, 4),
                "value_pinned": round(n_local / host_scalars["pinned"] / 1e3, 3), "unit": "ms per MSM; value_pinned in Mscalar-mul/s",
                "note": "the headline keeps bases AND scalars resident (pcdhip_msm_dev: `h` is made on the device, the assignment crosses once per proof "
                        "for five MSMs); this is pcdhip_msm, the same MSM with its 2^20 x 40 B of scalars crossing PCIe inside every call"},
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
        if kzg_info:
            out["kzg_multi_msm"] = kzg_info
        if step_info:
            out["pcd_step"] = step_info
        if step_753:
            out["pcd_step_753"] = step_753
        if multi_info:
            out["pcd_step_multi_device"] = multi_info
        if pre_info:
            out["preflight"] = pre_info
        sys.stdout.flush()
        os.dup2(stdout_fd, 1)
        print(json.dumps(out), flush=True)
        os.dup2(2, 1)
    if use_dist:
        dist.destroy_process_group()
    return exit_code


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


def fft_source_sha16():
    import hashlib
    h = hashlib.sha256()
    for f in ("fft.hip.h", "fp.hip.h", "inst_field.hip"):
        h.update(open(os.path.join(ROOT, "pcd_amd", "csrc", f), "rb").read())
    return h.hexdigest()[:16]


def fft_roofline(fid, n, eb, N, passes, prods):
    """[{pass, bound, hbm: {achieved GB/s, frac}, mad: {frac}, traffic ...}] -- the binding roof of a transform pass is the integer multiply-add
    issue rate (its butterflies), the HBM figure is what north_star asks rocprof to show: algorithmic bytes 2 n s per pass over the LIVE pass
    time; `traffic` (FETCH x 2 + WRITE, separate --pmc runs) and the profiler's own duration come from profiles/traffic_fft_pass.json,
    reported only while its source hash matches the kernels that ran (tools/profile_fft.sh re-measures)."""
    tpath = os.path.join(ROOT, "profiles", "traffic_fft_pass.json")
    prof, note = None, "profiles/traffic_fft_pass.json is missing"
    if os.path.exists(tpath):
        t = json.load(open(tpath))
        if t.get("source_sha16") == fft_source_sha16() and t.get("n") == n:
            key = [k for k in t["kernels"] if ("F298B" if fid == 1 else "F753B") in k]
            prof = t["kernels"][key[0]] if key else None
            note = "profiles/traffic_fft_pass.json (rocprofv3, separate --pmc runs: FETCH_SIZE x 2 + WRITE_SIZE; same kernel sources)"
        else:
            note = f"profiles/traffic_fft_pass.json was measured on sources {t.get('source_sha16')}, this build is {fft_source_sha16()}: stale, not reported"
    out = []
    for i, (p, k) in enumerate(zip(passes, prods)):
        alg = 2 * n * eb
        e = {"pass": i, "bound": "valu_int32_mad", "frac": round(n * k * 2 * N * N / (p * 1e-3) / MAD_PEAK, 3), "kernel_ms": round(p, 4),
             "hbm": {"achieved": round(alg / (p * 1e-3) / 1e9, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(alg / (p * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                     "algorithmic_bytes": alg},
             "traffic": None, "traffic_source": note}
        if prof and i < len(prof):
            e["traffic"] = prof[i].get("hbm_bytes")
            e["traffic_over_algorithmic_bytes"] = prof[i].get("traffic_over_algorithmic")
            e["profiler_kernel_us"] = prof[i].get("mean_us")
        out.append(e)
    return out


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
        # seam S2 (rust/src/s2.rs hooks ark-poly's fft_in_place to pcdhip_fft on the HOST vector): the same transform through the host-vector
        # entry point -- upload, conversion, passes, conversion, download -- against the resident one (pcdhip_fft_dev), wall clock around the
        # C-ABI call, median of 5; pageable memory (a Rust Vec) and page-locked
        from pcd_amd import capi as _capi
        import ctypes as _C
        host = np.ascontiguousarray(co.gen_field(fid, n, seed=SEED + 44 + fid))
        pinned = _capi.pinned_like(host)

        def wall(fn, reps=5):
            fn()
            ts = []
            for _ in range(reps):
                t0 = time.perf_counter(); fn(); ts.append((time.perf_counter() - t0) * 1e3)
            return float(np.median(ts))
        lib, cp = _capi.lib(), ctx._ctx
        call = lambda arr: (lambda: ctx._check(lib.pcdhip_fft(cp, fid, arr.ctypes.data_as(_C.c_void_p), log_n, 0, 1)))
        t_page, t_pin = wall(call(host)), wall(call(pinned))
        t_dev = wall(lambda: (ctx.fft(fid, x, coset=True), ctx.sync()))
        x.free()
        passes = [float(np.median([r[i] for r in runs])) for i in range(len(runs[0]))]
        layers = [log_n // len(passes) + (1 if i < log_n % len(passes) else 0) for i in range(len(passes))]
        prods = [d / 2 + 1 + (1 if i == 0 else 0) for i, d in enumerate(layers)]
        out[name] = {"pass_ms": [round(p, 4) for p in passes], "transform_ms": round(sum(passes), 4),
                     "pass_GBs": [round(2 * n * eb / (p * 1e-3) / 1e9, 1) for p in passes],
                     "pass_hbm_frac": [round(2 * n * eb / (p * 1e-3) / 1e9 / HBM_PEAK_GBS, 4) for p in passes],
                     "pass_mad_frac": [round(n * k * 2 * N * N / (p * 1e-3) / MAD_PEAK, 3) for p, k in zip(passes, prods)],
                     # per pass, both roofs of the LIVE durations above; `traffic` = counter bytes of the same pass from the committed rocprofv3 runs
                     "roofline": fft_roofline(fid, n, eb, N, passes, prods),
                     "call_wall_ms": {"resident_vector (pcdhip_fft_dev)": round(t_dev, 3), "host_vector_pinned (pcdhip_fft)": round(t_pin, 3),
                                      "host_vector_pageable (pcdhip_fft)": round(t_page, 3), "host_bytes_each_way": n * (40 if N == 11 else 96),
                                      "note": "the S2 hook moves the vector over PCIe both ways per transform; a host that chains transforms "
                                              "keeps it resident (pcdhip_buf_upload / pcdhip_fft_dev / pcdhip_buf_download, or pcdhip_fft_seq)"}}
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


def kzg_section(ctx, co, log_n=20):
    """BASELINE configs[3] (SURVEY.md 8d "C4"): the KZG multi-MSM commit pattern of a Marlin prover over MNT4-298 G1 (K7; reference path
    tests/mnt4_marlin.rs:72-75 -> MarlinKZG10::commit -> KZG10::commit = a prefix MSM over powers_of_g + a hiding MSM over powers_of_gamma_g):
    ONE resident powers vector of 6n points and one gamma vector of n points, n = 2^20; a round of 11 commitments over prefixes of the
    powers -- 7 of length n (w, z_a, z_b, mask, t, g_1, h_1-like), 2 of length 6n (g_2, h_2-like), 2 of length n (opening witnesses) -- each
    with an n-point hiding MSM: 22 MSMs, submitted four at a time (pcdhip_msm_submit / collect), every result read back, the pairs of partial
    results added on the device (pcdhip_points_sum); plus the transforms at n and 4n such a round performs on resident vectors.  The shape is
    an approximation of Marlin's commit pattern (exact counts come from the Rust host).  CPU port: ONE n-point commitment (prefix MSM +
    hiding MSM) timed on the host cores and scaled by the pair count; that commitment is also the parity check."""
    curve, fr, n = 0, co.CURVE_FR[0], 1 << log_n
    t0 = time.time()
    powers = co.gen_points_mt(curve, 1, 6 * n, seed=SEED + 600)   # stands in for [tau^i] g: any points do for timing and parity
    gamma = co.gen_points_mt(curve, 1, n, seed=SEED + 601)
    polys = co.gen_scalars(fr, 6 * n, seed=SEED + 602)            # coefficient vectors are read as prefixes of this one
    blind = co.gen_scalars(fr, n, seed=SEED + 603)
    gen_s = time.time() - t0
    t0 = time.time()
    P = ctx.bases_upload(curve, 1, powers)
    G = ctx.bases_upload(curve, 1, gamma)
    up_s = time.time() - t0
    S, B = ctx.buf_upload(fr, polys), ctx.buf_upload(fr, blind)
    lengths = [n] * 7 + [6 * n] * 2 + [n] * 2
    jobs = []
    for L in lengths:
        jobs.append((P, S, L)); jobs.append((G, B, n))

    def round_pipelined(depth):
        res, pending = [], []
        for (bs, sc_, L) in jobs:
            pending.append(ctx.msm_submit(bs, sc_, offset=0, n=L))
            if len(pending) == depth:
                res.append(ctx.msm_collect(pending.pop(0)))
        while pending:
            res.append(ctx.msm_collect(pending.pop(0)))
        return [ctx.points_sum(curve, 1, np.stack([res[2 * i], res[2 * i + 1]])) for i in range(len(lengths))]

    def timed(fn, reps=3):
        fn()
        ts, out = [], None
        for _ in range(reps):
            t0 = time.perf_counter(); out = fn(); ts.append((time.perf_counter() - t0) * 1e3)
        return float(np.median(ts)), out
    ms4, outs = timed(lambda: round_pipelined(4))
    ms1, outs1 = timed(lambda: round_pipelined(1))
    if not all(np.array_equal(co.to_affine(curve, 1, a)[0], co.to_affine(curve, 1, b)[0]) for a, b in zip(outs, outs1)):
        raise SystemExit("KZG round: pipelined and one-at-a-time commitments differ")
    ffts = {}
    for ln in (log_n, log_n + 2):
        x = ctx.buf_upload(fr, co.gen_field(fr, 1 << ln, seed=SEED + 604))
        ctx.fft(fr, x); ctx.sync()
        ts = []
        for _ in range(3):
            ctx.timer_start(); ctx.fft(fr, x); ts.append(ctx.timer_stop())
        ffts[f"fft_2^{ln}_ms"] = round(float(np.median(ts)), 3)
        x.free()
    threads = max(1, min(os.cpu_count() or 1, 20))
    t0 = time.perf_counter()
    want = co.jac_add(curve, 1, co.msm(curve, 1, powers[:n], polys[:n], nthreads=threads), co.msm(curve, 1, gamma, blind, nthreads=threads))
    cpu_one_s = time.perf_counter() - t0
    if not np.array_equal(co.to_affine(curve, 1, outs[0])[0], co.to_affine(curve, 1, want)[0]):
        raise SystemExit("KZG commitment differs from the CPU oracle")
    P.free(); G.free(); S.free(); B.free()
    pairs = sum(lengths) + n * len(lengths)
    return {"workload": f"MNT4-298 G1, n = 2^{log_n}: 11 KZG commitments over prefixes of one resident 6n-point powers vector (9 of n, 2 of 6n), "
                        "each with an n-point hiding MSM: 22 MSMs",
            "pairs": pairs, "round_ms": round(ms4, 2), "round_ms_one_msm_at_a_time": round(ms1, 2), "msms_in_flight": 4,
            "Mpairs_per_s": round(pairs / ms4 / 1e3, 1), **ffts,
            "cpu_baseline": {"value_s_scaled": round(cpu_one_s * pairs / (2 * n), 1), "cores": threads, "kind": "port",
                             "sample": f"ONE n-point commitment (prefix MSM + hiding MSM, 2^{log_n + 1} pairs) on the host cores, {cpu_one_s:.2f} s, "
                                       "scaled by the round's pair count; the same commitment is the parity check"},
            "speedup_vs_cpu_port": round(cpu_one_s * pairs / (2 * n) / (ms4 / 1e3), 1),
            "key_upload_precompute_s": round(up_s, 2), "input_gen_s": round(gen_s, 1)}


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


def pcd_step(ctx, co, proofs, max_threads, roofline_curve=None, variants=True):
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
    total_dense, total_wit, total_wit_cpu = 0.0, 0.0, 0.0
    step_work = {}
    threads = min(os.cpu_count() or 1, max_threads)

    def variant(curve, r, keys, rs, cpu, sparse_bits=None):
        """one more (assignment, key) combination of the same proof shape: median of 5 proves, both assembly forms must give the same
        bytes, and the CPU port on the same inputs when asked"""
        # the key's second layout for a shorter window is opt-in (pcdhip_groth16_set_sparse_window; off by default): the witness-like leg asks for the
        # automatic rule, or for a window outright where that rule would not spend the memory; the dense leg's assignment never takes it
        ctx.groth16_set_sparse_window(sparse_bits if sparse_bits is not None else -1)
        vpk = ctx.g16_pk_upload(keys.host_struct(), curve)
        ctx.groth16_set_sparse_window(0)
        ctx.g16_pk_set_r1cs(vpk, r)
        r.z = capi.pinned_like(r.z)
        ctx.groth16_prove(vpk, r, rs[0], rs[1], resident_r1cs=True)
        wall, proof, tm, walls = median_prove(ctx, vpk, r, rs)
        vmem = ctx.g16_pk_memory(vpk)
        sparse_used, general = ctx.groth16_last_plan()   # (pcdhip_groth16_set_sparse_window: the copies for a shorter window, taken when <= 1/8 of z is general)
        ctx.groth16_set_assembly(2)
        p_chained = ctx.groth16_prove(vpk, r, rs[0], rs[1], resident_r1cs=True)[0]
        ctx.groth16_set_assembly(1)
        p_folded = ctx.groth16_prove(vpk, r, rs[0], rs[1], resident_r1cs=True)[0]
        ctx.groth16_set_assembly(0)
        vpk.free()
        if not (np.array_equal(proof, p_chained) and np.array_equal(proof, p_folded)):
            raise SystemExit("the two assembly forms of a Groth16 proof differ")
        res = {"gpu_wall_ms": round(wall, 2), "gpu_wall_ms_min_max": [round(min(walls), 2), round(max(walls), 2)],
               "gpu_device_ms": {k: round(float(v), 3) for k, v in tm.items()},
               "query_infinity_frac": {"a": round(float(np.mean(keys.a_inf)), 4), "b": round(float(np.mean(keys.b_g2_inf)), 4)},
               "key_bytes": vmem,
               "sparse_window_plan": {"used": bool(sparse_used), "general_scalars": int(general),
                                      "window": "off (library default)" if sparse_bits == 0 else "forced %d bits" % sparse_bits if sparse_bits is not None else "opt-in, automatic rule (four bits below the key's, when the copies fit a quarter of the free memory)"}}
        if cpu:
            t0 = time.perf_counter()
            want, _ = co.groth16_prove(keys, r, rs[0], rs[1], nthreads=threads)
            res["cpu_port_ms"] = round((time.perf_counter() - t0) * 1e3, 1)
            res["cpu_threads"] = threads
            if not np.array_equal(proof, want):
                raise SystemExit("GPU Groth16 proof (variant) differs from the CPU oracle")
        else:
            res["check"] = "chained == folded assembly bytes (two different computations of s*A, r*B_1); this key model against the CPU oracle: tests/"
        return res

    for name, curve, nc in proofs:
        fr = co.CURVE_FR[curve]
        t0 = time.time()
        # the constraint matrices have the shape `cs.finalize()` leaves of a verifier circuit: power-law row lengths (a few rows above 4096
        # entries), >= 80 % unit coefficients (coracle.skewed_r1cs); the assignment stays uniformly random field elements
        r = co.skewed_r1cs(fr, nc, 2, seed=SEED + curve)
        # the key: seeded points, with the points at infinity a real setup leaves in the a / b queries (variables that no row of A / B
        # mentions: a_i(tau) G = O); the densest key there can be (every entry finite: what rounds 1-3 measured) is timed beside it below
        keys = co.synthetic_keys(curve, r, seed=SEED + 10 + curve, consistent=os.environ.get("PCD_BENCH_DENSE_KEYS") != "1")
        rs = co.gen_field(fr, 2, seed=SEED + 20)
        gen_s = time.time() - t0
        t0 = time.time()
        pk = ctx.g16_pk_upload(keys.host_struct(), curve)
        ctx.g16_pk_set_r1cs(pk, r)                              # matrices are fixed per circuit: resident like the key
        plan = ctx.g16_pk_info(pk)                              # (window bits, windows) of the key's five queries
        key_mem = ctx.g16_pk_memory(pk)                         # device bytes of the five queries with their window-shifted copies
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
                      "key_upload_precompute_s": round(up_s, 2), "input_gen_s": round(gen_s, 2),
                      # HBM this proof's key holds (pcdhip_g16_pk_memory): the five queries x their window-shifted copies; `sparse_window` = the second
                      # layout for a shorter window (opt-in, 0 here: only the witness-like variant below asks for it)
                      "key_bytes": key_mem}
        if roofline_curve is None:
            # executed multiply-adds of this proof's five accumulations (pairs whose base is the point at infinity are left out of the A / B_1 / B
            # lists when enough of them are: capi.hip launch_assignment) + its 7 transforms (3 passes, ~5 products per element and pass): the
            # numerator of `whole_step_int_frac` -- sorts, fix-ups, bucket reductions, conversions and the assembly count as zero work
            N = 11
            m4, nd = int(keys.a_query.shape[0]) + 4, int(keys.domain_size)
            fa, fb = float(np.mean(keys.a_inf)), float(np.mean(keys.b_g2_inf))
            ka, kb = (1.0 - fa if fa > 1 / 16 else 1.0), (1.0 - fb if fb > 1 / 16 else 1.0)
            g2m = G2_MADS[curve]
            work = ((nd - 1) * plan["h"][1] + m4 * plan["l"][1] + ka * m4 * plan["a"][1] + kb * m4 * plan["b_g1"][1]) * madd_mads(curve) \
                + kb * m4 * plan["b_g2"][1] * g2m + 7 * 3 * nd * 5 * 2 * N * N
            step_work[name] = (int(work), wall)
            info[name]["key_windows"] = {k: list(v) for k, v in plan.items()}
        if variants and os.environ.get("PCD_BENCH_DENSE_KEYS") != "1":
            # (b) the same assignment under the DENSEST key (no point at infinity in any query): GPU only
            pk.free()
            pk = None
            dense = co.synthetic_keys(curve, r, seed=SEED + 10 + curve, consistent=False, mt=curve >= 2)
            info[name]["dense_keys"] = variant(curve, r, dense, rs, cpu=False, sparse_bits=0)
            total_dense += info[name]["dense_keys"]["gpu_wall_ms"]
            del dense
            # (c) the assignment a verifier circuit really produces (coracle.witness_r1cs: runs of bits with booleanity rows, packed words, a few
            # products -- ~45 % zeros, ~35 % ones; data_structures.rs:269-304), key as a setup over THAT system makes it, CPU port on the same inputs
            rw = co.witness_r1cs(fr, nc, 2, seed=SEED + 400 + curve)
            zc = co.fp_op(fr, "to_canonical", np.ascontiguousarray(rw.z))
            f0 = float((~zc.any(axis=1)).mean())
            f1 = float(((zc[:, 0] == 1) & ~zc[:, 1:].any(axis=1)).mean())
            del zc
            kw = co.synthetic_keys(curve, rw, seed=SEED + 410 + curve, mt=curve >= 2)
            # (large 753-bit keys: the copies for the shorter window take 54 GB at 2^20 entries -- more than the automatic rule spends beside the
            #  other keys this bench keeps resident, so this variant asks for them; 288 GB of HBM is what they are for)
            info[name]["witness_like"] = variant(curve, rw, kw, rs, cpu=True, sparse_bits=16 if (curve >= 2 and nc >= (1 << 18)) else None)
            info[name]["witness_like"]["assignment"] = {"zero_frac": round(f0, 4), "one_frac": round(f1, 4)}
            total_wit += info[name]["witness_like"]["gpu_wall_ms"]
            total_wit_cpu += info[name]["witness_like"]["cpu_port_ms"]
            del rw, kw
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
            if pk is not None:
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
    if step_work:
        tot_w, tot_t = sum(w for w, _ in step_work.values()), sum(t for _, t in step_work.values())
        info["whole_step_int_frac"] = dict({k: round(w / (t * 1e-3) / MAD_PEAK, 4) for k, (w, t) in step_work.items()},
                                           step=round(tot_w / (tot_t * 1e-3) / MAD_PEAK, 4), executed_mads={k: w for k, (w, _) in step_work.items()},
                                           note="executed multiply-adds of the accumulations (per key window plan, infinite bases left out) and of the "
                                                "transforms over the proofs' wall time; everything else counted as zero work")
    if total_dense:
        info["pcd_step_prover_ms_dense_keys"] = round(total_dense, 2)
    if total_wit:
        info["pcd_step_prover_ms_witness_like"] = round(total_wit, 2)
        info["cpu_port_ms_witness_like"] = round(total_wit_cpu, 1)
        info["speedup_vs_cpu_port_witness_like"] = round(total_wit_cpu / total_wit, 2)
        info["assignments"] = ("main figures: uniformly random field elements (the densest MSMs a system of this size can ask for); `witness_like`: "
                               ">= 70 % of z is 0 or 1 in runs, as the in-circuit verifier of the reference produces (coracle.witness_r1cs) -- "
                               "zeros never enter a bucket list, ones go to the pseudo bucket; h stays dense either way")
    return info


if __name__ == "__main__":
    sys.exit(main() or 0)
