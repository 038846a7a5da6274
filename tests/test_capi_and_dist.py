"""CPU: the C-ABI library loads and exports every symbol include/pcdhip.h declares; host-side facts;
the product refuses to run without a GPU (no CPU fallback); world_size-2 gloo test of the multi-GPU path."""
import os
import re
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    from pcd_amd import capi
    lib = capi.lib()
    hdr = open(os.path.join(ROOT, "include", "pcdhip.h")).read()
    declared = sorted(set(re.findall(r"\b(pcdhip_[a-z0-9_]+)\s*\(", hdr)))
    assert len(declared) >= 30
    missing = [n for n in declared if not hasattr(lib, n)]
    assert not missing, missing
    assert sorted(capi.EXPORTS) == declared


def test_static_facts():
    from pcd_amd import capi
    lib = capi.lib()
    assert [lib.pcdhip_field_limbs(f) for f in range(4)] == [5, 5, 12, 12]
    assert [lib.pcdhip_curve_base_field(c) for c in range(4)] == [0, 1, 2, 3]
    assert [lib.pcdhip_curve_scalar_field(c) for c in range(4)] == [1, 0, 3, 2]
    assert [lib.pcdhip_point_limbs(c, 1) for c in range(4)] == [10, 10, 24, 24]
    assert [lib.pcdhip_point_limbs(c, 2) for c in range(4)] == [20, 30, 48, 72]
    assert lib.pcdhip_field_limbs(9) < 0 and lib.pcdhip_point_limbs(0, 3) < 0
    assert b"no CPU fallback" in lib.pcdhip_strerror(-3)


def test_no_cpu_fallback():
    """Without a GPU the product fails loudly instead of computing on the host."""
    from pcd_amd import capi
    if capi.lib().pcdhip_device_count() > 0:
        pytest.skip("a GPU is present")
    with pytest.raises(capi.PcdHipError):
        capi.Context(0)


def test_product_does_not_use_oracle():
    """Nothing under pcd_amd/ or include/ may import, include, link or load anything from oracle/."""
    bad = re.compile(r"(import\s+oracle|from\s+oracle|from\s+\.+oracle|liboracle|oracle/|coracle|pyoracle|orc_[a-z])")
    for top in ("pcd_amd", "include"):
        for dirpath, _, files in os.walk(os.path.join(ROOT, top)):
            for f in files:
                if f.endswith((".py", ".hip", ".h", ".cpp", "Makefile")):
                    src = open(os.path.join(dirpath, f), errors="ignore").read()
                    assert not bad.search(src), (dirpath, f, bad.search(src).group(0))


def test_shard_ranges():
    from pcd_amd.dist import shard_range
    for n in (0, 1, 7, 1 << 20, (1 << 20) + 5):
        for world in (1, 2, 3, 8):
            rs = [shard_range(n, r, world) for r in range(world)]
            assert rs[0][0] == 0 and rs[-1][1] == n
            assert all(rs[i][1] == rs[i + 1][0] for i in range(world - 1))
            assert max(h - l for l, h in rs) - min(h - l for l, h in rs) <= 1


WORKER = r'''
import os, sys
import numpy as np
sys.path.insert(0, sys.argv[1])
import torch.distributed as dist
from oracle import coracle as co
from pcd_amd.dist import sharded_msm
dist.init_process_group("gloo")
cid, grp, n = 0, 1, 3001
pts = co.gen_points(cid, grp, n, seed=5)
sc = co.gen_scalars(co.CURVE_FR[cid], n, seed=6, dist=1)
# the oracle stands in for the per-rank GPU pipeline: this test covers partition + exchange + combine
def msm_fn(lo, hi): return co.msm(cid, grp, pts[lo:hi], sc[lo:hi])
def sum_fn(parts):
    acc = parts[0]
    for p in parts[1:]: acc = co.jac_add(cid, grp, acc, p)
    return acc
full = sharded_msm(msm_fn, sum_fn, n)
want = co.msm(cid, grp, pts, sc)
ok = np.array_equal(co.to_affine(cid, grp, full)[0], co.to_affine(cid, grp, want)[0])
print("RANK", dist.get_rank(), "OK" if ok else "MISMATCH", flush=True)
dist.destroy_process_group()
sys.exit(0 if ok else 1)
'''


def test_sharded_msm_two_ranks_gloo(co, tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
           "--master-port", "29541", str(script), ROOT]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout + out.stderr
    assert out.stdout.count("OK") == 2


GPU_WORKER = r'''
import os, sys
import numpy as np
sys.path.insert(0, sys.argv[1])
import torch
import torch.distributed as dist
from oracle import coracle as co
from pcd_amd import capi
from pcd_amd.dist import sharded_msm, shard_range, DeviceExchange
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
ngpu = torch.cuda.device_count()
two = ngpu >= world
dev = rank if two else 0                      # one GPU per rank when there are enough; otherwise the ranks share cuda:0
torch.cuda.set_device(dev)
torch.zeros(1, device=f"cuda:{dev}")
dist.init_process_group("nccl" if two else "gloo")   # RCCL needs distinct devices; with one GPU the exchange runs over gloo
ctx = capi.Context(dev)
cid, grp, n = 0, 1, 40001
fr = co.CURVE_FR[cid]
pts = co.gen_points(cid, grp, n, seed=5)
sc = co.gen_scalars(fr, n, seed=6, dist=1)
lo, hi = shard_range(n, rank, world)
bases = ctx.bases_upload(cid, grp, pts[lo:hi])             # this rank's point range of the key is resident on ITS device
sbuf = ctx.buf_upload(fr, sc[lo:hi])
if two:
    ex = DeviceExchange(ctx, cid, grp, torch.device("cuda", dev))
    full = ex.msm(bases, sbuf)                                # partial -> RCCL all-gather -> EC sum, on device
    pending, outs = [], []                                    # and the pipelined form bench.py times: four shard MSMs in flight
    for _ in range(6):
        pending.append(ex.submit(bases, sbuf))
        if len(pending) == 4:
            outs.append(ex.collect(pending.pop(0)))
    while pending:
        outs.append(ex.collect(pending.pop(0)))
    assert all(np.array_equal(o, full) for o in outs)
else:
    full = sharded_msm(lambda a, b: ctx.msm(bases, sbuf), lambda parts: ctx.points_sum(cid, grp, parts), n)
want = co.msm(cid, grp, pts, sc, nthreads=4)
ok = np.array_equal(co.to_affine(cid, grp, full)[0], co.to_affine(cid, grp, want)[0])
print("RANK", rank, "backend", "nccl" if two else "gloo", "OK" if ok else "MISMATCH", flush=True)
dist.destroy_process_group()
sys.exit(0 if ok else 1)
'''


@pytest.mark.gpu
def test_sharded_msm_two_ranks_gpu(co, tmp_path):
    """world_size 2 with the GPU pipeline per rank: RCCL + the device-resident exchange when two devices are visible, otherwise both
    ranks on cuda:0 with the exchange over gloo -- partition, per-rank HIP MSM, gather, EC sum == the oracle's full MSM on both ranks."""
    script = tmp_path / "gpu_worker.py"
    script.write_text(GPU_WORKER)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
           "--master-port", "29543", str(script), ROOT]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env)
    assert out.returncode == 0, out.stdout[-3000:] + out.stderr[-3000:]
    assert out.stdout.count("OK") == 2


def test_bench_multi_gpu_failure_still_prints_one_line(tmp_path):
    """VERDICT r04 #5 (first contact with a multi-GPU node): whatever fails on the N > 1 path -- here the very first step, on a box without
    a GPU -- rank 0 still prints exactly ONE JSON line (the N = 1 figures of a fresh child process, or, when that fails too, a line that
    says so) with the error attached, and the exit code is non-zero.  No hang: the run is bounded."""
    import json
    import subprocess
    import sys
    env = dict(os.environ, PCD_BENCH_FORCE_DIST="1", PCD_BENCH_INJECT_FAILURE="init", PCD_BENCH_DIST_TIMEOUT_S="120")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        env.pop(k, None)
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    p = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "1", "--steps", "2", "--warmup", "1"], env=env,
                       capture_output=True, text=True, timeout=900)
    lines = [ln for ln in p.stdout.splitlines() if ln.strip()]
    assert p.returncode != 0, p.stderr[-500:]
    assert len(lines) == 1, p.stdout
    out = json.loads(lines[0])
    assert out["metric"] == "msm_mscalar_mul_per_s" and "multi_gpu_error" in out and out["n_gpus_requested"] == 1


def test_toolchain_the_copyprop_workaround_was_validated_on():
    """pcd_amd/csrc/Makefile builds with `-mllvm -disable-copyprop` because ROCm 7.2's MachineCopyPropagation deletes a live copy in one kernel
    (tests/gpucheck re-runs the reproducer on the GPU).  The flag, and the conclusion that it costs nothing, were validated on
    HIP 7.2 only: a different toolchain must not pass silently -- re-run tests/test_gpu_mailbox.py and
    tools/ab_libs.sh with and without the flag on the new toolchain, then update the version here."""
    import re
    import subprocess
    out = subprocess.run(["hipcc", "--version"], capture_output=True, text=True).stdout
    m = re.search(r"HIP version:\s*(\d+)\.(\d+)", out)
    assert m, out
    assert (int(m.group(1)), int(m.group(2))) == (7, 2), f"hipcc is HIP {m.group(1)}.{m.group(2)}: the -disable-copyprop work-around was validated on 7.2 only"
    mk = open(os.path.join(ROOT, "pcd_amd", "csrc", "Makefile")).read()
    assert "-mllvm -disable-copyprop" in mk
