"""Developer A/B: the main MNT4-298 proof at 2^20 (five MSMs on five streams + the witness map, all concurrent) -- median wall of 7
proves and the device stage times; run once per library (PCDHIP_LIB=...) on the same box.  Used for questions that only show under
concurrency, e.g. whether the 88-byte base records (a gather touches 1.7 lines) cost anything when five MSM streams share HBM."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
torch.zeros(1, device="cuda:0")
from oracle import coracle as co
from pcd_amd import capi
curve = int(sys.argv[1]) if len(sys.argv) > 1 else 0
nc = int(os.environ['AB_NC']) if os.environ.get('AB_NC') else (1 << int(sys.argv[2])) - 8 if len(sys.argv) > 2 else (1 << 20) - 8
ctx = capi.Context(0)
fr = co.CURVE_FR[curve]
r = (co.witness_r1cs if os.environ.get("AB_WITNESS") else co.skewed_r1cs)(fr, nc, 2, seed=77)   # AB_WITNESS=1: >= 70 % of z is 0 / 1
if os.environ.get("AB_BENCH_INPUTS"):   # exactly what bench.py's pcd_step proves (its seeds; keys from the sequential point walk, not gen_points_mt)
    SEED = 0x5043443031
    r = co.skewed_r1cs(fr, nc, 2, seed=SEED + curve)
    keys = co.synthetic_keys(curve, r, seed=SEED + 10 + curve)
else:
    keys = co.synthetic_keys(curve, r, seed=78, mt=True, consistent=os.environ.get("AB_DENSE", "0") != "1")
rs = co.gen_field(fr, 2, seed=79)
if os.environ.get("AB_SPARSE_WINDOW"):   # pcdhip_groth16_set_sparse_window: -1 automatic, 0 off, 6..22 forced bits
    ctx.groth16_set_sparse_window(int(os.environ["AB_SPARSE_WINDOW"]))
pk = ctx.g16_pk_upload(keys.host_struct(), curve)
ctx.g16_pk_set_r1cs(pk, r)
r.z = capi.pinned_like(r.z)
if os.environ.get("AB_ASSEMBLY"):   # 1: folded (two more MSMs), 2: chained (one-point products), 0: by size
    ctx.groth16_set_assembly(int(os.environ["AB_ASSEMBLY"]))
first = None
# AB_SCHEDULES: comma list of `schedule` or `schedule:reserved CUs` (2: accumulate lane, the default; 0: rounds 1-4; 1: map first, all MSMs at once)
for item in os.environ.get("AB_SCHEDULES", "0,2:8,0,2:8").split(","):
    sched, _, res = item.partition(":")
    sched = int(sched)
    ctx.groth16_set_schedule(sched)
    if res:
        ctx.set_lane_reserve(int(res))
    for _ in range(3):
        proof, _ = ctx.groth16_prove(pk, r, rs[0], rs[1], resident_r1cs=True)
    first = proof if first is None else first
    assert np.array_equal(proof, first)
    walls = []
    for _ in range(7):
        t0 = time.perf_counter(); ctx.groth16_prove(pk, r, rs[0], rs[1], resident_r1cs=True); walls.append((time.perf_counter() - t0) * 1e3)
    tm = {k: round(float(v), 2) for k, v in ctx.groth16_last_timings().items()}
    plan = ctx.groth16_last_plan() if hasattr(ctx, "groth16_last_plan") else None
    print(f"lib={os.environ.get('PCDHIP_LIB', 'in-tree')} curve={curve} schedule={item} prove wall ms median {np.median(walls):.3f} min {min(walls):.3f}; device {tm}; sparse plan {plan}", flush=True)
if os.environ.get("AB_MAP_ALONE"):   # the witness map with the device to itself, same matrices and assignment
    wm = [ctx.witness_map_resident(pk, r, want_h=False)[1] for _ in range(5)][-1]
    print(f"witness map alone, device ms: { {k: round(float(v), 3) for k, v in wm.items()} }", flush=True)
