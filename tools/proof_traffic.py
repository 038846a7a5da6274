"""Developer measurement (run under rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes): HBM traffic of whole Groth16 proofs --
five MSM streams + the witness map sharing the device -- so that the bytes every kernel of a proof moves can be set against the
proof's wall time ("nothing measures HBM contention there").   python tools/proof_traffic.py [curve=0] [log rows=20] [proofs=10]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from oracle import coracle as co
from pcd_amd import capi
curve = int(sys.argv[1]) if len(sys.argv) > 1 else 0
nc = (1 << (int(sys.argv[2]) if len(sys.argv) > 2 else 20)) - 8
K = int(sys.argv[3]) if len(sys.argv) > 3 else 10
ctx = capi.Context(0)
fr = co.CURVE_FR[curve]
r = co.skewed_r1cs(fr, nc, 2, seed=77)
keys = co.synthetic_keys(curve, r, seed=78, mt=True)
rs = co.gen_field(fr, 2, seed=79)
pk = ctx.g16_pk_upload(keys.host_struct(), curve)
ctx.g16_pk_set_r1cs(pk, r)
r.z = capi.pinned_like(r.z)
walls = []
for _ in range(K):
    t0 = time.perf_counter(); ctx.groth16_prove(pk, r, rs[0], rs[1], resident_r1cs=True); walls.append((time.perf_counter() - t0) * 1e3)
print(f"PROOFS curve={curve} rows={nc} proofs={K} wall_ms_median={np.median(walls[2:]):.3f} wall_ms_all={[round(w, 2) for w in walls]}", flush=True)
