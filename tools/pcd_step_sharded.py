"""BASELINE configs[4] on whatever devices are visible: one Groth16 proof whose MSMs run on ALL devices of a multi-device context
(pcdhip_init_devices: every query sharded by point range, the witness map on device 0, partial results summed on device 0), next to
independent DAG branches, one per device, each in its own host thread with its own context.  Proofs are compared with the CPU oracle.

    python tools/pcd_step_sharded.py [curve=0] [log_n=20] [devices=all visible, e.g. 0,1,2,3]

On a 1-GPU box the device list may name the device twice (0,0): same code path, no speed-up.  Not run by the driver."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from oracle import coracle as co
from pcd_amd import capi, dag

curve = int(sys.argv[1]) if len(sys.argv) > 1 else 0
log_n = int(sys.argv[2]) if len(sys.argv) > 2 else 20
ndev = capi.lib().pcdhip_device_count()
devices = [int(x) for x in sys.argv[3].split(",")] if len(sys.argv) > 3 else list(range(ndev))
fr = co.CURVE_FR[curve]
r = co.synthetic_r1cs(fr, (1 << log_n) - 8, 2, seed=31)
keys = co.synthetic_keys(curve, r, seed=32)
rs = co.gen_field(fr, 2, seed=33)
want, _ = co.groth16_prove(keys, r, rs[0], rs[1], nthreads=min(os.cpu_count() or 1, 64))
out = {"curve": co.CURVE_NAMES[curve], "domain": int(keys.domain_size), "devices": devices}

def timed(ctx, pk):
    ctx.groth16_prove(pk, r, rs[0], rs[1], resident_r1cs=True)
    ts = []
    for _ in range(5):
        t = time.perf_counter(); proof, _ = ctx.groth16_prove(pk, r, rs[0], rs[1], resident_r1cs=True); ts.append((time.perf_counter() - t) * 1e3)
    assert np.array_equal(proof, want), "proof differs from the oracle"
    return float(np.median(ts))

one = capi.Context(devices[0])
pk = one.g16_pk_upload(keys.host_struct(), curve); one.g16_pk_set_r1cs(pk, r)
out["one_device_ms"] = round(timed(one, pk), 2)
pk.free(); one.close()
multi = capi.Context(devices=devices)
t = time.time(); mpk = multi.g16_pk_upload(keys.host_struct(), curve); multi.g16_pk_set_r1cs(mpk, r); out["sharded_key_upload_s"] = round(time.time() - t, 2)
out["all_devices_ms"] = round(timed(multi, mpk), 2)
mpk.free(); multi.close()

def branch(ctx):
    bpk = ctx.g16_pk_upload(keys.host_struct(), curve); ctx.g16_pk_set_r1cs(bpk, r)
    ms = timed(ctx, bpk); bpk.free(); return ms
t = time.perf_counter(); per = dag.run_branches([branch] * len(devices), devices); wall = time.perf_counter() - t
out["branches_ms_each"] = [round(v, 2) for v in per]
print(json.dumps(out))
