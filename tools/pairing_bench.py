"""Developer measurement / profiling target for K6: Groth16 verification latency (process_vk + prepared verification) on the two
forms of the pairing kernels -- one wave per pairing (pairing_vm.hip.h, mode 0) and one lane per pairing (pairing.hip.h, mode 1) --
single, batch of 8 and batch of 64, beside the CPU oracle.  Run under rocprofv3 --kernel-trace --stats for profiles/.

    python tools/pairing_bench.py [curves, default 0,2] [modes, default 0,1]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
torch.zeros(1, device="cuda:0")
from oracle import coracle as co
from pcd_amd import capi

curves = [int(x) for x in (sys.argv[1] if len(sys.argv) > 1 else "0,2").split(",")]
modes = [int(x) for x in (sys.argv[2] if len(sys.argv) > 2 else "0,1").split(",")]
ctx = capi.Context(0)
for cid in curves:
    fr = co.CURVE_FR[cid]
    r = co.synthetic_r1cs(fr, 60, 3, seed=50 + cid)
    keys = co.groth16_setup(cid, r, co.gen_field(fr, 5, seed=51), nthreads=16)
    pk = ctx.g16_pk_upload(keys.host_struct(), cid)
    proofs = np.stack([ctx.groth16_prove(pk, r, *co.gen_field(fr, 2, seed=60 + i))[0] for i in range(8)])
    pk.free()
    pub_m = np.ascontiguousarray(r.z[1:r.num_inputs])
    pub = co.fp_op(fr, "to_canonical", pub_m)
    t0 = time.perf_counter(); assert co.groth16_verify(keys, pub_m, proofs[0]); cpu1 = (time.perf_counter() - t0) * 1e3
    for mode in modes:
        ctx.pairing_set_mode(mode)
        pvk = ctx.process_vk(cid, keys.alpha_g1, keys.beta_g2, keys.gamma_g2, keys.delta_g2, keys.gamma_abc_g1)
        res = {}
        for k in (1, 8, 64):
            P, U = np.concatenate([proofs] * ((k + 7) // 8))[:k], np.stack([pub] * k)
            assert ctx.groth16_verify_prepared(pvk, U, P).all()
            ts = []
            for _ in range(5):
                t0 = time.perf_counter(); ctx.groth16_verify_prepared(pvk, U, P); ts.append((time.perf_counter() - t0) * 1e3)
            res[k] = float(np.median(ts))
        t0 = time.perf_counter(); g = ctx.multi_pairing(cid, keys.alpha_g1, keys.beta_g2); one = (time.perf_counter() - t0) * 1e3
        pvk.free()
        print(f"curve {co.CURVE_NAMES[cid]} mode {'wave' if mode == 0 else 'lane'}-per-pairing: verify x1 {res[1]:.2f} ms, x8 {res[8]:.2f} ms, x64 {res[64]:.2f} ms; "
              f"one pairing {one:.2f} ms; CPU oracle verify (1 core) {cpu1:.2f} ms", flush=True)
    ctx.pairing_set_mode(0)
