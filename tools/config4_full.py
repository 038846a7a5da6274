#!/usr/bin/env python3
"""BASELINE configs[4] at its stated size, by hand (about ten minutes, most of it the CPU oracle): ONE MNT4-753 Groth16 proof over
2^22 constraints through a context of eight (logical) devices -- the merge node of an arity-8 PCD DAG
(/root/reference src/ec_cycle_pcd/data_structures.rs:269-304, mod.rs:92-181) -- compared bit for bit with the oracle's proof.

    python tools/config4_full.py [log2 constraints = 22] [shards = 8] [budget GB per vector and device = 3]

With one GPU the eight shards share it, so the window-shifted copies are capped per vector (pcdhip_set_precompute_budget): on an
8-GPU node every device holds its 26 GB share of the key with all copies.  tests/test_gpu_config4.py covers the same path at
2^17 constraints plus the eight branch proofs under `-m gpu`."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    log_nc = int(sys.argv[1]) if len(sys.argv) > 1 else 22
    shards = int(sys.argv[2]) if len(sys.argv) > 2 else 8
    budget_gb = float(sys.argv[3]) if len(sys.argv) > 3 else 3.0
    import torch
    torch.zeros(1, device="cuda:0")
    from oracle import coracle as co
    from pcd_amd import capi
    co.build(); co.lib()
    threads = min(os.cpu_count() or 1, 64)
    curve, fr = 2, co.CURVE_FR[2]
    nc = (1 << log_nc) - 8
    t0 = time.time()
    r = co.synthetic_r1cs(fr, nc, 2, seed=2200)
    keys = co.synthetic_keys(curve, r, seed=2201, mt=True)
    rs = co.gen_field(fr, 2, seed=2202)
    print(f"inputs: MNT4-753, {nc} constraints, {r.num_vars} variables, domain {keys.domain_size}; generated in {time.time() - t0:.1f} s", flush=True)
    ndev = capi.lib().pcdhip_device_count()
    devs = [i % ndev for i in range(shards)]
    ctx = capi.Context(devices=devs)
    if ndev < shards:   # the shards share one device: fewer copies per vector, and a smaller window so that the bucket arrays of 8 x 5 MSMs fit
        ctx.set_precompute_budget(int(budget_gb * (1 << 30)))
        ctx.msm_config(17, 0)
    t0 = time.time()
    pk = ctx.g16_pk_upload(keys.host_struct(), curve)
    ctx.g16_pk_set_r1cs(pk, r)
    print(f"key upload + window-shifted copies over devices {devs}: {time.time() - t0:.1f} s", flush=True)
    walls = []
    for _ in range(3):
        t0 = time.perf_counter()
        proof, inf = ctx.groth16_prove(pk, r, rs[0], rs[1], resident_r1cs=True)
        walls.append((time.perf_counter() - t0) * 1e3)
    tm = ctx.groth16_last_timings()
    print(f"sharded prove wall ms: {[round(w, 1) for w in walls]}; device: witness map (device 0) {tm['witness_map']:.1f} ms, total {tm['total']:.1f} ms", flush=True)
    pk.free(); ctx.close()
    t0 = time.time()
    want, winf = co.groth16_prove(keys, r, rs[0], rs[1], nthreads=threads)
    print(f"CPU oracle prove ({threads} threads): {time.time() - t0:.1f} s", flush=True)
    ok = np.array_equal(proof, want) and np.array_equal(inf, winf)
    print("RESULT:", "sharded proof == oracle proof (bit-exact)" if ok else "MISMATCH")
    return 0 if ok else 1


if __name__ == "__main__":
    sys.exit(main())
