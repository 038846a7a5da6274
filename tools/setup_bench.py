"""Measurement of the key-generation side: the fixed-base batch primitive at n = 2^20 on
MNT4-298 G1 / G2, and a complete Groth16 `generate_parameters` (circuit_specific_setup, src/ec_cycle_pcd/mod.rs:69,78)
for a 2^18-constraint synthetic circuit, each next to the CPU oracle (restatement of ark-ec FixedBaseMSM, all host
threads) on the same inputs, outputs compared bit-exact.  Not part of the default bench."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from oracle import coracle as co
from pcd_amd import capi

ctx = capi.Context(0)
threads = min(os.cpu_count() or 1, 64)
out = {"cpu_threads": threads}
for curve, group, log_n in ((0, 1, 20), (0, 2, 18), (2, 1, 18)):
    fr = co.CURVE_FR[curve]
    n = 1 << log_n
    sc = co.gen_scalars(fr, n, seed=3)
    base = co.generator(curve, group)
    ctx.fixed_base_mul(curve, group, base, sc[:1024])  # warm-up
    t = time.perf_counter(); got, inf = ctx.fixed_base_mul(curve, group, base, sc); gpu = time.perf_counter() - t
    t = time.perf_counter(); want, winf = co.fixed_base_mul(curve, group, base, sc, nthreads=threads); cpu = time.perf_counter() - t
    key = f"fixed_base_curve{curve}_g{group}_2^{log_n}"
    out[key] = {"ok_vs_oracle": bool(np.array_equal(got, want) and np.array_equal(inf, winf)), "gpu_s_incl_pcie": round(gpu, 4),
                "cpu_port_s": round(cpu, 3), "gpu_Mmul_per_s": round(n / gpu / 1e6, 2), "speedup": round(cpu / gpu, 1)}
    print(key, json.dumps(out[key]), flush=True)
for curve, nc in ((0, (1 << 18) - 8), (1, (1 << 16) - 8)):
    fr = co.CURVE_FR[curve]
    r = co.synthetic_r1cs(fr, nc, 2, seed=11 + curve)
    toxic = co.gen_field(fr, 5, seed=12)
    g1, g2 = co.generator(curve, 1), co.generator(curve, 2)
    ctx.groth16_setup(curve, co.synthetic_r1cs(fr, 1000, 2, seed=1), g1, g2, toxic)  # warm-up
    t = time.perf_counter(); K = ctx.groth16_setup(curve, r, g1, g2, toxic); gpu = time.perf_counter() - t
    t = time.perf_counter(); want = co.groth16_setup(curve, r, toxic, nthreads=threads); cpu = time.perf_counter() - t
    ok = all(np.array_equal(K[k], getattr(want, k)) for k in K if k != "domain_size")
    key = f"groth16_setup_curve{curve}_nc{nc}"
    out[key] = {"ok_vs_oracle": bool(ok), "gpu_s_incl_pcie_and_transposes": round(gpu, 3), "cpu_port_s": round(cpu, 2), "speedup": round(cpu / gpu, 1)}
    print(key, json.dumps(out[key]), flush=True)
print(json.dumps(out))
