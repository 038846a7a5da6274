"""One-off measurement of BASELINE configs[2]: prover arithmetic of a PCD step on the 753-bit cycle
(main proof MNT4-753, domain 2^20; help proof MNT6-753 on the mixed-radix domain 5 * 2^15 its circuit size
forces), synthetic keys, proof bytes checked against the CPU oracle.  Not part of the default bench (the CPU
check alone takes minutes)."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from oracle import coracle as co
from pcd_amd import capi
import bench

ctx = capi.Context(0)
out = {}
for name, curve, nc in (("main_mnt4_753", 2, (1 << 20) - 8), ("help_mnt6_753", 3, (1 << 15) + 20000)):
    fr = co.CURVE_FR[curve]
    t = time.time()
    r = co.synthetic_r1cs(fr, nc, 2, seed=5 + curve)
    dom = co.domain_size(fr, nc + 2)
    # Keys sized for the actual domain (h_query has dom - 1 entries)
    class R: pass
    rr = r
    keys = bench.synthetic_keys(co, curve, r, seed=77 + curve) if dom == (1 << r.domain_log) else None
    if keys is None:
        # mixed-radix domain: build the key with the right h_query length
        m, ni = r.num_vars, r.num_inputs
        g1 = co.gen_points(curve, 1, 2 * m + (dom - 1) + (m - ni) + 3, seed=77 + curve)
        g2 = co.gen_points(curve, 2, m + 2, seed=78 + curve)
        z8 = lambda k: np.zeros(k, dtype=np.uint8)
        o = [0]
        def take(k):
            v = np.ascontiguousarray(g1[o[0]:o[0] + k]); o[0] += k; return v
        A = dict(a_query=take(m), b_g1_query=take(m), h_query=take(dom - 1), l_query=take(m - ni))
        A.update(alpha_g1=take(1)[0], beta_g1=take(1)[0], delta_g1=take(1)[0])
        A.update(b_g2_query=np.ascontiguousarray(g2[:m]), beta_g2=np.ascontiguousarray(g2[m]), delta_g2=np.ascontiguousarray(g2[m + 1]),
                 gamma_g2=np.ascontiguousarray(g2[m + 1]), gamma_abc_g1=np.ascontiguousarray(g1[:ni]), gamma_abc_inf=z8(ni),
                 a_inf=z8(m), b_g1_inf=z8(m), b_g2_inf=z8(m), h_inf=z8(dom - 1), l_inf=z8(m - ni))
        keys = co.Keys(curve, r, A)
        keys.domain_size = dom
    rs = co.gen_field(fr, 2, seed=9)
    r.z = capi.pinned_like(r.z)
    tsetup = time.time() - t
    t = time.time(); pk = ctx.g16_pk_upload(keys.host_struct(), curve); ctx.g16_pk_set_r1cs(pk, r); tup = time.time() - t
    ctx.groth16_set_assembly(2)
    ctx.groth16_prove(pk, r, rs[0], rs[1], resident_r1cs=True)
    t = time.perf_counter(); proof, inf = ctx.groth16_prove(pk, r, rs[0], rs[1], resident_r1cs=True); wall = (time.perf_counter() - t) * 1e3
    tm = ctx.groth16_last_timings()
    ctx.groth16_set_assembly(1)
    ctx.groth16_prove(pk, r, rs[0], rs[1], resident_r1cs=True)
    t = time.perf_counter(); proof_f, _ = ctx.groth16_prove(pk, r, rs[0], rs[1], resident_r1cs=True); wall_f = (time.perf_counter() - t) * 1e3
    ctx.groth16_set_assembly(0)
    threads = min(os.cpu_count() or 1, 64)
    t = time.perf_counter(); want, _ = co.groth16_prove(keys, r, rs[0], rs[1], nthreads=threads); cpu_ms = (time.perf_counter() - t) * 1e3
    ok = bool(np.array_equal(proof, want) and np.array_equal(proof_f, want))
    out[name] = {"domain": int(dom), "ok_vs_oracle": ok, "gpu_wall_ms": round(min(wall, wall_f) if curve >= 2 and dom <= (1 << 18) else wall, 1), "gpu_wall_ms_chained_assembly": round(wall, 1), "gpu_wall_ms_folded_assembly": round(wall_f, 1), "gpu_device_ms": {k: round(float(v), 2) for k, v in tm.items()},
                 "cpu_port_ms": round(cpu_ms), "cpu_threads": threads, "key_upload_precompute_s": round(tup, 1), "input_gen_s": round(tsetup, 1)}
    print(name, json.dumps(out[name]), flush=True)
    pk.free()
out["pcd_step_prover_ms"] = round(sum(v["gpu_wall_ms"] for v in out.values() if isinstance(v, dict)), 1)
print(json.dumps(out))
