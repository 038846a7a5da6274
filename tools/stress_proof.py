"""Randomised differential soak of the proof path against the CPU oracle (developer tool, not a test): random curve, constraint count
(so that the domains are radix-2 on the main curves and radix-2 or mixed-radix on the help curves), banded or skewed constraint
matrices, resident or per-call matrices, either assembly form, running sums or pair trees -- the proof bytes must equal the oracle's;
then the proof is verified on the GPU in both pairing forms, with a wrong public input as the negative case, and through the prepared
key.   python tools/stress_proof.py [seconds = 300] [seed = 1]"""
import os, sys, time, random
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from oracle import coracle as co
from pcd_amd import capi

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 300.0
rnd = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
ctx = capi.Context(0)
t0 = time.time()
cases = 0
per_curve = {}
while time.time() - t0 < budget:
    cid = rnd.choice((0, 0, 1, 1, 2, 3))
    fr = co.CURVE_FR[cid]
    nc = rnd.randrange(20, 250) if cid >= 2 else rnd.randrange(20, 4000)
    ni = rnd.randrange(2, 5)
    u = rnd.random()   # banded / skewed matrices with a uniform assignment, or (round 5) the witness-like system: >= 70 % of z is 0 / 1
    make = co.synthetic_r1cs if nc < 50 or u < 0.35 else co.skewed_r1cs if u < 0.65 else co.witness_r1cs
    r = make(fr, nc, ni, seed=rnd.randrange(1 << 30))
    keys = co.groth16_setup(cid, r, co.gen_field(fr, 5, seed=rnd.randrange(1 << 30)), nthreads=32)
    rs = co.gen_field(fr, 2, seed=rnd.randrange(1 << 30))
    want, winf = co.groth16_prove(keys, r, rs[0], rs[1], nthreads=16)
    spw = rnd.choice((-1, 0, 7, 9, 11, 13))   # (round 5) the key's second layout for a shorter window, taken by proofs over sparse assignments
    ctx.groth16_set_sparse_window(spw)
    pk = ctx.g16_pk_upload(keys.host_struct(), cid)
    ctx.groth16_set_sparse_window(0)
    resident = rnd.random() < 0.5
    if resident: ctx.g16_pk_set_r1cs(pk, r)
    asm = rnd.randrange(3)
    acc = rnd.choice(((0, 0, 0), (2, rnd.randrange(2, 200), rnd.randrange(0, 16))))
    sched = rnd.choice((0, 0, 1, 2))   # (2: the accumulate lane)
    bad_at = rnd.randrange(r.num_inputs - 1)
    if os.environ.get("STRESS_ONLY_CASE") and cases != int(os.environ["STRESS_ONLY_CASE"]):   # replay the random choices, skip the GPU work
        pk.free(); cases += 1
        continue
    if os.environ.get("STRESS_VERBOSE"):
        import faulthandler
        faulthandler.dump_traceback_later(45, exit=False)   # (a stalled case says where it stalls)
        print("case", cases, dict(cid=cid, nc=nc, ni=ni, make=make.__name__, resident=resident, asm=asm, acc=acc, sched=sched, spw=spw), flush=True)
    ctx.groth16_set_assembly(asm); ctx.msm_set_accumulate(*acc); ctx.groth16_set_schedule(sched)
    proof, inf = ctx.groth16_prove(pk, r, rs[0], rs[1], resident_r1cs=resident)
    ctx.groth16_set_assembly(0); ctx.msm_set_accumulate(0); ctx.groth16_set_schedule(0)
    pk.free()
    if not (np.array_equal(proof, want) and np.array_equal(inf, winf)):
        print("PROOF MISMATCH", dict(cid=cid, nc=nc, ni=ni, make=make.__name__, resident=resident, asm=asm, acc=acc, sched=sched, spw=spw, case=cases), flush=True)
        sys.exit(1)
    pub = co.fp_op(fr, "to_canonical", np.ascontiguousarray(r.z[1:r.num_inputs]))
    args = (cid, keys.alpha_g1, keys.beta_g2, keys.gamma_g2, keys.delta_g2, keys.gamma_abc_g1)
    bad = pub.copy(); bad[bad_at, 0] ^= 1
    for mode in (0, 1):
        ctx.pairing_set_mode(mode)
        if not ctx.groth16_verify(*args, pub, proof) or ctx.groth16_verify(*args, bad, proof):
            print("VERIFY MISMATCH", dict(cid=cid, nc=nc, ni=ni, mode=mode, case=cases), flush=True)
            sys.exit(1)
    ctx.pairing_set_mode(0)
    pvk = ctx.process_vk(*args)
    both = np.stack([proof, proof])
    ok = ctx.groth16_verify_prepared(pvk, np.concatenate([pub, bad]), both)
    pvk.free()
    if list(ok) != [1, 0]:
        print("PREPARED VERIFY MISMATCH", dict(cid=cid, nc=nc, ni=ni, ok=list(ok), case=cases), flush=True)
        sys.exit(1)
    cases += 1
    per_curve[cid] = per_curve.get(cid, 0) + 1
print(f"stress ok: {cases} random Groth16 prove + verify cases in {time.time() - t0:.0f} s, per curve: {sorted(per_curve.items())}", flush=True)
